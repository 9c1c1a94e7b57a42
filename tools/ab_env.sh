#!/bin/bash
# measurement only: interleaved in-step A/B of environment switches on one box: tools/ab_env.sh "" "BCP_X=1" "BCP_Y=2 BCP_Z=3" ...
# (each argument is one variant's environment; three rounds, LA bench, ms/step and volumes/s)
for r in 1 2 3; do
  for v in "$@"; do
    echo -n "[${v:-default}] "; env $v python bench.py --no-cpu-baseline --steps 40 --warmup 10 ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
