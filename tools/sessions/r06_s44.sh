#!/bin/bash
# round 6, GPU session 44: does the timed value depend on the number of warm-up / timed steps? (20 / 3 = the default, 20 / 10, 40 / 10, 100 / 10), interleaved
out=$PWD/gpurun_out/r06_s44; mkdir -p $out
for r in 1 2 3; do for sw in "20 3" "20 10" "40 10" "100 10"; do set -- $sw
  python bench.py --no-extra --no-cpu-baseline --no-roofline --steps $1 --warmup $2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('steps $1 warmup $2', d['value'], d['ms_per_step'])"; done; done | tee $out/ab.txt
