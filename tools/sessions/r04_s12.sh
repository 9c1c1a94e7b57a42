#!/bin/bash
# round 4, GPU session 12: backward statistics in the dgrad epilogue everywhere (1) vs not at the 3-D 32-channel slabs (2)
out=$PWD/gpurun_out/r04_s12; mkdir -p $out
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep la base $(ab) bw2 $(ab --opt fuse_bwd_stats=2) | panc base $(ab --workload pancreas) bw2 $(ab --workload pancreas --opt fuse_bwd_stats=2)"
done 2>&1 | tee $out/ab.txt
