#!/bin/bash
# round 4, GPU session 28: does a 20-step window depend on how warm the GPU is?  (the driver runs bench.py --steps 20 --warmup 5; the standalone
# ACDC line and the one inside the LA run's extra_workloads differed by 3 % in r04_t4)
out=$PWD/gpurun_out/r04_s28; mkdir -p $out
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep acdc w5/k20 $(ab --workload acdc --steps 20 --warmup 5) w40/k20 $(ab --workload acdc --steps 20 --warmup 40) w5/k80 $(ab --workload acdc --steps 80 --warmup 5) | la w5/k20 $(ab --steps 20 --warmup 5) w40/k20 $(ab --steps 20 --warmup 40) w5/k80 $(ab --steps 80 --warmup 5)"
done 2>&1 | tee $out/ab.txt
