#!/bin/bash
# round 4, GPU session 32: forward passes as HIP graphs (plan.GRAPHS = 1) against per-launch replays (0) -- cost of giving the graphs up, after
# tools/_abl/flaky3.py showed the teacher's graph output deviating (1e-5, 3-6 pseudo-label pixels) in 24 of 150 small ACDC runs when another
# stream runs beside it, 0 of 150 as per-launch replays, 0 of 150 with a host synchronisation behind every graph launch
out=$PWD/gpurun_out/r04_s32; mkdir -p $out
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_enqueue_ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep la graphs $(ab) replay $(ab --opt graphs=0) | acdc graphs $(ab --workload acdc) replay $(ab --workload acdc --opt graphs=0) | panc graphs $(ab --workload pancreas) replay $(ab --workload pancreas --opt graphs=0)"
done 2>&1 | tee $out/ab.txt
