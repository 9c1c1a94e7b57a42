#!/bin/bash
# round 4, GPU session 29: why is extra_workloads.acdc inside the LA run slower than the standalone ACDC line (6848 vs 7068 in r04_t4)?
out=$PWD/gpurun_out/r04_s29; mkdir -p $out
ex() { python bench.py --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d.get('extra_workloads',{}); print(d['ms_per_step'], {k:(v.get('ms_per_step'), v.get('host_enqueue_ms_per_step')) for k,v in e.items()})"; }
for rep in 1 2; do
  echo "rep $rep full: $(ex)"
  echo "rep $rep no cpu baseline: $(ex --no-cpu-baseline)"
  echo "rep $rep standalone acdc: $(ex --workload acdc --no-cpu-baseline)"
done 2>&1 | tee $out/ab.txt
