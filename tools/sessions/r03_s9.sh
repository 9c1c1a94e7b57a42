#!/bin/bash
out=$PWD/gpurun_out/s9; mkdir -p $out
( time python -m pytest tests -m gpu -q ) > $out/pytest_gpu.txt 2>&1; tail -4 $out/pytest_gpu.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_ms_per_step_empty_queue'])"; }
for rep in 1 2; do
  echo "rep $rep la fuse_c1=0 $(ab --opt fuse_c1=0)"; echo "rep $rep la default   $(ab)"
  echo "rep $rep acdc fuse_c1=0 $(ab --workload acdc --opt fuse_c1=0)"; echo "rep $rep acdc default   $(ab --workload acdc)"
  echo "rep $rep panc fuse_c1=0 $(ab --workload pancreas --opt fuse_c1=0)"; echo "rep $rep panc default   $(ab --workload pancreas)"
done > $out/ab.txt 2>&1; cat $out/ab.txt
