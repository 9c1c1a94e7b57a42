#!/bin/bash
out=$PWD/gpurun_out/s8; mkdir -p $out
for v in new w6old; do
  lib=""; [ $v != new ] && lib="--lib tools/_abl/$v.so"
  echo "== $v"
  python tools/bench_conv.py $lib --levels 16,32,64,128,256 --ops wgrad,fwd_stats --rounds 4 --json $out/c_$v.json --variants "d:" 2>&1 | grep -v amdgpu
done > $out/w6.txt 2>&1
cat $out/w6.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_ms_per_step_empty_queue'])"; }
for rep in 1 2; do echo "rep $rep step default $(ab)"; echo "rep $rep acdc $(ab --workload acdc)";  echo "rep $rep panc $(ab --workload pancreas)"; done
