#!/bin/bash
out=$PWD/gpurun_out/s7; mkdir -p $out
for v in stream nostream fd2 mp1 mp4; do
  lib=""; [ $v != stream ] && lib="--lib tools/_abl/$v.so"
  echo "== $v"
  python tools/bench_conv.py $lib --levels 16,32 --ops fwd_stats,dgrad --rounds 4 --json $out/c_$v.json --variants "d:" 2>&1 | grep -v amdgpu
done > $out/stream.txt 2>&1
cat $out/stream.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_ms_per_step_empty_queue'])"; }
for rep in 1 2; do echo "rep $rep step default $(ab)"; done
