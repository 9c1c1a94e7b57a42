#!/bin/bash
# round 3, GPU session 16: stage reorder in the LDS-staged kernels (weights stash + refill before the MFMAs, scalar stage base): probe + alone timings + step A/B
out=$PWD/gpurun_out/s16; mkdir -p $out
(python tools/ts_probe.py tools/_abl/ts.so 64; python tools/ts_probe.py tools/_abl/ts.so 128) 2>&1 | grep -v amdgpu | grep -v "^wg" > $out/ts.txt; cat $out/ts.txt
python tools/bench_conv.py --levels 64,128,256 --ops fwd_stats,dgrad --json $out/c_new.json > $out/c.txt 2>&1
python tools/bench_conv.py --levels 64,128,256 --ops fwd_stats,dgrad --lib tools/_abl/prev.so --json $out/c_prev.json >> $out/c.txt 2>&1; cat $out/c.txt
python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3" 2>&1 | tail -3
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_ms_per_step_empty_queue'])"; }
cp bcp_amd/csrc/libbcp_hip.so /tmp/new.so
for rep in 1 2; do
  cp tools/_abl/prev.so bcp_amd/csrc/libbcp_hip.so
  echo "rep $rep la prev $(ab)"; echo "rep $rep acdc prev $(ab --workload acdc)"; echo "rep $rep panc prev $(ab --workload pancreas)"
  cp /tmp/new.so bcp_amd/csrc/libbcp_hip.so
  echo "rep $rep la new  $(ab)"; echo "rep $rep acdc new  $(ab --workload acdc)"; echo "rep $rep panc new  $(ab --workload pancreas)"
done > $out/ab.txt 2>&1; cat $out/ab.txt
