#!/bin/bash
# round 3, GPU session 18: k_c3p (LDS-DMA software pipeline of the 64 x 64 tile): parity, probe, alone timings, step A/B
out=$PWD/gpurun_out/s18; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3" 2>&1 | tail -5
(python tools/ts_probe.py tools/_abl/ts.so 64) 2>&1 | grep -v amdgpu | grep -v "^wg" > $out/ts.txt; cut -c1-700 $out/ts.txt
python tools/bench_conv.py --levels 64 --ops fwd_stats,dgrad,fwd_chain,bwd_chain --json $out/c.json --variants "pipe:;c3h:conv3_b6_pipe=0" > $out/c.txt 2>&1; cat $out/c.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_ms_per_step_empty_queue'])"; }
for rep in 1 2; do
  echo "rep $rep la c3h  $(ab --opt conv3_b6_pipe=0)"; echo "rep $rep la pipe $(ab)"
  echo "rep $rep panc c3h  $(ab --workload pancreas --opt conv3_b6_pipe=0)"; echo "rep $rep panc pipe $(ab --workload pancreas)"
done > $out/ab.txt 2>&1; cat $out/ab.txt
