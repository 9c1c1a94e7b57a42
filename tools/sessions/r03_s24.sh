#!/bin/bash
# round 3, GPU session 24: k_c3q weight streams dealt to XCDs; split-K at the 128-channel level
out=$PWD/gpurun_out/s24; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3" 2>&1 | tail -3
python tools/bench_conv.py --levels 128,256 --ops fwd_stats,dgrad,fwd_chain --json $out/c_new.json --variants "new:;sk2:conv3_b6_flat_sk=2;sk8:conv3_b6_flat_sk=8;noxcd:conv3_xcd=0" > $out/c.txt 2>&1
python tools/bench_conv.py --levels 128,256 --ops fwd_stats,dgrad,fwd_chain --lib tools/_abl/prev.so --json $out/c_prev.json >> $out/c.txt 2>&1; grep -v amdgpu $out/c.txt | grep -v fp32
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_ms_per_step_empty_queue'])"; }
cp bcp_amd/csrc/libbcp_hip.so /tmp/new.so
for rep in 1 2; do
  cp tools/_abl/prev.so bcp_amd/csrc/libbcp_hip.so
  echo "rep $rep la prev $(ab)"; echo "rep $rep panc prev $(ab --workload pancreas)"
  cp /tmp/new.so bcp_amd/csrc/libbcp_hip.so
  echo "rep $rep la new  $(ab)"; echo "rep $rep panc new  $(ab --workload pancreas)"
done > $out/ab.txt 2>&1; cat $out/ab.txt
