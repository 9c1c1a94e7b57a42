#!/bin/bash
# round 3, GPU session 25: first-layer kernel with the next tile's halo prefetched in registers; plan / replay ADVICE fixes under the full GPU suite
out=$PWD/gpurun_out/s25; mkdir -p $out
( time python -m pytest tests -m gpu -q -x ) 2>&1 | tail -6
python tools/bench_conv.py --levels 16 --ops c1_norm_fwd,c1_norm_bwd --json $out/c_new.json > $out/c.txt 2>&1
python tools/bench_conv.py --levels 16 --ops c1_norm_fwd,c1_norm_bwd --lib tools/_abl/prev.so --json $out/c_prev.json >> $out/c.txt 2>&1; grep -v amdgpu $out/c.txt | tail -12
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_ms_per_step_empty_queue'])"; }
cp bcp_amd/csrc/libbcp_hip.so /tmp/new.so
for rep in 1 2; do
  cp tools/_abl/prev.so bcp_amd/csrc/libbcp_hip.so
  echo "rep $rep la prev $(ab)"; echo "rep $rep acdc prev $(ab --workload acdc)"; echo "rep $rep panc prev $(ab --workload pancreas)"
  cp /tmp/new.so bcp_amd/csrc/libbcp_hip.so
  echo "rep $rep la new  $(ab)"; echo "rep $rep acdc new  $(ab --workload acdc)"; echo "rep $rep panc new  $(ab --workload pancreas)"
done > $out/ab.txt 2>&1; cat $out/ab.txt
