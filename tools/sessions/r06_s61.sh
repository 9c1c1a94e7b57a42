#!/bin/bash
# round 6, GPU session 61: VERDICT r05 item 4 priced -- the fp16 split arithmetic compiled out of the bf16-pipe kernels (B6_ABLATE = 128: fetch and LDS stores stay), alone and in the step (results wrong, timing valid; largest-CC left out in every arm: its time depends on the data)
out=$PWD/gpurun_out/r06_s61; mkdir -p $out
for v in none fwd all; do
  echo "== split_$v"; timeout 600 python tools/bench_conv.py --lib tools/_abl/split_$v.so --levels 16,32,64 --ops fwd_stats,dgrad,wgrad --rounds 3 --iters 20 --variants "f16:" --json $out/alone_$v.json 2>&1 | grep -v "^$"
done 2>&1 | tee $out/alone.txt
cp bcp_amd/csrc/libbcp_hip.so /tmp/libbcp_keep2.so
for r in 1 2 3; do for v in none fwd all; do
  cp tools/_abl/split_$v.so bcp_amd/csrc/libbcp_hip.so
  for w in la acdc; do echo -n "$w split_$v "; python bench.py --workload $w --no-extra --no-cpu-baseline --no-roofline --steps 40 --warmup 10 --opt-late whatif=4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done
done; done 2>&1 | tee $out/ab.txt
cp /tmp/libbcp_keep2.so bcp_amd/csrc/libbcp_hip.so
