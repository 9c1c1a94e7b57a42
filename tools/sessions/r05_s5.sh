#!/bin/bash
# round 5, GPU session 5: s4 put the first divergent values into the OUTPUT OF bcp_bilinear2x_fwd (component .y of 8-16 consecutive lanes' float4,
# plausible values) in the round-4 build; the rebuilt kernel (another amax expression, same arithmetic) never deviates.  Is the old kernel wrong
# on its own under load (reproducer), and what do the wrong values correspond to (dump)?
out=$PWD/gpurun_out/r05_s5; mkdir -p $out
R=$PWD
cd tools/_abl/r04head
for cfg in "workers=2 load=1" "workers=2 load=0" "workers=1 load=1" "workers=3 load=1 C=64 H=8" "workers=2 load=1 C=128 H=4"; do
  timeout 300 python tools/probe/bilinear_race_probe.py rounds=30 $cfg 2>&1 | grep -v amdgpu.ids | tee -a $out/probe_old.txt | tail -8 | cut -c1-400
done
cd $R
for cfg in "workers=2 load=1" "workers=3 load=1 C=64 H=8"; do
  timeout 300 python tools/probe/bilinear_race_probe.py rounds=30 $cfg 2>&1 | grep -v amdgpu.ids | tee -a $out/probe_new.txt | tail -3 | cut -c1-400
done
cd tools/_abl/r04head
timeout 600 python tools/probe/replay_stress.py --what acdc --mode replay --load 1 --runs 40 --deep 1 --show 3 --showt 4 --dump $out/dump.pt --tag old_dump 2>&1 | grep -v "^     got\|^     ref" | tee $out/e1.txt | tail -30 | cut -c1-600
