#!/bin/bash
# round 5, GPU session 6: the upsample kernel alone never deviates (s5: 0 of 25 k launches).  The in-step deviation needs the bf16-pipe convs
# (conv3_b6=0: 0 of 100 runs).  Reproducer: upsample [+ the GEMM in front of it] on two streams BESIDE conv launches on other streams.
out=$PWD/gpurun_out/r05_s6; mkdir -p $out
R=$PWD
cd tools/_abl/r04head
for cfg in "workers=2 conv=2 load=1" "workers=2 conv=2 load=0" "workers=2 conv=2 load=1 gemm=1" "workers=1 conv=1 load=1 gemm=1" "workers=2 conv=3 load=1 C=64 H=8 gemm=1" "workers=2 conv=0 load=1 gemm=1"; do
  timeout 300 python tools/probe/bilinear_race_probe.py rounds=30 $cfg 2>&1 | grep -v amdgpu.ids | tee -a $out/probe_old.txt | tail -7 | cut -c1-420
done
cd $R
for cfg in "workers=2 conv=2 load=1 gemm=1" "workers=2 conv=3 load=1 C=64 H=8 gemm=1"; do
  timeout 300 python tools/probe/bilinear_race_probe.py rounds=30 $cfg 2>&1 | grep -v amdgpu.ids | tee -a $out/probe_new.txt | tail -3 | cut -c1-420
done
