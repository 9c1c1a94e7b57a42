#!/bin/bash
# round 5, GPU session 8: the round-4 upsample code with ONE change -- a full s_waitcnt vmcnt(0) behind its four loads (vC) -- against the unchanged
# code (vB) in the reproducer (upsample + GEMM on one stream beside the bf16-pipe convs + a copy / GEMM load)
out=$PWD/gpurun_out/r05_s8; mkdir -p $out
R=$PWD
for v in vB vC vB vC; do
  cd $R/tools/_abl/$v
  timeout 300 python tools/probe/bilinear_race_probe.py rounds=60 workers=1 conv=1 load=1 gemm=1 2>&1 | grep RESULT | sed "s/^/$v: /" | tee -a $out/probe.txt
done
cd $R
timeout 300 python tools/probe/bilinear_race_probe.py rounds=60 workers=1 conv=1 load=1 gemm=1 2>&1 | grep RESULT | sed "s/^/new: /" | tee -a $out/probe.txt
