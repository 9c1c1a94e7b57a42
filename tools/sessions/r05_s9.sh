#!/bin/bash
# round 5, GPU session 9: vD = the round-4 upsample source compiled WITHOUT SLP vectorisation (no v_pk_* instruction in the kernel) against vB
out=$PWD/gpurun_out/r05_s9; mkdir -p $out
R=$PWD
for v in vB vD vB vD; do
  cd $R/tools/_abl/$v
  timeout 300 python tools/probe/bilinear_race_probe.py rounds=60 workers=1 conv=1 load=1 gemm=1 2>&1 | grep RESULT | sed "s/^/$v: /" | tee -a $out/probe.txt
done
