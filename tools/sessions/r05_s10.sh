#!/bin/bash
out=$PWD/gpurun_out/r05_s10; mkdir -p $out
timeout 600 python tools/probe/pkmov_hazard.py rounds=60 conv=1 load=1 variants=6,7,0,2 2>&1 | grep -v amdgpu.ids | tee $out/pk.txt
