#!/bin/bash
# round 5, GPU session 7: micro-reproducer -- which consumer instruction behind a partial s_waitcnt vmcnt reads a loaded VGPR too early?
out=$PWD/gpurun_out/r05_s7; mkdir -p $out
timeout 600 python tools/probe/pkmov_hazard.py rounds=30 conv=2 load=1 2>&1 | grep -v amdgpu.ids | tee $out/pk_conv2_load1.txt
timeout 600 python tools/probe/pkmov_hazard.py rounds=15 conv=0 load=1 2>&1 | grep -v amdgpu.ids | tee $out/pk_conv0_load1.txt
timeout 600 python tools/probe/pkmov_hazard.py rounds=15 conv=2 load=0 2>&1 | grep -v amdgpu.ids | tee $out/pk_conv2_load0.txt
