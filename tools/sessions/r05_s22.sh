#!/bin/bash
# round 5, GPU session 22: is the fuse_head test's 5.5e-5 (two steps, first-layer weights) a defect of bcp_pw16_bwd_norm_bwd or two steps'
# amplification of last-bit differences in dy?  Weight differences after 1, 2, 3 steps for both head-backward paths, and the two paths at full size
out=$PWD/gpurun_out/r05_s22; mkdir -p $out
python tools/probe/head_fusion_diff.py la 3 2>&1 | grep -v amdgpu.ids | tee $out/la.txt
python tools/probe/head_fusion_diff.py pancreas 3 2>&1 | grep -v amdgpu.ids | grep -v "^full size" | tee $out/pancreas.txt
