#!/bin/bash
# round 6, GPU session 52: the apply passes' grid shape in the step (probe r06_s51: one float4 per thread beats the 2048-workgroup grid-stride form by ~20 % alone)
out=$PWD/gpurun_out/r06_s52; mkdir -p $out
WL="la" tools/ab_opts.sh "" "--opt norm_apply_cap=8192" "--opt norm_apply_cap=65536" "--opt norm_apply_cap=65536 --opt norm_apply_vec=2" "--opt norm_apply_cap=65536 --opt norm_apply_vec=1" 2>&1 | tee $out/ab.txt
