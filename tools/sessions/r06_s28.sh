#!/bin/bash
# round 6, GPU session 28: partial rows per group in front of the fused apply passes (norm_fin_rows 128 / 64 / 32 / 16): A/B
out=$PWD/gpurun_out/r06_s28; mkdir -p $out
WL="la pancreas" tools/ab_opts.sh "--opt norm_fin_rows=128" "--opt norm_fin_rows=64" "--opt norm_fin_rows=32" "--opt norm_fin_rows=16" 2>&1 | tee $out/ab.txt
