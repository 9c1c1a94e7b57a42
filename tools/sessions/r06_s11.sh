#!/bin/bash
# round 6, GPU session 11: step A/B of the statistics-in-the-GEMM-epilogue variant (fp32 lane partials) against the norm's own passes
out=$PWD/gpurun_out/r06_s11; mkdir -p $out
WL="la pancreas" tools/ab_opts.sh "" "--opt k2_stats=0" "--opt k2_stats=2" 2>&1 | tee $out/ab.txt
