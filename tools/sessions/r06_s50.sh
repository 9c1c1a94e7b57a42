#!/bin/bash
# round 6, GPU session 50: the recomputing transposed conv + norm in the TEACHER's forward only (no backward pass there): top level (2) / every level (1) against off
out=$PWD/gpurun_out/r06_s50; mkdir -p $out
WL="la pancreas" tools/ab_opts.sh "" "--opt up_recompute=2" "--opt up_recompute=1" 2>&1 | tee $out/ab.txt
