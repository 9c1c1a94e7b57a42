#!/bin/bash
# round 6, GPU session 18: upper bound of what skipping unused weight-pack sections can buy -- the LA step reads only the fp16 planes
# (every conv3 launch of the step is a PL = 2 instance: profiles/r06_t1_kernel_stats.csv), so packing ONLY those is valid there
out=$PWD/gpurun_out/r06_s18; mkdir -p $out
WL="la" tools/ab_opts.sh "" "--opt pack_sections=4" 2>&1 | tee $out/ab.txt
