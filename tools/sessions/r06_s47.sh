#!/bin/bash
# round 6, GPU session 47: option sweep on the final tree (LA; have the optima moved with the fused apply passes / batched forks?)
out=$PWD/gpurun_out/r06_s47; mkdir -p $out
WL="la" tools/ab_opts.sh "" "--opt k2_stats=1" "--opt k2_stats=0" "--opt conv3_xcd=3" "--opt conv3_xcd=27" "--opt wgrad_b6_deep_slots=512" "--opt fuse_bwd_stats=0" "--opt graphs=2" "--opt teacher_prio=-1" "--opt wgrad_prio=-1" "--opt cc_select_blocks=1024" "--opt wgrad_defer=4" "--opt pack_partial=0" 2>&1 | tee $out/ab.txt
