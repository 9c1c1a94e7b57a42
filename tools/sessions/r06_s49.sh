#!/bin/bash
# round 6, GPU session 49: option sweep on the final tree, ACDC and pancreas
out=$PWD/gpurun_out/r06_s49; mkdir -p $out
WL="acdc" tools/ab_opts.sh "" "--opt conv3_b6_cfg2d=0" "--opt conv3_b6_cfg2d=2" "--opt conv3_xcd=3" "--opt wgrad_defer=3" "--opt wgrad_defer=4" "--opt conv3_sk_elems=2097152" "--opt splitk=2" "--opt fuse_bwd_stats=0" "--opt inline_dropout=0" "--opt skip_in_concat=0" 2>&1 | grep MEAN | tee $out/ab_acdc.txt
WL="pancreas" tools/ab_opts.sh "" "--opt wgrad_defer=3" "--opt wgrad_defer=4" "--opt k2_stats=1" "--opt conv3_xcd=3" "--opt wgrad_b6_deep_slots=512" 2>&1 | grep MEAN | tee $out/ab_pancreas.txt
