#!/bin/bash
# round 6, GPU session 40: the top level's weight gradients held back until the main chain is at the deep levels (WGRAD_HOLD n:k): A/B
out=$PWD/gpurun_out/r06_s40; mkdir -p $out
WL="la" tools/ab_opts.sh "" "--opt wgrad_hold=1:6" "--opt wgrad_hold=1:10" "--opt wgrad_hold=2:10" "--opt wgrad_hold=5:10" "--opt wgrad_hold=5:14" 2>&1 | tee $out/ab.txt
