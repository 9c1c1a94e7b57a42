#!/bin/bash
# round 6, GPU session 25: apply passes that finalise the statistics themselves (k_norm_apply_fin / k_norm_bwd_apply_fin, option norm_fuse_fin): suite + A/B
out=$PWD/gpurun_out/r06_s25; mkdir -p $out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
tools/ab_opts.sh "" "--opt norm_fuse_fin=0" 2>&1 | tee $out/ab.txt
