#!/bin/bash
# round 6, GPU session 46: the suite with check_norm_fuse_fin (fused apply passes against the finalize-launch chain) on the final tree
out=$PWD/gpurun_out/r06_s46; mkdir -p $out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
