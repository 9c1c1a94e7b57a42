#!/bin/bash
# round 6, GPU session 26: fused apply passes, third form (one-round-trip prologue) (runner workgroups of their own, first trip's loads in front of the prologue): kernel checks + A/B
out=$PWD/gpurun_out/r06_s27; mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "norm" ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
tools/ab_opts.sh "" "--opt norm_fuse_fin=0" 2>&1 | tee $out/ab.txt
