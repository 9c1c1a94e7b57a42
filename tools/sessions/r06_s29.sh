#!/bin/bash
# round 6, GPU session 29: statistics pass with predicated U-row trips (no row-by-row / slab-by-slab tail): kernel checks + A/B against the fused-apply switch
out=$PWD/gpurun_out/r06_s29; mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "norm" ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
tools/ab_opts.sh "" "--opt norm_fuse_fin=0" 2>&1 | tee $out/ab.txt
