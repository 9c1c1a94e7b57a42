#!/bin/bash
# round 6, GPU session 34: slab-contiguous weight-gradient slab sums (k_wgrad_reduce_flat): kernel checks + A/B
out=$PWD/gpurun_out/r06_s34; mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "wgrad or conv3" ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
tools/ab_opts.sh "" "--opt wgrad_reduce_flat=0" 2>&1 | tee $out/ab.txt
