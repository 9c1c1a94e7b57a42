#!/bin/bash
# round 6, GPU session 32: small layers' weight gradients cross to the side stream in batches (WGRAD_DEFER 1 / 2 / 3 / 5): suite, then A/B
out=$PWD/gpurun_out/r06_s32; mkdir -p $out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
tools/ab_opts.sh "--opt wgrad_defer=1" "--opt wgrad_defer=2" "--opt wgrad_defer=3" "--opt wgrad_defer=5" 2>&1 | tee $out/ab.txt
