#!/bin/bash
# round 6, GPU session 9: norm statistics of the k2s2 / transposed conv outputs from the GEMM epilogue (k_gemm_nn<.., STATS>): the device
# suite, then the step A/B against the norm's own statistics passes (--opt k2_stats=0)
out=$PWD/gpurun_out/r06_s9; mkdir -p $out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
tools/ab_opts.sh "" "--opt k2_stats=0" 2>&1 | tee $out/ab.txt
