#!/bin/bash
# round 6, GPU session 16: backward statistics of the norm layers in front of the k2s2 / transposed convs from the dgrad GEMM's epilogue
# (k_gemm_nn<.., 2>): the device suite, then the step A/B over the three modes of k2_bwd_stats
out=$PWD/gpurun_out/r06_s16; mkdir -p $out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
WL="la pancreas" tools/ab_opts.sh "" "--opt k2_bwd_stats=0" "--opt k2_bwd_stats=1" 2>&1 | tee $out/ab.txt
