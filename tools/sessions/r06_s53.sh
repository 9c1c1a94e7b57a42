#!/bin/bash
# round 6, GPU session 53: k_gemm_nn's row-block walk with the next block's operands in flight (gemm_pipe) and the walk length (gemm_stat_r): kernel checks, the launch alone, the step
out=$PWD/gpurun_out/r06_s53; mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "k2 or up_norm" ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
for o in "gemm_pipe=0" "gemm_pipe=1" "gemm_pipe=1 gemm_stat_r=4" "gemm_pipe=1 gemm_stat_r=16" "gemm_pipe=0 gemm_stat_r=4"; do timeout 300 python tools/probe/k2_stats_probe.py $o 2>&1 | grep -E "OPTIONS|RESULT"; done | tee $out/probe.txt
WL="la pancreas" tools/ab_opts.sh "--opt gemm_pipe=0" "" "--opt gemm_stat_r=4" "--opt gemm_stat_r=16" 2>&1 | tee $out/ab.txt
