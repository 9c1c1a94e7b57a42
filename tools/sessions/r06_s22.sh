#!/bin/bash
# round 6, GPU session 22: accumulate operand of k_gemm_nn requested in front of the K loop (down conv dgrad += skip gradient)
out=$PWD/gpurun_out/r06_s22; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "k2" 2>&1 | tail -2
WL="la pancreas" tools/ab_opts.sh "" "--opt gemm_acc_early=0" 2>&1 | tee $out/ab.txt
