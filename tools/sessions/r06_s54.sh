#!/bin/bash
# round 6, GPU session 54: the PLAIN k_gemm_nn launches as row-block walks (gemm_walk = target workgroups): check, the step
out=$PWD/gpurun_out/r06_s54; mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "k2 or up_norm or gemm_walk" ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
for o in "gemm_walk=0" "gemm_walk=1024" "gemm_walk=2048"; do timeout 300 python tools/probe/k2_stats_probe.py $o 2>&1 | grep -E "OPTIONS|RESULT"; done | tee $out/probe.txt
WL="la pancreas" tools/ab_opts.sh "" "--opt gemm_walk=768" "--opt gemm_walk=1280" "--opt gemm_walk=2048" 2>&1 | tee $out/ab.txt
