#!/bin/bash
# round 5, GPU session 18: gemm.hip with / without the SLP vectoriser (interleaved A/B on one box, three workloads) + the GEMM kernel checks
out=$PWD/gpurun_out/r05_s18; mkdir -p $out
for w in la pancreas acdc; do echo "== $w"; bash tools/ab_libs.sh tools/_abl/gemm_slp.so tools/_abl/gemm_noslp.so --no-extra --no-roofline --workload $w; done 2>&1 | tee $out/ab.txt
cp tools/_abl/gemm_noslp.so bcp_amd/csrc/libbcp_hip.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -2 | tee $out/pytest_kernels.txt
