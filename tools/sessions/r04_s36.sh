#!/bin/bash
# round 4, GPU session 36: weight fragments of k_c3d requested TWO tap pairs ahead (a third register set, -DBCP_C3D_BD=2; not the persistent
# 16-channel instance) against one pair ahead: conv checks and network suites on the BD = 2 build, then the three workloads on both builds
out=$PWD/gpurun_out/r04_s36; mkdir -p $out
cp bcp_amd/csrc/libbcp_hip.so /tmp/keep.so; cp tools/_abl/bd2.so bcp_amd/csrc/libbcp_hip.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv3" 2>&1 | tail -1 | tee $out/pytest_k.txt
timeout 900 python -m pytest tests/test_gpu_vnet.py tests/test_gpu_unet.py -m gpu -q 2>&1 | tail -1 | tee $out/pytest_n.txt
cp /tmp/keep.so bcp_amd/csrc/libbcp_hip.so
bash tools/ab_libs.sh tools/_abl/bd1.so tools/_abl/bd2.so --no-extra --no-roofline 2>&1 | tee $out/ab_la.txt
bash tools/ab_libs.sh tools/_abl/bd1.so tools/_abl/bd2.so --no-extra --no-roofline --workload acdc 2>&1 | tee $out/ab_acdc.txt
bash tools/ab_libs.sh tools/_abl/bd1.so tools/_abl/bd2.so --no-extra --no-roofline --workload pancreas 2>&1 | tee $out/ab_panc.txt
