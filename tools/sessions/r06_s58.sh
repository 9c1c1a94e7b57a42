#!/bin/bash
# round 6, GPU session 58: mixloss kernels with one exponential per two-channel voxel and one (d, h, w) decomposition per trip: checks, alone, build against build
out=$PWD/gpurun_out/r06_s58; mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "mixloss or diceloss" ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
cp bcp_amd/csrc/libbcp_hip.so /tmp/keep.so
for v in prev new prev new; do cp tools/_abl/libbcp_$v.so bcp_amd/csrc/libbcp_hip.so; echo "== $v"; timeout 200 python tools/probe/mixloss_probe.py 2>&1 | grep RESULT; done | tee $out/probe.txt
cp /tmp/keep.so bcp_amd/csrc/libbcp_hip.so
for w in la acdc pancreas; do echo "== $w"; tools/ab_libs.sh tools/_abl/libbcp_prev.so tools/_abl/libbcp_new.so --workload $w --no-extra --no-roofline; done 2>&1 | tee $out/ab.txt
