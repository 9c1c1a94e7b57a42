#!/bin/bash
# round 6, GPU session 42: predicated U-row trips in the SLAB-SUMMING statistics pass only (the deep levels' instances): build against build, kernel checks
out=$PWD/gpurun_out/r06_s42; mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "norm" ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
for w in la pancreas acdc; do echo "== $w"; tools/ab_libs.sh tools/_abl/libbcp_prev.so tools/_abl/libbcp_new.so --workload $w --no-extra --no-roofline; done 2>&1 | tee $out/ab.txt
