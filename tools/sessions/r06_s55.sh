#!/bin/bash
# round 6, GPU session 55: the teacher's tail -- pseudo-label inside the first largest-CC kernel (bcp_plabel_cc_largest), the selection inside the size count (k_cc_count_select): checks, the chain alone, the step
out=$PWD/gpurun_out/r06_s55; mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "cc or plabel" ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
timeout 300 python tools/cc_probe.py 2>&1 | tail -12 | tee $out/probe.txt
tools/ab_opts.sh "--opt plabel_cc_fused=0 --opt cc_fuse_select=0" "--opt plabel_cc_fused=0" "" 2>&1 | tee $out/ab.txt
