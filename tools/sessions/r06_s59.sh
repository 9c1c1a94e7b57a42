#!/bin/bash
# round 6, GPU session 59: the largest-CC size count per tile with an LDS table (one global atomic per tile and global root; cc_count_tile): checks, the chain alone, the step
out=$PWD/gpurun_out/r06_s59; mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "cc or plabel" ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
for o in cc_count_tile=0 cc_count_tile=1; do echo "== $o"; timeout 300 python tools/cc_probe.py $o 2>&1 | tail -4; done | tee $out/probe.txt
tools/ab_opts.sh "--opt cc_count_tile=0" "" 2>&1 | tee $out/ab.txt
