#!/bin/bash
# round 6, GPU session 56: k_cc_border with the wave-level pair exchange (cc_border_dedupe): checks, the chain alone, the step
out=$PWD/gpurun_out/r06_s56; mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "cc or plabel" ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
for o in cc_border_dedupe=0 cc_border_dedupe=1; do echo "== $o"; timeout 300 python tools/cc_probe.py $o 2>&1 | tail -4; done | tee $out/probe.txt
tools/ab_opts.sh "--opt cc_border_dedupe=0" "" 2>&1 | tee $out/ab.txt
