#!/bin/bash
# round 6, GPU session 60: k_cc_local labelling quads from the logits (16-byte loads, one 4-byte store of the map): checks, the chain alone, the step against the two-call form
out=$PWD/gpurun_out/r06_s60; mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "cc or plabel" ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
timeout 300 python tools/cc_probe.py 2>&1 | tail -4 | tee $out/probe.txt
tools/ab_opts.sh "--opt plabel_cc_fused=0" "" 2>&1 | tee $out/ab.txt
