#!/bin/bash
# round 6, GPU session 14: weight packs inside the recorded training forward on a side stream (HipNet.PACK_IN_PLAN) + the early gradient memset:
# the device suite, then the step A/B (--opt pack_in_plan=0)
out=$PWD/gpurun_out/r06_s14; mkdir -p $out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
tools/ab_opts.sh "" "--opt pack_in_plan=0" 2>&1 | tee $out/ab.txt
