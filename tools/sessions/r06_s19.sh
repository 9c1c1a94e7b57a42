#!/bin/bash
# round 6, GPU session 19: partial weight packs in front of replays (HipNet.PACK_PARTIAL): the device suite, then the step A/B
out=$PWD/gpurun_out/r06_s19; mkdir -p $out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
tools/ab_opts.sh "" "--opt pack_partial=0" 2>&1 | tee $out/ab.txt
