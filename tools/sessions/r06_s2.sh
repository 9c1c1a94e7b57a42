#!/bin/bash
# round 6, GPU session 2: deep-level weight gradient -- slab launches of rounds 2-5 against few groups / direct write
out=$PWD/gpurun_out/r06_s2; mkdir -p $out
timeout 600 python tools/probe/wgrad_deep_probe.py 2>&1 | grep -E "RESULT|Error|error" | tee $out/wgrad_deep.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv3" 2>&1 | tail -3 | tee $out/pytest_conv3.txt
