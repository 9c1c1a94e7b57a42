#!/bin/bash
# round 6, GPU session 63: the bilinear x2 backward with six candidate rows / columns per input pixel instead of eight: checks (incl. the upsample-beside-convs load tests), alone, ACDC step
out=$PWD/gpurun_out/r06_s63; mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -m gpu -x -q -k "pool2d or upsample or full_size_properties_acdc or hazard" ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
for o in bilinear_bwd_window=8 bilinear_bwd_window=6; do echo "== $o"; timeout 200 python tools/probe/pool_probe.py $o 2>&1 | grep "RESULT bilinear"; done | tee $out/probe.txt
WL="acdc" tools/ab_opts.sh "--opt bilinear_bwd_window=8" "" 2>&1 | tee $out/ab.txt
