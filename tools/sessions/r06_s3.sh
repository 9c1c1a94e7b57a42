#!/bin/bash
# round 6, GPU session 3: deep-level weight gradients without partial slabs (k_w6 DIR, 128-voxel tiles, 256 slots) -- the kernel checks on
# the device, then the step: interleaved A/B against rounds 2-5's launches (wgrad_b6_deep=0), and the 64-voxel-tile / wide-slab variants
out=$PWD/gpurun_out/r06_s3; mkdir -p $out
timeout 600 python tools/probe/wgrad_deep_probe.py 2>&1 | grep -E "RESULT|Error|error" > $out/wgrad_deep.txt; head -3 $out/wgrad_deep.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -3 | tee $out/pytest_kernels.txt
tools/ab_opts.sh "" "--opt wgrad_b6_deep=0" "--opt wgrad_b6_deep_tile=0" "--opt wgrad_b6_deep_nt=2" "--opt wgrad_b6_deep_slots=512" 2>&1 | tee $out/ab.txt
