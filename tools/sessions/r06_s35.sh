#!/bin/bash
# round 6, GPU session 35: the smallest levels' norm in ONE launch (k_norm_own_fwd / _bwd, option norm_own), chunk width 8 / 16 / 32 by the 32-workgroup bound: kernel checks + A/B
out=$PWD/gpurun_out/r06_s35; mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "norm" ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
tools/ab_opts.sh "" "--opt norm_own=0" 2>&1 | tee $out/ab.txt
