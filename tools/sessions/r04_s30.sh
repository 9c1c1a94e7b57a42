#!/bin/bash
# round 4, GPU session 30: the 2-D 32-channel-slab k_c3d as a persistent kernel (one workgroup per CU: 251 + 44 registers) against one tile per
# workgroup (two per CU): conv checks with it forced from 1 tile up, then the ACDC step with thresholds 1024 (the 3072-tile dgrad only),
# 512 (also the 768-tile launches) and off
out=$PWD/gpurun_out/r04_s30; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv3" 2>&1 | tail -3 | tee $out/pytest_k.txt
timeout 900 python -m pytest tests/test_gpu_unet.py -m gpu -q 2>&1 | tail -3 | tee $out/pytest_n.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 80 --warmup 5 --workload acdc "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep acdc per>=1024 $(ab) per>=512 $(ab --opt conv3_per2d=512) off $(ab --opt conv3_per2d=0)"
done 2>&1 | tee $out/ab.txt
