#!/bin/bash
# round 4, GPU session 26: nn.Dropout keep bits evaluated inside the norm kernels (hip_ops.SeedMask, UNet_2d.inline_dropout): kernel + network
# checks, launch-plan bit identity, then the ACDC step with and without
out=$PWD/gpurun_out/r04_s26; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "inline_dropout or norm or c1" 2>&1 | tail -3 | tee $out/pytest_k.txt
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_vnet.py -m gpu -q 2>&1 | tail -5 | tee $out/pytest_n.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 80 --warmup 5 --workload acdc "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3 4; do
  echo "rep $rep acdc inline $(ab) masks $(ab --opt inline_dropout=0)"
done 2>&1 | tee $out/ab.txt
