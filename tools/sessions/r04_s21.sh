#!/bin/bash
# round 4, GPU session 21: U-Net skip written straight into the concat buffer (bcp_norm_fwd out_ld, bcp_maxpool2d ldx) + pool backward joining the
# skip gradient: kernel + network checks, then ACDC with and without (host switch skip_in_concat)
out=$PWD/gpurun_out/r04_s21; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "pool2d or norm" 2>&1 | tail -3 | tee $out/pytest_k.txt
timeout 900 python -m pytest tests/test_gpu_unet.py -m gpu -q 2>&1 | tail -5 | tee $out/pytest_n.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 80 --warmup 5 --workload acdc "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3 4; do
  echo "rep $rep acdc direct $(ab) copy $(ab --opt skip_in_concat=0)"
done 2>&1 | tee $out/ab.txt
