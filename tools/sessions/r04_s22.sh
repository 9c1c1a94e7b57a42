#!/bin/bash
# round 4, GPU session 22: the step's total loss summed by the second mix_loss launch + cached unit gradient (train_step.STEP_TOTAL), on top of
# the U-Net skip-in-concat change: kernel / network suites, then each workload with and without
out=$PWD/gpurun_out/r04_s22; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_vnet.py -m gpu -q 2>&1 | tail -5 | tee $out/pytest.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 80 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep acdc total $(ab --workload acdc) torch $(ab --workload acdc --opt step_total=0) | la total $(ab) torch $(ab --opt step_total=0) | panc total $(ab --workload pancreas) torch $(ab --workload pancreas --opt step_total=0)"
done 2>&1 | tee $out/ab.txt
