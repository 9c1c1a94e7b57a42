#!/bin/bash
# round 4, GPU session 2: slab sum folded into the norm statistics pass (norm_slabs): parity on the device, step A/B on one box
out=$PWD/gpurun_out/r04_s2; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vnet.py tests/test_gpu_unet.py -q -x 2>&1 | tail -3 | tee $out/pytest.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep la slabs0 $(ab --opt norm_slabs=0) slabs1 $(ab) | panc slabs0 $(ab --workload pancreas --opt norm_slabs=0) slabs1 $(ab --workload pancreas) | acdc slabs0 $(ab --workload acdc --opt norm_slabs=0) slabs1 $(ab --workload acdc)"
done 2>&1 | tee $out/ab.txt
