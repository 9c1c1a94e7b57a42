#!/bin/bash
# round 4, GPU session 34: split-K raw slabs + norm-from-slabs for the U-Net's MID levels (128 ch @32x32, 64 ch @64x64: 192-384 workgroups per
# launch on 256 CUs, 25-30 us each, 29 launches per step) -- limits raised through the measurement switches raw_out_max / norm_slabs_rows
out=$PWD/gpurun_out/r04_s34; mkdir -p $out
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 --workload acdc "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
timeout 600 python -m pytest tests/test_gpu_unet.py -m gpu -q -k "smooth or golden" 2>&1 | tail -1
for rep in 1 2 3; do
  echo "rep $rep acdc default $(ab) 128ch-level $(ab --opt raw_out_max=2097152 --opt norm_slabs_rows=8192) +64ch-level $(ab --opt raw_out_max=4194304 --opt norm_slabs_rows=32768)"
done 2>&1 | tee $out/ab.txt
