#!/bin/bash
# round 4, GPU session 35: dispatch switches chosen on the three-plane bf16 kernels (rounds 2-3), re-measured on the two-plane fp16 ones
out=$PWD/gpurun_out/r04_s35; mkdir -p $out
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 50 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do
  echo "rep $rep LA default $(ab) direct=2 $(ab --opt conv3_b6_direct=2) direct=0 $(ab --opt conv3_b6_direct=0) flat=0 $(ab --opt conv3_b6_flat=0) pipe=0 $(ab --opt conv3_b6_pipe=0) wg_levels=11 $(ab --opt wgrad_b6_levels=11) wslots=768 $(ab --opt wgrad_b6_slots=768) default $(ab)"
  echo "rep $rep ACDC default $(ab --workload acdc) cfg2d=0 $(ab --workload acdc --opt conv3_b6_cfg2d=0) cfg2d=2 $(ab --workload acdc --opt conv3_b6_cfg2d=2) cin16max=16 $(ab --workload acdc --opt conv3_b6_cin16max=16) direct=0 $(ab --workload acdc --opt conv3_b6_direct=0) direct=2 $(ab --workload acdc --opt conv3_b6_direct=2) w22=0 $(ab --workload acdc --opt conv3_b6_w22=0) default $(ab --workload acdc)"
done 2>&1 | tee $out/ab.txt
