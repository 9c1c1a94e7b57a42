#!/bin/bash
# round 4, GPU session 16: weight-gradient workgroup count (k_w6: 512 = two per CU; its 16-channel instance has registers for three)
out=$PWD/gpurun_out/r04_s16; mkdir -p $out
python tools/bench_conv.py --levels 16,32,64,128,256 --ops wgrad --variants "s512:;s768:wgrad_b6_slots=768;s1024:wgrad_b6_slots=1024;s256:wgrad_b6_slots=256" --json $out/c.json 2>&1 | grep -v "amdgpu\|fp32" | tee $out/c.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep la s512 $(ab) s768 $(ab --opt wgrad_b6_slots=768) s1024 $(ab --opt wgrad_b6_slots=1024)"
done 2>&1 | tee $out/ab.txt
