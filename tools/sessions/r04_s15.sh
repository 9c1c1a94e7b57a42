#!/bin/bash
# round 4, GPU session 15: XCD-aware orders re-measured now that the convs issue half the MFMAs (conv3_xcd bits: 2 = persistent 16-channel
# kernel walks contiguous eighths, 8 = one weight stream per XCD at the 128-channel level)
out=$PWD/gpurun_out/r04_s15; mkdir -p $out
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep la x1 $(ab) x3 $(ab --opt conv3_xcd=3) x9 $(ab --opt conv3_xcd=9) x11 $(ab --opt conv3_xcd=11) | panc x1 $(ab --workload pancreas) x9 $(ab --workload pancreas --opt conv3_xcd=9)"
done 2>&1 | tee $out/ab.txt
python tools/bench_conv.py --levels 16,128 --ops fwd_stats,dgrad --variants "x1:;x3:conv3_xcd=3;x9:conv3_xcd=9" --json $out/c.json 2>&1 | grep -v "amdgpu\|fp32" | tee $out/c.txt
