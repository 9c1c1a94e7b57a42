#!/bin/bash
# round 4, GPU session 17: pack launch shape; kernel stats
out=$PWD/gpurun_out/r04_s17; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "pack or conv3_f16" 2>&1 | tail -2
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "la $(ab) $(ab) $(ab) | acdc $(ab --workload acdc) $(ab --workload acdc) | panc $(ab --workload pancreas) $(ab --workload pancreas)" | tee $out/ab.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev -o ev --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 10 --warmup 2 > /tmp/ev.log 2>&1
f=$(find /tmp/ev -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv
cd $R; grep -E "pack|wamax" $out/kernel_stats.csv | cut -c1-50,60-140
