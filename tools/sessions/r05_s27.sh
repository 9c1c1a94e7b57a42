#!/bin/bash
# round 5, GPU session 27: k_conv3_c1's backward passes with the tile's four da rows requested together at the tile top (session 26:
# <3,4,4,16,3> 52.1 us, <..,5> 65.4 us, 2-D <..,3> 36.5 us, <..,5> 31.9 us) -- the -m gpu suite, kernel durations, bench lines
out=$PWD/gpurun_out/r05_s27; mkdir -p $out
( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
for w in la acdc; do
  rocprofv3 --kernel-trace --stats -d /tmp/ks$w -o ev --output-format csv -- python $R/bench.py --workload $w --no-cpu-baseline --no-extra --no-roofline --steps 10 --warmup 2 > /tmp/ks$w.log 2>&1
  f=$(find /tmp/ks$w -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_$w.csv
  grep -E "k_conv3_c1" $out/kernel_stats_$w.csv | cut -c1-60,150-260
done
cd $R; for i in 1 2; do for w in la acdc pancreas; do python bench.py --workload $w --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'])" | tee -a $out/bench.txt; done; done
