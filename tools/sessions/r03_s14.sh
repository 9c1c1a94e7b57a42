#!/bin/bash
# round 3, GPU session 14: timeline + kernel stats of the current tree (LA), to pick the next kernel.
out=$PWD/gpurun_out/s19; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev -o ev --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 > /tmp/ev.log 2>&1
f=$(find /tmp/ev -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv
rocprofv3 --kernel-trace -d /tmp/ev2 -o run --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 8 --warmup 4 > /tmp/ev2.log 2>&1
cd $R
python tools/timeline_attrib.py $(find /tmp/ev2 -name "*kernel_trace.csv" | head -1) --steps 4 --json $out/timeline.json > $out/timeline.txt; head -70 $out/timeline.txt
