#!/bin/bash
# round 6, GPU session 31: kernel trace of the LA step after the fused apply passes: one step as an ordered list + kernel statistics
out=$PWD/gpurun_out/r06_s31; mkdir -p $out
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev2 -o run --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 8 --warmup 4 > /tmp/ev2.log 2>&1
cd $R
python tools/step_sequence.py $(find /tmp/ev2 -name "*kernel_trace.csv" | head -1) > $out/step_sequence.txt 2>&1
python tools/timeline_attrib.py $(find /tmp/ev2 -name "*kernel_trace.csv" | head -1) --steps 4 --json $out/timeline.json > $out/timeline.txt
cp $(find /tmp/ev2 -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
head -5 $out/timeline.txt
