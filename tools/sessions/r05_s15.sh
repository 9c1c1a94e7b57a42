#!/bin/bash
# round 5, GPU session 15: single-channel mix kernel A/B on one box (interleaved), and a kernel trace of the LA step for the critical path
out=$PWD/gpurun_out/r05_s15; mkdir -p $out
for rep in 1 2 3; do for m in 0 1; do for w in la acdc pancreas; do
  python bench.py --workload $w --opt mix_c1=$m --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w mix_c1=$m', d['value'], d['ms_per_step'])" | tee -a $out/mix_ab.txt
done; done; done
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/tr -o run --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 8 --warmup 3 > /tmp/tr.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); cd $R
python tools/step_sequence.py $f > $out/la_step_sequence.txt 2>&1; head -5 $out/la_step_sequence.txt
python tools/timeline_attrib.py $f --steps 4 > $out/la_timeline.txt 2>&1; head -4 $out/la_timeline.txt
cd /tmp; rocprofv3 --kernel-trace -d /tmp/tr2 -o run --output-format csv -- python $R/bench.py --workload acdc --no-cpu-baseline --no-extra --no-roofline --steps 8 --warmup 3 > /tmp/tr2.log 2>&1
f=$(find /tmp/tr2 -name "*kernel_trace.csv" | head -1); cd $R
python tools/step_sequence.py $f > $out/acdc_step_sequence.txt 2>&1; head -3 $out/acdc_step_sequence.txt
