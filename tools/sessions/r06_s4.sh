#!/bin/bash
# round 6, GPU session 4: one step of each workload as an ordered kernel list (rocprofv3 --kernel-trace; queues, gaps, workgroup counts)
out=$PWD/gpurun_out/r06_s4; mkdir -p $out
R=$PWD; cd /tmp && export TMPDIR=/tmp
for w in la acdc pancreas; do
  rm -rf /tmp/ev_$w
  rocprofv3 --kernel-trace -d /tmp/ev_$w -o run --output-format csv -- python $R/bench.py --workload $w --no-cpu-baseline --no-extra --no-roofline --steps 8 --warmup 4 > /tmp/ev_$w.log 2>&1
  f=$(find /tmp/ev_$w -name "*kernel_trace.csv" | head -1)
  python $R/tools/step_sequence.py $f > $out/${w}_step_sequence.txt 2>&1
  python $R/tools/timeline_attrib.py $f --steps 4 --json $out/${w}_timeline.json > $out/${w}_timeline.txt 2>&1
  head -3 $out/${w}_step_sequence.txt; head -12 $out/${w}_timeline.txt
done
