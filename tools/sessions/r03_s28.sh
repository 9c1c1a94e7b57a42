#!/bin/bash
# round 3, GPU session 28: where the ACDC and pancreas steps' time goes (kernel timeline attribution)
out=$PWD/gpurun_out/s28; mkdir -p $out; R=$PWD
cd /tmp && export TMPDIR=/tmp
for wl in acdc pancreas; do
  rm -rf /tmp/ev_$wl
  rocprofv3 --kernel-trace -d /tmp/ev_$wl -o run --output-format csv -- python $R/bench.py --workload $wl --no-cpu-baseline --no-extra --no-roofline --steps 8 --warmup 4 > /tmp/ev_$wl.log 2>&1
  python $R/tools/timeline_attrib.py $(find /tmp/ev_$wl -name "*kernel_trace.csv" | head -1) --steps 4 --json $out/timeline_$wl.json > $out/timeline_$wl.txt
  head -28 $out/timeline_$wl.txt | cut -c1-140
done
