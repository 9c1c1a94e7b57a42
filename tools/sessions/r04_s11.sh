#!/bin/bash
# round 4, GPU session 11: ACDC kernel breakdown with the fp16 instances; loss reduce with eight loads in flight
out=$PWD/gpurun_out/r04_s11; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "mixloss or diceloss" 2>&1 | tail -2
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev -o ev --output-format csv -- python $R/bench.py --workload acdc --no-cpu-baseline --no-extra --no-roofline --steps 10 --warmup 2 > /tmp/ev.log 2>&1
f=$(find /tmp/ev -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_acdc.csv
f=$(find /tmp/ev -name "*kernel_trace.csv" | head -1)
cd $R; python tools/timeline_attrib.py $f --steps 4 --json $out/timeline_acdc.json > $out/timeline_acdc.txt; head -50 $out/timeline_acdc.txt
python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value']); print(json.dumps(d['roofline'])[:600])
for k in d['kernels'][:14]: print(k['op'], k['shape'], k['avg_us'], k.get('pipe','')[:12], k.get('frac'))"
