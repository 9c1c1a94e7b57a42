#!/bin/bash
# round 4, GPU session 1: DiceLoss class seam + vectorised mixloss on the device, baseline step, fresh per-step kernel sequence
out=$PWD/gpurun_out/r04_s1; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $out/pytest.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "la $(ab) $(ab) | acdc $(ab --workload acdc) | panc $(ab --workload pancreas)" | tee $out/ab.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/ev2 -o run --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 8 --warmup 4 > /tmp/ev2.log 2>&1
cd $R
f=$(find /tmp/ev2 -name "*kernel_trace.csv" | head -1)
python tools/timeline_attrib.py $f --steps 4 --json $out/timeline.json > $out/timeline.txt
python tools/step_sequence.py $f > $out/step_seq.txt
cd /tmp
rocprofv3 --kernel-trace -d /tmp/ev3 -o run --output-format csv -- python $R/bench.py --workload acdc --no-cpu-baseline --no-extra --no-roofline --steps 8 --warmup 4 > /tmp/ev3.log 2>&1
cd $R
f=$(find /tmp/ev3 -name "*kernel_trace.csv" | head -1)
python tools/timeline_attrib.py $f --steps 4 --json $out/timeline_acdc.json > $out/timeline_acdc.txt
python tools/step_sequence.py $f > $out/step_seq_acdc.txt
head -30 $out/timeline.txt
