#!/bin/bash
# round 6, GPU session 38: suite on the current tree; kernel traces of the ACDC and pancreas steps as ordered lists
out=$PWD/gpurun_out/r06_s38; mkdir -p $out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
for w in acdc pancreas; do
  rm -rf /tmp/ev_$w
  rocprofv3 --kernel-trace -d /tmp/ev_$w -o run --output-format csv -- python $R/bench.py --workload $w --no-cpu-baseline --no-extra --no-roofline --steps 8 --warmup 4 > /tmp/ev_$w.log 2>&1
  python $R/tools/step_sequence.py $(find /tmp/ev_$w -name "*kernel_trace.csv" | head -1) > $out/step_sequence_$w.txt 2>&1
done
cd $R; head -3 $out/step_sequence_acdc.txt
