#!/bin/bash
# round 5, GPU session 35: volatile_io (no copies in and out of the recorded passes; what bench.py and the training scripts set) -- the -m gpu
# suite with its new check, interleaved A/B against defensive copies (BCP_VOLATILE_IO=0) on the three workloads, the default bench command
out=$PWD/gpurun_out/r05_s35; mkdir -p $out
( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
for rep in 1 2 3; do for m in 0 1; do for w in la acdc pancreas; do
  BCP_VOLATILE_IO=$m python bench.py --workload $w --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w volatile_io=$m', d['value'], d['ms_per_step'])" | tee -a $out/vol_ab.txt
done; done; done
( time python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | grep real | tee $out/bench_time.txt; cut -c1-300 $out/bench_default.json
