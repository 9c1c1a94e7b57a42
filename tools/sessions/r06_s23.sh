#!/bin/bash
# round 6, GPU session 23: |max| of a dgrad pack from its forward twin's header (the weights are not read a second time): suite + bench
out=$PWD/gpurun_out/r06_s23; mkdir -p $out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
for r in 1 2 3; do for w in la acdc pancreas; do python bench.py --workload $w --no-extra --no-cpu-baseline --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'])"; done; done | tee $out/bench.txt
