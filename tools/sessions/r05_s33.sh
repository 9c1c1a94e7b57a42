#!/bin/bash
# round 5, GPU session 33: both mix_loss calls of a step as one launch pair (bcp_mixloss_pair_fwd / _bwd) -- the -m gpu suite (its kernel
# check asserts bit-identity with the two calls), interleaved A/B against the two-call path (BCP_MIXLOSS_PAIR=0) on the three workloads
out=$PWD/gpurun_out/r05_s33; mkdir -p $out
( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
for rep in 1 2 3; do for m in 0 1; do for w in la acdc pancreas; do
  BCP_MIXLOSS_PAIR=$m python bench.py --workload $w --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w mixloss_pair=$m', d['value'], d['ms_per_step'])" | tee -a $out/pair_ab.txt
done; done; done
