#!/bin/bash
# round 5, GPU session 26: the first layer's backward with its weight gradient folded in (bcp_conv3_c1_norm_bwd_wgrad, kernel EPI 5) --
# the -m gpu suite, interleaved A/B against conv3_c1_norm_bwd + conv3_c1_wgrad (BCP_C1_BWD_FUSED=0) on all three workloads, kernel durations
out=$PWD/gpurun_out/r05_s26; mkdir -p $out
( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
for rep in 1 2 3; do for m in 0 1; do for w in la acdc pancreas; do
  BCP_C1_BWD_FUSED=$m python bench.py --workload $w --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w c1_bwd_fused=$m', d['value'], d['ms_per_step'])" | tee -a $out/c1_ab.txt
done; done; done
R=$PWD; cd /tmp && export TMPDIR=/tmp
for w in la acdc; do
  rocprofv3 --kernel-trace --stats -d /tmp/ks$w -o ev --output-format csv -- python $R/bench.py --workload $w --no-cpu-baseline --no-extra --no-roofline --steps 10 --warmup 2 > /tmp/ks$w.log 2>&1
  f=$(find /tmp/ks$w -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_$w.csv
  grep -E "k_conv3_c1" $out/kernel_stats_$w.csv | cut -c1-60,150-260
done
