#!/bin/bash
# round 5, GPU session 21: the head's backward through the norm (bcp_pw16_bwd_norm_bwd) -- parity on the device, the -m gpu suite, interleaved
# A/B against the round-4 chain (BCP_HEAD_BWD_FUSED=0: pw16_bwd_norm + norm_bwd) on LA and pancreas, and the new kernels' durations
out=$PWD/gpurun_out/r05_s21; mkdir -p $out
( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
for rep in 1 2 3; do for m in 0 1; do for w in la pancreas; do
  BCP_HEAD_BWD_FUSED=$m python bench.py --workload $w --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w head_bwd_fused=$m', d['value'], d['ms_per_step'])" | tee -a $out/head_ab.txt
done; done; done
R=$PWD; cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  BCP_HEAD_BWD_FUSED=$m rocprofv3 --kernel-trace --stats -d /tmp/ks$m -o ev --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 10 --warmup 2 > /tmp/ks$m.log 2>&1
  f=$(find /tmp/ks$m -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_fused$m.csv
  grep -E "k_pw16_bwd|k_col_partial<1, false>|k_norm_bwd_apply|k_norm_bwd_finalize" $out/kernel_stats_fused$m.csv | cut -c1-60,100-260
done
