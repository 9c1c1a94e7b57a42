#!/bin/bash
# round 5, GPU session 23: k_pw16_bwd_stats with per-trip fp32 sums (was 43.4 us with element-wise fp64), the revised fuse_head test, LA number
out=$PWD/gpurun_out/r05_s23; mkdir -p $out
( time timeout 600 python -m pytest tests/test_gpu_vnet.py tests/test_gpu_kernels.py -m gpu -x -q -k "head_fused or pw16" ) 2>&1 | grep -E "passed|failed|^real|^FAILED|Error" | tee $out/pytest.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ks -o ev --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 10 --warmup 2 > /tmp/ks.log 2>&1
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv
grep -E "k_pw16_bwd|k_col_partial<1, false>" $out/kernel_stats.csv | cut -c1-60,100-260
cd $R; for i in 1 2; do python bench.py --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-120 | tee -a $out/la.txt; done
