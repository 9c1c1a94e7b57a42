#!/bin/bash
# round 6, GPU session 17: the largest-CC chain (on the step's critical path behind the teacher's forward): per-kernel times, root-selection grid
out=$PWD/gpurun_out/r06_s17; mkdir -p $out
R=$PWD; cd /tmp && export TMPDIR=/tmp
for v in cc_select_blocks=256 cc_select_blocks=1024 cc_select_blocks=2048; do
  rm -rf /tmp/cc_$v; rocprofv3 --kernel-trace --stats -d /tmp/cc_$v -o cc --output-format csv -- python $R/tools/cc_probe.py $v > /tmp/cc_$v.log 2>&1
  echo "== $v"; grep "us per" /tmp/cc_$v.log; f=$(find /tmp/cc_$v -name "*kernel_stats.csv" | head -1); grep "k_cc\|fillBuffer" $f | cut -d, -f1-4 | cut -c1-140
done 2>&1 | tee $out/cc.txt
cd $R
WL="la acdc pancreas" tools/ab_opts.sh "" "--opt cc_select_blocks=256" 2>&1 | tee $out/ab.txt
