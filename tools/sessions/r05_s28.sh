#!/bin/bash
# round 5, GPU session 28: interleaved A/B on one box of the da-row hoist in k_conv3_c1's backward passes -- main tree (hoisted) against
# tools/_abl/nohoist (the commit before: one request per m-tile), LA / ACDC / pancreas
out=$PWD/gpurun_out/r05_s28; mkdir -p $out; R=$PWD
for rep in 1 2 3; do for v in nohoist main; do for w in la acdc pancreas; do
  if [ $v == main ]; then cd $R; else cd $R/tools/_abl/$v; fi
  python bench.py --workload $w --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w $v', d['value'], d['ms_per_step'])" | tee -a $out/hoist_ab.txt
done; done; done
