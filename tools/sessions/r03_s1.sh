#!/bin/bash
# round 3, GPU session 1 (run ON the GPU box from the repo root): parity suite, the driver-style bench line, in-step A/B of the
# round-3 switches, per-op counters, the step timeline.  Everything lands in gpurun_out/s1/.
out=$PWD/gpurun_out/s1; mkdir -p $out
R=$PWD
( time python -m pytest tests -m gpu -x -q ) > $out/pytest_gpu.txt 2>&1
tail -5 $out/pytest_gpu.txt
( time python bench.py > $out/bench.json 2> $out/bench.err ); tail -3 $out/bench.err; cut -c1-300 $out/bench.json
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_ms_per_step_empty_queue'])"; }
{
for rep in 1 2; do
  echo "rep $rep all_off        $(ab --opt norm_small=0 --opt fuse_bwd_stats=0 --opt conv3_xcd=0)"
  echo "rep $rep xcd_only       $(ab --opt norm_small=0 --opt fuse_bwd_stats=0)"
  echo "rep $rep small_only     $(ab --opt fuse_bwd_stats=0 --opt conv3_xcd=0)"
  echo "rep $rep bwdstats_only  $(ab --opt norm_small=0 --opt conv3_xcd=0)"
  echo "rep $rep all_on         $(ab)"
  echo "rep $rep all_on_flatsk2 $(ab --opt conv3_b6_flat_sk=2)"
  echo "rep $rep all_on_flatsk8 $(ab --opt conv3_b6_flat_sk=8)"
done
echo "acdc all_off  $(ab --workload acdc --opt norm_small=0 --opt fuse_bwd_stats=0 --opt conv3_xcd=0)"
echo "acdc all_on   $(ab --workload acdc)"
echo "panc all_off  $(ab --workload pancreas --opt norm_small=0 --opt fuse_bwd_stats=0 --opt conv3_xcd=0)"
echo "panc all_on   $(ab --workload pancreas)"
} > $out/ab.txt 2>&1
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev -o ev --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 > /tmp/ev.log 2>&1
f=$(find /tmp/ev -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; head -12 $out/kernel_stats.csv | cut -c1-170
rocprofv3 --kernel-trace -d /tmp/ev2 -o run --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 8 --warmup 4 > /tmp/ev2.log 2>&1
cd $R
python tools/timeline_attrib.py $(find /tmp/ev2 -name "*kernel_trace.csv" | head -1) --steps 4 --json $out/timeline.json > $out/timeline.txt; head -45 $out/timeline.txt
bash tools/collect_pmc_ops.sh $out/pmc_ops > $out/pmc_ops.txt 2>&1; cp $out/pmc_ops/summary.json $out/pmc_ops.json; rm -rf $out/pmc_ops/pass*; cat $out/pmc_ops.txt | tail -60
