#!/bin/bash
# round 3, GPU session 3: parity suite with the measured numbers printed (-rP), A/B of the two surviving switches, timeline, counters.
out=$PWD/gpurun_out/s3; mkdir -p $out
R=$PWD
( time python -m pytest tests -m gpu -q -rP ) > $out/pytest_gpu_full.txt 2>&1
tail -4 $out/pytest_gpu_full.txt; grep -E "full-size step|batch-8 step|traj5f step|la_traj5f|pancreas full" $out/pytest_gpu_full.txt | cut -c1-400
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_ms_per_step_empty_queue'])"; }
{
for rep in 1 2 3; do
  echo "rep $rep all_off        $(ab --opt fuse_bwd_stats=0 --opt conv3_xcd=0)"
  echo "rep $rep xcd_only       $(ab --opt fuse_bwd_stats=0)"
  echo "rep $rep default        $(ab)"
done
for rep in 1 2; do
echo "acdc all_off  $(ab --workload acdc --opt fuse_bwd_stats=0 --opt conv3_xcd=0)"
echo "acdc default  $(ab --workload acdc)"
echo "panc all_off  $(ab --workload pancreas --opt fuse_bwd_stats=0 --opt conv3_xcd=0)"
echo "panc default  $(ab --workload pancreas)"
done
} > $out/ab.txt 2>&1
cat $out/ab.txt
python tools/bench_conv.py --levels 16,32,64 --ops bwd_chain,dgrad --json $out/bench_conv_mid.json --variants "r2:conv3_xcd=0,fuse_bwd_stats=0;xcd:fuse_bwd_stats=0;xcd_bwdstats:" > $out/bench_conv_mid.txt 2>&1; cat $out/bench_conv_mid.txt
( time python bench.py > $out/bench.json 2> $out/bench.err ); tail -2 $out/bench.err; cut -c1-200 $out/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev -o ev --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 > /tmp/ev.log 2>&1
f=$(find /tmp/ev -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv
rocprofv3 --kernel-trace -d /tmp/ev2 -o run --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 8 --warmup 4 > /tmp/ev2.log 2>&1
cd $R
python tools/timeline_attrib.py $(find /tmp/ev2 -name "*kernel_trace.csv" | head -1) --steps 4 --json $out/timeline.json > $out/timeline.txt; head -60 $out/timeline.txt
bash tools/collect_pmc_ops.sh $out/pmc_ops > $out/pmc_ops.txt 2>&1; cp $out/pmc_ops/summary.json $out/pmc_ops.json; rm -rf $out/pmc_ops/pass*; tail -50 $out/pmc_ops.txt
