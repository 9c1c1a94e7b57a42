#!/bin/bash
# run ON the GPU box from the repo root: the round's evidence set -> gpurun_out/<tag>_*  (copy into profiles/ afterwards;
# the per-op counter table goes through tools/stamp_pmc.py, which stamps it with the commit)
#   tools/collect_evidence.sh r03_t1
tag=${1:-evidence}; out=$PWD/gpurun_out; mkdir -p $out
( time python -m pytest tests -m gpu -q -rP ) > $out/${tag}_pytest_gpu_full.txt 2>&1
{ grep -E "^[0-9]+ (passed|failed)|^FAILED|^ERROR" $out/${tag}_pytest_gpu_full.txt | tail -5; grep -E "full-size step|batch-8 step|traj5f step|la_traj5f|pancreas full|acdc full|flips|worst grad" $out/${tag}_pytest_gpu_full.txt | cut -c1-300; } > $out/${tag}_pytest_gpu.txt
# (round 6: the last stdout line is the compact record, the full report is bench_detail.json)
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench.out 2> $out/${tag}_bench.err; tail -1 $out/${tag}_bench.out > $out/${tag}_bench.json; cp $out/bench_detail.json $out/${tag}_bench_detail.json
python bench.py --workload acdc --no-extra > $out/${tag}_bench_acdc.out 2>> $out/${tag}_bench.err; tail -1 $out/${tag}_bench_acdc.out > $out/${tag}_bench_acdc.json
python bench.py --workload pancreas --no-extra > $out/${tag}_bench_pancreas.out 2>> $out/${tag}_bench.err; tail -1 $out/${tag}_bench_pancreas.out > $out/${tag}_bench_pancreas.json
rm -f $out/${tag}_bench.out $out/${tag}_bench_acdc.out $out/${tag}_bench_pancreas.out
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev -o ev --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 > /tmp/ev.log 2>&1
f=$(find /tmp/ev -name "*kernel_stats.csv" | head -1); cp $f $out/${tag}_kernel_stats.csv
cd $R; cat $out/${tag}_pytest_gpu.txt; cut -c1-400 $out/${tag}_bench.json; cut -c1-200 $out/${tag}_bench_acdc.json; cut -c1-200 $out/${tag}_bench_pancreas.json; head -8 $out/${tag}_kernel_stats.csv | cut -c1-160
# per-op counter passes (HBM bytes, MFMA busy, LDS conflicts) at the in-step shapes
bash tools/collect_pmc_ops.sh $out/${tag}_pmc_ops > $out/${tag}_pmc_ops.txt 2>&1; cp $out/${tag}_pmc_ops/summary.json $out/${tag}_pmc_ops.json; rm -rf $out/${tag}_pmc_ops
# where the step's wall time goes (kernels of the 2-3 streams overlap)
cd /tmp
rocprofv3 --kernel-trace -d /tmp/ev2 -o run --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 8 --warmup 4 > /tmp/ev2.log 2>&1
cd $R
python tools/timeline_attrib.py $(find /tmp/ev2 -name "*kernel_trace.csv" | head -1) --steps 4 --json $out/${tag}_timeline.json > $out/${tag}_timeline.txt
python tools/step_sequence.py $(find /tmp/ev2 -name "*kernel_trace.csv" | head -1) > $out/${tag}_step_sequence.txt 2>&1      # one step as an ordered list with queues and gaps
# round 5: the determinism evidence next to the numbers -- the stress harness (product configuration and per-launch replays, load generator)
# and the upsample reproducer on this build
{ timeout 600 python tools/probe/replay_stress.py --what acdc --mode replay --load 1 --runs 150 --graphs 1 --tag product_graphs1 2>&1 | grep RESULT
  timeout 600 python tools/probe/replay_stress.py --what acdc --mode replay --load 1 --runs 150 --graphs 0 --tag per_launch_replays 2>&1 | grep RESULT
  timeout 600 python tools/probe/replay_stress.py --what la --mode replay --load 1 --runs 80 --tag la_default 2>&1 | grep RESULT
  timeout 600 python tools/probe/replay_stress.py --what pancreas --mode replay --load 1 --runs 60 --tag pancreas_default 2>&1 | grep RESULT
  timeout 300 python tools/probe/bilinear_race_probe.py rounds=60 workers=1 conv=1 load=1 gemm=1 2>&1 | grep RESULT
  timeout 300 python tools/probe/bilinear_race_probe.py rounds=30 workers=2 conv=3 load=1 gemm=1 C=64 H=8 2>&1 | grep RESULT; } > $out/${tag}_determinism.txt 2>&1
cat $out/${tag}_determinism.txt | cut -c1-260
bash tools/probe/boxinfo.sh > $out/${tag}_box.txt 2>&1
# ACDC kernel statistics (the LA ones are above)
cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/ev3 -o ev --output-format csv -- python $R/bench.py --workload acdc --no-cpu-baseline --no-extra --steps 10 --warmup 2 > /tmp/ev3.log 2>&1
f=$(find /tmp/ev3 -name "*kernel_stats.csv" | head -1); cp $f $out/${tag}_kernel_stats_acdc.csv; cd $R
rm -f $out/${tag}_pytest_gpu_full.txt.gz; gzip -9 $out/${tag}_pytest_gpu_full.txt
tail -40 $out/${tag}_pmc_ops.txt; head -20 $out/${tag}_timeline.txt
