#!/bin/bash
# run ON the GPU box from the repo root: the round's evidence set -> gpurun_out/<tag>_*  (copy into profiles/ afterwards)
#   tools/collect_evidence.sh r01_t70
tag=${1:-evidence}; out=$PWD/gpurun_out; mkdir -p $out
python -m pytest tests -m gpu -x -q 2>&1 | tail -2 > $out/${tag}_pytest_gpu.txt
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python bench.py --workload acdc > $out/${tag}_bench_acdc.json 2>> $out/${tag}_bench.err
python bench.py --workload pancreas > $out/${tag}_bench_pancreas.json 2>> $out/${tag}_bench.err
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev -o ev --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 > /tmp/ev.log 2>&1
f=$(find /tmp/ev -name "*kernel_stats.csv" | head -1); cp $f $out/${tag}_kernel_stats.csv
cd $R; cat $out/${tag}_pytest_gpu.txt; cut -c1-400 $out/${tag}_bench.json; cut -c1-200 $out/${tag}_bench_acdc.json; cut -c1-200 $out/${tag}_bench_pancreas.json; head -8 $out/${tag}_kernel_stats.csv | cut -c1-160
# per-op counter passes (HBM bytes, MFMA busy, LDS conflicts) at the in-step shapes
bash tools/collect_pmc_ops.sh $out/${tag}_pmc_ops > $out/${tag}_pmc_ops.txt 2>&1; cp $out/${tag}_pmc_ops/summary.json $out/${tag}_pmc_ops.json
# where the step's wall time goes (kernels of the 2-3 streams overlap)
rocprofv3 --kernel-trace -d /tmp/ev2 -o run --output-format csv -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 4 > /tmp/ev2.log 2>&1
python tools/timeline_attrib.py $(find /tmp/ev2 -name "*kernel_trace.csv" | head -1) --steps 4 --json $out/${tag}_timeline.json > $out/${tag}_timeline.txt
tail -40 $out/${tag}_pmc_ops.txt; head -20 $out/${tag}_timeline.txt
