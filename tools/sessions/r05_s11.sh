#!/bin/bash
# round 5, GPU session 11: the rebuilt library (plumbing kernels without SLP-packed arithmetic): whole -m gpu suite twice, the stress harness
# with HIP graphs beside the teacher stream (round 4: 24 of 150 deviated), the three workloads' bench lines
out=$PWD/gpurun_out/r05_s11; mkdir -p $out
bash tools/probe/boxinfo.sh > $out/box.txt 2>&1
for i in 1 2; do ( time timeout 900 python -m pytest tests -m gpu -q -x ) 2>&1 | tail -6 | tee -a $out/pytest.txt; done
S="timeout 600 python tools/probe/replay_stress.py"
$S --what acdc --mode replay --load 1 --runs 150 --graphs 1 --tag graphs1 2>&1 | tee $out/g1.txt | tail -4 | cut -c1-400
$S --what acdc --mode replay --load 1 --runs 150 --graphs 2 --tag graphs2 2>&1 | tee $out/g2.txt | tail -4 | cut -c1-400
$S --what la --mode replay --load 1 --runs 80 --graphs 1 --tag la_graphs1 2>&1 | tee $out/g3.txt | tail -2 | cut -c1-400
$S --what acdc --mode replay --load 2 --runs 150 --tag load2 2>&1 | tee $out/g4.txt | tail -2 | cut -c1-400
$S --what acdc --mode eager --main null --load 1 --runs 100 --tag eager 2>&1 | tee $out/g5.txt | tail -2 | cut -c1-400
python bench.py > $out/bench_la.json 2> $out/bench.err; cut -c1-600 $out/bench_la.json
python bench.py --workload acdc > $out/bench_acdc.json 2>> $out/bench.err; cut -c1-300 $out/bench_acdc.json
python bench.py --workload pancreas > $out/bench_panc.json 2>> $out/bench.err; cut -c1-300 $out/bench_panc.json
