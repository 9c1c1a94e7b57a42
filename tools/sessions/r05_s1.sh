#!/bin/bash
# round 5, GPU session 1: reproduce the round-4 red test (replayed passes beside the teacher stream != eager path) and bisect it.
out=$PWD/gpurun_out/r05_s1; mkdir -p $out
S="timeout 400 python tools/probe/replay_stress.py"
for i in 1 2 3; do
  timeout 300 python -m pytest tests/test_gpu_vnet.py -m gpu -q -x -k "graph_replays or replayed_passes" 2>&1 | tail -3 | tee -a $out/pytest_repro.txt
done
$S --what acdc --mode replay --load 0 --runs 100 --deep 1 --tag base0      2>&1 | tee $out/a1.txt | tail -25
$S --what acdc --mode replay --load 1 --runs 100 --deep 1 --tag load1      2>&1 | tee $out/a2.txt | tail -25
$S --what acdc --mode replay --load 2 --runs 100 --deep 1 --tag load2      2>&1 | tee $out/a3.txt | tail -25
$S --what acdc --mode eager --main null --load 1 --runs 100 --tag eager_null 2>&1 | tee $out/a4.txt | tail -12
$S --what acdc --mode eager --main real --load 1 --runs 60 --tag eager_real  2>&1 | tee $out/a5.txt | tail -12
$S --what acdc --mode replay --load 1 --runs 100 --deep 1 --pregraph 1 --tag pregraph 2>&1 | tee $out/a6.txt | tail -25
$S --what la --mode replay --load 1 --runs 60 --deep 1 --tag la            2>&1 | tee $out/a7.txt | tail -25
$S --what acdc --mode replay --load 1 --runs 100 --amax 0 --tag amax0      2>&1 | tee $out/a8.txt | tail -12
$S --what acdc --mode replay --load 1 --runs 100 --overlap 0 --tag overlap0 2>&1 | tee $out/a9.txt | tail -12
$S --what acdc --mode replay --load 1 --runs 100 --wgrad 0 --tag wgrad0    2>&1 | tee $out/a10.txt | tail -12
$S --what acdc --mode replay --load 1 --runs 100 --opt conv3_f16=0 --tag f16off 2>&1 | tee $out/a11.txt | tail -12
grep -h RESULT $out/a*.txt > $out/summary.txt; cat $out/summary.txt
