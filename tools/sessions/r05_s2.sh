#!/bin/bash
# round 5, GPU session 2: the deviation reproduces under a load generator (s1: 21 of 100 replayed runs, eager too, |max| slots / fp16 planes /
# weight-gradient stream NOT involved, teacher on the main stream: 0).  Find the first divergent tensor and what kind of load matters.
out=$PWD/gpurun_out/r05_s2; mkdir -p $out
S="timeout 500 python tools/probe/replay_stress.py"
$S --what acdc --mode replay --load 1 --runs 80 --deep 1 --show 8 --tag deep      2>&1 | tee $out/b1.txt | tail -60 | cut -c1-600
$S --what acdc --mode replay --load 1 --runs 100 --loadkind copy --tag copy        2>&1 | tee $out/b2.txt | tail -2
$S --what acdc --mode replay --load 1 --runs 100 --loadkind mm --tag mm            2>&1 | tee $out/b3.txt | tail -2
HSA_ENABLE_SDMA=0 $S --what acdc --mode replay --load 1 --runs 100 --tag nosdma    2>&1 | tee $out/b4.txt | tail -2
$S --what acdc --mode replay --load 1 --runs 100 --opt conv3_b6=0 --tag b6off      2>&1 | tee $out/b5.txt | tail -2
$S --what acdc --mode replay --load 1 --runs 100 --opt norm_slabs=0 --tag slabs0   2>&1 | tee $out/b6.txt | tail -2
$S --what acdc --mode replay --load 1 --runs 100 --attr fuse_c1=0 --tag fusec1off  2>&1 | tee $out/b7.txt | tail -2
$S --what acdc --mode replay --load 1 --runs 100 --attr skip_in_concat=0 --tag skipcat0 2>&1 | tee $out/b8.txt | tail -2
$S --what acdc --mode replay --load 1 --runs 100 --attr inline_dropout=0 --tag inldrop0 2>&1 | tee $out/b9.txt | tail -2
$S --what la --mode replay --load 1 --runs 150 --tag la                             2>&1 | tee $out/b10.txt | tail -2
$S --what pancreas --mode replay --load 1 --runs 80 --tag pancreas                  2>&1 | tee $out/b11.txt | tail -2
AMD_SERIALIZE_KERNEL=3 $S --what acdc --mode replay --load 1 --runs 60 --tag serialize 2>&1 | tee $out/b12.txt | tail -2
grep -h RESULT $out/b*.txt > $out/summary.txt; cut -c1-400 $out/summary.txt
