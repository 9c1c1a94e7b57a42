#!/bin/bash
# round 5, GPU session 4: the round-4 tree deviates (s3: 40 / 37 of 150 loaded runs), the working tree does not (0 / 0) ON THE SAME BOX.
# Which of this round's changes is it -- and what is the first divergent tensor in the old tree?
out=$PWD/gpurun_out/r05_s4; mkdir -p $out
bash tools/probe/boxinfo.sh > $out/box.txt 2>&1
S="timeout 600 python tools/probe/replay_stress.py"
X="--what acdc --mode replay --load 1"
( cd tools/_abl/r04head && $S $X --runs 40 --deep 1 --show 5 --tag old_deep 2>&1 | tee $out/d1.txt | grep -v "^     got\|^     ref" | tail -60 | cut -c1-900 )
$S $X --runs 150 --amax 0 --tag new_amax0 2>&1 | tee $out/d2.txt | tail -1 | cut -c1-330
( cd tools/_abl/vA && $S $X --runs 150 --tag vA_aliased_slots 2>&1 | tee $out/d3.txt | tail -1 | cut -c1-330 )
( cd tools/_abl/vB && $S $X --runs 150 --tag vB_old_bilinear 2>&1 | tee $out/d4.txt | tail -1 | cut -c1-330 )
cd tools/_abl/r04head
$S $X --runs 100 --amax 0 --tag old_amax0 2>&1 | tee $out/d5.txt | tail -1 | cut -c1-330
$S $X --runs 100 --attr skip_in_concat=0 --tag old_skipcat0 2>&1 | tee $out/d6.txt | tail -1 | cut -c1-330
$S $X --runs 100 --attr fuse_c1=0 --tag old_fusec10 2>&1 | tee $out/d7.txt | tail -1 | cut -c1-330
$S $X --runs 100 --attr inline_dropout=0 --tag old_inldrop0 2>&1 | tee $out/d8.txt | tail -1 | cut -c1-330
GPU_MAX_HW_QUEUES=1 $S $X --runs 100 --tag old_hwq1 2>&1 | tee $out/d9.txt | tail -1 | cut -c1-330
GPU_MAX_HW_QUEUES=8 $S $X --runs 100 --tag old_hwq8 2>&1 | tee $out/d10.txt | tail -1 | cut -c1-330
$S --what la --mode replay --load 1 --runs 100 --tag old_la 2>&1 | tee $out/d11.txt | tail -1 | cut -c1-330
$S --what pancreas --mode replay --load 1 --runs 60 --tag old_pancreas 2>&1 | tee $out/d12.txt | tail -1 | cut -c1-330
$S $X --runs 100 --opt conv3_b6=0 --tag old_b6off 2>&1 | tee $out/d13.txt | tail -1 | cut -c1-330
$S $X --runs 100 --opt norm_slabs=0 --tag old_slabs0 2>&1 | tee $out/d14.txt | tail -1 | cut -c1-330
grep -h RESULT $out/d*.txt | cut -c1-330 > $out/summary.txt
