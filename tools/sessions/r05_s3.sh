#!/bin/bash
# round 5, GPU session 3: the deviation did not show at all on session 2's box.  Box identity + A/B of the round-4 HEAD tree against the working tree.
out=$PWD/gpurun_out/r05_s3; mkdir -p $out
bash tools/probe/boxinfo.sh > $out/box.txt 2>&1; head -50 $out/box.txt
R=$PWD
S="timeout 400 python tools/probe/replay_stress.py"
$S --what acdc --mode replay --load 1 --runs 150 --tag new_a 2>&1 | tee $out/c1.txt | tail -8 | cut -c1-400
( cd tools/_abl/r04head && $S --what acdc --mode replay --load 1 --runs 150 --tag old_a 2>&1 | tee $out/c2.txt | tail -8 | cut -c1-400 )
$S --what acdc --mode replay --load 1 --runs 150 --tag new_b 2>&1 | tee $out/c3.txt | tail -3 | cut -c1-400
( cd tools/_abl/r04head && $S --what acdc --mode replay --load 1 --runs 150 --tag old_b 2>&1 | tee $out/c4.txt | tail -3 | cut -c1-400 )
$S --what acdc --mode eager --main null --load 1 --runs 150 --tag new_eager 2>&1 | tee $out/c5.txt | tail -3 | cut -c1-400
grep -h RESULT $out/c*.txt | cut -c1-330 > $out/summary.txt; cat $out/summary.txt
