#!/bin/bash
out=$PWD/gpurun_out/s6; mkdir -p $out
V="s0:"
for bit in 8 0 3 5 6 7 9; do for n in 2 4 6; do V="$V;b${bit}n${n}:conv3_stagger=$n,conv3_stagger_bit=$bit"; done; done
python tools/bench_conv.py --levels 32,64,16 --ops fwd_stats,dgrad,wgrad --rounds 3 --json $out/stagger.json --variants "$V" 2>&1 | grep -v amdgpu > $out/stagger.txt
cat $out/stagger.txt
