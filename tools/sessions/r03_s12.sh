#!/bin/bash
out=$PWD/gpurun_out/s12; mkdir -p $out
for v in prod a64; do
  lib=""; [ $v != prod ] && lib="--lib tools/_abl/$v.so"
  echo "== $v"
  python tools/bench_conv.py $lib --levels 64,128,256 --ops dgrad --rounds 4 --json $out/c_$v.json --variants "c3f:;c3g4:conv3_b6_flatd=4;c3g2:conv3_b6_flatd=2;c3g4sk8:conv3_b6_flatd=4,conv3_b6_flat_sk=8;c3fsk8:conv3_b6_flat_sk=8;c3fsk2:conv3_b6_flat_sk=2" 2>&1 | grep -v amdgpu
done > $out/w.txt 2>&1
cat $out/w.txt
