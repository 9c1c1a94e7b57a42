#!/bin/bash
out=$PWD/gpurun_out/s5; mkdir -p $out
for m in 0 8 7 2 4 1 6 3; do
  lib=""; [ $m != 0 ] && lib="--lib tools/_abl/b6_$m.so"
  echo "== ablate mask $m"
  python tools/bench_conv.py $lib --levels 32,64,16 --ops fwd_stats,dgrad --rounds 3 --json $out/abl_$m.json --variants "d:" 2>&1 | grep -v amdgpu
done > $out/ablate.txt 2>&1
cat $out/ablate.txt
