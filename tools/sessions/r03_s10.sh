#!/bin/bash
out=$PWD/gpurun_out/s10; mkdir -p $out
for v in prod w3 w4; do
  lib=""; [ $v != prod ] && lib="--lib tools/_abl/$v.so"
  echo "== $v"
  python tools/bench_conv.py $lib --levels 32 --ops fwd_stats,dgrad --rounds 4 --json $out/c_$v.json --variants "t256:;t128:conv3_b6_cfg32=1" 2>&1 | grep -v amdgpu
done > $out/occ.txt 2>&1
cat $out/occ.txt
