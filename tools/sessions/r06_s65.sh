#!/bin/bash
# round 6, GPU session 65: the largest-CC local tile size in the step on the final tree (cc_tile 1 = 4x8x16, 2 = 8x16x16 = the default at this size), with and without the per-tile size table
out=$PWD/gpurun_out/r06_s65; mkdir -p $out
for o in cc_tile=2 cc_tile=1 "cc_tile=1 cc_count_tile=1"; do echo "== $o"; timeout 300 python tools/cc_probe.py $o 2>&1 | tail -4; done | tee $out/probe.txt
WL="la" tools/ab_opts.sh "" "--opt cc_tile=1" "--opt cc_tile=1 --opt cc_count_tile=1" 2>&1 | tee $out/ab.txt
