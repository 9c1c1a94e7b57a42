#!/bin/bash
# round 6, GPU session 39: wider |max| / k2 pack launches at the head of the forward passes, largest-CC table cleared by k_cc_local, flat weight-gradient sums only for large slabs: suite + bench against the previous build
out=$PWD/gpurun_out/r06_s39; mkdir -p $out

for w in la acdc pancreas; do echo "== $w"; tools/ab_libs.sh tools/_abl/libbcp_prev.so tools/_abl/libbcp_new.so --workload $w --no-extra --no-roofline; done 2>&1 | tee $out/ab.txt
