#!/bin/bash
# round 6, GPU session 30: statistics pass, old (row-by-row tail) against new (predicated U-row trips) build, interleaved on one box
out=$PWD/gpurun_out/r06_s30; mkdir -p $out
for w in la pancreas acdc; do echo "== $w"; tools/ab_libs.sh tools/_abl/libbcp_k1old.so tools/_abl/libbcp_k1new.so --workload $w --no-extra --no-roofline; done 2>&1 | tee $out/ab.txt
