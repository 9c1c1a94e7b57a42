#!/bin/bash
# round 4, GPU session 27: upper bound of "weight fragments resident" for the ACDC step: the whole step with the weight-fragment loads of the
# bf16-pipe kernels compiled out (B6_ABLATE = 4: results are wrong, only the time matters) against the same build with them
out=$PWD/gpurun_out/r04_s27; mkdir -p $out
bash tools/ab_libs.sh tools/_abl/b6_0.so tools/_abl/b6_4.so --no-extra --no-roofline --workload acdc 2>&1 | tee $out/ab_acdc.txt
bash tools/ab_libs.sh tools/_abl/b6_0.so tools/_abl/b6_4.so --no-extra --no-roofline 2>&1 | tee $out/ab_la.txt
