#!/bin/bash
# round 6, GPU session 6: stream priorities re-measured (teacher's side stream high, weight-gradient stream low)
out=$PWD/gpurun_out/r06_s6; mkdir -p $out
tools/ab_opts.sh "" "--opt teacher_prio=-1" "--opt wgrad_prio=1" 2>&1 | tee $out/ab.txt
