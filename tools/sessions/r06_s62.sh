#!/bin/bash
# round 6, GPU session 62: does the teacher stream's HIP priority reach its kernels when the forward is NOT a graph launch? (graphs = 0 / 1 x teacher_prio 0 / -1)
out=$PWD/gpurun_out/r06_s62; mkdir -p $out
WL="la" tools/ab_opts.sh "" "--opt teacher_prio=-1" "--opt graphs=0" "--opt graphs=0 --opt teacher_prio=-1" 2>&1 | tee $out/ab.txt
