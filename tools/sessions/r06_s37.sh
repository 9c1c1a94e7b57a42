#!/bin/bash
# round 6, GPU session 37: the weight-gradient side stream / the teacher's side stream restricted to a subset of the CUs (CU-masked HIP streams): A/B
out=$PWD/gpurun_out/r06_s37; mkdir -p $out
F=ffffffff; Z=00000000; H=55555555; Q=77777777
WL="la" tools/ab_opts.sh "" "--opt wgrad_cumask=$F:$F:$F:$F:$Z:$Z:$Z:$Z" "--opt wgrad_cumask=$H:$H:$H:$H:$H:$H:$H:$H" "--opt wgrad_cumask=$Q:$Q:$Q:$Q:$Q:$Q:$Q:$Q" "--opt wgrad_cumask=$F:$F:$F:$F:$F:$F:$Z:$Z" "--opt teacher_cumask=$H:$H:$H:$H:$H:$H:$H:$H" 2>&1 | tee $out/ab.txt
