#!/bin/bash
# round 6, GPU session 48: the teacher's side stream on a CONTIGUOUS half / quarter-complement of the CU mask (whole XCDs, if the mask is XCD-major): A/B
out=$PWD/gpurun_out/r06_s48; mkdir -p $out
F=ffffffff; Z=00000000
WL="la" tools/ab_opts.sh "" "--opt teacher_cumask=$F:$F:$F:$F:$Z:$Z:$Z:$Z" "--opt teacher_cumask=$Z:$Z:$Z:$Z:$F:$F:$F:$F" "--opt teacher_cumask=$F:$F:$F:$F:$F:$F:$Z:$Z" "--opt teacher_cumask=$F:$Z:$F:$Z:$F:$Z:$F:$Z" 2>&1 | tee $out/ab.txt
