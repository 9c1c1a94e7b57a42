#!/bin/bash
# round 6, GPU session 33: what-if pricing (timing only, wrong results): no largest-CC (4), no weight gradients (8), no apply passes at the two top levels (16), no backward statistics pass there (32)
out=$PWD/gpurun_out/r06_s33; mkdir -p $out
WL="la" tools/ab_opts.sh "" "--opt-late whatif=4" "--opt-late whatif=8" "--opt-late whatif=16" "--opt-late whatif=32" 2>&1 | tee $out/ab.txt
