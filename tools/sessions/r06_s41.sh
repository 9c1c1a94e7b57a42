#!/bin/bash
# round 6, GPU session 41: the what-if table for ACDC and pancreas (timing only): deep-level finalize + apply (2), largest-CC (4), weight gradients (8), top-level apply passes (16), top-level backward statistics (32)
out=$PWD/gpurun_out/r06_s41; mkdir -p $out
WL="acdc pancreas" tools/ab_opts.sh "" "--opt-late whatif=2" "--opt-late whatif=4" "--opt-late whatif=8" "--opt-late whatif=16" "--opt-late whatif=32" 2>&1 | tee $out/ab.txt
