#!/bin/bash
# round 6, GPU session 24: what-if pricing of launch fusion in the norm layers (results wrong, timing only): finalize launches left out
# (whatif=1), finalize + apply left out at the deep levels (whatif=2), both (3)
out=$PWD/gpurun_out/r06_s24; mkdir -p $out
tools/ab_opts.sh "" "--opt-late whatif=1" "--opt-late whatif=2" "--opt-late whatif=3" 2>&1 | tee $out/ab.txt
