#!/bin/bash
# round 6, GPU session 45: per-step GPU time of the first steps behind a synchronisation (where do the ~2.2 ms per timed region go?)
out=$PWD/gpurun_out/r06_s45; mkdir -p $out
python tools/probe/first_steps_probe.py 24 2>&1 | grep "^rep" | tee $out/steps.txt
