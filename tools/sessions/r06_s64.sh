#!/bin/bash
# round 6, GPU session 64: anatomy of the LA step on the final tree without the profiler (HIP events on both streams)
out=$PWD/gpurun_out/r06_s64; mkdir -p $out
timeout 300 python tools/step_segments2.py 2>&1 | tail -16 | tee $out/segments.txt
