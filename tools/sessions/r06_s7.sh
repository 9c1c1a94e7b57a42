#!/bin/bash
# round 6, GPU session 7: anatomy of the LA step without the profiler (HIP events on both streams)
out=$PWD/gpurun_out/r06_s7; mkdir -p $out
timeout 300 python tools/step_segments2.py 2>&1 | tail -14 | tee $out/segments.txt
