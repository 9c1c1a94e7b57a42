#!/bin/bash
# round 6, GPU session 10: the statistics-in-the-GEMM-epilogue variant alone, level by level
out=$PWD/gpurun_out/r06_s10; mkdir -p $out
timeout 600 python tools/probe/k2_stats_probe.py 2>&1 | grep -E "RESULT|Error" | tee $out/k2_stats.txt
