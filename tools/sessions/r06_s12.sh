#!/bin/bash
out=$PWD/gpurun_out/r06_s12; mkdir -p $out
timeout 300 python tools/probe/loss_region_probe.py 2>&1 | tail -40 | tee $out/torch_ops.txt
