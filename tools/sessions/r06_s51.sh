#!/bin/bash
# round 6, GPU session 51: HBM write bandwidth by store flavour / grid shape (every pure writer of a 128 MB tensor sits at ~3 TB/s)
out=$PWD/gpurun_out/r06_s51; mkdir -p $out
timeout 300 tools/probe/_bin/write_bw_probe 128 6 2>&1 | tee $out/write_bw_128.txt | tail -80
timeout 300 tools/probe/_bin/write_bw_probe 32 12 2>&1 > $out/write_bw_32.txt
