#!/bin/bash
# round 6, GPU session 15: split-K over the cin chunks for conv outputs up to 2^21 / 2^22 elements (the cap was 2^20: the U-Net's 32x32x128 level
# at a grouped batch of 12 -- 192 workgroups, 29-49 us chains -- was just above it)
out=$PWD/gpurun_out/r06_s15; mkdir -p $out
WL="acdc la pancreas" tools/ab_opts.sh "" "--opt conv3_sk_elems=2097152" "--opt conv3_sk_elems=4194304" 2>&1 | tee $out/ab.txt
