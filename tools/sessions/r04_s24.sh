#!/bin/bash
# round 4, GPU session 24: what bounds the persistent 16-channel kernel (k_c3d<..,NT = 1, PER>, 123 us alone, mfma_busy 0.30, HBM at 0.3)?
# B6_ABLATE variants alone at 2x112x112x80x16: 1 no stores, 2 no halo fetch / split, 4 no weight-fragment loads, 8 no MFMAs / fragment reads
out=$PWD/gpurun_out/r04_s24; mkdir -p $out
for m in 0 1 2 4 6 8; do
  echo "== B6_ABLATE=$m"; python tools/bench_conv.py --lib tools/_abl/b6_$m.so --levels 16,32 --ops fwd_stats,dgrad --rounds 3 --iters 20 --variants "f16:" --json $out/abl_$m.json 2>&1 | grep -v "^$"
done 2>&1 | tee $out/ablate.txt
