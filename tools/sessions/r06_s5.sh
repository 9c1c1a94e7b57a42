#!/bin/bash
# round 6, GPU session 5: dgrad weight packs on the side stream (HipNet.DEFER_DGRAD_PACK, round 4: measured, not adopted) re-measured
out=$PWD/gpurun_out/r06_s5; mkdir -p $out
tools/ab_opts.sh "" "--opt defer_dgrad_pack=1" 2>&1 | tee $out/ab.txt
