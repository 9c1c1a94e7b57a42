#!/bin/bash
# round 5, GPU session 32: the final build under the stress harness for longer than the evidence set's runs -- product configuration (forward
# graphs), load generator on a third stream, every run against the serial eager reference
out=$PWD/gpurun_out/r05_s32; mkdir -p $out
{ timeout 900 python tools/probe/replay_stress.py --what acdc --mode replay --load 1 --runs 500 --graphs 1 --tag product_acdc_500 2>&1 | grep RESULT
  timeout 900 python tools/probe/replay_stress.py --what la --mode replay --load 1 --runs 200 --graphs 1 --tag product_la_200 2>&1 | grep RESULT
  timeout 900 python tools/probe/replay_stress.py --what pancreas --mode replay --load 1 --runs 150 --graphs 1 --tag product_pancreas_150 2>&1 | grep RESULT
  timeout 900 python tools/probe/replay_stress.py --what la --mode replay --load 2 --runs 100 --graphs 2 --tag la_fwd_bwd_graphs_100 2>&1 | grep RESULT
  hipcc --offload-arch=gfx950 -O3 -o /tmp/pkmul_mfma_repro tools/probe/pkmul_mfma_repro.hip 2>/dev/null && /tmp/pkmul_mfma_repro; } 2>&1 | tee $out/determinism_long.txt
bash tools/probe/boxinfo.sh > $out/box.txt 2>&1
