#!/bin/bash
# round 4, GPU session 23: first layer (k_conv3_c1) with the next tile's halo prefetched under the MFMAs -- checks, then base vs new library
out=$PWD/gpurun_out/r04_s23; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "c1" 2>&1 | tail -3 | tee $out/pytest_k.txt
bash tools/ab_libs.sh tools/_abl/base.so tools/_abl/c1pre.so --no-extra --no-roofline 2>&1 | tee $out/ab_la.txt
bash tools/ab_libs.sh tools/_abl/base.so tools/_abl/c1pre.so --no-extra --no-roofline --workload acdc 2>&1 | tee $out/ab_acdc.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev -o ev --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 10 --warmup 2 > /tmp/ev.log 2>&1
grep -h "k_conv3_c1\|k_gemm_nn\|k_gemm_tn" $(find /tmp/ev -name "*kernel_stats.csv" | head -1) | cut -c1-200 | tee $out/stats_c1.txt
