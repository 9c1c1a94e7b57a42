#!/bin/bash
# round 4, GPU session 31: the fused head's backward leaves the last norm's backward statistics (bcp_pw16_bwd_norm stat_partial; VNet.head_bwd_stats):
# checks, then LA / pancreas with and without
out=$PWD/gpurun_out/r04_s31; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "pw16 or norm" 2>&1 | tail -3 | tee $out/pytest_k.txt
timeout 1200 python -m pytest tests/test_gpu_vnet.py -m gpu -q 2>&1 | tail -4 | tee $out/pytest_n.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep la fused $(ab) pass $(ab --opt head_bwd_stats=0) | panc fused $(ab --workload pancreas) pass $(ab --workload pancreas --opt head_bwd_stats=0)"
done 2>&1 | tee $out/ab.txt
