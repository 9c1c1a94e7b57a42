#!/bin/bash
# round 4, GPU session 33: plan.GRAPHS = 0 by default + the head's backward statistics: the network suites five times over (the graph-less
# replay must never deviate), kernel checks, the probe both ways, then LA / pancreas with and without the head statistics
out=$PWD/gpurun_out/r04_s33; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -2 | tee $out/pytest_k.txt
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_gpu_vnet.py tests/test_gpu_unet.py -m gpu -q 2>&1 | tail -1; done | tee $out/pytest_n.txt
python tools/probe/graph_concurrency_probe.py 100 graphs=0 2>&1 | tail -1 | tee $out/probe.txt
python tools/probe/graph_concurrency_probe.py 100 graphs=1 2>&1 | tail -1 | tee -a $out/probe.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep la fused $(ab) pass $(ab --opt head_bwd_stats=0) | panc fused $(ab --workload pancreas) pass $(ab --workload pancreas --opt head_bwd_stats=0)"
done 2>&1 | tee $out/ab.txt
