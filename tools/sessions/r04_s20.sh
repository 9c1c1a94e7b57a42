#!/bin/bash
# round 4, GPU session 20: the statistics pass finalises its own groups (last-arriver tree, option norm_tree) -- kernel checks, the network
# suites, then the step with and without it
out=$PWD/gpurun_out/r04_s20; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "norm or pw16 or c1_norm or bwdstats" 2>&1 | tail -3 | tee $out/pytest_k.txt
timeout 900 python -m pytest tests/test_gpu_vnet.py tests/test_gpu_unet.py -m gpu -q 2>&1 | tail -3 | tee $out/pytest_n.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep la tree $(ab) launch $(ab --opt norm_tree=0) | acdc tree $(ab --workload acdc) launch $(ab --workload acdc --opt norm_tree=0) | panc tree $(ab --workload pancreas) launch $(ab --workload pancreas --opt norm_tree=0)"
done 2>&1 | tee $out/ab.txt
