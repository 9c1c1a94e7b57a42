#!/bin/bash
# round 4, GPU session 3: norm_slabs with more statistics blocks, deferred dgrad packs, mix / loss launch shapes: parity + A/B on one box
out=$PWD/gpurun_out/r04_s3; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vnet.py tests/test_gpu_unet.py tests/test_gpu_scripts.py -q -x 2>&1 | tail -3 | tee $out/pytest.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep la base $(ab) slabs0 $(ab --opt norm_slabs=0) nodefer $(ab --opt defer_dgrad_pack=0) | panc base $(ab --workload pancreas) slabs0 $(ab --workload pancreas --opt norm_slabs=0) | acdc base $(ab --workload acdc) slabs0 $(ab --workload acdc --opt norm_slabs=0) nodefer $(ab --workload acdc --opt defer_dgrad_pack=0)"
done 2>&1 | tee $out/ab.txt
python tools/bench_conv.py --levels 128,256 --ops fwd_chain,bwd_chain --json $out/c.json --variants "slabs:;sum:norm_slabs=0" 2>&1 | grep -v "amdgpu\|fp32" | tee $out/c.txt
