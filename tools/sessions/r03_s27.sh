#!/bin/bash
# round 3, GPU session 27: split-K slab sum (+ statistics row) in the tile's last workgroup (k_c3q): parity, alone, step A/B
out=$PWD/gpurun_out/s27; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3 or norm or dgrad" 2>&1 | tail -3
python tools/bench_conv.py --levels 128,256 --ops fwd_stats,dgrad,fwd_chain,bwd_chain --json $out/c.json --variants "fuse:;nofuse:conv3_fuse_slabs=0" 2>&1 | grep -v "amdgpu\|fp32" | tee $out/c.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep la nofuse $(ab --opt conv3_fuse_slabs=0) fuse $(ab) | panc nofuse $(ab --workload pancreas --opt conv3_fuse_slabs=0) fuse $(ab --workload pancreas) | acdc nofuse $(ab --workload acdc --opt conv3_fuse_slabs=0) fuse $(ab --workload acdc)"
done 2>&1 | tee $out/ab.txt
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
