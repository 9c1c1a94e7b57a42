#!/bin/bash
# round 4, GPU session 18: deferred dgrad packs now that the switch fires (it tested torch.is_grad_enabled() inside autograd.Function.forward)
out=$PWD/gpurun_out/r04_s18; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $out/pytest.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep la defer $(ab) nodefer $(ab --opt defer_dgrad_pack=0) | acdc defer $(ab --workload acdc) nodefer $(ab --workload acdc --opt defer_dgrad_pack=0) | panc defer $(ab --workload pancreas) nodefer $(ab --workload pancreas --opt defer_dgrad_pack=0)"
done 2>&1 | tee $out/ab.txt
