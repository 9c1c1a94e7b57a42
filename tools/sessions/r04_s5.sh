#!/bin/bash
# round 4, GPU session 5: the trajectory test under the fp16 instances (numbers), faster weight-amax pass, A/B
out=$PWD/gpurun_out/r04_s5; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_vnet.py -q -x -s -k "five_step_trajectory" 2>&1 | grep -v "^$" | tail -25 | tee $out/traj.txt
python - <<'PY' 2>&1 | tee $out/traj_bf16.txt
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import torch, net_checks as NC
from bcp_amd.hip_ops import Ops
ops = Ops.product(); ops.set_option("conv3_f16", 0)
rep = []
try:
    NC.check_la_traj5(ops, torch.device("cuda:0"), "tests/golden", report=rep, fixture="la_traj5f.npz")
except AssertionError as e:
    print("bf16 FAIL", e)
for r in rep: print("bf16x3 la_traj5f step %d: |hip - ref32| %.2e  |hip - ref64f| %.2e  (ens median %.2e) plab %d (ref %d)" % r)
PY
timeout 1500 python -m pytest tests/test_gpu_vnet.py tests/test_gpu_unet.py tests/test_gpu_scripts.py -q 2>&1 | tail -8 | tee $out/pytest_n.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do
  echo "rep $rep la f16 $(ab) bf16 $(ab --opt conv3_f16=0) | panc f16 $(ab --workload pancreas) bf16 $(ab --workload pancreas --opt conv3_f16=0) | acdc f16 $(ab --workload acdc) bf16 $(ab --workload acdc --opt conv3_f16=0)"
done 2>&1 | tee $out/ab.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev -o ev --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 10 --warmup 2 > /tmp/ev.log 2>&1
f=$(find /tmp/ev -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv
cd $R; grep -E "wamax|pack_conv3|k_c3d" $out/kernel_stats.csv | cut -c1-60,180-260
