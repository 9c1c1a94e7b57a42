#!/bin/bash
# round 6, GPU session 8: skewed start -- the student's forward delayed by a spin kernel so that the teacher's pseudo-label + largest-CC tail
# (150 us exposed after the student's forward, gpurun_out/r06_s7) runs under the student's last layers
out=$PWD/gpurun_out/r06_s8; mkdir -p $out
python - <<'PY' | tee $out/calib.txt
import torch, time
torch.cuda._sleep(1000); torch.cuda.synchronize()
for c in (100000, 1000000):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(c); e1.record(); torch.cuda.synchronize()
    print("sleep", c, "cycles =", e0.elapsed_time(e1) * 1e3, "us")
PY
WL=la tools/ab_opts.sh "" "--opt student_delay=200000" "--opt student_delay=350000" "--opt student_delay=500000" 2>&1 | tee $out/ab.txt
