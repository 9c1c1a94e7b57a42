#!/bin/bash
# round 4, GPU session 8: + fp16 instances of k_c3q and of the weight gradient (k_w6): kernel checks, network suite, A/B, kernel stats
out=$PWD/gpurun_out/r04_s8; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x 2>&1 | tail -3 | tee $out/pytest_k.txt
timeout 1500 python -m pytest tests/test_gpu_vnet.py tests/test_gpu_unet.py tests/test_gpu_scripts.py -q 2>&1 | tail -6 | tee $out/pytest_n.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do
  echo "rep $rep la f16 $(ab) bf16 $(ab --opt conv3_f16=0) | panc f16 $(ab --workload pancreas) bf16 $(ab --workload pancreas --opt conv3_f16=0) | acdc f16 $(ab --workload acdc) bf16 $(ab --workload acdc --opt conv3_f16=0)"
done 2>&1 | tee $out/ab.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev -o ev --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 10 --warmup 2 > /tmp/ev.log 2>&1
f=$(find /tmp/ev -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv
cd $R; python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r04_s8/kernel_stats.csv")))
for r in rows[:45]:
    print("%-70s x%5s avg %8.1f us  %5s%%" % (r["Name"].replace("bcp::", "").replace("void ", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
python tools/diag/f16_parity_diag.py 2>&1 | grep -v amdgpu.ids | grep "f16x2" | tee $out/diag.txt
