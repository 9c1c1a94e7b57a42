#!/bin/bash
# round 4, GPU session 10: |max| slots (32 lines instead of one address), loss finalize out of the LDS; A/B fuse_bwd_stats now that the convs are faster
out=$PWD/gpurun_out/r04_s10; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $out/pytest.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep la base $(ab) nobwstats $(ab --opt fuse_bwd_stats=0) | panc base $(ab --workload pancreas) nobwstats $(ab --workload pancreas --opt fuse_bwd_stats=0) | acdc base $(ab --workload acdc) nobwstats $(ab --workload acdc --opt fuse_bwd_stats=0)"
done 2>&1 | tee $out/ab.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev -o ev --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --no-roofline --steps 10 --warmup 2 > /tmp/ev.log 2>&1
f=$(find /tmp/ev -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv
cd $R; python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r04_s10/kernel_stats.csv")))
for r in rows[:60]:
    if any(k in r["Name"] for k in ("norm", "mixloss", "col_partial", "mix_box", "pack", "wamax")):
        print("%-70s x%5s avg %8.1f us  %5s%%" % (r["Name"].replace("bcp::", "").replace("void ", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
