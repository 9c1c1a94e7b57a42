#!/bin/bash
# round 3, GPU session 26: k_c3q at the 128-channel level under the three workgroup -> XCD orders: time alone and fabric fetch
out=$PWD/gpurun_out/s26; mkdir -p $out; R=$PWD
python tools/bench_conv.py --levels 128 --ops fwd_stats,dgrad --lib tools/_abl/xcdm.so --rounds 5 --variants "plain:;streams:conv3_xcd=3;tiles:conv3_xcd=5" 2>&1 | grep -v "amdgpu\|fp32" | tee $out/c.txt
cd /tmp; export TMPDIR=/tmp
for v in 1 3 5; do
  rm -rf /tmp/pm$v
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pm$v -o run --output-format csv -- python $R/tools/bench_conv.py --levels 128 --ops dgrad --lib $R/tools/_abl/xcdm.so --rounds 1 --iters 5 --variants "v:conv3_xcd=$v" > /tmp/pm$v.log 2>&1
  f=$(find /tmp/pm$v -name "*counter_collection.csv" | head -1)
  python - "$f" $v <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
v = [float(r["Counter_Value"]) for r in rows if "k_c3q" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print(f"conv3_xcd={sys.argv[2]}: k_c3q launches {len(v)}, FETCH_SIZE median {sorted(v)[len(v)//2]:.0f} KB raw -> {2*sorted(v)[len(v)//2]/1024:.1f} MB fetched per launch")
PY
done 2>&1 | tee $out/fetch.txt
