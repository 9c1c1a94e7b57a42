#!/bin/bash
# round 4, GPU session 19: the two tests that failed in s17 / s18 with the committed defaults; where ACDC's idle time sits (gaps by kernel
# pair, last step's launch sequence); backward pass as a graph on ACDC (plan.GRAPHS = 2 was only ever measured on LA)
out=$PWD/gpurun_out/r04_s19; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_vnet.py tests/test_gpu_unet.py -m gpu -q 2>&1 | tail -3 | tee $out/pytest.txt
ab() { python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"; }
for rep in 1 2; do
  echo "rep $rep acdc g1 $(ab --workload acdc) g2 $(ab --workload acdc --opt graphs=2) | la g1 $(ab) g2 $(ab --opt graphs=2) | panc g1 $(ab --workload pancreas) g2 $(ab --workload pancreas --opt graphs=2)"
done 2>&1 | tee $out/ab.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
for wl in acdc la; do
  rocprofv3 --kernel-trace -d /tmp/tr_$wl -o run --output-format csv -- python $R/bench.py --workload $wl --no-cpu-baseline --no-extra --no-roofline --steps 8 --warmup 4 > /tmp/tr_$wl.log 2>&1
  python $R/tools/timeline_attrib.py $(find /tmp/tr_$wl -name "*kernel_trace.csv" | head -1) --steps 4 --top 30 --gaps 40 --dump $out/seq_$wl.txt > $out/timeline_$wl.txt
done
head -3 $out/timeline_acdc.txt; grep -A12 "idle gaps" $out/timeline_acdc.txt
