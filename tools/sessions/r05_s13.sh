#!/bin/bash
# round 5, GPU session 13: -m gpu suite three times in a row on the rebuilt library; forward graphs on / off for the three workloads
out=$PWD/gpurun_out/r05_s13; mkdir -p $out
for i in 1 2 3; do ( time timeout 900 python -m pytest tests -m gpu -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED" | tee -a $out/pytest.txt; done
for rep in 1 2; do for g in 0 1; do for w in la acdc pancreas; do
  python bench.py --workload $w --opt graphs=$g --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w graphs=$g', d['value'], d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'))" | tee -a $out/graphs_ab.txt
done; done; done
