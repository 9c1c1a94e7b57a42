#!/bin/bash
# round 4, GPU session 25: what the driver runs at round end, as it runs it (smoke, bench with its own step counts), and the 300-step soak
out=$PWD/gpurun_out/r04_s25; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $out/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 2> $out/bench.err | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('driver-style LA', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], 'acdc', d['extra_workloads']['acdc'].get('value'), 'panc', d['extra_workloads']['pancreas'].get('value'), 'cpu', d['cpu_baseline']['value'])" | tee $out/bench.txt
python tools/soak.py 300 2>&1 | tail -4 | tee $out/soak.txt
