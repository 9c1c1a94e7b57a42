#!/bin/bash
# round 5, GPU session 29: the final build once more on a fresh box -- smoke(), the -m gpu suite three times in a row, the driver's default
# bench command, and the two-ranks-on-one-GPU bench mode (the bucketed exchange with the fused backward paths)
out=$PWD/gpurun_out/r05_s29; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $out/smoke.txt
for i in 1 2 3; do ( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | grep -E "passed|failed|^real|^FAILED" | tee -a $out/pytest.txt; done
( time python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | grep real | tee $out/bench_time.txt; cut -c1-420 $out/bench_default.json
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --share-gpu --steps 6 --warmup 2 > $out/share2_la.json 2> $out/share2_la.err; cut -c1-300 $out/share2_la.json; python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r05_s29/share2_la.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("n_gpus", "ms_per_step", "exposed_allreduce_ms_per_step", "ranks_seen", "bucket_report")})
except Exception as e:
    print("share-gpu line:", e)
P
