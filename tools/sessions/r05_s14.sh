#!/bin/bash
# round 5, GPU session 14: graphs default on + single-channel mix kernel + fp16-backward long run; suite, benches, two ranks sharing the GPU
out=$PWD/gpurun_out/r05_s14; mkdir -p $out
( time timeout 1200 python -m pytest tests -m gpu -q -rP ) > $out/pytest_full.txt 2>&1; grep -E "passed|failed|^real|^FAILED|long run step" $out/pytest_full.txt | tee $out/pytest.txt
python bench.py > $out/bench_la.json 2> $out/bench.err; cut -c1-330 $out/bench_la.json
python bench.py --workload acdc > $out/bench_acdc.json 2>> $out/bench.err; cut -c1-300 $out/bench_acdc.json
python bench.py --workload pancreas > $out/bench_panc.json 2>> $out/bench.err; cut -c1-300 $out/bench_panc.json
for w in la acdc; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --share-gpu --workload $w --no-extra --no-cpu-baseline --no-roofline > $out/share2_$w.json 2> $out/share2_$w.err
  python -c "
import json; d = json.loads(open('$out/share2_$w.json').read().strip().splitlines()[-1]); print('$w share-gpu x2:', d['value'], d['ms_per_step'], 'exposed', d.get('exposed_allreduce_ms_per_step'), d.get('allreduce_buckets'))" || tail -5 $out/share2_$w.err
done
gzip -9 -f $out/pytest_full.txt
