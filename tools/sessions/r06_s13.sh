#!/bin/bash
# round 6, GPU session 13: the glue between the student's forward and backward -- the logits gradient written into the backward plan's input
# again (the provider is reachable from autograd's thread), no zero-filled gradients for the loss terms, labels converted in front of the
# forward: this tree against HEAD's Python (tools/_abl/base, same library), interleaved
out=$PWD/gpurun_out/r06_s13; mkdir -p $out
timeout 300 python tools/probe/loss_region_probe.py 2>&1 | grep "aten::\|dout is" | tail -8 | tee $out/torch_ops.txt
R=$PWD
for r in 1 2 3; do for v in base new; do for w in la acdc pancreas; do
  if [ $v == base ]; then cd $R/tools/_abl/base; else cd $R; fi
  python bench.py --workload $w --no-extra --no-cpu-baseline --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w [$v]', d['value'], d['ms_per_step'])" | tee -a $out/ab.txt
done; done; done
cd $R
python - $out/ab.txt <<'PY'
import sys, collections
acc = collections.defaultdict(list)
for l in open(sys.argv[1]):
    k, _, rest = l.rpartition("] ")
    v, ms = rest.split()
    acc[k + "]"].append((float(v), float(ms)))
for k, xs in acc.items():
    print(f"MEAN {k:30s} {sum(x[0] for x in xs) / len(xs):10.2f}  {sum(x[1] for x in xs) / len(xs):8.4f} ms  (n={len(xs)})")
PY
