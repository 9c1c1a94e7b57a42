#!/bin/bash
# measurement only: interleaved in-step A/B of library / host options on one box, three rounds, the three workloads (WL="la acdc pancreas"):
#   tools/ab_opts.sh "" "--opt wgrad_b6_deep=0" "--opt x=1 --opt y=2" ...     (each argument is one variant's bench.py flags)
# prints  <workload> [<variant>] <value> <ms_per_step>  per run and the per-variant means at the end
tmp=$(mktemp)
for r in 1 2 3; do
  for v in "$@"; do
    for w in ${WL:-la acdc pancreas}; do
      python bench.py --workload $w --no-extra --no-cpu-baseline --no-roofline --steps ${STEPS:-40} --warmup 10 $v 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w [${v:-default}]', d['value'], d['ms_per_step'])" | tee -a $tmp
    done
  done
done
python - $tmp <<'PY'
import sys, collections
acc = collections.defaultdict(list)
for l in open(sys.argv[1]):
    k, _, rest = l.rpartition("] ")
    v, ms = rest.split()
    acc[k + "]"].append((float(v), float(ms)))
for k, xs in acc.items():
    print(f"MEAN {k:60s} {sum(x[0] for x in xs) / len(xs):10.2f}  {sum(x[1] for x in xs) / len(xs):8.4f} ms  (n={len(xs)})")
PY
rm -f $tmp
