"""Print the kernel sequence of ONE steady-state LA step from a rocprofv3 --kernel-trace csv (measurement only):
python tools/trace_seq.py <kernel_trace.csv> [pattern]  -> neighbours of every kernel whose name contains `pattern`"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0][-60:] for r in rows]
ctx = collections.Counter()
for i, n in enumerate(names):
    if pat in n:
        prev = names[i - 1] if i else "-"
        nxt = names[i + 1] if i + 1 < len(names) else "-"
        ctx[(prev, nxt)] += 1
for (p, n), c in ctx.most_common(25):
    print(f"{c:5d}  after {p:60s} before {n}")
