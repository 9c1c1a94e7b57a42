"""One steady-state step of a rocprofv3 --kernel-trace csv as an ordered list (measurement only): start offset, duration, queue,
gap since the previous kernel on the same queue ended, short kernel name, workgroups.
python tools/step_sequence.py <kernel_trace.csv> [--step K] [--marker k_ema]"""
import argparse, csv, re
ap = argparse.ArgumentParser()
ap.add_argument("csv"); ap.add_argument("--step", type=int, default=-2); ap.add_argument("--marker", default="k_ema")
a = ap.parse_args()
rows = list(csv.DictReader(open(a.csv)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
lo, hi = marks[a.step - 1] + 1, marks[a.step] + 1
step = rows[lo:hi]
t0 = step[0]["s"]
last_end = {}
print(f"step of {len(step)} kernels, {(step[-1]['e'] - t0) / 1e3:.1f} us")
for r in step:
    q = r.get("Queue_Id", "?")
    name = re.sub(r"^void |bcp::|\(.*$", "", r["Kernel_Name"])[:46]
    gap = (r["s"] - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = max(last_end.get(q, 0), r["e"])
    wg = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)) or 256), 1)
    print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f} q{q:>3} gap {gap:7.1f}  {name:46s} wg {wg}")
