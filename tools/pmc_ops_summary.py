"""Per-op counter summary of the passes written by tools/collect_pmc_ops.sh: the dispatch list of every pass is cut at the
marker launches (k_ema on 64 elements: before the warm-up call, before the measured calls, behind them), the middle segment of triple i
belongs to op i of <dir>/ops.json, counters are summed over all kernels of
the segment and divided by the repetitions.  FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled for the HBM estimate
(MI355X_MICROARCH.md: gfx950 tallies 128-B read requests at 64 B for 16-B/lane streams).
   python tools/pmc_ops_summary.py <dir>  -> JSON {op: {counter: per-launch value, ...}}"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
plan = json.load(open(os.path.join(d, "ops.json")))
out = {p["op"]: {} for p in plan}
for f in sorted(glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    disp = defaultdict(dict)
    meta = {}
    for r in rows:
        i = int(r["Dispatch_Id"])
        disp[i][r["Counter_Name"]] = float(r["Counter_Value"])
        meta[i] = (r["Kernel_Name"], int(r["Grid_Size"]) if r.get("Grid_Size") else 0)
    order = sorted(disp)
    segs, cur, started = [], [], False
    for i in order:
        name, grid = meta[i]
        if "k_ema" in name and grid <= 4096:
            if started:
                segs.append(cur)
            cur, started = [], True
        elif started:
            cur.append(i)
    segs = segs[1::3]          # [warm-up, measured, set-up of the next op] triples: keep the measured ones (round 5: third marker)
    if len(segs) != len(plan):
        print(f"warning: {f}: {len(segs)} segments for {len(plan)} ops", file=sys.stderr)
    for p, seg in zip(plan, segs):
        tot = defaultdict(float)
        for i in seg:
            for c, v in disp[i].items():
                tot[c] += v
        for c, v in tot.items():
            out[p["op"]][c] = v / p["reps"]
        out[p["op"]]["kernels_per_launch"] = len(seg) / p["reps"]
# kernel-trace durations from any pass
for f in sorted(glob.glob(os.path.join(d, "pass1", "**", "*kernel_trace.csv"), recursive=True)):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    segs, cur, started = [], [], False
    for r in rows:
        if "k_ema" in r["Kernel_Name"] and int(r["Grid_Size_X"]) <= 4096:
            if started:
                segs.append(cur)
            cur, started = [], True
        elif started:
            cur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for p, seg in zip(plan, segs[1::3]):
        out[p["op"]]["avg_us"] = sum(seg) / p["reps"] / 1e3
for p in plan:
    o = out[p["op"]]
    o["algorithmic_mb"] = p["bytes"] / 1e6
    if "FETCH_SIZE" in o:
        o["hbm_fetch_mb"] = 2 * o["FETCH_SIZE"] * 1024 / 1e6
    if "WRITE_SIZE" in o:
        o["hbm_write_mb"] = o["WRITE_SIZE"] * 1024 / 1e6
    if "SQ_VALU_MFMA_BUSY_CYCLES" in o and "GRBM_GUI_ACTIVE" in o and o["GRBM_GUI_ACTIVE"] > 0:
        o["mfma_busy_frac"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (o["GRBM_GUI_ACTIVE"] / 8)
    if o.get("avg_us"):
        if p["flop"]:
            o["achieved"] = f"{p['flop'] / o['avg_us'] / 1e6:.1f} TFLOP/s"
            o["achieved_tflops"] = p["flop"] / o["avg_us"] / 1e6
        else:
            o["achieved"] = f"{p['bytes'] / o['avg_us'] / 1e3:.0f} GB/s"
            o["achieved_gbs"] = p["bytes"] / o["avg_us"] / 1e3
print(json.dumps(out, indent=1))
