#!/usr/bin/env python
"""Wall-clock attribution of one training step from a rocprofv3 kernel trace (kernels of 2-3 streams overlap, so per-kernel
durations add up to more than the step): every instant of the window is split equally among the kernels running at that
instant and the shares are summed per kernel family.  The result adds up to the window length: it says where the STEP's time
goes, which the per-kernel averages of `--stats` cannot.

  python tools/timeline_attrib.py <run_kernel_trace.csv> [--steps K] [--skip S]

The window is the last K steps of the trace, a step boundary being a launch of the optimiser kernel (k_sgd / k_adam).
"""
import argparse
import collections
import csv
import json
import re


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("bcp::", "")
    return name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--json", default=None)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--gaps", type=int, default=0, help="also list the N largest (previous kernel -> next kernel) idle-gap families")
    ap.add_argument("--dump", default=None, help="write the LAST step's launches (offset us, duration us, stream, workgroups, kernel) to this file")
    args = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(args.trace)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                     int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])),
                     int(r["Stream_Id"])))
    rows.sort()
    marks = [e for (s, e, n, g, st) in rows if n.startswith("k_sgd") or n.startswith("k_adam")]
    if len(marks) < args.steps + 1:
        raise SystemExit("not enough optimiser launches in the trace")
    t0, t1 = marks[-args.steps - 1], marks[-1]
    win = [(max(s, t0), min(e, t1), n, g, st) for (s, e, n, g, st) in rows if e > t0 and s < t1]
    ev = []
    for i, (s, e, n, g, st) in enumerate(win):
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort()
    share = collections.defaultdict(float)
    busy = collections.defaultdict(float)
    count = collections.defaultdict(int)
    running = set()
    idle = 0.0
    conc = collections.defaultdict(float)
    last = t0
    gaps = collections.defaultdict(lambda: [0.0, 0])       # (kernel that ended last, kernel that starts next) -> [ns, count]
    last_ended = None
    for t, kind, i in ev:
        dt = t - last
        if dt > 0:
            if running:
                for j in running:
                    share[(win[j][2], win[j][3])] += dt / len(running)
            else:
                idle += dt
                if kind == 1 and last_ended is not None:
                    g = gaps[(win[last_ended][2], win[i][2])]
                    g[0] += dt
                    g[1] += 1
            conc[len(running)] += dt
        last = t
        if kind == 1:
            running.add(i)
        else:
            running.discard(i)
            last_ended = i
    for (s, e, n, g, st) in win:
        busy[(n, g)] += e - s
        count[(n, g)] += 1
    total = float(t1 - t0)
    K = args.steps
    out = []
    for key, v in sorted(share.items(), key=lambda kv: -kv[1]):
        out.append({"kernel": key[0], "workgroups": key[1], "launches_per_step": count[key] / K, "avg_us": busy[key] / count[key] / 1e3,
                    "busy_ms_per_step": busy[key] / K / 1e6, "wall_share_ms_per_step": v / K / 1e6, "wall_frac": v / total})
    print(f"window: {K} steps, {total / K / 1e6:.3f} ms per step; no kernel running {idle / K / 1e6:.3f} ms per step")
    print("concurrency (ms per step with n kernels running):", {k: round(v / K / 1e6, 3) for k, v in sorted(conc.items())})
    for o in out[:args.top]:
        print(f"{o['wall_share_ms_per_step']:7.3f} ms {o['wall_frac'] * 100:5.1f}%  busy {o['busy_ms_per_step']:6.3f}  x{o['launches_per_step']:5.1f}  avg {o['avg_us']:7.1f} us  wgs {o['workgroups']:6d}  {o['kernel']}")
    if args.gaps:
        print(f"idle gaps by (kernel that ended -> kernel that started), per step:")
        for (a, b), (ns, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:args.gaps]:
            print(f"  {ns / K / 1e3:8.1f} us  x{c / K:5.1f}  avg {ns / c / 1e3:6.1f} us   {a[:44]} -> {b[:44]}")
    if args.dump:
        s0 = marks[-2]
        with open(args.dump, "w") as f:
            for (s, e, n, g, st) in rows:
                if e > s0 and s < t1:
                    f.write(f"{(s - s0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} s{st} {g:6d} {n}\n")
    if args.json:
        json.dump({"ms_per_step": total / K / 1e6, "idle_ms_per_step": idle / K / 1e6,
                   "concurrency_ms": {str(k): v / K / 1e6 for k, v in sorted(conc.items())}, "kernels": out}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
