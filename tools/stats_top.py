"""per-step view of a rocprofv3 kernel_stats.csv: stats_top.py <csv> <steps> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step: {tot / steps / 1e6:.3f} ms")
for r in rows[:top]:
    print(f'{r["Name"].split("(")[0][-46:]:46s} n/step {int(r["Calls"]) / steps:6.1f} avg {float(r["AverageNs"]) / 1e3:7.1f} us  per step {int(r["TotalDurationNs"]) / steps / 1e3:7.1f} us {float(r["Percentage"]):5.1f}%')
