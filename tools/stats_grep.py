"""print rows of a rocprofv3 kernel_stats.csv whose kernel name contains a pattern: stats_grep.py <csv> <pattern>"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print(f'{r["Name"].split("(")[0][-48:]:48s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {int(r["MinNs"])/1e3:8.1f}  max {int(r["MaxNs"])/1e3:8.1f}')
