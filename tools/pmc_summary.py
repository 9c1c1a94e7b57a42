"""Average per-launch counter values per kernel from the rocprofv3 --pmc passes written by tools/collect_pmc.sh.
   python tools/pmc_summary.py <dir>  -> JSON {kernel-short-name: {counter: mean over launches}}"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_\w+)<([^>]*)>", name)
    return f"{m.group(1)}<{m.group(2).replace(' ', '')}>" if m else name.split("(")[0]


acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k.startswith("k_conv3"):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} | {"launches": len(next(iter(cs.values())))} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
