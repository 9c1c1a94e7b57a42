"""Copy a collected per-op counter summary (gpurun_out/<tag>_pmc_ops.json) into profiles/ as the round's traffic table bench.py
reads (roofline.traffic), stamped with the commit it was measured at.   python tools/stamp_pmc.py gpurun_out/r03_t1_pmc_ops.json r03"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, rnd = sys.argv[1], sys.argv[2]
d = json.load(open(src))
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "bcp_amd"], capture_output=True, text=True).stdout.strip())
d["_meta"] = {"commit": commit + ("+uncommitted" if dirty else ""), "evidence_set": os.path.basename(src).replace("_pmc_ops.json", ""),
              "how": "tools/collect_pmc_ops.sh (rocprofv3 --pmc, FETCH_SIZE / WRITE_SIZE in separate passes), tools/pmc_ops_summary.py"}
dst = os.path.join(ROOT, "profiles", f"{rnd}_pmc_ops.json")
json.dump(d, open(dst, "w"), indent=1)
print(dst, d["_meta"])
