#!/bin/bash
# rocprofv3 counter passes over every op family of the LA step at its in-step shape (run ON the GPU box from the repo root):
#   tools/collect_pmc_ops.sh <out_dir>
# Separate passes as /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters
# are never combined with --sys-trace / hip / hsa tracing).  Summarise with tools/pmc_ops_summary.py.
set -u
out=${1:-gpurun_out/pmc_ops}
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $set -d "$out/pass$i" -o run --output-format csv -- python tools/prof_ops.py "$out" 3 > "$out/pass$i.log" 2>&1 || echo "pass $i ($set) failed"
done
python tools/pmc_ops_summary.py "$out" > "$out/summary.json"
python - "$out/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(f"{k:34s} {v.get('avg_us', 0):8.1f} us  fetch {v.get('hbm_fetch_mb', 0):8.1f} MB  write {v.get('hbm_write_mb', 0):8.1f} MB  alg {v.get('algorithmic_mb', 0):8.1f} MB  mfma_busy {v.get('mfma_busy_frac', 0):.3f}  {v.get('achieved', '')}")
PY
