#!/bin/bash
# rocprofv3 counter passes for the dominant conv kernels (run ON the GPU box from the repo root):
#   tools/collect_pmc.sh <out_dir> [C]
# Separate passes as /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# counters are never combined with --sys-trace / hip / hsa tracing).  Summarise with tools/pmc_summary.py.
set -u
out=${1:-gpurun_out/pmc}; C=${2:-16}
mkdir -p "$out"
export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  tag=$(echo "$set" | tr ' ' '+' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set -d "$out/$tag" -o run --output-format csv -- python tools/prof_conv.py "$C" > "$out/$tag.log" 2>&1 || echo "pass $tag failed"
done
python tools/pmc_summary.py "$out" > "$out/summary.json"
cat "$out/summary.json"
