#!/bin/bash
# Weak-scaling sweep of the headline benchmark on ONE node: bench.py at N = 1, 2, 4, 8 ranks (one process per GPU, RCCL all-reduce
# of the flat gradient buffer between backward and SGD).  Prints the four JSON lines; efficiency = value(N) / (N * value(1)).
#   tools/scale.sh [steps] [warmup] [workload]
# NOT measured by the builder (no multi-GPU box in the build budget): the driver's SCALE run is the first measurement.
set -u
cd "$(dirname "$0")/.."
STEPS=${1:-20}; WARM=${2:-5}; WL=${3:-la}
NG=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  [ "$N" -gt "$NG" ] && { echo "{\"skipped\": \"N=$N needs $N GPUs, $NG visible\"}"; continue; }
  if [ "$N" = 1 ]; then
    python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARM" --workload "$WL" --no-cpu-baseline
  else
    HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARM" --workload "$WL"
  fi
done
