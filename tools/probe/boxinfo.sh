#!/bin/bash
# identity of the GPU box a session ran on (the round-5 deviation is box dependent: session 1 reproduced it in 21-35 % of the loaded
# runs, session 2 -- same script, another box -- in 0 of 1200)
echo "== host"; hostname; uname -r; nproc; grep -m1 "model name" /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max 2>/dev/null
echo "== amdgpu"; cat /sys/module/amdgpu/version 2>/dev/null; cat /sys/module/amdgpu/parameters/mtype_local 2>/dev/null
for p in sched_policy hws_max_conc_proc num_kcq queue_preemption_timeout_ms; do echo "$p=$(cat /sys/module/amdgpu/parameters/$p 2>/dev/null)"; done
echo "== smi"; rocm-smi --showserial --showuniqueid --showvbios --showcomputepartition --showmemorypartition --showperflevel --showclocks 2>&1 | grep -v "^=\|^$" | head -40
rocm-smi --showfwinfo 2>&1 | grep -i "mec\|sdma\|smc\|rlc\|cp_" | head -12
rocm-smi --showrasinfo all 2>&1 | grep -iv "^=\|^$" | head -30
echo "== rocminfo"; rocminfo 2>/dev/null | grep -i "Marketing\|Compute Unit\|Max Waves\|Uuid\|Internal Node\|Node:" | head -20
echo "== env"; env | grep -i "^HSA\|^HIP\|^AMD\|^ROC\|^GPU_\|^NCCL\|^RCCL" | sort
