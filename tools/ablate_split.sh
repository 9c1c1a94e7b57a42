#!/bin/bash
# measurement only (round 6): libbcp_hip.so with the fp16 split arithmetic of the bf16-pipe kernels compiled out (B6_ABLATE = 128, conv3_defs.h
# split_store4_f16: the fetch and the LDS stores stay) -- forward / dgrad only (tools/_abl/split_fwd.so) and with the weight-gradient kernels
# too (tools/_abl/split_all.so).  What planes written by the producing apply pass would save.  Results are WRONG.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_abl
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DB6_ABLATE=128"
/opt/rocm/bin/hipcc $F -c bcp_amd/csrc/conv3b.hip -o tools/_abl/conv3b_128.o &
/opt/rocm/bin/hipcc $F -c bcp_amd/csrc/conv3bw.hip -o tools/_abl/conv3bw_128.o &
wait
O1=$(ls bcp_amd/csrc/build/*.o | grep -v "/conv3b.o")
O2=$(ls bcp_amd/csrc/build/*.o | grep -v "/conv3b.o" | grep -v "/conv3bw.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_abl/split_fwd.so $O1 tools/_abl/conv3b_128.o -ldl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_abl/split_all.so $O2 tools/_abl/conv3b_128.o tools/_abl/conv3bw_128.o -ldl
cp bcp_amd/csrc/libbcp_hip.so tools/_abl/split_none.so
ls -la tools/_abl/split_*.so
