#!/bin/bash
# measurement only: build variants of libbcp_hip.so with parts of k_conv3_res switched off (BCP_ABLATE bit mask, conv3.hip)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_abl
for m in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DBCP_ABLATE=$m -c bcp_amd/csrc/conv3.hip -o tools/_abl/conv3_$m.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_abl/libbcp_abl_$m.so tools/_abl/conv3_$m.o $(ls bcp_amd/csrc/build/*.o | grep -v conv3.o) ) &
done
wait
ls -la tools/_abl/*.so
