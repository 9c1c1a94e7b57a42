#!/bin/bash
# measurement only: variants of libbcp_hip.so with parts of k_c3p compiled out (P8_ABLATE bit mask, conv3p.hip) -> tools/_abl/p8_<mask>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_abl
OBJS=$(ls bcp_amd/csrc/build/*.o | grep -v conv3p.o)
for m in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DP8_ABLATE=$m -c bcp_amd/csrc/conv3p.hip -o tools/_abl/conv3p_$m.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_abl/p8_$m.so $OBJS tools/_abl/conv3p_$m.o -ldl ) &
done
wait
ls -la tools/_abl/p8_*.so
