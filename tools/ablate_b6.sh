#!/bin/bash
# measurement only: variants of libbcp_hip.so with parts of k_c3b compiled out (B6_ABLATE bit mask, conv3b.hip) -> tools/_abl/b6_<mask>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_abl
OBJS=$(ls bcp_amd/csrc/build/*.o | grep -v conv3b.o)
for m in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DB6_ABLATE=$m -c bcp_amd/csrc/conv3b.hip -o tools/_abl/conv3b_$m.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_abl/b6_$m.so $OBJS tools/_abl/conv3b_$m.o -ldl ) &
done
wait
ls -la tools/_abl/b6_*.so
