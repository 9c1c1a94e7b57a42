#!/bin/bash
# measurement only: libbcp_hip.so with the s_memtime stamps of conv3b.hip compiled in (-DBCP_TS_DEBUG=1) -> tools/_abl/ts.so, for
# tools/ts_probe.py.  Needs the product objects (python -c "import __graft_entry__ as g; g.build()") to link against.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_abl
OBJS=$(ls bcp_amd/csrc/build/*.o | grep -v conv3b.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DBCP_TS_DEBUG=1 -c bcp_amd/csrc/conv3b.hip -o tools/_abl/conv3b_ts.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_abl/ts.so $OBJS tools/_abl/conv3b_ts.o -ldl
rm -f tools/_abl/conv3b_ts.o
ls -la tools/_abl/ts.so
