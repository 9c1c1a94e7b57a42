#!/bin/bash
# Build the kernel sources for the HOST simulator (test infrastructure; see tools/emu/hip/hip_runtime.h).
# Output: tests/_emu/libbcp_emu.so -- loaded only by tests/ (never by the product loader bcp_amd/_lib.py).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/tests/_emu"
mkdir -p "$OUT"
CXX=/opt/rocm/lib/llvm/bin/clang++
[ -x "$CXX" ] || CXX=clang++
SRCS="api elementwise loss norm conv3 conv3b conv3bw gemm cc pool2d eval comm replay"
OBJS=""
for s in $SRCS; do
  $CXX -x c++ -O2 -std=c++17 -fPIC -w -I "$ROOT/tools/emu" -c "$ROOT/bcp_amd/csrc/$s.hip" -o "$OUT/$s.o" &
  OBJS="$OBJS $OUT/$s.o"
done
$CXX -O2 -std=c++17 -fPIC -w -I "$ROOT/tools/emu" -c "$ROOT/tools/emu/emu_runtime.cpp" -o "$OUT/emu_runtime.o" &
wait
$CXX -shared -o "$OUT/libbcp_emu.so" $OBJS "$OUT/emu_runtime.o" -lpthread -ldl
echo "built $OUT/libbcp_emu.so"
