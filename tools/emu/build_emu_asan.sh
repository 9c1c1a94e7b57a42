#!/bin/bash
# AddressSanitizer build of the host simulator (test infrastructure): the kernel sources with every global / LDS access checked.
#   tools/emu/build_emu_asan.sh && tools/emu/run_asan.sh -k "conv3 or norm"
# Output: tests/_emu/asan/libbcp_emu.so (run_asan.sh points the emulator fixture at it through BCP_EMU_LIB)
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/tests/_emu/asan"
mkdir -p "$OUT"
CXX=/opt/rocm/lib/llvm/bin/clang++
SRCS="api elementwise loss norm conv3 conv3b conv3bw gemm cc pool2d eval comm replay"
FLAGS="-x c++ -O1 -g -std=c++17 -fPIC -w -fsanitize=address -shared-libasan -fno-omit-frame-pointer -I $ROOT/tools/emu"
OBJS=""
for s in $SRCS; do
  $CXX $FLAGS -c "$ROOT/bcp_amd/csrc/$s.hip" -o "$OUT/$s.o" &
  OBJS="$OBJS $OUT/$s.o"
done
$CXX $FLAGS -c "$ROOT/tools/emu/emu_runtime.cpp" -o "$OUT/emu_runtime.o" &
wait
$CXX -shared -fsanitize=address -shared-libasan -o "$OUT/libbcp_emu.so" $OBJS "$OUT/emu_runtime.o" -lpthread -ldl
echo "built $OUT/libbcp_emu.so"
