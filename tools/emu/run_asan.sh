#!/bin/bash
# run the simulator kernel checks against the AddressSanitizer build (see build_emu_asan.sh); extra arguments go to pytest
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd "$ROOT"
BCP_EMU_LIB="$ROOT/tests/_emu/asan/libbcp_emu.so" LD_PRELOAD="$RT" \
ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:use_sigaltstack=0:handle_segv=1:abort_on_error=0:halt_on_error=1:allocator_may_return_null=1 \
python -m pytest tests/test_emu_kernels.py -x -q -p no:cacheprovider "$@"
