// tools/emu/hip/hip_runtime.h -- HOST-SIDE KERNEL-LOGIC SIMULATOR (TEST INFRASTRUCTURE ONLY).
//
// This header shadows <hip/hip_runtime.h> when the kernel sources under bcp_amd/csrc are
// compiled for x86 by tools/emu/build_emu.sh (clang++ -I tools/emu).  It executes a HIP grid
// on the CPU: one OS thread per in-flight workgroup, one fibre per HIP thread, cooperative
// switching at __syncthreads() / wave-level ops, and bit-faithful software models of the two
// fp32 MFMA instructions the kernels use (k-ordered fmaf chain, see
// /opt/skills/guides/cdna_hip_programming.md "FP32-input MFMA").
//
// Purpose: there is no GPU in the build container and GPU minutes are scarce, so indexing /
// tiling / fragment-layout logic is debugged here first.  The resulting library
// (tests/_emu/libbcp_emu.so) is loaded ONLY by tests/emu_*; the product loader
// (bcp_amd/_lib.py) never looks at it and fails loudly when libbcp_hip.so is missing.
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
#include <algorithm>
#include <sys/mman.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
// LDS-only workgroup barrier of the product code (common.h): a plain barrier on the host
#define BCP_LDS_BARRIER() ::bcpemu::block_sync()
#define BCP_DRAIN_VMEM() ((void)0)
#define BCP_S_SLEEP(n) ((void)0)      /* a timing-only instruction */
// LDS-DMA (global_load_lds_dwordx4): 16 bytes per lane to a wave-uniform LDS base + 16 * lane; synchronous on the host (a fiber runs
// from barrier to barrier, so a slot refilled too early shows up as wrong data in the fibers scheduled later)
#define BCP_GLDS16(gsrc, lds_wave_base) __builtin_memcpy(reinterpret_cast<char*>(lds_wave_base) + 16 * (threadIdx.x & 63), (gsrc), 16)
#define BCP_VM_LDS_BARRIER(N) ::bcpemu::block_sync()
// v_cvt_pk_bf16_f32 (round to nearest even, finite values): software on the host
static inline unsigned bcpemu_rne_bf16(float x) { unsigned u; __builtin_memcpy(&u, &x, 4); u += 0x7FFFu + ((u >> 16) & 1u); return u >> 16; }
// ds_read_b64_tr_b16: lane i of a 16-lane group gets element (i & 3) of the 8 bytes addressed by lane 4 j + (i >> 2) as its element j
// (measured on gfx950: tools/probe/tr_probe.hip)
#define BCP_DS_READ_TR16_B64(p) ::bcpemu::ds_read_tr16_b64(p)
#define BCP_CVT_PK_BF16(a, b) (bcpemu_rne_bf16(a) | (bcpemu_rne_bf16(b) << 16))
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(::bcpemu::dyn_lds());

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyDeviceToDevice = 3, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDefault = 4 };
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <class F> inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
struct hipDeviceProp_t { char gcnArchName[256]; int multiProcessorCount; };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { strcpy(p->gcnArchName, "emu-x86"); p->multiProcessorCount = 8; return hipSuccess; }
// events (bench-only API; the emulator just reports 0)
typedef void* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
// HIP graphs do not exist on the host simulator: capture reports failure and the caller keeps its per-launch replay
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return 1; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return 1; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return 1; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 1; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };

struct float2 { float x, y; };
struct double2 { double x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline double2 make_double2(double x, double y) { return {x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline int2 make_int2(int x, int y) { return {x, y}; }
inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return {x, y, z, w}; }

namespace bcpemu {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Fiber {
  void* sp;          // saved stack pointer
  char* stack;       // base of stack mapping
  bool done;
  uint3_ tid;
};

struct WaveScratch {
  float a[2][64], b[2][64];
  uint16_t a16[2][64][8], b16[2][64][8];   // bf16 operand fragments (v_mfma_f32_16x16x32_bf16)
  uint64_t u[2][64];
  unsigned phase;          // per-lane phase counters live in lane_phase
  unsigned arrived;        // lanes arrived at current wave sync
  unsigned gen;
};

struct BlockCtx {
  std::vector<Fiber> fibers;
  std::vector<WaveScratch> waves;
  std::vector<unsigned> lane_phase;  // per-thread op counter for double-buffered scratch
  unsigned nthreads = 0, alive = 0;
  unsigned bar_arrived = 0, bar_gen = 0;
  int cur = -1;
  void* sched_sp = nullptr;
  const std::function<void()>* body = nullptr;
  char* dyn = nullptr;
  size_t dyn_cap = 0;
  char* stacks = nullptr;
  size_t stacks_cap = 0;
};

extern thread_local BlockCtx* g_ctx;
extern thread_local uint3_ g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

void run_grid(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void block_sync();
void wave_sync();           // all live lanes of the calling wave
void* dyn_lds();
int lane_id();
WaveScratch& my_wave();
unsigned next_phase();      // returns 0/1 alternating per wave-collective op of this lane

template <class T> inline T shfl_generic(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "shfl width");
  WaveScratch& w = my_wave();
  unsigned p = next_phase();
  int l = lane_id();
  uint64_t bits = 0;
  memcpy(&bits, &v, sizeof(T));
  w.u[p][l] = bits;
  wave_sync();
  uint64_t r = w.u[p][src_lane & 63];
  T out;
  memcpy(&out, &r, sizeof(T));
  return out;
}

f32x4 mfma_16x16x4(float a, float b, f32x4 c);
f32x16 mfma_32x32x2(float a, float b, f32x16 c);
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
f32x4 mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c);
typedef _Float16 f16x8_emu __attribute__((ext_vector_type(8)));
f32x4 mfma_16x16x32_f16(f16x8_emu a, f16x8_emu b, f32x4 c);
uint64_t ds_read_tr16_b64(const void* p);   // ds_read_b64_tr_b16

}  // namespace bcpemu

#define threadIdx (::bcpemu::g_threadIdx)
#define blockIdx (::bcpemu::g_blockIdx)
#define blockDim (::bcpemu::g_blockDim)
#define gridDim (::bcpemu::g_gridDim)
static const int warpSize = 64;

inline void __syncthreads() { ::bcpemu::block_sync(); }
inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
inline void __threadfence_block() {}

template <class T> inline T __shfl(T v, int lane, int width = 64) {
  int l = ::bcpemu::lane_id();
  int base = l & ~(width - 1);
  return ::bcpemu::shfl_generic(v, base + (lane & (width - 1)));
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  return ::bcpemu::shfl_generic(v, ::bcpemu::lane_id() ^ mask);
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
  int l = ::bcpemu::lane_id();
  int src = l + (int)d;
  if ((src & ~(width - 1)) != (l & ~(width - 1))) src = l;
  return ::bcpemu::shfl_generic(v, src);
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
  int l = ::bcpemu::lane_id();
  int src = l - (int)d;
  if (src < 0 || (src & ~(width - 1)) != (l & ~(width - 1))) src = l;
  return ::bcpemu::shfl_generic(v, src);
}
inline unsigned long long __ballot(int pred) {
  unsigned long long bit = pred ? 1ull : 0ull;
  unsigned long long m = 0;
  // gather via 64 shuffles is slow; use scratch directly
  ::bcpemu::WaveScratch& w = ::bcpemu::my_wave();
  unsigned p = ::bcpemu::next_phase();
  int l = ::bcpemu::lane_id();
  w.u[p][l] = bit | 2ull;  // bit1 marks "participating"
  ::bcpemu::wave_sync();
  for (int i = 0; i < 64; ++i)
    if (w.u[p][i] & 1ull) m |= (1ull << i);
  return m;
}

// ---- atomics on "global" memory (blocks run concurrently on different OS threads)
inline float atomicAdd(float* p, float v) {
  uint32_t* ip = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw;
  float f;
  do {
    memcpy(&f, &old, 4);
    f += v;
    memcpy(&nw, &f, 4);
  } while (!__atomic_compare_exchange_n(ip, &old, nw, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED));
  memcpy(&f, &old, 4);
  return f;
}
inline double atomicAdd(double* p, double v) {
  uint64_t* ip = reinterpret_cast<uint64_t*>(p);
  uint64_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw;
  double f;
  do {
    memcpy(&f, &old, 8);
    f += v;
    memcpy(&nw, &f, 8);
  } while (!__atomic_compare_exchange_n(ip, &old, nw, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED));
  memcpy(&f, &old, 8);
  return f;
}
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline int atomicMin(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) {}
  return old;
}
inline unsigned atomicMin(unsigned* p, unsigned v) {
  unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) {}
  return old;
}
inline int atomicMax(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) {}
  return old;
}
inline unsigned atomicMax(unsigned* p, unsigned v) {
  unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) {}
  return old;
}
inline int atomicCAS(int* p, int cmp, int v) {
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED);
  return cmp;
}
inline unsigned atomicCAS(unsigned* p, unsigned cmp, unsigned v) {
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED);
  return cmp;
}
inline int atomicExch(int* p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }

// ---- device math shims
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
using std::max;
using std::min;

// ---- MFMA builtins (the product code calls the real __builtin_amdgcn_* names)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) ::bcpemu::mfma_16x16x4((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) ::bcpemu::mfma_32x32x2((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) ::bcpemu::mfma_16x16x32_bf16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) ::bcpemu::mfma_16x16x32_f16((a), (b), (c))
#define __builtin_amdgcn_readfirstlane(v) (::bcpemu::shfl_generic((int)(v), 0))
#define __builtin_amdgcn_s_barrier() ::bcpemu::block_sync()
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)

// ---- launch
template <class K, class... Args>
inline void bcpemu_launch(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
  std::function<void()> body = [=]() { kernel(args...); };
  ::bcpemu::run_grid(grid, block, shmem, body);
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  bcpemu_launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), (hipStream_t)(stream), ##__VA_ARGS__)

// ---- relaxed agent-scope loads (L1-bypassing on the device): plain loads here
// (clang's __hip_atomic_load builtin exists on the host too; only the scope constant may be missing)
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
// ---- added: rounding-mode intrinsics and 64-bit atomicMax used by the kernels
inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
  unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) {}
  return old;
}
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
