// tools/probe/write_bw_probe.hip -- measurement only (round 6): what bounds the step's pure WRITERS of 128 MB tensors at ~3 TB/s
// ?  Fill / read / copy kernels with plain and nontemporal 16-byte accesses, several grid shapes,
// over a ring of distinct buffers (nothing served from the L2 / MALL of a previous launch) and over ONE buffer (the MALL case).
//   hipcc --offload-arch=gfx950 -O3 -o write_bw_probe write_bw_probe.hip && ./write_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0 fill, 1 read (sum), 2 copy, 3 apply-like (read y, fma + max, write a)
// ST: 0 plain store, 1 nontemporal store;  LD: 0 plain load, 1 nontemporal load
template <int MODE, int ST, int LD, int U>
__global__ __launch_bounds__(256) void k(f4* __restrict__ dst, const f4* __restrict__ src, long long n4, float s, float* sink) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 0) v[u] = (f4){s, s, s, s};
      else v[u] = LD ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 1) { acc += v[u]; continue; }
      if (MODE == 3) {
        v[u] = v[u] * s + 0.25f;
        v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f); v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f);
      }
      if (ST) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n4; i += stride) {
    f4 v = MODE == 0 ? (f4){s, s, s, s} : src[i];
    if (MODE == 1) { acc += v; continue; }
    dst[i] = v;
  }
  if (MODE == 1 && acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

// contiguous chunk per workgroup (each block owns bytes [b * chunk, (b + 1) * chunk)): what the conv epilogues / apply passes look like
template <int ST>
__global__ __launch_bounds__(256) void k_fill_chunk(f4* __restrict__ dst, long long n4, float s) {
  const long long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long long b0 = (long long)blockIdx.x * per, b1 = b0 + per < n4 ? b0 + per : n4;
  const f4 v = {s, s, s, s};
  for (long long i = b0 + threadIdx.x; i < b1; i += 256) {
    if (ST) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
  }
}

struct Ring { std::vector<f4*> a, b; };

template <typename F>
static double time_it(F launch, int reps, hipStream_t st) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch(i);
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch(i);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1000.0 / reps;   // us per launch
}

int main(int argc, char** argv) {
  const long long MB = argc > 1 ? atoll(argv[1]) : 128;
  const int NR = argc > 2 ? atoi(argv[2]) : 6;          // ring size (distinct buffers)
  const long long bytes = MB << 20, n4 = bytes / 16;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  Ring r;
  for (int i = 0; i < NR; ++i) {
    f4 *p, *q;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes));
    CK(hipMemsetAsync(p, 0, bytes, st)); CK(hipMemsetAsync(q, 0, bytes, st));
    r.a.push_back(p); r.b.push_back(q);
  }
  float* sink;
  CK(hipMalloc(&sink, 64));
  CK(hipStreamSynchronize(st));
  const int reps = 30;
  printf("# %lld MB per tensor, ring of %d; us per launch, TB/s of (bytes read + bytes written)\n", MB, NR);
  const int grids[] = {1024, 2048, 4096, 8192, 0};      // 0 = one 16-byte element per thread
  for (int ring = 1; ring >= 0; --ring) {
    printf("## %s\n", ring ? "ring of distinct buffers" : "ONE buffer pair (MALL-resident between launches where it fits)");
    for (int gi = 0; gi < 5; ++gi) {
      const int g = grids[gi] ? grids[gi] : (int)((n4 + 255) / 256);
#define RUN(NAME, MODE, ST, LD, U, RW)                                                                                    \
      { double us = time_it([&](int i) { const int j = ring ? i % NR : 0;                                                 \
          hipLaunchKernelGGL((k<MODE, ST, LD, U>), dim3(g), dim3(256), 0, st, r.a[j], r.b[j], n4, 0.5f, sink); }, reps, st); \
        printf("%-28s grid %7d  %8.2f us  %6.3f TB/s\n", NAME, g, us, (double)(RW) * bytes / us * 1e-6); }
      RUN("fill plain U1", 0, 0, 0, 1, 1)
      RUN("fill nt    U1", 0, 1, 0, 1, 1)
      RUN("fill plain U4", 0, 0, 0, 4, 1)
      RUN("fill nt    U4", 0, 1, 0, 4, 1)
      RUN("read plain U4", 1, 0, 0, 4, 1)
      RUN("read nt    U4", 1, 0, 1, 4, 1)
      RUN("copy plain U4", 2, 0, 0, 4, 2)
      RUN("copy ntst  U4", 2, 1, 0, 4, 2)
      RUN("copy ntld  U4", 2, 0, 1, 4, 2)
      RUN("copy ntboth U4", 2, 1, 1, 4, 2)
      RUN("apply plain U4", 3, 0, 0, 4, 2)
      RUN("apply ntst  U4", 3, 1, 0, 4, 2)
      RUN("apply ntboth U2", 3, 1, 1, 2, 2)
      if (grids[gi]) {
        double us = time_it([&](int i) { const int j = ring ? i % NR : 0;
          hipLaunchKernelGGL((k_fill_chunk<0>), dim3(g), dim3(256), 0, st, r.a[j], n4, 0.5f); }, reps, st);
        printf("%-28s grid %7d  %8.2f us  %6.3f TB/s\n", "fill chunk plain", g, us, (double)bytes / us * 1e-6);
        us = time_it([&](int i) { const int j = ring ? i % NR : 0;
          hipLaunchKernelGGL((k_fill_chunk<1>), dim3(g), dim3(256), 0, st, r.a[j], n4, 0.5f); }, reps, st);
        printf("%-28s grid %7d  %8.2f us  %6.3f TB/s\n", "fill chunk nt", g, us, (double)bytes / us * 1e-6);
      }
    }
  }
  // hipMemsetAsync as the runtime's own writer
  { double us = time_it([&](int i) { CK(hipMemsetAsync(r.a[i % NR], 0, bytes, st)); }, reps, st);
    printf("%-28s               %8.2f us  %6.3f TB/s\n", "hipMemsetAsync (ring)", us, (double)bytes / us * 1e-6); }
  { double us = time_it([&](int i) { CK(hipMemcpyAsync(r.a[i % NR], r.b[i % NR], bytes, hipMemcpyDeviceToDevice, st)); }, reps, st);
    printf("%-28s               %8.2f us  %6.3f TB/s\n", "hipMemcpyAsync d2d (ring)", us, 2.0 * bytes / us * 1e-6); }
  return 0;
}
