// Measurement only: issue rate of v_mfma_f32_16x16x4_f32 from 1 / 2 / 4 waves per SIMD with 1..8 independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/mfma_probe tools/probe/mfma_probe.hip && tools/probe/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345f) out[threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, float* d) {
  const int iters = 4096, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * NACC * 2048.0;
  printf("waves/SIMD %d  NACC %d : %7.1f us  %6.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", blocks_per_cu, NACC, ms * 1e3, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / ((double)blocks_per_cu * iters * NACC));
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  for (int w = 1; w <= 4; w *= 2) { run<1>(w, d); run<2>(w, d); run<4>(w, d); run<8>(w, d); }
  return 0;
}
