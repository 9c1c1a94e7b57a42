// Measurement only: the conv tap loop in isolation -- per "tap" 4 A-fragment + 1 B-fragment ds_read_b128 and 16 MFMAs on 4
// accumulators (the k_conv3_res<3,4,4,16,1> inner loop), LDS filled once, no global traffic, no barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int XS, bool PIPE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ float4 smem4[];
  float* smem = (float*)smem4;
  for (int i = threadIdx.x; i < 648 * XS + 27 * 256; i += 256) smem[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  const float* Xs = smem;
  const float* Ws = smem + 648 * XS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  int voff[4];
  for (int mt = 0; mt < 4; ++mt) { const int m = (wave * 4 + mt) * 16 + li; voff[mt] = ((m / 64 * 6 + (m / 16) % 4) * 18 + m % 16) * XS + lg * 4; }
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll 9
    for (int tap = 0; tap < 27; ++tap) {
      const int toff = (((tap / 9) * 6 + (tap / 3) % 3) * 18 + tap % 3) * XS;
      float4 a[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a[mt] = *(const float4*)(Xs + voff[mt] + toff);
      const float4 b = *(const float4*)(Ws + ((tap * 4 + lg) * 16 + li) * 4);
      if (PIPE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b.x, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b.y, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, b.z, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, b.w, acc[mt], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345f) out[threadIdx.x] = s;
}

template <int XS, bool PIPE>
void run(int bpc, float* d) {
  const int iters = 8, grid = 256 * bpc;
  const size_t lds = (648 * XS + 27 * 256) * 4;
  hipFuncSetAttribute((const void*)k<XS, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<XS, PIPE>), dim3(grid), dim3(256), lds, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<XS, PIPE>), dim3(grid), dim3(256), lds, 0, d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * 27 * 16 * 2048.0;
  printf("XS %d  blocks/CU %d  sched_barrier %d : %7.1f us  %6.1f TFLOP/s\n", XS, bpc, (int)PIPE, ms * 1e3, flops / ms / 1e9);
}

int main() {
  float* d;
  (void)hipMalloc(&d, 4096);
  for (int b = 1; b <= 2; ++b) { run<20, false>(b, d); run<20, true>(b, d); run<16, false>(b, d); run<24, false>(b, d); }
  return 0;
}
