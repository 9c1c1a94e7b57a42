// Measurement only: the C=32 conv tap loop (k_conv3_res<3,4,4,8,2>: 128-voxel tile, 32 -> 32 channels, resident weights, one
// workgroup per CU) in isolation, with the two MFMA shapes: MODE 0 = 16x16x4 (2 x 2 register blocking per wave, the product
// kernel's loop), MODE 1 = 32x32x2 (one 32 x 32 block per wave, two accumulators), MODE 2 = as 1 with ONE accumulator.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int XS = 36, HALO = 6 * 6 * 10;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ float4 smem4[];
  float* smem = (float*)smem4;
  for (int i = threadIdx.x; i < HALO * XS + 27 * 1024; i += 256) smem[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  const float* Xs = smem;
  const float* Ws = smem + HALO * XS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0.f;
  if (MODE == 0) {
    const int li = lane & 15, lg = lane >> 4;
    int voff[2];
    for (int mt = 0; mt < 2; ++mt) { const int m = (wave * 2 + mt) * 16 + li; voff[mt] = ((m / 32 * 6 + (m / 8) % 4) * 10 + m % 8) * XS + lg * 4; }
    f32x4 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll 3
      for (int tap = 0; tap < 27; ++tap) {
        const int toff = (((tap / 9) * 6 + (tap / 3) % 3) * 10 + tap % 3) * XS;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float4 a[2], b[2];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) a[mt] = *(const float4*)(Xs + voff[mt] + toff + h * 16);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) b[nt] = *(const float4*)(Ws + ((tap * 8 + h * 4 + lg) * 32 + nt * 16 + li) * 4);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b[nt].x, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b[nt].y, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, b[nt].z, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, b[nt].w, acc[mt][nt], 0, 0, 0);
            }
        }
      }
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  } else {
    const int l32 = lane & 31, g = lane >> 5;
    const int m = wave * 32 + l32;
    const int voff = ((m / 32 * 6 + (m / 8) % 4) * 10 + m % 8) * XS + g * 4;
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 3
      for (int tap = 0; tap < 27; ++tap) {
        const int toff = (((tap / 9) * 6 + (tap / 3) % 3) * 10 + tap % 3) * XS;
        float4 a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = *(const float4*)(Xs + voff + toff + q * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = *(const float4*)(Ws + ((tap * 8 + q * 2 + g) * 32 + l32) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = MODE == 1 ? (q & 1) : 0;
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q].x, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[q].y, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[q].z, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[q].w, acc[c], 0, 0, 0);
        }
      }
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  }
  if (s == 1.2345f) out[threadIdx.x] = s;
}

template <int MODE>
void run(float* d) {
  const int iters = 64, grid = 256;
  const size_t lds = (HALO * XS + 27 * 1024) * 4;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), lds, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), lds, 0, d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * iters * 27.0 * 128 * 32 * 32 * 2;
  printf("mode %d : %7.1f us  %6.1f TFLOP/s\n", MODE, ms * 1e3, flops / ms / 1e9);
}

int main() {
  float* d;
  (void)hipMalloc(&d, 4096);
  for (int r = 0; r < 2; ++r) { run<0>(d); run<1>(d); run<2>(d); }
  return 0;
}
