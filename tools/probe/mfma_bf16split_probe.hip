// Measurement only (a lead for the next round): the C=16 conv tap loop with fp32 operands split into
// bf16 pieces on the bf16 matrix pipe -- LDS reads + MFMAs only, no global traffic, no barriers, like mfma_lds_probe.hip.
//   NS = 3 pieces, 6 products (a1b1 a1b2 a2b1 a1b3 a2b2 a3b1): fp32-equivalent accuracy
//   NS = 2 pieces, 3 products (a1b1 a1b2 a2b1): ~2^-16 relative per product
// One MFMA = v_mfma_f32_16x16x32_bf16 over TWO taps x 16 input channels (K = 32).  Reported: fp32-EQUIVALENT TFLOP/s
// (2 * voxels * 27 * 16 * 16 per tile; 28 taps are computed, the odd one is padding).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int HALO = 6 * 6 * 18, XSB = 24;   // bf16 elements per halo voxel row (16 channels + 8 pad = 48 B: conflict-free 16-B reads)

template <int NS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ float4 smem4[];
  __bf16* sm = (__bf16*)smem4;
  const int nx = NS * HALO * XSB, nw = NS * 14 * 16 * 32;
  for (int i = threadIdx.x; i < nx + nw; i += 256) sm[i] = (__bf16)((float)(i & 7) * 0.125f);
  __syncthreads();
  const __bf16* Xs = sm;
  const __bf16* Ws = sm + nx;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  int voff[4];
  for (int mt = 0; mt < 4; ++mt) { const int m = (wave * 4 + mt) * 16 + li; voff[mt] = ((m / 64 * 6 + (m / 16) % 4) * 18 + m % 16) * XSB + (lg & 1) * 8; }
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll 2
    for (int tp = 0; tp < 14; ++tp) {
      const int tap = tp * 2 + (lg >> 1);            // lanes 0-31: first tap of the pair, lanes 32-63: second
      const int t = tap < 27 ? tap : 26;
      const int toff = (((t / 9) * 6 + (t / 3) % 3) * 18 + t % 3) * XSB;
      bf16x8 a[4][NS], b[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a[mt][s] = *(const bf16x8*)(Xs + s * HALO * XSB + voff[mt] + toff);
        b[s] = *(const bf16x8*)(Ws + ((s * 14 + tp) * 16 + li) * 32 + lg * 8);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][0], b[0], acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][0], b[1], acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][1], b[0], acc[mt], 0, 0, 0);
        if (NS == 3) {
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][0], b[2], acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][1], b[1], acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][2], b[0], acc[mt], 0, 0, 0);
        }
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345f) out[threadIdx.x] = s;
}

template <int NS>
void run(int bpc, float* d) {
  const int iters = 64, grid = 256 * bpc;
  const size_t lds = (size_t)(NS * HALO * XSB + NS * 14 * 16 * 32) * 2;
  hipFuncSetAttribute((const void*)k<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NS>), dim3(grid), dim3(256), lds, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NS>), dim3(grid), dim3(256), lds, 0, d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * iters * 2.0 * 256 * 27 * 16 * 16;
  printf("pieces %d (%d products)  blocks/CU %d  LDS %zu KB : %7.1f us  %6.1f fp32-equivalent TFLOP/s\n", NS, NS == 3 ? 6 : 3, bpc, lds >> 10, ms * 1e3,
         flops / ms / 1e9);
}

int main() {
  float* d;
  (void)hipMalloc(&d, 4096);
  for (int r = 0; r < 2; ++r)
    for (int b = 1; b <= 2; ++b) { run<3>(b, d); run<2>(b, d); }
  return 0;
}
