// tools/probe/pkmul_mfma_repro.hip -- the cause of round 4's red GPU test in one file (MI355X gfx950, ROCm 7.2).
// A wave that executes  v_pk_mul_f32 D, S0, S1 op_sel:[0,1] op_sel_hi:[1,0]  (packed fp32 multiply, the halves of the SECOND source crossed:
// D.lo = S0.lo * S1.hi, D.hi = S0.hi * S1.lo) gets D.lo = +-0 in lanes 48..63 whenever a wave of ANOTHER kernel on the same CU is issuing dense
// 16-bit MFMAs (v_mfma_f32_16x16x32_bf16 / _f16).  Alone, beside fp32 MFMAs, LDS-DMA, LDS or VALU traffic, or with the plain operand order: never.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pkmul_mfma_repro tools/probe/pkmul_mfma_repro.hip && /tmp/pkmul_mfma_repro
// prints e.g. "crossed: 1649 of 1920 launches wrong (lanes 48..63 only: 1); plain: 0 of 1920".   hipcc's SLP vectoriser emits the crossed form
// for float2 shuffles (k_bilinear2x_fwd, round-4 build); the product build keeps it out of the kernels that run beside the convs (tools/isa_scan.py).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
__global__ __launch_bounds__(256) void neighbour(float* sink, int iters) {          // nothing but dense bf16 MFMAs
  bf16x8 a, b; f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(threadIdx.x * 1e-3f + k); b[k] = (__bf16)(1.0f + k); }
  for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[threadIdx.x] = acc[0];
}
template <bool CROSSED> __global__ __launch_bounds__(256) void victim(const float* x, float* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float lo;                                                                           // v[20:21] = (x, x), v[22:23] = (2.0, 1.0) or (1.0, 2.0)
  if (CROSSED) asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, 2.0\n\tv_mov_b32 v23, 1.0\n\t"
                            "v_pk_mul_f32 v[24:25], v[20:21], v[22:23] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32 %0, v24"
                            : "=v"(lo) : "v"(x[i]) : "v20", "v21", "v22", "v23", "v24", "v25");
  else asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, 1.0\n\tv_mov_b32 v23, 2.0\n\t"
                    "v_pk_mul_f32 v[24:25], v[20:21], v[22:23]\n\tv_mov_b32 %0, v24" : "=v"(lo) : "v"(x[i]) : "v20", "v21", "v22", "v23", "v24", "v25");
  out[i] = lo;                                                                        // must be x[i] * 1.0
}
int main() {
  const int n = 32768, launches = 1920;
  std::vector<float> hx(n), ho(n);
  for (int i = 0; i < n; ++i) hx[i] = 0.25f + (float)(i % 977) * 0.01f;
  float *x, *o, *sink; hipMalloc(&x, n * 4); hipMalloc(&o, n * 4); hipMalloc(&sink, 1024);
  hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
  hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  for (int crossed = 1; crossed >= 0; --crossed) {
    int bad = 0, outside = 0;
    for (int l = 0; l < launches; ++l) {
      hipLaunchKernelGGL(neighbour, dim3(1024), dim3(256), 0, sb, sink, 400);
      if (crossed) hipLaunchKernelGGL(victim<true>, dim3(n / 256), dim3(256), 0, sa, x, o); else hipLaunchKernelGGL(victim<false>, dim3(n / 256), dim3(256), 0, sa, x, o);
      hipStreamSynchronize(sa); hipMemcpy(ho.data(), o, n * 4, hipMemcpyDeviceToHost);
      int w = 0; for (int i = 0; i < n; ++i) if (ho[i] != hx[i]) { ++w; if ((i & 63) < 48) ++outside; }
      bad += w > 0;
    }
    hipDeviceSynchronize();
    printf("%s: %d of %d launches wrong (lanes 48..63 only: %d)%s", crossed ? "crossed" : "plain", bad, launches, outside == 0, crossed ? "; " : "\n");
  }
  return 0;
}
