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
// FORM 0: plain operand order; 1: the second source's halves crossed (op_sel:[0,1] op_sel_hi:[1,0]) -- the form found WRONG; 2-5 (round 6): the
// operand selections the product library still contains (tools/isa_scan.py watch list: low results from low halves) -- op_sel_hi:[0,1] and
// op_sel_hi:[1,0] on v_pk_mul_f32, op_sel_hi:[0,1,1] and op_sel_hi:[1,0,1] on v_pk_fma_f32.  S0 = (x, x), S1 = (1, 2) [form 1: (2, 1)], S2 = (0.5, 0.25);
// both result halves are checked.  tests/test_gpu_kernels.py runs this as a canary: forms 0, 2-5 must stay at 0 wrong launches.
template <int FORM> __global__ __launch_bounds__(256) void victim(const float* x, float* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float lo, hi;
#define BCP_SETUP "v_mov_b32 v20, %2\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v22, 1.0\n\tv_mov_b32 v23, 2.0\n\tv_mov_b32 v26, 0.5\n\tv_mov_b32 v27, 0.25\n\t"
#define BCP_TAIL "\n\tv_mov_b32 %0, v24\n\tv_mov_b32 %1, v25"
#define BCP_CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"
  if (FORM == 0) asm volatile(BCP_SETUP "v_pk_mul_f32 v[24:25], v[20:21], v[22:23]" BCP_TAIL : "=v"(lo), "=v"(hi) : "v"(x[i]) : BCP_CLOB);
  else if (FORM == 1) asm volatile("v_mov_b32 v20, %2\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v22, 2.0\n\tv_mov_b32 v23, 1.0\n\t"
                                   "v_pk_mul_f32 v[24:25], v[20:21], v[22:23] op_sel:[0,1] op_sel_hi:[1,0]" BCP_TAIL : "=v"(lo), "=v"(hi) : "v"(x[i]) : BCP_CLOB);
  else if (FORM == 2) asm volatile(BCP_SETUP "v_pk_mul_f32 v[24:25], v[20:21], v[22:23] op_sel_hi:[0,1]" BCP_TAIL : "=v"(lo), "=v"(hi) : "v"(x[i]) : BCP_CLOB);
  else if (FORM == 3) asm volatile(BCP_SETUP "v_pk_mul_f32 v[24:25], v[20:21], v[22:23] op_sel_hi:[1,0]" BCP_TAIL : "=v"(lo), "=v"(hi) : "v"(x[i]) : BCP_CLOB);
  else if (FORM == 4) asm volatile(BCP_SETUP "v_pk_fma_f32 v[24:25], v[20:21], v[22:23], v[26:27] op_sel_hi:[0,1,1]" BCP_TAIL : "=v"(lo), "=v"(hi) : "v"(x[i]) : BCP_CLOB);
  else asm volatile(BCP_SETUP "v_pk_fma_f32 v[24:25], v[20:21], v[22:23], v[26:27] op_sel_hi:[1,0,1]" BCP_TAIL : "=v"(lo), "=v"(hi) : "v"(x[i]) : BCP_CLOB);
  out[2 * i] = lo;
  out[2 * i + 1] = hi;
}
static void expect(int form, float x, float& lo, float& hi) {      // S0 = (x, x), S1 = (1, 2), S2 = (0.5, 0.25)
  switch (form) {
    case 0: lo = x; hi = 2.f * x; break;
    case 1: lo = x; hi = 2.f * x; break;                 // S1 = (2, 1) crossed: lo = x * 1, hi = x * 2
    case 2: lo = x; hi = 2.f * x; break;                 // op_sel_hi:[0,1]: S0.lo in both halves
    case 3: lo = x; hi = x; break;                       // op_sel_hi:[1,0]: S1.lo in both halves
    case 4: lo = x + 0.5f; hi = 2.f * x + 0.25f; break;  // fma, S0.lo in both halves
    default: lo = x + 0.5f; hi = x + 0.25f; break;       // fma, S1.lo in both halves
  }
}
int main() {
  const int n = 32768, launches = 1920;
  std::vector<float> hx(n), ho(2 * n);
  for (int i = 0; i < n; ++i) hx[i] = 0.25f + (float)(i % 977) * 0.01f;
  float *x, *o, *sink; hipMalloc(&x, n * 4); hipMalloc(&o, 2 * n * 4); hipMalloc(&sink, 1024);
  hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
  hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  const char* names[6] = {"plain", "crossed", "mul op_sel_hi:[0,1]", "mul op_sel_hi:[1,0]", "fma op_sel_hi:[0,1,1]", "fma op_sel_hi:[1,0,1]"};
  const int order[6] = {1, 0, 2, 3, 4, 5};
  for (int oi = 0; oi < 6; ++oi) {
    const int form = order[oi];
    int bad = 0, outside = 0;
    for (int l = 0; l < launches; ++l) {
      hipLaunchKernelGGL(neighbour, dim3(1024), dim3(256), 0, sb, sink, 400);
      const dim3 g(n / 256), b(256);
      switch (form) {
        case 0: hipLaunchKernelGGL(victim<0>, g, b, 0, sa, x, o); break;
        case 1: hipLaunchKernelGGL(victim<1>, g, b, 0, sa, x, o); break;
        case 2: hipLaunchKernelGGL(victim<2>, g, b, 0, sa, x, o); break;
        case 3: hipLaunchKernelGGL(victim<3>, g, b, 0, sa, x, o); break;
        case 4: hipLaunchKernelGGL(victim<4>, g, b, 0, sa, x, o); break;
        default: hipLaunchKernelGGL(victim<5>, g, b, 0, sa, x, o); break;
      }
      hipStreamSynchronize(sa); hipMemcpy(ho.data(), o, 2 * n * 4, hipMemcpyDeviceToHost);
      int w = 0;
      for (int i = 0; i < n; ++i) {
        float lo, hi; expect(form, hx[i], lo, hi);
        if (ho[2 * i] != lo || ho[2 * i + 1] != hi) { ++w; if ((i & 63) < 48) ++outside; }
      }
      bad += w > 0;
    }
    hipDeviceSynchronize();
    // (the first two in the round-5 format; then one line per watched form)
    if (oi == 0) printf("%s: %d of %d launches wrong (lanes 48..63 only: %d); ", names[form], bad, launches, outside == 0);
    else if (oi == 1) printf("%s: %d of %d\n", names[form], bad, launches);
    else printf("FORM %s: %d of %d\n", names[form], bad, launches);
  }
  return 0;
}
