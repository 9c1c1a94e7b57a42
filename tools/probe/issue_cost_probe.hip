// tools/probe/issue_cost_probe.hip -- measurement only: how many matrix-pipe cycles does ONE vector-side instruction cost a
// wave that otherwise issues back-to-back v_mfma_f32_16x16x4_f32?  (tools/probe/overlap_probe.hip showed that on gfx950 the
// fp32 MFMA shares its issue / datapath with every other vector instruction of the SIMD: times ADD, only SALU overlaps.)
// Each wave runs ITER x [ G MFMAs on 4 accumulators + K copies of the filler ]; cost = (t(filler) - t(none)) / (ITER * K).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))

template <int FILL, int K, int G>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters) {
  __shared__ float4 lds[4096];
  const int tid = threadIdx.x;
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x = in[tid & 63], y = in[(tid & 63) + 64];
  unsigned laddr = (unsigned)((tid & 255) * 16);
  const float* gp = in + ((blockIdx.x * 512 + tid) & 4095) * 4;
  float* sp = out + (1 << 20) + (long long)(blockIdx.x * 512 + tid) * 4;
  f32x4 t0 = {1, 2, 3, 4};
  float s0 = 1.f;
  unsigned keep;
  unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)((tid >> 6) * 1024 + 32768));
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int g = 0; g < G / 4; ++g) { MFMA(a0); MFMA(a1); MFMA(a2); MFMA(a3); }
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      if (FILL == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(t0) : "v"(laddr));
      if (FILL == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(s0) : "v"(laddr));
      if (FILL == 3) asm volatile("ds_write_b128 %0, %1" ::"v"(laddr), "v"(t0));
      if (FILL == 4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t0) : "v"(gp));
      if (FILL == 5) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gp), "s"(lbase) : "memory");
      if (FILL == 6) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(sp), "v"(t0));
      if (FILL == 7) asm volatile("global_store_dword %0, %1, off" ::"v"(sp), "v"(s0));
      if (FILL == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s0) : "v"(x));
      if (FILL == 9) asm volatile("ds_read_b64 %0, %1" : "=v"(*reinterpret_cast<double*>(&t0)) : "v"(laddr));
      if (FILL == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(s0) : "v"(x));
      if (FILL == 11) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(keep));
      if (FILL == 12) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(*reinterpret_cast<double*>(&t0)));
      if (FILL == 13) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(s0) : "v"(x));
      if (FILL == 14) asm volatile("ds_write_b32 %0, %1" ::"v"(laddr), "v"(s0));
    }
    if (FILL != 0 && (i & 3) == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  out[blockIdx.x * 512 + tid] = a0[0] + a1[1] + a2[2] + a3[3] + t0[0] + s0 + lds[tid].x;
}

static float* g_out; static float* g_in;
template <int FILL, int K, int G>
float run(int threads, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<FILL, K, G>), dim3(256), dim3(threads), 0, 0, g_out, g_in, iters);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<FILL, K, G>), dim3(256), dim3(threads), 0, 0, g_out, g_in, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 3 * 1e3f;
}

template <int FILL>
void line(const char* name, int iters) {
  for (int threads = 256; threads <= 512; threads += 256) {
    const float base = run<0, 1, 8>(threads, iters), t1 = run<FILL, 1, 8>(threads, iters), t2 = run<FILL, 2, 8>(threads, iters);
    // time per iteration per SIMD in ns; waves per SIMD = threads / 256
    const float w = threads / 256.f;
    printf("%-26s %d wave/SIMD: 8 MFMA %7.1f us | +1 filler %7.1f us (+%5.1f ns each = %5.1f clk @2.3GHz) | +2 fillers %7.1f us (+%5.1f ns each)\n", name, threads / 256, base, t1,
           (t1 - base) * 1e3f / iters / w, (t1 - base) * 1e3f / iters / w * 2.3f, t2, (t2 - base) * 1e3f / iters / w / 2);
  }
}

int main() {
  (void)hipMalloc(&g_out, (size_t)(1 << 23) * 4);
  (void)hipMalloc(&g_in, 1 << 20);
  std::vector<float> h(1 << 18, 1.0f);
  (void)hipMemcpy(g_in, h.data(), 1 << 20, hipMemcpyHostToDevice);
  const int iters = 2048;
  line<1>("ds_read_b128", iters);
  line<9>("ds_read_b64", iters);
  line<2>("ds_read_b32", iters);
  line<3>("ds_write_b128", iters);
  line<14>("ds_write_b32", iters);
  line<4>("global_load_dwordx4", iters);
  line<5>("global_load_lds_dwordx4", iters);
  line<6>("global_store_dwordx4", iters);
  line<7>("global_store_dword", iters);
  line<8>("v_cndmask_b32", iters);
  line<10>("v_mov_b32", iters);
  line<13>("v_fma_f32", iters);
  line<12>("v_fma_f64", iters);
  line<11>("s_mul_i32", iters);
  return 0;
}
