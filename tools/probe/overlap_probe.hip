// tools/probe/overlap_probe.hip -- measurement only: what overlaps with a wave that issues back-to-back v_mfma_f32_16x16x4_f32?
// One 512-thread workgroup per CU (2 waves per SIMD).  Waves 0-3 run an MFMA loop, waves 4-7 run `mode` work:
//   0 nothing   1 VALU (v_fma_f32)   2 SALU   3 global loads (L2-resident)   4 LDS writes   5 global stores   6 f64 VALU   7 VALU ints (v_mad_u64)
// Prints the time of MFMA alone, the partner alone, and both together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE, int MM = 0, int PRIO = 0>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int mfma_iters, int other_iters, int run_mfma, int run_other) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4) {
    if (!run_mfma) return;
    float x = in[tid & 63], y = in[(tid & 63) + 64];
    if (MM == 0) {
      f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
      for (int i = 0; i < mfma_iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
      }
      out[blockIdx.x * 512 + tid] = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (MM == 1) {
      f32x16 a0 = {0}, a1 = a0;
      for (int i = 0; i < mfma_iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      }
      out[blockIdx.x * 512 + tid] = a0[0] + a1[1];
    } else {
      f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
      bf16x8 xb, yb;
      for (int j = 0; j < 8; ++j) { xb[j] = (__bf16)x; yb[j] = (__bf16)y; }
      for (int i = 0; i < mfma_iters * 4; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a3, 0, 0, 0);
      }
      out[blockIdx.x * 512 + tid] = a0[0] + a1[1] + a2[2] + a3[3];
    }
  } else {
    if (!run_other) return;
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    float acc = in[tid & 127];
    if (MODE == 1) {
      float b = in[1], c = in[2];
#pragma unroll 16
      for (int i = 0; i < other_iters * 16; ++i) acc = __builtin_fmaf(acc, b, c);
    } else if (MODE == 2) {
      int s = blockIdx.x + 7, t = blockIdx.x;
#pragma unroll 16
      for (int i = 0; i < other_iters * 16; ++i) { s = s * 3 + t; s ^= (s >> 3); }
      acc += (float)s;
    } else if (MODE == 3) {
      const float4* p = reinterpret_cast<const float4*>(in);
      float4 v = {0, 0, 0, 0};
      for (int i = 0; i < other_iters; ++i) {
        float4 w = p[((i * 512 + tid) & 16383)];
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
      acc += v.x + v.y + v.z + v.w;
    } else if (MODE == 4) {
      float4 v = {acc, acc, acc, acc};
      for (int i = 0; i < other_iters; ++i) {
        *reinterpret_cast<float4*>(&lds[((tid & 255) * 4 + (i & 7) * 1024) & 8191]) = v;
        asm volatile("" ::: "memory");
      }
      acc += lds[tid & 255];
    } else if (MODE == 5) {
      for (int i = 0; i < other_iters; ++i) out[(long long)(1 << 20) + ((blockIdx.x * 512 + tid + i * 131072) & ((1 << 22) - 1))] = acc;
    } else if (MODE == 6) {
      double d = acc, b = in[1], c = in[2];
#pragma unroll 16
      for (int i = 0; i < other_iters * 16; ++i) d = __builtin_fma(d, b, c);
      acc = (float)d;
    } else if (MODE == 7) {
      long long d = (long long)acc; int b = (int)in[1];
#pragma unroll 16
      for (int i = 0; i < other_iters * 16; ++i) d = d * b + i;
      acc = (float)d;
    }
    out[blockIdx.x * 512 + tid] = acc;
  }
}

template <int MODE, int MM = 0, int PRIO = 0>
void run(const char* name, float* out, float* in, int mi, int oi) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float t[3];
  for (int c = 0; c < 3; ++c) {
    const int rm = c != 1, ro = c != 0;
    hipLaunchKernelGGL((k<MODE, MM, PRIO>), dim3(256), dim3(512), 0, 0, out, in, mi, oi, rm, ro);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, MM, PRIO>), dim3(256), dim3(512), 0, 0, out, in, mi, oi, rm, ro);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&t[c], e0, e1);
    t[c] /= 5;
  }
  printf("%-22s mfma alone %8.1f us   other alone %8.1f us   together %8.1f us   (sum %8.1f, max %8.1f)\n", name, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3,
         (t[0] + t[1]) * 1e3, (t[0] > t[1] ? t[0] : t[1]) * 1e3);
}

int main() {
  float *out, *in;
  hipMalloc(&out, (size_t)(1 << 23) * 4);
  hipMalloc(&in, 1 << 20);
  std::vector<float> h(1 << 18, 1.0f);
  hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice);
  const int mi = 4096;   // 16384 MFMAs per wave = 524 K cycles
  run<0>("nothing", out, in, mi, 0);
  run<1>("VALU v_fma_f32", out, in, mi, 4096);
  run<2>("SALU", out, in, mi, 4096);
  run<3>("global loads 16 B", out, in, mi, 4096);
  run<4>("LDS writes b128", out, in, mi, 8192);
  run<5>("global stores 4 B", out, in, mi, 2048);
  run<6>("VALU f64 fma", out, in, mi, 2048);
  run<7>("VALU int64 mad", out, in, mi, 1024);
  printf("-- partner at s_setprio 3\n");
  run<1, 0, 1>("VALU v_fma_f32", out, in, mi, 4096);
  run<3, 0, 1>("global loads 16 B", out, in, mi, 4096);
  run<4, 0, 1>("LDS writes b128", out, in, mi, 8192);
  printf("-- v_mfma_f32_32x32x2_f32\n");
  run<1, 1>("VALU v_fma_f32", out, in, mi / 2, 4096);
  run<3, 1>("global loads 16 B", out, in, mi / 2, 4096);
  run<4, 1>("LDS writes b128", out, in, mi / 2, 8192);
  printf("-- v_mfma_f32_16x16x32_bf16 (4x the instructions)\n");
  run<1, 2>("VALU v_fma_f32", out, in, mi, 4096);
  run<3, 2>("global loads 16 B", out, in, mi, 4096);
  run<4, 2>("LDS writes b128", out, in, mi, 8192);
  return 0;
}
