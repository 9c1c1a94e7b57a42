// bcp_amd/csrc/elementwise.hip -- HBM-bound BCP ops for gfx950: box copy-paste mix (A3+A4),
// pseudo-label (A5), EMA (A10), SGD/Adam (A11), label casts, weight packing.
// Every kernel is a 16-B-per-lane vectorised grid-stride stream (guide Appendix B "element-wise").
#include "common.h"
#include "../../include/bcp_hip.h"

// EMA / SGD must round every product and sum separately (bit-exact vs the reference's mul_/add_ chain):
// hipcc defaults to -ffp-contract=fast, which would fuse __fmul_rn + __fadd_rn into v_fma_f32.
#pragma clang fp contract(off)
namespace bcp {
__device__ __forceinline__ float rn_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float rn_add(float a, float b) { return a + b; }
}  // namespace bcp

namespace bcp {

static inline int stream_grid(long long n_items, int block) {
  long long g = (n_items + block - 1) / block;
  if (g > 2048) g = 2048;  // 256 CUs x 8 blocks, grid-stride the rest (guide G11)
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------- mix (A3+A4)
// out = a outside the box, b inside (reference: a*mask + b*(1-mask), mask = 1 outside the box;
// LA_BCP_train.py:248-251, ACDC_BCP_train.py:372-373, utils/BCP_utils.py:18-28).
// Layout [N][D][H][W][C] with C*W % 4 == 0; the box is in (d,h,w) voxel coordinates.
template <int U>
__global__ __launch_bounds__(256) void k_mix_box(const float* __restrict__ a, const float* __restrict__ b,
                                                 float* __restrict__ out, long long n_vec, int D, int H, int W, int C,
                                                 int d0, int d1, int h0, int h1, int w0, int w1) {
  // (round 3: 32-bit index arithmetic -- the first version spent three 64-bit divisions per 16 bytes -- and U independent
  //  vectors per thread; the launcher guarantees n_vec < 2^29.  Round 4: U = 1 for launches under 4 M floats -- one LA volume with
  //  U = 4 is 245 workgroups, less than one per CU, and the launch is latency: 31 us alone for 12 MB)
  const unsigned nv = (unsigned)n_vec, stride = gridDim.x * blockDim.x;
  const unsigned rowlen = (unsigned)(W * C);  // floats per (n,d,h) row
  for (unsigned i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < nv; i0 += U * stride) {
    float4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned i = i0 + u * stride;
      if (i < nv) { va[u] = ld4(a + (size_t)i * 4); vb[u] = ld4(b + (size_t)i * 4); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned i = i0 + u * stride;
      if (i >= nv) continue;
      const unsigned e = i * 4u;
      const unsigned row = e / rowlen, inrow = e - row * rowlen;
      const int h = (int)(row % (unsigned)H), d = (int)((row / (unsigned)H) % (unsigned)D);
      const bool dh_in = (d >= d0) & (d < d1) & (h >= h0) & (h < h1);
      float4 vo;
      float* po = &vo.x;
      const float* pa = &va[u].x;
      const float* pb = &vb[u].x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int w = (int)((inrow + k) / (unsigned)C);
        const bool in = dh_in & (w >= w0) & (w < w1);
        po[k] = in ? pb[k] : pa[k];
      }
      st4(out + (size_t)i * 4, vo);
    }
  }
}

// Single-channel images (C == 1, W % 4 == 0: every input of the three training loops), round 5.  The general kernel above spends six
// 32-bit divisions per 16 bytes on its (d, h, w) coordinates and reads both inputs everywhere; here a workgroup row (blockIdx.y) is one
// (n, d) slice, the row index inside it is one multiply-high by a host-computed reciprocal, and `b` is read only where the box is:
// algorithmic bytes 8 (a) + 8 (out) + 8 x box fraction per voxel -- 5.0 us for one 8 MB LA batch, 3.6 TB/s (profiles/r05_t2_pmc_ops.txt).
// (The "30 us, 58.5 MB written" rounds 2-4 reported for the mix were the profiling script's own torch.randn launches, DESIGN.md section 5;
// inside the step the two kernels are indistinguishable: 5.541 vs 5.524 ms.)
__global__ __launch_bounds__(256) void k_mix_box_c1(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                    int HW4 /* float4 per slice */, int W4 /* float4 per image row */,
                                                    unsigned magic /* ceil(2^32 / W4) */, int D, int d0, int d1, int h0, int h1, int w0,
                                                    int w1) {
  const int slice = blockIdx.y;                     // n * D + d
  const int d = slice % D;
  const bool d_in = (d >= d0) & (d < d1);           // workgroup-uniform
  const long long base = (long long)slice * HW4;
  for (int s = blockIdx.x * 256 + threadIdx.x; s < HW4; s += gridDim.x * 256) {
    float4 v = ld4(a + (base + s) * 4);
    if (d_in) {
      const int h = (int)__umulhi((unsigned)s, magic);          // s / W4 (exact: the launcher checks HW4 * W4 < 2^32)
      const int w = (s - h * W4) * 4;
      if ((h >= h0) & (h < h1) & (w + 3 >= w0) & (w < w1)) {
        const float4 vb = ld4(b + (base + s) * 4);
        if ((w >= w0) & (w < w1)) v.x = vb.x;
        if ((w + 1 >= w0) & (w + 1 < w1)) v.y = vb.y;
        if ((w + 2 >= w0) & (w + 2 < w1)) v.z = vb.z;
        if ((w + 3 >= w0) & (w + 3 < w1)) v.w = vb.w;
      }
    }
    st4(out + (base + s) * 4, v);
  }
}

// ---------------------------------------------------------------- pseudo-label (A5)
// LA / pancreas: softmax over 2 channels, (p1 >= thres) -> uint8 (LA_BCP_train.py:57-60).
__global__ __launch_bounds__(256) void k_plabel_bin(const float* __restrict__ logits, uint8_t* __restrict__ out,
                                                    long long n_vec, float thres) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    const float4 l0 = ld4(logits + i * 8), l1 = ld4(logits + i * 8 + 4);
    const float x[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
    uchar4 r;
    unsigned char* pr = &r.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) pr[k] = plabel_bin_of(x[2 * k], x[2 * k + 1], thres);
    *reinterpret_cast<uchar4*>(out + i * 4) = r;
  }
}

// ACDC: softmax over 4 channels then argmax, first maximum wins (ACDC_BCP_train.py:112-114).
__global__ __launch_bounds__(256) void k_plabel_argmax4(const float* __restrict__ logits, uint8_t* __restrict__ out,
                                                        long long n_pix) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
    const float4 l = ld4(logits + i * 4);
    out[i] = plabel_argmax4_of(l.x, l.y, l.z, l.w);
  }
}

// ---------------------------------------------------------------- EMA (A10)
// dst = alpha*dst + (1-alpha)*src with the reference's two roundings per term
// (ema.mul_(alpha).add_((1-alpha)*param), utils/BCP_utils.py:78-81); no FMA contraction.
__global__ __launch_bounds__(256) void k_ema(float* __restrict__ dst, const float* __restrict__ src, long long n,
                                             float alpha, float one_minus_alpha) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long nv = n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    float4 d = ld4(dst + i * 4);
    const float4 s = ld4(src + i * 4);
    d.x = rn_add(rn_mul(d.x, alpha), rn_mul(one_minus_alpha, s.x));
    d.y = rn_add(rn_mul(d.y, alpha), rn_mul(one_minus_alpha, s.y));
    d.z = rn_add(rn_mul(d.z, alpha), rn_mul(one_minus_alpha, s.z));
    d.w = rn_add(rn_mul(d.w, alpha), rn_mul(one_minus_alpha, s.w));
    st4(dst + i * 4, d);
  }
  for (long long i = nv * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = rn_add(rn_mul(dst[i], alpha), rn_mul(one_minus_alpha, src[i]));
}

// ---------------------------------------------------------------- SGD / Adam (A11)
// torch.optim.SGD (LA_BCP_train.py:218): g += wd*p; buf = m*buf + g (first step: buf = g); p -= lr*buf.
// Optional fused EMA of a teacher copy (28 B/param instead of 20 + 12).
__global__ __launch_bounds__(256) void k_sgd(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                             float* __restrict__ ema, long long n, float lr, float momentum, float wd,
                                             float gscale, int first_step, float alpha, float one_minus_alpha) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float pv = p[i];
    const float gv = rn_add(g[i] * gscale, rn_mul(wd, pv));
    const float bv = first_step ? gv : rn_add(rn_mul(momentum, buf[i]), gv);
    buf[i] = bv;
    pv = rn_add(pv, rn_mul(-lr, bv));
    p[i] = pv;
    if (ema) ema[i] = rn_add(rn_mul(ema[i], alpha), rn_mul(one_minus_alpha, pv));
  }
}

// torch.optim.Adam defaults (pancreas/dataloaders.py:182): bias-corrected, eps outside the sqrt.
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                              float bc1, float bc2_sqrt, float gscale) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gv = g[i] * gscale;
    const float mv = m[i] * b1 + (1.f - b1) * gv;
    const float vv = v[i] * b2 + (1.f - b2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = p[i] - (lr / bc1) * (mv / denom);
  }
}

// ---------------------------------------------------------------- casts / fills
__global__ __launch_bounds__(256) void k_i64_to_u8(const long long* __restrict__ in, uint8_t* __restrict__ out, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (uint8_t)in[i];
}
__global__ __launch_bounds__(256) void k_f32_to_u8(const float* __restrict__ in, uint8_t* __restrict__ out, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (uint8_t)in[i];
}
__global__ __launch_bounds__(256) void k_u8_to_f32(const uint8_t* __restrict__ in, float* __restrict__ out, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (float)in[i];
}
__global__ __launch_bounds__(256) void k_u8_to_i64(const uint8_t* __restrict__ in, long long* __restrict__ out, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (long long)in[i];
}
__global__ __launch_bounds__(256) void k_axpy(float* __restrict__ y, const float* __restrict__ x, long long n, float a) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = y[i] + a * x[i];
}

// Counter-hash Bernoulli keep-masks for throughput runs (parity runs inject masks instead).
// seed_dev != nullptr: the seed lives in device memory (a launch captured in a HIP graph keeps its arguments; the seed still changes
// every pass -- bcp_store_u64 writes it ahead of the graph launch)
__global__ __launch_bounds__(256) void k_bernoulli_f32(float* __restrict__ out, long long n, float p_keep, float keep_value,
                                                       unsigned seed_lo, unsigned seed_hi, const unsigned long long* __restrict__ seed_dev) {
  if (seed_dev) { const unsigned long long s = *seed_dev; seed_lo = (unsigned)s; seed_hi = (unsigned)(s >> 32); }
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    out[i] = bern_keep(i, seed_lo, seed_hi, p_keep) ? keep_value : 0.f;
  }
}
__global__ __launch_bounds__(256) void k_bernoulli_u8(uint8_t* __restrict__ out, long long n, float p_keep,
                                                      unsigned seed_lo, unsigned seed_hi, const unsigned long long* __restrict__ seed_dev) {
  if (seed_dev) { const unsigned long long s = *seed_dev; seed_lo = (unsigned)s; seed_hi = (unsigned)(s >> 32); }
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    out[i] = bern_keep(i, seed_lo, seed_hi, p_keep) ? 1 : 0;
  }
}

}  // namespace bcp

using namespace bcp;

extern "C" int bcp_mix_box(const float* a, const float* b, float* out, int N, int D, int H, int W, int C,
                           const int* box6, void* stream) {
  BCP_REQUIRE(a && b && out && box6, "bcp_mix_box: null pointer");
  BCP_REQUIRE(aligned16(a) && aligned16(b) && aligned16(out), "bcp_mix_box: pointers must be 16-B aligned");
  BCP_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && C > 0, "bcp_mix_box: bad extents");
  BCP_REQUIRE((W * C) % 4 == 0, "bcp_mix_box: W*C must be a multiple of 4");
  const long long n_vec = (long long)N * D * H * W * C / 4;
  BCP_REQUIRE(n_vec < (1LL << 29), "bcp_mix_box: tensor too large (>= 2^31 floats)");
  const long long HW4 = (long long)H * (W / 4);
  if (options().mix_c1 && C == 1 && W % 4 == 0 && HW4 * (W / 4) < (1LL << 32) && (long long)N * D <= 65535) {
    const int W4 = W / 4;
    const unsigned magic = (unsigned)(((1ULL << 32) + W4 - 1) / W4);      // W4 >= 1; W4 == 1: 2^32 does not fit -> h = s below
    const int gx = (int)((HW4 + 255) / 256 > 64 ? 64 : (HW4 + 255) / 256);
    if (W4 > 1) {
      hipLaunchKernelGGL(k_mix_box_c1, dim3(gx, N * D), dim3(256), 0, (hipStream_t)stream, a, b, out, (int)HW4, W4, magic, D, box6[0],
                         box6[0] + box6[3], box6[1], box6[1] + box6[4], box6[2], box6[2] + box6[5]);
      BCP_CHECK_LAUNCH("bcp_mix_box");
      return BCP_OK;
    }
  }
  if (n_vec <= (1LL << 20))
    hipLaunchKernelGGL(k_mix_box<1>, dim3(stream_grid(n_vec, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n_vec, D, H,
                       W, C, box6[0], box6[0] + box6[3], box6[1], box6[1] + box6[4], box6[2], box6[2] + box6[5]);
  else
    hipLaunchKernelGGL(k_mix_box<4>, dim3(stream_grid((n_vec + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n_vec, D, H,
                       W, C, box6[0], box6[0] + box6[3], box6[1], box6[1] + box6[4], box6[2], box6[2] + box6[5]);
  BCP_CHECK_LAUNCH("bcp_mix_box");
  return BCP_OK;
}

extern "C" int bcp_plabel_bin(const float* logits, uint8_t* out, long long n_vox, float thres, void* stream) {
  BCP_REQUIRE(logits && out, "bcp_plabel_bin: null pointer");
  BCP_REQUIRE(n_vox > 0 && n_vox % 4 == 0, "bcp_plabel_bin: n_vox must be a positive multiple of 4");
  BCP_REQUIRE(aligned16(logits) && (reinterpret_cast<uintptr_t>(out) & 3u) == 0, "bcp_plabel_bin: alignment");
  hipLaunchKernelGGL(k_plabel_bin, dim3(stream_grid(n_vox / 4, 256)), dim3(256), 0, (hipStream_t)stream, logits, out,
                     n_vox / 4, thres);
  BCP_CHECK_LAUNCH("bcp_plabel_bin");
  return BCP_OK;
}

extern "C" int bcp_plabel_argmax4(const float* logits, uint8_t* out, long long n_pix, void* stream) {
  BCP_REQUIRE(logits && out && n_pix > 0, "bcp_plabel_argmax4: bad argument");
  BCP_REQUIRE(aligned16(logits), "bcp_plabel_argmax4: alignment");
  hipLaunchKernelGGL(k_plabel_argmax4, dim3(stream_grid(n_pix, 256)), dim3(256), 0, (hipStream_t)stream, logits, out,
                     n_pix);
  BCP_CHECK_LAUNCH("bcp_plabel_argmax4");
  return BCP_OK;
}

extern "C" int bcp_ema(float* dst, const float* src, long long n, double alpha, void* stream) {
  BCP_REQUIRE(dst && src && n > 0, "bcp_ema: bad argument");
  BCP_REQUIRE(aligned16(dst) && aligned16(src), "bcp_ema: alignment");
  // (1 - alpha) is formed in double then rounded to f32, as python does for (1 - alpha) * tensor
  const float oma = (float)(1.0 - alpha);
  hipLaunchKernelGGL(k_ema, dim3(stream_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, dst, src, n, (float)alpha, oma);
  BCP_CHECK_LAUNCH("bcp_ema");
  return BCP_OK;
}

extern "C" int bcp_sgd(float* p, const float* g, float* buf, float* ema_or_null, long long n, float lr, float momentum,
                       float weight_decay, float grad_scale, int first_step, double ema_alpha, void* stream) {
  BCP_REQUIRE(p && g && buf && n > 0, "bcp_sgd: bad argument");
  const float oma = (float)(1.0 - ema_alpha);
  hipLaunchKernelGGL(k_sgd, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, buf, ema_or_null, n, lr,
                     momentum, weight_decay, grad_scale, first_step, (float)ema_alpha, oma);
  BCP_CHECK_LAUNCH("bcp_sgd");
  return BCP_OK;
}

extern "C" int bcp_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                        float eps, int step, float grad_scale, void* stream) {
  BCP_REQUIRE(p && g && m && v && n > 0 && step >= 1, "bcp_adam: bad argument");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(k_adam, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale);
  BCP_CHECK_LAUNCH("bcp_adam");
  return BCP_OK;
}

extern "C" int bcp_cast(const void* in, void* out, long long n, int kind, void* stream) {
  BCP_REQUIRE(in && out && n > 0, "bcp_cast: bad argument");
  const dim3 g(stream_grid(n, 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
  switch (kind) {
    case BCP_CAST_I64_U8: hipLaunchKernelGGL(k_i64_to_u8, g, b, 0, s, (const long long*)in, (uint8_t*)out, n); break;
    case BCP_CAST_F32_U8: hipLaunchKernelGGL(k_f32_to_u8, g, b, 0, s, (const float*)in, (uint8_t*)out, n); break;
    case BCP_CAST_U8_F32: hipLaunchKernelGGL(k_u8_to_f32, g, b, 0, s, (const uint8_t*)in, (float*)out, n); break;
    case BCP_CAST_U8_I64: hipLaunchKernelGGL(k_u8_to_i64, g, b, 0, s, (const uint8_t*)in, (long long*)out, n); break;
    default: BCP_REQUIRE(false, "bcp_cast: unknown kind %d", kind);
  }
  BCP_CHECK_LAUNCH("bcp_cast");
  return BCP_OK;
}

extern "C" int bcp_axpy(float* y, const float* x, long long n, float a, void* stream) {
  BCP_REQUIRE(y && x && n > 0, "bcp_axpy: bad argument");
  hipLaunchKernelGGL(k_axpy, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, y, x, n, a);
  BCP_CHECK_LAUNCH("bcp_axpy");
  return BCP_OK;
}

static void bernoulli_launch(void* out, long long n, float p_keep, float keep_value, int as_u8, unsigned long long seed,
                             const unsigned long long* seed_dev, hipStream_t s) {
  const dim3 g(stream_grid(n, 256)), b(256);
  if (as_u8)
    hipLaunchKernelGGL(k_bernoulli_u8, g, b, 0, s, (uint8_t*)out, n, p_keep, (unsigned)seed, (unsigned)(seed >> 32), seed_dev);
  else
    hipLaunchKernelGGL(k_bernoulli_f32, g, b, 0, s, (float*)out, n, p_keep, keep_value, (unsigned)seed, (unsigned)(seed >> 32), seed_dev);
}

extern "C" int bcp_bernoulli(void* out, long long n, float p_keep, float keep_value, int as_u8, unsigned long long seed,
                             void* stream) {
  BCP_REQUIRE(out && n > 0, "bcp_bernoulli: bad argument");
  bernoulli_launch(out, n, p_keep, keep_value, as_u8, seed, nullptr, (hipStream_t)stream);
  BCP_CHECK_LAUNCH("bcp_bernoulli");
  return BCP_OK;
}

extern "C" int bcp_bernoulli_dev(void* out, long long n, float p_keep, float keep_value, int as_u8, const unsigned long long* seed_dev,
                                 void* stream) {
  BCP_REQUIRE(out && n > 0 && seed_dev, "bcp_bernoulli_dev: bad argument");
  bernoulli_launch(out, n, p_keep, keep_value, as_u8, 0ull, seed_dev, (hipStream_t)stream);
  BCP_CHECK_LAUNCH("bcp_bernoulli_dev");
  return BCP_OK;
}

namespace bcp {
struct U64x16 { unsigned long long v[16]; };
__global__ void k_store_u64(unsigned long long* __restrict__ dst, int n, U64x16 vals) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}
}  // namespace bcp

extern "C" int bcp_store_u64(unsigned long long* dst, int n, const unsigned long long* host_values, void* stream) {
  BCP_REQUIRE(dst && host_values && n >= 1 && n <= 16, "bcp_store_u64: 1..16 values (they travel as kernel arguments)");
  bcp::U64x16 v{};
  for (int i = 0; i < n; ++i) v.v[i] = host_values[i];
  hipLaunchKernelGGL(bcp::k_store_u64, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, n, v);
  BCP_CHECK_LAUNCH("bcp_store_u64");
  return BCP_OK;
}
