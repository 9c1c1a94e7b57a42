// bcp_amd/csrc/pool2d.hip -- 2-D U-Net plumbing, channels-last [N][H][W][C] fp32:
//   nn.MaxPool2d(2)                                             networks/unet.py:36
//   nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)   networks/unet.py:50
//   torch.cat([skip, up], dim=1)                                networks/unet.py:56
// The upsample writes straight into its half of the concat buffer (row stride ld, channel offset),
// so the concat never costs an extra pass; its backward is a deterministic gather (no atomics).
#include "common.h"
#include "../../include/bcp_hip.h"

namespace bcp {

// ldx: row stride of x in floats -- C, or the width of the concat buffer whose first C channels x is (round 4: an encoder block's
// output lives only there; no copy into the concat buffer)
// amax_src / amax_dst (round 5): x is a skip tensor living in the leading channels of a concat buffer; the buffer gets |max| slots OF ITS
// OWN, started here as a copy of x's (final since x's producing pass; this launch sits between that pass and the upsample that
// max-reduces its half into them).  x's slots stay what its producer left: every reader of x -- the next encoder conv through the pooled
// tensor, its weight-gradient kernel a backward pass later -- sees ONE scale (they used to be the concat buffer's slots as well).
__global__ __launch_bounds__(256) void k_maxpool2d_fwd(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W,
                                                       int C, int ldx, const float* __restrict__ amax_src, float* __restrict__ amax_dst) {
  if (amax_dst && blockIdx.x == 0 && threadIdx.x < kAmaxSlots) amax_dst[threadIdx.x * kAmaxStride] = amax_src[threadIdx.x * kAmaxStride];
  const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
  const long long total = (long long)N * Ho * Wo * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const int wo = (int)((i / C4) % Wo), ho = (int)((i / ((long long)C4 * Wo)) % Ho), n = (int)(i / ((long long)C4 * Wo * Ho));
    const float* p = x + ((((long long)n * H + 2 * ho) * W + 2 * wo) * ldx) + c4 * 4;
    const float4 a = ld4(p), b = ld4(p + ldx), c = ld4(p + (long long)W * ldx), d = ld4(p + (long long)W * ldx + ldx);
    float4 o;
    o.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x));
    o.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
    o.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z));
    o.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
    st4(y + i * 4, o);
  }
}

// nn.MaxPool3d(3, stride=2), no padding (networks/VNet.py:246,288: the V-Net's second return value, pool(x5) -- forward only, the
// reference never differentiates it in any train script).  One thread per (output voxel, float4 channel group).
__global__ __launch_bounds__(256) void k_maxpool3d_k3s2_fwd(const float* __restrict__ x, float* __restrict__ y, int N, int D, int H, int W, int C) {
  const int Do = (D - 3) / 2 + 1, Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1, C4 = C >> 2;
  const long long total = (long long)N * Do * Ho * Wo * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long long r = i / C4;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho); r /= Ho;
    const int d_o = (int)(r % Do), n = (int)(r / Do);
    float4 o = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int kd = 0; kd < 3; ++kd)
      for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
          const float4 v = ld4(x + ((((long long)n * D + 2 * d_o + kd) * H + 2 * ho + kh) * W + 2 * wo + kw) * C + c4 * 4);
          // NaN-propagating like torch's MaxPool3d (fmaxf would drop a NaN)
          o.x = (v.x > o.x || v.x != v.x) ? v.x : o.x; o.y = (v.y > o.y || v.y != v.y) ? v.y : o.y;
          o.z = (v.z > o.z || v.z != v.z) ? v.z : o.z; o.w = (v.w > o.w || v.w != v.w) ? v.w : o.w;
        }
    st4(y + i * 4, o);
  }
}

// gradient goes to the FIRST maximal element of each 2x2 window (row-major), as torch's max_pool2d does.  One thread per (window, float4
// channel group).  add (nullable, row stride ld_add): a second gradient of x joined on the way out -- the decoder-side skip gradient,
// the first C channels of the concat buffer's gradient (dx = scatter + add: the += launch that followed is gone)
__global__ __launch_bounds__(256) void k_maxpool2d_bwd(const float* __restrict__ x, int ldx, const float* __restrict__ dy,
                                                       float* __restrict__ dx, int N, int H, int W, int C, int accumulate,
                                                       const float* __restrict__ add, int ld_add) {
  const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
  const long long total = (long long)N * Ho * Wo * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const int wo = (int)((i / C4) % Wo), ho = (int)((i / ((long long)C4 * Wo)) % Ho), n = (int)(i / ((long long)C4 * Wo * Ho));
    const long long row = ((long long)n * H + 2 * ho) * W + 2 * wo;
    const long long roff[4] = {0, 1, W, (long long)W + 1};
    float xv[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = ld4(x + (row + roff[k]) * ldx + c4 * 4);
      xv[k][0] = v.x; xv[k][1] = v.y; xv[k][2] = v.z; xv[k][3] = v.w;
    }
    const float4 g4 = ld4(dy + i * 4);
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    int best[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int b = 0;
      float bv = xv[0][q];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (xv[k][q] > bv) { bv = xv[k][q]; b = k; }
      best[q] = b;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = (best[q] == k) ? g[q] : 0.f;
      float* dp = dx + (row + roff[k]) * C + c4 * 4;
      if (accumulate) { const float4 t = ld4(dp); o[0] = t.x + o[0]; o[1] = t.y + o[1]; o[2] = t.z + o[2]; o[3] = t.w + o[3]; }
      if (add) { const float4 t = ld4(add + (row + roff[k]) * ld_add + c4 * 4); o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w; }
      st4(dp, make_float4(o[0], o[1], o[2], o[3]));
    }
  }
}

struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_ac(int o, int in, int out) {  // align_corners=True source coordinate (torch upsample math, fp32)
#pragma clang fp contract(off)
  const float scale = (out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f;
  // the source coordinate is ROUNDED to fp32 before the integer part is taken off (torch: area_pixel_compute_source_index, then
  // `lambda = src - floor`): no contraction of the product into the subtraction in here (the pragma; the _rn intrinsics alone do not stop
  // -ffp-contract=fast, see elementwise.hip) -- with the fused form the weights of
  // a 128 -> 256 upsample moved by up to an ulp of `src` (2.7e-5 in the output against torch; round 5, when the kernel lost its
  // SLP-packed arithmetic and the contraction became possible)
  const float src = scale * (float)o;
  Lerp r;
  r.i0 = (int)src;
  r.i1 = r.i0 + ((r.i0 < in - 1) ? 1 : 0);
  r.l1 = src - (float)r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}

__global__ __launch_bounds__(256) void k_bilinear2x_fwd(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W,
                                                        int C, int ldy, int y_off, float* __restrict__ amax_io) {
  float amax = 0.f;            // max |o| of what this thread writes, max-reduced INTO amax_io (round 4: see bcp_copy_channels)
  const int Ho = 2 * H, Wo = 2 * W, C4 = C >> 2;
  const long long total = (long long)N * Ho * Wo * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const int wo = (int)((i / C4) % Wo), ho = (int)((i / ((long long)C4 * Wo)) % Ho), n = (int)(i / ((long long)C4 * Wo * Ho));
    const Lerp lh = lerp_ac(ho, H, Ho), lw = lerp_ac(wo, W, Wo);
    const float* p = x + (long long)n * H * W * C + c4 * 4;
    const float4 a = ld4(p + ((long long)lh.i0 * W + lw.i0) * C), b = ld4(p + ((long long)lh.i0 * W + lw.i1) * C);
    const float4 c = ld4(p + ((long long)lh.i1 * W + lw.i0) * C), d = ld4(p + ((long long)lh.i1 * W + lw.i1) * C);
    float4 o;
    o.x = lh.l0 * (lw.l0 * a.x + lw.l1 * b.x) + lh.l1 * (lw.l0 * c.x + lw.l1 * d.x);
    o.y = lh.l0 * (lw.l0 * a.y + lw.l1 * b.y) + lh.l1 * (lw.l0 * c.y + lw.l1 * d.y);
    o.z = lh.l0 * (lw.l0 * a.z + lw.l1 * b.z) + lh.l1 * (lw.l0 * c.z + lw.l1 * d.z);
    o.w = lh.l0 * (lw.l0 * a.w + lw.l1 * b.w) + lh.l1 * (lw.l0 * c.w + lw.l1 * d.w);
    st4(y + (((long long)n * Ho + ho) * Wo + wo) * ldy + y_off + c4 * 4, o);
    // NaN-propagating maximum (fmaxf drops NaNs -- also the one a previous iteration stored): once NaN, `t > amax` is false for every
    // later t and the NaN stays; block_amax_publish forwards it to the slot
    const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float t = fabsf(ov[k]); amax = (t > amax || t != t) ? t : amax; }
  }
  if (amax_io) block_amax_publish(amax, amax_io);
}

// dx[h][w] = sum over output pixels whose stencil touches (h, w) of weight * dy  (gather; deterministic)
__global__ __launch_bounds__(256) void k_bilinear2x_bwd(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W,
                                                        int C, int lddy, int dy_off) {
  const int Ho = 2 * H, Wo = 2 * W, C4 = C >> 2;
  const long long total = (long long)N * H * W * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const int w = (int)((i / C4) % W), h = (int)((i / ((long long)C4 * W)) % H), n = (int)(i / ((long long)C4 * W * H));
    // output rows / columns whose source coordinate lies in (h-1, h+1): o in [2h-3, 2h+4] is a safe superset for the
    // align_corners scale (H-1)/(2H-1) in (0, 0.5]; the 8 row and 8 column weights are evaluated once each
    float wr[8], wc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ho = 2 * h - 3 + k, wo = 2 * w - 3 + k;
      float a = 0.f, b = 0.f;
      if (ho >= 0 && ho < Ho) {
        const Lerp l = lerp_ac(ho, H, Ho);
        if (l.i0 == h) a += l.l0;
        if (l.i1 == h) a += l.l1;
      }
      if (wo >= 0 && wo < Wo) {
        const Lerp l = lerp_ac(wo, W, Wo);
        if (l.i0 == w) b += l.l0;
        if (l.i1 == w) b += l.l1;
      }
      wr[k] = a;
      wc[k] = b;
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (wr[k] == 0.f) continue;
      const float* row = dy + (((long long)n * Ho + (2 * h - 3 + k)) * Wo) * lddy + dy_off + c4 * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (wc[j] == 0.f) continue;
        const float4 v = ld4(row + (long long)(2 * w - 3 + j) * lddy);
        const float f = wr[k] * wc[j];
        s.x += f * v.x; s.y += f * v.y; s.z += f * v.z; s.w += f * v.w;
      }
    }
    st4(dx + i * 4, s);
  }
}

__global__ __launch_bounds__(256) void k_copy_channels(const float* __restrict__ src, float* __restrict__ dst, long long rows, int C,
                                                       int ld_src, int src_off, int ld_dst, int dst_off, int accumulate,
                                                       const float* __restrict__ amax_src, float* __restrict__ amax_dst) {
  if (amax_dst && blockIdx.x == 0 && threadIdx.x < kAmaxSlots)      // the concat buffer's slots start as the skip tensor's
    amax_dst[threadIdx.x * kAmaxStride] = amax_src ? amax_src[threadIdx.x * kAmaxStride] : 0.f;
  const int C4 = C >> 2;
  const long long total = rows * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C4;
    const int c4 = (int)(i - r * C4);
    float4 v = ld4(src + r * ld_src + src_off + c4 * 4);
    float* d = dst + r * ld_dst + dst_off + c4 * 4;
    if (accumulate) {
      const float4 o = ld4(d);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    st4(d, v);
  }
}

static inline int sgrid(long long n) {
  long long g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace bcp

using namespace bcp;

extern "C" int bcp_maxpool2d_fwd(const float* x, int ldx, float* y, int N, int H, int W, int C, const float* amax_src_or_null,
                                 float* amax_dst_or_null, void* stream) {
  if (ldx == 0) ldx = C;
  BCP_REQUIRE(x && y && N > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && ldx >= C && ldx % 4 == 0 && aligned16(x), "bcp_maxpool2d_fwd: bad argument");
  BCP_REQUIRE((amax_src_or_null == nullptr) == (amax_dst_or_null == nullptr), "bcp_maxpool2d_fwd: the |max| slot copy needs source and destination");
  hipLaunchKernelGGL(k_maxpool2d_fwd, dim3(sgrid((long long)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y,
                     N, H, W, C, ldx, amax_src_or_null, amax_dst_or_null);
  BCP_CHECK_LAUNCH("bcp_maxpool2d_fwd");
  return BCP_OK;
}
extern "C" int bcp_maxpool3d_k3s2_fwd(const float* x, float* y, int N, int D, int H, int W, int C, void* stream) {
  BCP_REQUIRE(x && y && N > 0 && D >= 3 && H >= 3 && W >= 3 && C >= 4 && (C & 3) == 0, "bcp_maxpool3d_k3s2_fwd: need D,H,W >= 3 and C %% 4 == 0");
  const long long total = (long long)N * ((D - 3) / 2 + 1) * ((H - 3) / 2 + 1) * ((W - 3) / 2 + 1) * (C / 4);
  hipLaunchKernelGGL(k_maxpool3d_k3s2_fwd, dim3((int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x, y, N, D, H, W, C);
  BCP_CHECK_LAUNCH("bcp_maxpool3d_k3s2_fwd");
  return BCP_OK;
}

extern "C" int bcp_maxpool2d_bwd(const float* x, int ldx, const float* dy, float* dx, int N, int H, int W, int C, int accumulate,
                                 const float* add_or_null, int ld_add, void* stream) {
  if (ldx == 0) ldx = C;
  if (add_or_null && ld_add == 0) ld_add = C;
  BCP_REQUIRE(x && dy && dx && N > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && ldx >= C && ldx % 4 == 0, "bcp_maxpool2d_bwd: bad argument");
  BCP_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dx) && (!add_or_null || (aligned16(add_or_null) && ld_add >= C && ld_add % 4 == 0)),
              "bcp_maxpool2d_bwd: alignment / add stride");
  hipLaunchKernelGGL(k_maxpool2d_bwd, dim3(sgrid((long long)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, dy, dx,
                     N, H, W, C, accumulate, add_or_null, ld_add);
  BCP_CHECK_LAUNCH("bcp_maxpool2d_bwd");
  return BCP_OK;
}
extern "C" int bcp_bilinear2x_fwd(const float* x, float* y, int N, int H, int W, int C, int ldy, int y_off, float* amax_io_or_null, void* stream) {
  BCP_REQUIRE(x && y && N > 0 && C % 4 == 0 && ldy % 4 == 0 && y_off % 4 == 0, "bcp_bilinear2x_fwd: bad argument");
  hipLaunchKernelGGL(k_bilinear2x_fwd, dim3(sgrid((long long)N * 4 * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W,
                     C, ldy, y_off, amax_io_or_null);
  BCP_CHECK_LAUNCH("bcp_bilinear2x_fwd");
  return BCP_OK;
}
extern "C" int bcp_bilinear2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, int lddy, int dy_off, void* stream) {
  BCP_REQUIRE(dy && dx && N > 0 && C % 4 == 0 && lddy % 4 == 0 && dy_off % 4 == 0, "bcp_bilinear2x_bwd: bad argument");
  hipLaunchKernelGGL(k_bilinear2x_bwd, dim3(sgrid((long long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, dx, N, H, W, C, lddy,
                     dy_off);
  BCP_CHECK_LAUNCH("bcp_bilinear2x_bwd");
  return BCP_OK;
}
extern "C" int bcp_copy_channels(const float* src, float* dst, long long rows, int C, int ld_src, int src_off, int ld_dst, int dst_off,
                                 int accumulate, const float* amax_src_or_null, float* amax_dst_or_null, void* stream) {
  BCP_REQUIRE(src && dst && rows > 0 && C % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && src_off % 4 == 0 && dst_off % 4 == 0,
              "bcp_copy_channels: bad argument");
  hipLaunchKernelGGL(k_copy_channels, dim3(sgrid(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, src, dst, rows, C, ld_src, src_off,
                     ld_dst, dst_off, accumulate, amax_src_or_null, amax_dst_or_null);
  BCP_CHECK_LAUNCH("bcp_copy_channels");
  return BCP_OK;
}
