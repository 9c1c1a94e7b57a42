// bcp_amd/csrc/eval.hip -- validation path on the device: eval-mode BatchNorm, the sliding-window
// accumulation of utils/test_3d_patch.py:test_single_case, and the Dice metric.
//
// Reference: nn.BatchNorm3d/2d in eval() mode (running statistics; networks/VNet.py:18-26 under model.eval()),
// test_single_case (utils/test_3d_patch.py:82-141): per patch softmax -> class-1 probability added into score_map and
// a visit count, score_map / cnt, label = score > 0.5; medpy.metric.binary.dc (2|A&B| / (|A| + |B|)).
#include "common.h"
#include "../../include/bcp_hip.h"

namespace bcp {

// a = act((y - running_mean) * gamma / sqrt(running_var + eps) + beta) [+ residual]
__global__ __launch_bounds__(256) void k_norm_eval(const float* __restrict__ y, long long rows, int C, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ rmean,
                                                   const float* __restrict__ rvar, float eps, int act,
                                                   const float* __restrict__ residual, float* __restrict__ out) {
  const int C4 = C >> 2;
  const int col = threadIdx.x % C4;             // 256 % C4 == 0 and the stride below is a multiple of C4: fixed column
  const float4 mu = ld4(rmean + col * 4), va = ld4(rvar + col * 4);
  float4 ga = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
  if (gamma) ga = ld4(gamma + col * 4);
  if (beta) be = ld4(beta + col * 4);
  // torch: invstd = 1 / sqrt(var + eps) in the compute type (fp32), then (x - mean) * invstd * weight + bias
  const float sx = ga.x * (1.f / sqrtf(va.x + eps)), sy = ga.y * (1.f / sqrtf(va.y + eps));
  const float sz = ga.z * (1.f / sqrtf(va.z + eps)), sw = ga.w * (1.f / sqrtf(va.w + eps));
  const long long nv = rows * C4, stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) {
    const float4 v = ld4(y + i * 4);
    float4 o = make_float4(act_fwd((v.x - mu.x) * sx + be.x, act), act_fwd((v.y - mu.y) * sy + be.y, act),
                           act_fwd((v.z - mu.z) * sz + be.z, act), act_fwd((v.w - mu.w) * sw + be.w, act));
    if (residual) {
      const float4 r = ld4(residual + i * 4);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    st4(out + i * 4, o);
  }
}

// score[x0+i][y0+j][z0+k] += softmax(logits[i][j][k][:])[cls];  cnt[...] += 1        (one patch, C in {2, 4})
template <int C>
__global__ __launch_bounds__(256) void k_sw_accumulate(const float* __restrict__ logits, float* __restrict__ score,
                                                       float* __restrict__ cnt, int Y, int Z, int px, int py, int pz, int x0,
                                                       int y0, int z0, int cls) {
  const long long n = (long long)px * py * pz;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % pz), j = (int)((i / pz) % py), ii = (int)(i / ((long long)pz * py));
    float l[C];
    float m = -3.4e38f;
#pragma unroll
    for (int c = 0; c < C; ++c) { l[c] = logits[i * C + c]; m = fmaxf(m, l[c]); }
    float den = 0.f, num = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { const float e = expf(l[c] - m); den += e; if (c == cls) num = e; }
    const long long o = ((long long)(x0 + ii) * Y + (y0 + j)) * Z + (z0 + k);
    score[o] += num / den;
    cnt[o] += 1.f;
  }
}

// score /= cnt;  label = score > thres
__global__ __launch_bounds__(256) void k_sw_finish(float* __restrict__ score, const float* __restrict__ cnt, uint8_t* __restrict__ label,
                                                   long long n, float thres) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float s = score[i] / cnt[i];
    score[i] = s;
    label[i] = s > thres ? 1 : 0;
  }
}

// counts[0] = |A & B|, counts[1] = |A|, counts[2] = |B|   (A = pred != 0, B = gt != 0, or == cls when cls > 0); integer atomics: deterministic
__global__ __launch_bounds__(256) void k_overlap_counts(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, long long n, int cls,
                                                        unsigned long long* __restrict__ counts) {
  unsigned long long c0 = 0, c1 = 0, c2 = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const bool x = cls ? a[i] == cls : a[i] != 0, y = cls ? b[i] == cls : b[i] != 0;
    c0 += (x && y); c1 += x; c2 += y;
  }
  c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counts[0], c0); atomicAdd(&counts[1], c1); atomicAdd(&counts[2], c2);
  }
}

static inline int egrid(long long n) {
  long long g = (n + 255) / 256;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace bcp

using namespace bcp;

extern "C" int bcp_norm_eval(const float* y, long long rows, int C, const float* gamma, const float* beta, const float* running_mean,
                             const float* running_var, float eps, int act, const float* residual, float* out, void* stream) {
  BCP_REQUIRE(y && running_mean && running_var && out && rows > 0, "bcp_norm_eval: bad argument");
  BCP_REQUIRE(C >= 16 && C <= 1024 && (C & (C - 1)) == 0, "bcp_norm_eval: C=%d unsupported (need a power of two in 16..1024)", C);
  hipLaunchKernelGGL(k_norm_eval, dim3(egrid(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, y, rows, C, gamma, beta, running_mean,
                     running_var, eps, act, residual, out);
  BCP_CHECK_LAUNCH("bcp_norm_eval");
  return BCP_OK;
}

extern "C" int bcp_sw_accumulate(const float* logits, float* score, float* cnt, int X, int Y, int Z, int px, int py, int pz, int x0, int y0,
                                 int z0, int C, int cls, void* stream) {
  BCP_REQUIRE(logits && score && cnt, "bcp_sw_accumulate: null pointer");
  BCP_REQUIRE(x0 >= 0 && y0 >= 0 && z0 >= 0 && x0 + px <= X && y0 + py <= Y && z0 + pz <= Z && cls >= 0 && cls < C,
              "bcp_sw_accumulate: patch outside the volume");
  const long long n = (long long)px * py * pz;
  if (C == 2) hipLaunchKernelGGL((k_sw_accumulate<2>), dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, logits, score, cnt, Y, Z, px, py, pz, x0, y0, z0, cls);
  else if (C == 4) hipLaunchKernelGGL((k_sw_accumulate<4>), dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, logits, score, cnt, Y, Z, px, py, pz, x0, y0, z0, cls);
  else BCP_REQUIRE(false, "bcp_sw_accumulate: C=%d unsupported (2 or 4)", C);
  BCP_CHECK_LAUNCH("bcp_sw_accumulate");
  return BCP_OK;
}

extern "C" int bcp_sw_finish(float* score, const float* cnt, uint8_t* label, long long n, float thres, void* stream) {
  BCP_REQUIRE(score && cnt && label && n > 0, "bcp_sw_finish: bad argument");
  hipLaunchKernelGGL(k_sw_finish, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, score, cnt, label, n, thres);
  BCP_CHECK_LAUNCH("bcp_sw_finish");
  return BCP_OK;
}

// counts: device uint64[3], zeroed here
extern "C" int bcp_overlap_counts(const uint8_t* pred, const uint8_t* gt, long long n, int cls, unsigned long long* counts, void* stream) {
  BCP_REQUIRE(pred && gt && counts && n > 0 && cls >= 0 && cls < 256, "bcp_overlap_counts: bad argument");
  hipMemsetAsync(counts, 0, 3 * sizeof(unsigned long long), (hipStream_t)stream);
  hipLaunchKernelGGL(k_overlap_counts, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, pred, gt, n, cls, counts);
  BCP_CHECK_LAUNCH("bcp_overlap_counts");
  return BCP_OK;
}

// ------------------------------------------------------------------------------------------------
// Device-side input pipeline for LA: RandomRotFlip + RandomCrop (dataloaders/dataset.py:52-59, 173-214)
// as ONE gather: dst[i][j][l] = pad(flip(rot90(src, k), axis))[w1+i][h1+j][d1+l], zero in the padding.
// ------------------------------------------------------------------------------------------------
namespace bcp {
template <typename T>
__global__ __launch_bounds__(256) void k_crop_rotflip(const T* __restrict__ src, T* __restrict__ dst, int n0, int n1, int n2, int k,
                                                      int axis, int pw, int ph, int pd, int w1, int h1, int d1, int P0, int P1, int P2) {
  const int b0 = (k & 1) ? n1 : n0, b1 = (k & 1) ? n0 : n1;      // shape of rot90(src, k) in the (0, 1) plane
  const long long n = (long long)P0 * P1 * P2;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
    const int l = (int)(q % P2), j = (int)((q / P2) % P1), i = (int)(q / ((long long)P2 * P1));
    int a = i + w1 - pw, b = j + h1 - ph;                          // coordinates in flip(rot90(src))
    const int c = l + d1 - pd;
    T v = 0;
    if (a >= 0 && a < b0 && b >= 0 && b < b1 && c >= 0 && c < n2) {
      if (axis == 0) a = b0 - 1 - a; else if (axis == 1) b = b1 - 1 - b;   // undo np.flip (axis -1: no flip)
      int s0, s1;                                                  // undo np.rot90(m, k, axes=(0, 1))
      switch (k & 3) {
        case 0: s0 = a; s1 = b; break;
        case 1: s0 = b; s1 = n1 - 1 - a; break;
        case 2: s0 = n0 - 1 - a; s1 = n1 - 1 - b; break;
        default: s0 = n0 - 1 - b; s1 = a; break;
      }
      v = src[((long long)s0 * n1 + s1) * n2 + c];
    }
    dst[q] = v;
  }
}
}  // namespace bcp

extern "C" int bcp_crop_rotflip(const void* src, void* dst, int elem_bytes, int n0, int n1, int n2, int k, int flip_axis, int pw, int ph,
                                int pd, int w1, int h1, int d1, int P0, int P1, int P2, void* stream) {
  BCP_REQUIRE(src && dst && n0 > 0 && n1 > 0 && n2 > 0 && P0 > 0 && P1 > 0 && P2 > 0, "bcp_crop_rotflip: bad argument");
  BCP_REQUIRE((flip_axis >= -1 && flip_axis <= 1) && k >= 0 && k < 4 && (elem_bytes == 4 || elem_bytes == 1), "bcp_crop_rotflip: bad mode");
  const long long n = (long long)P0 * P1 * P2;
  if (elem_bytes == 4)
    hipLaunchKernelGGL((k_crop_rotflip<float>), dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, (const float*)src, (float*)dst, n0, n1,
                       n2, k, flip_axis, pw, ph, pd, w1, h1, d1, P0, P1, P2);
  else
    hipLaunchKernelGGL((k_crop_rotflip<uint8_t>), dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, (uint8_t*)dst, n0,
                       n1, n2, k, flip_axis, pw, ph, pd, w1, h1, d1, P0, P1, P2);
  BCP_CHECK_LAUNCH("bcp_crop_rotflip");
  return BCP_OK;
}

// ------------------------------------------------------------------------------------------------
// Device-side input pipeline for ACDC: RandomGenerator of dataloaders/dataset.py:52-88 -- either
// rot90 + flip, or a rotation by a whole number of degrees (scipy.ndimage.rotate, order=0, reshape=False, cval 0), or nothing --
// followed by the nearest-neighbour zoom to the training resolution (scipy.ndimage.zoom, order=0) as ONE gather.
// The coordinate arithmetic restates scipy 1.15's NI_ZoomShift / NI_GeometricTransform for order 0 in fp64, operation for
// operation (no FMA contraction): zoom source = floor(o * (in - 1) / (out - 1) + 0.5); rotation source = floor(c + 0.5) with
// c = ((0 + x * m0) + y * m1) + offset, outside [0, len - 1] -> cval.  Pinned against the reference's class (which calls scipy)
// by tests/golden/aug_acdc.npz.
// ------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
namespace bcp {
struct Affine2 { double m00, m01, m10, m11, o0, o1, zx, zy; };

template <typename T>
__global__ __launch_bounds__(256) void k_acdc_augment(const T* __restrict__ src, T* __restrict__ dst, int H, int W, int mode, int k,
                                                      int axis, Affine2 A, int OH, int OW) {
  const int IH = (mode == 1 && (k & 1)) ? W : H, IW = (mode == 1 && (k & 1)) ? H : W;   // shape after the first stage
  const long long n = (long long)OH * OW;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
    const int u = (int)(q / OW), v = (int)(q % OW);
    // zoom (order 0): nearest source pixel in the intermediate image
    const double cx = (double)u * A.zx, cy = (double)v * A.zy;
    int a = (int)floor(cx + 0.5), b = (int)floor(cy + 0.5);
    a = a < 0 ? 0 : (a > IH - 1 ? IH - 1 : a);
    b = b < 0 ? 0 : (b > IW - 1 ? IW - 1 : b);
    T val = 0;
    if (mode == 0) {
      val = src[(long long)a * W + b];
    } else if (mode == 1) {
      if (axis == 0) a = IH - 1 - a; else b = IW - 1 - b;          // undo np.flip
      int s0, s1;                                                  // undo np.rot90(m, k)
      switch (k & 3) {
        case 0: s0 = a; s1 = b; break;
        case 1: s0 = b; s1 = W - 1 - a; break;
        case 2: s0 = H - 1 - a; s1 = W - 1 - b; break;
        default: s0 = H - 1 - b; s1 = a; break;
      }
      val = src[(long long)s0 * W + s1];
    } else {
      const double x = (double)a, y = (double)b;
      double c0 = 0.0, c1 = 0.0;
      c0 = c0 + x * A.m00; c0 = c0 + y * A.m01; c0 = c0 + A.o0;
      c1 = c1 + x * A.m10; c1 = c1 + y * A.m11; c1 = c1 + A.o1;
      if (!(c0 < 0.0 || c0 > (double)(H - 1) || c1 < 0.0 || c1 > (double)(W - 1))) {
        const int p = (int)floor(c0 + 0.5), r = (int)floor(c1 + 0.5);
        val = src[(long long)p * W + r];
      }
    }
    dst[q] = val;
  }
}
}  // namespace bcp

extern "C" int bcp_acdc_augment(const void* src, void* dst, int elem_bytes, int H, int W, int mode, int k, int flip_axis,
                                const double* affine6, int OH, int OW, void* stream) {
  BCP_REQUIRE(src && dst && H > 1 && W > 1 && OH > 1 && OW > 1, "bcp_acdc_augment: bad extents (every side must be > 1)");
  BCP_REQUIRE(mode >= 0 && mode <= 2 && (elem_bytes == 4 || elem_bytes == 1), "bcp_acdc_augment: bad mode");
  BCP_REQUIRE(mode != 1 || ((flip_axis == 0 || flip_axis == 1) && k >= 0 && k < 4), "bcp_acdc_augment: bad rot90 / flip arguments");
  BCP_REQUIRE(mode != 2 || affine6, "bcp_acdc_augment: the rotation needs its matrix and offset (host pointer to 6 doubles)");
  Affine2 A{1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0};
  if (mode == 2) { A.m00 = affine6[0]; A.m01 = affine6[1]; A.m10 = affine6[2]; A.m11 = affine6[3]; A.o0 = affine6[4]; A.o1 = affine6[5]; }
  const int IH = (mode == 1 && (k & 1)) ? W : H, IW = (mode == 1 && (k & 1)) ? H : W;
  A.zx = (double)(IH - 1) / (double)(OH - 1);
  A.zy = (double)(IW - 1) / (double)(OW - 1);
  const long long n = (long long)OH * OW;
  if (elem_bytes == 4)
    hipLaunchKernelGGL((k_acdc_augment<float>), dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, (const float*)src, (float*)dst, H, W, mode,
                       k, flip_axis, A, OH, OW);
  else
    hipLaunchKernelGGL((k_acdc_augment<uint8_t>), dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, (uint8_t*)dst, H, W,
                       mode, k, flip_axis, A, OH, OW);
  BCP_CHECK_LAUNCH("bcp_acdc_augment");
  return BCP_OK;
}
