// bcp_amd/csrc/conv3_defs.h -- tile geometry, halo fetch and fused-statistics helpers shared by the 3x3(x3) convolution
// kernels (conv3.hip: fp32-MFMA streaming / resident / wgrad kernels; conv3b.hip / conv3bw.hip: the bf16-pipe kernels).
#pragma once
#include "common.h"
#include <type_traits>

namespace bcp {

#ifndef BCP_XS
#define BCP_XS 20
#endif
static constexpr int XS = BCP_XS;  // LDS floats per halo voxel: 16 channels + 4 pad (16-B aligned rows)
// halo row stride of the wgrad kernel (ds_read_b32, lane (li = channel, lg = voxel))
static constexpr int XSW = 20;   // measured: 16 (conflict-free b32 reads) is 6 % SLOWER at C=16 -- LDS conflicts are not what bounds wgrad


template <int KD, int TD, int TH, int TW>
struct Tile {
  static constexpr int M = TD * TH * TW;
  static constexpr int TD_ = TD, TH_ = TH, TW_ = TW;
  static constexpr int MT = M / 64;
  static constexpr int PD = (KD == 3) ? 1 : 0;
  static constexpr int HD = TD + 2 * PD, HH = TH + 2, HW = TW + 2;
  static constexpr int HV = HD * HH * HW;
  static constexpr int T = KD * 9;
  static_assert(M % 64 == 0, "tile must hold a multiple of 64 voxels");
  __device__ static __forceinline__ int voff(int m) {  // halo-local voxel index of tile voxel m at tap (0,0,0)
    const int tw = m % TW, th = (m / TW) % TH, td = m / (TW * TH);
    return (td * HH + th) * HW + tw;
  }
  __device__ static __forceinline__ int tapoff(int tap) {
    const int kw = tap % 3, kh = (tap / 3) % 3, kd = tap / 9;
    return (kd * HH + kh) * HW + kw;
  }
};

struct ConvDims {
  int N, D, H, W;
  int Cin, Cout;        // real channel counts (row strides of X and Y)
  int Cin16, Cout16;    // padded to multiples of 16 (packed-weight extents)
  int tiles_d, tiles_h, tiles_w;
  int xcd;              // != 0: XCD-aware workgroup -> tile order in the bf16-pipe kernels (option conv3_xcd)
  const float* yamax;   // weight gradient only: max |dy| of the second operand (bcp_conv3_wgrad ... dy_amax); see xamax
  const float* xamax;   // device float: max |x| of the input tensor (from the norm apply pass that wrote it), or NULL.  Non-NULL selects the
                        // two-plane fp16 instances where they exist (option conv3_f16); the three-plane bf16 ones need no scale
};

// XCD-aware tile order for one-tile-per-workgroup grids.  Workgroups are dealt round-robin to the 8 XCDs in linear-id order, so
// workgroup x of a grid row whose first workgroup has linear id `row_lin` runs on XCD (x + row_lin) % 8.  The workgroups of XCD c
// get a CONTIGUOUS range of the row's tile list (x-th of its class -> start_c + x / 8): tiles in flight on an XCD are spatial
// neighbours and their halo overlap (2.3x per brick tile, ~6x per flat deep-level tile) hits that XCD's own 4 MB L2 instead of
// the fabric.  A bijection of [0, gx) for every gx (classes hold gx / 8 or gx / 8 + 1 tiles).  All SALU (block-uniform).
__device__ __forceinline__ int xcd_tile(int x, int gx, int row_lin) {
  const int o = row_lin & 7, c = (x + o) & 7;
  int start = 0;
#pragma unroll
  for (int cc = 0; cc < 7; ++cc) {
    const int x0 = (cc - o) & 7;
    if (cc < c && x0 < gx) start += (gx - x0 + 7) >> 3;
  }
  return start + (x >> 3);
}

__device__ __forceinline__ void tile_origin(const ConvDims& cd, int bx, int TD, int TH, int TW, int& n, int& d0, int& h0,
                                            int& w0) {
  const int tw = bx % cd.tiles_w;
  const int th = (bx / cd.tiles_w) % cd.tiles_h;
  const int td = (bx / (cd.tiles_w * cd.tiles_h)) % cd.tiles_d;
  n = bx / (cd.tiles_w * cd.tiles_h * cd.tiles_d);
  d0 = td * TD;
  h0 = th * TH;
  w0 = tw * TW;
}

// Halo fetch with launch-invariant indexing.  A halo "row" is one (hd, hh) line of HW voxels = RW float4 columns; a
// pass moves RPP rows with threads (r0, col).  A thread's column is fixed for the whole launch, so everything but the
// tile origin is computed once: the per-pass global offsets grel[] and the LDS slot; per tile there is one uniform
// 64-bit row-validity mask (SALU).  Per-tile vector work of a fetch: ~2 VALU per float4 (was ~25: div/mod of the flat
// index + three range checks + 64-bit address arithmetic per element).
template <class TL, int XSP = XS>
struct HaloFetch {
  static constexpr int RW = TL::HW * 4, RPP = 256 / RW, HR = TL::HD * TL::HH, NP = (HR + RPP - 1) / RPP;
  static_assert(HR <= 128 && RW <= 256, "halo rows must fit the 128-bit validity mask");
  int r0, hw, part;
  bool act;
  unsigned grel[NP];
  float* lds;

  __device__ __forceinline__ void init(const ConvDims& cd, float* Xs, int tid = -1) {
    if (tid < 0) tid = threadIdx.x;      // 256 fetching threads; the wave-specialised kernel passes its helper-local id
    r0 = tid / RW;
    const int col = tid - r0 * RW;
    hw = col >> 2;
    part = col & 3;
    act = tid < RPP * RW;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int row = u * RPP + r0, hd = row / TL::HH, hh = row - hd * TL::HH;
      grel[u] = (unsigned)(((hd * cd.H + hh) * cd.W + hw) * cd.Cin + part * 4);
    }
    lds = Xs + (r0 * TL::HW + hw) * XSP + part * 4;
  }
  __device__ __forceinline__ void stash_to(float* Xbuf, const float* Xbase, const float4 (&pre)[NP]) const {   // other halo buffer
#pragma unroll
    for (int u = 0; u < NP; ++u)
      if (act && u * RPP + r0 < HR) st4(Xbuf + (lds - Xbase) + u * RPP * TL::HW * XSP, pre[u]);
  }
  // halo of the tile at (n, d0, h0, w0), cin chunk c -> registers; zero outside the volume / beyond Cin
  __device__ __forceinline__ void fetch(const float* __restrict__ X, const ConvDims& cd, int n, int d0, int h0, int w0, int c,
                                        float4 (&pre)[NP]) const {
    // uniform: valid hh range, valid hd range -> one bit per halo row
    const int hlo = (h0 >= 1) ? 0 : 1 - h0, hhi = (cd.H - h0 + 1 < TL::HH) ? cd.H - h0 + 1 : TL::HH;
    const int dlo = (d0 >= TL::PD) ? 0 : TL::PD - d0, dhi = (cd.D - d0 + TL::PD < TL::HD) ? cd.D - d0 + TL::PD : TL::HD;
    const unsigned mh = (hhi > hlo) ? (((1u << hhi) - 1u) & ~((1u << hlo) - 1u)) : 0u;
    unsigned long long M0 = 0, M1 = 0;   // bit (hd * HH + hh), rows 0..63 / 64..127
#pragma unroll
    for (int hd = 0; hd < TL::HD; ++hd) {
      constexpr int HHc = TL::HH;
      const int pos = hd * HHc;
      if (hd >= dlo && hd < dhi) {
        if (pos < 64) M0 |= (unsigned long long)mh << pos;
        if (pos < 64 && pos + HHc > 64) M1 |= (unsigned long long)mh >> (64 - pos);
        if (pos >= 64) M1 |= (unsigned long long)mh << (pos - 64);
      }
    }
    const bool col_ok = act && (unsigned)(w0 - 1 + hw) < (unsigned)cd.W && c * 16 + part * 4 < cd.Cin;
    unsigned long long Mt0 = M0 >> r0, Mt1 = 0;
    if (HR > 64) {
      if (r0) Mt0 |= M1 << (64 - r0);
      Mt1 = M1 >> r0;
    }
    if (!col_ok) { Mt0 = 0; Mt1 = 0; }
    const float* xb = X + ((((long long)n * cd.D + (d0 - TL::PD)) * cd.H + (h0 - 1)) * cd.W + (w0 - 1)) * cd.Cin + c * 16;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (((u * RPP < 64 ? Mt0 >> ((u * RPP) & 63) : Mt1 >> ((u * RPP - 64) & 63)) & 1ull)) v = ld4(xb + grel[u]);
      pre[u] = v;
    }
  }
  // Branch-free variant: EVERY load is issued (rows outside the volume read the tensor's first element instead) and the
  // returned bit mask says which registers hold halo data; the caller applies it when the registers are consumed.  `if (valid)
  // v = load` compiles to a branch per row, and hipcc's wait-count pass, which does not follow paths, then drains vmcnt in front
  // of the block -- fatal for a prefetch issued in the middle of a pipelined loop.  Offsets are 32-bit (tensors < 2^32 floats).
  __device__ __forceinline__ unsigned fetch_nb(const float* __restrict__ X, const ConvDims& cd, int n, int d0, int h0, int w0, int c,
                                               float4 (&pre)[NP]) const {
    static_assert(NP <= 32, "validity bits of a fetch must fit 32 bits");
    const int hlo = (h0 >= 1) ? 0 : 1 - h0, hhi = (cd.H - h0 + 1 < TL::HH) ? cd.H - h0 + 1 : TL::HH;
    const int dlo = (d0 >= TL::PD) ? 0 : TL::PD - d0, dhi = (cd.D - d0 + TL::PD < TL::HD) ? cd.D - d0 + TL::PD : TL::HD;
    const unsigned mh = (hhi > hlo) ? (((1u << hhi) - 1u) & ~((1u << hlo) - 1u)) : 0u;
    unsigned long long M0 = 0, M1 = 0;
#pragma unroll
    for (int hd = 0; hd < TL::HD; ++hd) {
      constexpr int HHc = TL::HH;
      const int pos = hd * HHc;
      if (hd >= dlo && hd < dhi) {
        if (pos < 64) M0 |= (unsigned long long)mh << pos;
        if (pos < 64 && pos + HHc > 64) M1 |= (unsigned long long)mh >> (64 - pos);
        if (pos >= 64) M1 |= (unsigned long long)mh << (pos - 64);
      }
    }
    const bool col_ok = act && (unsigned)(w0 - 1 + hw) < (unsigned)cd.W && c * 16 + part * 4 < cd.Cin;
    unsigned long long Mt0 = M0 >> r0, Mt1 = 0;
    if (HR > 64) {
      if (r0) Mt0 |= M1 << (64 - r0);
      Mt1 = M1 >> r0;
    }
    if (!col_ok) { Mt0 = 0; Mt1 = 0; }
    const unsigned boff = (unsigned)(((((long long)n * cd.D + (d0 - TL::PD)) * cd.H + (h0 - 1)) * cd.W + (w0 - 1)) * cd.Cin + c * 16);
    unsigned vm = 0;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const unsigned bit = (unsigned)((u * RPP < 64 ? Mt0 >> ((u * RPP) & 63) : Mt1 >> ((u * RPP - 64) & 63)) & 1ull);
      const unsigned off = bit ? boff + grel[u] : 0u;
      pre[u] = ld4(X + off);
      vm |= bit << u;
    }
    return vm;
  }
  __device__ __forceinline__ void stash(const float4 (&pre)[NP]) const {
#pragma unroll
    for (int u = 0; u < NP; ++u)
      if (act && u * RPP + r0 < HR) st4(lds + u * RPP * TL::HW * XSP, pre[u]);
  }
};

// Fused BatchNorm / InstanceNorm statistics: the conv epilogue already holds y = conv + bias in registers, so the
// per-channel (sum, sum of squares) partials the norm needs are produced here instead of by a second pass over y
// (csrc/norm.hip k_col_partial<0>).  partial[g][row][C][2] doubles, fp64 accumulation as in the standalone pass.
struct StatsArg {
  double* partial;       // nullptr: disabled
  int rows;              // rows per group (nb of the norm finalize)
  int tiles_per_group;   // spatial tiles per normalisation group (tiles are sample-major)
  int C;                 // channel count of the partial rows (= Cout)
  int G;                 // normalisation groups
  // BACKWARD statistics (bf16-pipe dgrad kernels, bcp_conv3_dgrad_bwdstats): the conv output IS da of the consumer's norm layer; with
  // that layer's pre-norm tensor `by` (same [voxel][C] layout) and statistics table the epilogue accumulates (sum dz, sum dz * xhat),
  // dz = da * act'(z) -- what k_col_partial<1> would re-read y and da for.  by == nullptr: forward statistics (sum y, sum y^2).
  const float* by;
  const float* bstats;   // float[5][G][C]: mean, rstd, scale, shift(, unbiased variance)
  int bact;
};
struct BwdStatsIn { const float* y; const float* stats; int act; };     // host side of StatsArg's backward fields

template <int MODE>
__device__ __forceinline__ void stat_add(double& s1, double& s2, float v) {
  if (MODE == 1) {
    s1 += (double)v;
    s2 += (double)v * (double)v;
  }
}

// block-wide sum over the 4 lg lane groups and the 4 waves, then one (s1, s2) pair per channel of the slab
template <int NT>
__device__ __forceinline__ void stats_flush(double (&s1)[NT], double (&s2)[NT], double* __restrict__ Ss /* [4][NT*16][2] */,
                                            double* __restrict__ dst_row /* &partial[g][row][0][0] */, int cout0, int Cout) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    double a = s1[nt], b = s2[nt];
    a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
    b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
    if (lg == 0) { Ss[(wave * NT * 16 + nt * 16 + li) * 2] = a; Ss[(wave * NT * 16 + nt * 16 + li) * 2 + 1] = b; }
    s1[nt] = 0.0; s2[nt] = 0.0;
  }
  __syncthreads();
  if ((int)threadIdx.x < NT * 16 && cout0 + (int)threadIdx.x < Cout) {
    const int c = threadIdx.x;
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { a += Ss[(w * NT * 16 + c) * 2]; b += Ss[(w * NT * 16 + c) * 2 + 1]; }
    dst_row[(cout0 + c) * 2] = a;
    dst_row[(cout0 + c) * 2 + 1] = b;
  }
  __syncthreads();
}

// transposed accumulators (lane = voxel li, channels lg*4 + r): sum over the 16 voxel lanes, then over the 4 waves
template <int NT>
__device__ __forceinline__ void stats_flush_t(double (&s1)[NT][4], double (&s2)[NT][4], double* __restrict__ Ss /* [4][NT*16][2] */,
                                              double* __restrict__ dst_row /* &partial[g][row][0][0] */, int cout0, int Cout) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double a = s1[nt][r], b = s2[nt][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
      if (li == 0) { Ss[(wave * NT * 16 + nt * 16 + lg * 4 + r) * 2] = a; Ss[(wave * NT * 16 + nt * 16 + lg * 4 + r) * 2 + 1] = b; }
      s1[nt][r] = 0.0; s2[nt][r] = 0.0;
    }
  __syncthreads();
  if ((int)threadIdx.x < NT * 16 && cout0 + (int)threadIdx.x < Cout) {
    const int c = threadIdx.x;
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { a += Ss[(w * NT * 16 + c) * 2]; b += Ss[(w * NT * 16 + c) * 2 + 1]; }
    dst_row[(cout0 + c) * 2] = a;
    dst_row[(cout0 + c) * 2 + 1] = b;
  }
  __syncthreads();
}

// x -> three bf16 pieces, largest first: x = p0 + p1 + p2 up to 2^-26 |x| (round to nearest even at every level; the two
// subtractions are exact).  Integer arithmetic: bit-identical to v_cvt_pk_bf16_f32 for finite values.
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split3_bf16(float x, unsigned short (&p)[3]) {
  p[0] = f32_to_bf16_rne(x);
  float r = x - bf16_to_f32(p[0]);
  p[1] = f32_to_bf16_rne(r);
  r -= bf16_to_f32(p[1]);
  p[2] = f32_to_bf16_rne(r);
}

// two floats -> packed bf16 pair (low half = a), round to nearest even: v_cvt_pk_bf16_f32 (the host simulator supplies BCP_CVT_PK_BF16)
#ifndef BCP_CVT_PK_BF16
typedef __bf16 bcp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float bcp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  const bcp_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bcp_bf16x2));
}
#else
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) { return BCP_CVT_PK_BF16(a, b); }
#endif
// four consecutive channels of one voxel row -> the three piece planes (8-byte stores): 6 conversions + 8 unpacks + 8 subtractions
__device__ __forceinline__ void split_store4(const float4& v, unsigned short* base, int plane_stride) {
  unsigned h0 = cvt_pk_bf16(v.x, v.y), h1 = cvt_pk_bf16(v.z, v.w);
  float r0 = v.x - __uint_as_float(h0 << 16), r1 = v.y - __uint_as_float(h0 & 0xffff0000u);
  float r2 = v.z - __uint_as_float(h1 << 16), r3 = v.w - __uint_as_float(h1 & 0xffff0000u);
  uint2 w;
  w.x = h0; w.y = h1;
  *reinterpret_cast<uint2*>(base) = w;
  h0 = cvt_pk_bf16(r0, r1); h1 = cvt_pk_bf16(r2, r3);
  r0 -= __uint_as_float(h0 << 16); r1 -= __uint_as_float(h0 & 0xffff0000u);
  r2 -= __uint_as_float(h1 << 16); r3 -= __uint_as_float(h1 & 0xffff0000u);
  w.x = h0; w.y = h1;
  *reinterpret_cast<uint2*>(base + plane_stride) = w;
  w.x = cvt_pk_bf16(r0, r1); w.y = cvt_pk_bf16(r2, r3);
  *reinterpret_cast<uint2*>(base + 2 * plane_stride) = w;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 4: TWO fp16 planes instead of three bf16 ones (conv3b.hip PL = 2).  x * s = h0 + h1 with h0 = fp16(x * s), h1 = fp16(x * s - h0)
// (22 mantissa bits + the sign of the residual), a * b ~ a0 b0 + a0 b1 + a1 b0: THREE v_mfma_f32_16x16x32_f16 per K block instead of
// six bf16 ones, two thirds of the LDS fragment reads.  fp16 has 5 exponent bits, so both operands are pre-scaled by POWERS OF TWO taken
// from the tensor's own |max| (exact; undone on the accumulator in the epilogue): the activation's comes from the norm apply pass that
// wrote it (bcp_norm_fwd ... amax_out), the weights' from the packer (pack header).  With max * s in [2^13, 2^14) nothing overflows
// (fp16 max 65504) and every element keeps an ABSOLUTE error <= 2^-25 in scaled units = 2^-38 of the tensor's max; measured error of a
// conv against fp64: 1.0-1.3x the fp32 kernel's (tools/probe/f16split_numeric.py, tests/kernel_checks.py check_conv3_f16).
// Without the per-tensor scale the split is NOT fp32-equivalent (uniformly small tensors lose up to 3 decimal digits): no scale, no PL = 2.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 bcp_f16x2 __attribute__((ext_vector_type(2)));
#ifndef BCP_F32X2_DEFINED
typedef float bcp_f32x2b __attribute__((ext_vector_type(2)));
#endif

// power of two s with amax * s in [2^13, 2^14); 1 for amax = 0 / NaN / Inf.  Exponent clamped to +-60 so that the product of an
// activation scale and a weight scale (and its reciprocal) stays a normal fp32 number.
__host__ __device__ __forceinline__ int f16_scale_exp(float amax) {
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 0;
  int e;
  (void)frexpf(amax, &e);                  // amax = m * 2^e, m in [0.5, 1)
  e = 14 - e;
  return e > 60 ? 60 : (e < -60 ? -60 : e);
}
__host__ __device__ __forceinline__ float f16_scale(float amax) { return ldexpf(1.f, f16_scale_exp(amax)); }

// two floats -> packed fp16 pair (low half = a), round to nearest even (v_cvt_pk_f16_f32 on gfx950)
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
  const bcp_f32x2b v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bcp_f16x2));
}
__device__ __forceinline__ float f16lo_to_f32(unsigned h) { return (float)__builtin_bit_cast(bcp_f16x2, h)[0]; }
__device__ __forceinline__ float f16hi_to_f32(unsigned h) { return (float)__builtin_bit_cast(bcp_f16x2, h)[1]; }

// four consecutive channels of one voxel row, pre-scaled by s -> the two fp16 planes (8-byte stores)
__device__ __forceinline__ void split_store4_f16(const float4& v, float s, unsigned short* base, int plane_stride) {
#if defined(B6_ABLATE) && (B6_ABLATE & 128)
  // MEASUREMENT ONLY (wrong results; tools/ablate_b6.sh 128): the operand's bits go to the two planes as they are -- the fetch and both LDS
  // stores stay, the 16 VALU operations of the split go: what planes written by the PRODUCER would save this kernel
  *reinterpret_cast<uint2*>(base) = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
  *reinterpret_cast<uint2*>(base + plane_stride) = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
  return;
#endif
  const float x0 = v.x * s, x1 = v.y * s, x2 = v.z * s, x3 = v.w * s;
  const unsigned h0 = cvt_pk_f16(x0, x1), h1 = cvt_pk_f16(x2, x3);
  uint2 w;
  w.x = h0; w.y = h1;
  *reinterpret_cast<uint2*>(base) = w;
  w.x = cvt_pk_f16(x0 - f16lo_to_f32(h0), x1 - f16hi_to_f32(h0));
  w.y = cvt_pk_f16(x2 - f16lo_to_f32(h1), x3 - f16hi_to_f32(h1));
  *reinterpret_cast<uint2*>(base + plane_stride) = w;
}

// pack header behind the fp32 / bf16 / fp16 sections of a packed conv weight (bcp_conv3_packed_weight_floats): 32 floats,
// [0] = max |w| of the layer, [1 .. 16] = the per-block partial maxima k_wamax_many left, rest reserved
static constexpr int kPackHeaderFloats = 32;
// float offsets of the sections of one packed weight: fp32 [T][K16/4][N16][4] | bf16 [K16/16][TP][3][N16][32] | fp16 [K16/16][TP][2][N16][32] | header
__host__ __device__ __forceinline__ long long pack_off_bf16(int T, int K16, int N16) { return (long long)T * K16 * N16; }
__host__ __device__ __forceinline__ long long pack_off_f16(int T, int K16, int N16) { return (long long)(T + 3 * ((T + 1) / 2)) * K16 * N16; }
__host__ __device__ __forceinline__ long long pack_off_hdr(int T, int K16, int N16) { return (long long)(T + 5 * ((T + 1) / 2)) * K16 * N16; }

struct Cfg { int KD, TD, TH, TW, NT, WT; };

}  // namespace bcp
