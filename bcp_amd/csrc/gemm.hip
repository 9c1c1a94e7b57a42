// bcp_amd/csrc/gemm.hip -- row-gather / row-scatter fp32 MFMA GEMMs for the non-overlapping convs:
//   * k=2,s=2 down conv  (nn.Conv3d(k=2,stride=2),          networks/VNet.py:74)   fwd / dgrad / wgrad
//   * k=2,s=2 up conv    (nn.ConvTranspose3d(k=2,stride=2), networks/VNet.py:101)  fwd / dgrad / wgrad
//   * 1x1 conv           (nn.Conv2d(k=1), networks/unet.py:48)                     fwd / dgrad / wgrad
//   * the 16 -> n_classes 1x1x1 output conv (networks/VNet.py:210), a tiny VALU stream.
// A 2x2x2/stride-2 conv has no halo: it is a plain GEMM whose A rows are gathered from the 8 fine
// voxels under a coarse voxel ("PATCH" row map), and the transposed conv is the same GEMM with the
// C rows scattered to them.  These layers are HBM-bound at the top V-Net level (AI ~ 13 FLOP/B).
//
// NN kernel:  C(m, n) = sum_k A(m, k) * B[k][n] (+ bias)      B packed [K/4][N][4] (k%4 innermost)
// TN kernel:  P[k][n] = sum_m A(m, k) * B(m, n)                (weight gradients; M = voxels)
// Both use v_mfma_f32_16x16x4_f32; lane (i = l&15, g = l>>4).
#include "common.h"
#include <cstring>
#include <cstdlib>
#include "../../include/bcp_hip.h"

namespace bcp {

enum { MAP_PLAIN = 0, MAP_PATCH = 1 };

// Row map: logical matrix [M][L]; PLAIN: p + m*L + q.  PATCH: m indexes a coarse voxel of a fine
// [N][D][H][W][Cs] tensor, q = s*Cs + c with s = (pd*2 + ph)*2 + pw.
//
// Index-math discipline (measured: the first version spent ~1500 VALU instructions per lane per block on runtime
// integer divisions in base()/off() and ran the HBM-bound top-level layers at 0.8-1.7 TB/s): base() -- three div/mod
// pairs -- is evaluated at most once per row per block (row tables in LDS for the TN kernel, one row + carry-increments
// in the NN epilogue); off16() is only ever called with block-uniform arguments, because Cs % 16 == 0 makes every
// aligned 16-column group lie inside one patch segment.
struct RowMap {
  float* p;
  int mode;
  int L;            // logical row length
  int Cs;           // channels of the fine tensor (PATCH)
  int D, H, W;      // fine dims (PATCH)
  __device__ __forceinline__ long long base(int m) const {  // offset of (m, s=0, c=0)
    if (mode == MAP_PLAIN) return (long long)m * L;
    const int Wc = W >> 1, Hc = H >> 1, Dc = D >> 1;
    const int wc = m % Wc, hc = (m / Wc) % Hc, dc = (m / (Wc * Hc)) % Dc, n = m / (Wc * Hc * Dc);
    return ((((long long)n * D + 2 * dc) * H + 2 * hc) * W + 2 * wc) * Cs;
  }
  // offset of the aligned 16-column group starting at logical column q16 (q16 % 16 == 0): uniform when q16 is
  __device__ __forceinline__ long long off16(int q16) const {
    if (mode == MAP_PLAIN) return q16;
    const int s = q16 / Cs, c = q16 - s * Cs;
    const int pw = s & 1, ph = (s >> 1) & 1, pd = s >> 2;
    return (((long long)pd * H + ph) * W + pw) * Cs + c;
  }
  // bases of 4 consecutive rows m..m+3 with one decomposition + carries (m % 4 == 0 is NOT required)
  __device__ __forceinline__ void base4(int m, long long (&b)[4]) const {
    if (mode == MAP_PLAIN) {
#pragma unroll
      for (int r = 0; r < 4; ++r) b[r] = (long long)(m + r) * L;
      return;
    }
    const int Wc = W >> 1, Hc = H >> 1, Dc = D >> 1;
    int wc = m % Wc, hc = (m / Wc) % Hc, dc = (m / (Wc * Hc)) % Dc, n = m / (Wc * Hc * Dc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      b[r] = ((((long long)n * D + 2 * dc) * H + 2 * hc) * W + 2 * wc) * Cs;
      if (++wc == Wc) { wc = 0; if (++hc == Hc) { hc = 0; if (++dc == Dc) { dc = 0; ++n; } } }
    }
  }
};

static constexpr int KC = 32;       // k per pipeline stage of the NN kernel
static constexpr int AS = KC + 4;   // LDS row stride of the A chunk (+4 pad keeps rows 16-B aligned and spreads banks)

// ------------------------------------------------------------------------------------------------
// NN: block = 64 rows x (16*NT) cols, waves split the rows (one 16-row m-tile each); K walks in stages of 32 with the
// next stage's A rows and B slab fetched into registers underneath the MFMAs of the current one.
// ------------------------------------------------------------------------------------------------
// STATS (round 6): the launch also leaves the per-channel (sum y, sum y^2) partial rows the norm layer behind a k2s2 / transposed conv needs
// (csrc/norm.hip bcp_norm_fwd partial_in) -- its statistics pass over y (k_col_partial<0>: 23 us / 128 MB at the top V-Net level, eight
// passes per forward) is skipped.  A workgroup then walks R consecutive 64-row blocks (one partial row per workgroup: the finalize kernel
// reads <= ~1000 rows per group) and sums in fp64 per lane, as the 3x3x3 kernels' epilogues do (conv3_defs.h stats_flush_t: the same lane ->
// (row, four channels) layout).  Columns fold onto channels: column n is channel n % Cout (the transposed conv's eight sub-positions).
struct GemmStats {
  double* partial;     // [G][nb][C][2]
  int nb;              // partial rows per normalisation group
  int C;               // channels (= Cout of the layer)
  int wpg;             // workgroup x-indices per group
  int R;               // 64-row blocks per workgroup
  int ysets;           // column slabs per channel set: gridDim.y / max(1, C / CT)
  // STATS == 2 (backward statistics, as bcp_conv3_dgrad_bwdstats): the GEMM output IS da of the consumer's norm layer; with that layer's
  // pre-norm tensor `by` (laid out like the output) and statistics table float[5][G][C] (mean, rstd, scale, shift, ..) the epilogue
  // accumulates (sum dz, sum dz * xhat), dz = da * act'(z) -- what k_col_partial<1> would re-read y and da for
  const float* by;
  const float* bstats;
  int G, act;
  // STATS 3 .. 6 (round 6, the transposed conv whose output is RECOMPUTED instead of stored: bcp_up_fwd_norm / bcp_up_norm_bwd; the
  // epilogues of k_conv3_c1's EPI 1 .. 4): 3 = forward statistics only, nothing stored; 4 = a = act((y - mean) * scale + shift)
  // (+ residual) -> C, |max| published; 5 = backward statistics (sum dz, sum dz * xhat) from aux = da, nothing stored; 6 = dy = scale *
  // (dz - c1 - xhat * c2) -> C.  bstats = the layer's own statistics table in all four.
  const float* aux;      // 4: residual (nullable), 5 / 6: da -- laid out like the output
  const float* c1c2;     // 6: float[2][G][C]
  float* amax;           // 4: nullable |max| slots of the activation written
  int pipe;              // (round 6) a workgroup that walks R > 1 row blocks requests the NEXT block's first A / B stage under the MFMAs and the
                         // epilogue of the current one (the walk was a serial chain per block: load -> LDS -> MFMA -> store, nothing in flight
                         // during two of its four phases); same arithmetic, same bits
};

template <int NT, int STATS = 0>
__global__ __launch_bounds__(256) void k_gemm_nn(RowMap A, const float* __restrict__ Bp, const float* __restrict__ bias,
                                                 RowMap C, int M, int K, int N, int bias_mod, int accumulate, GemmStats st) {
  constexpr int CT = NT * 16;
  constexpr int NB4 = (8 * CT + 255) / 256;          // B float4s per thread per stage
  __shared__ __attribute__((aligned(16))) float As[64 * AS];
  __shared__ __attribute__((aligned(16))) float Bs[8 * CT * 4];
  constexpr bool ACCUM = STATS == 1 || STATS == 2 || STATS == 3 || STATS == 5;      // the modes that leave partial statistics rows
  __shared__ double Ss[ACCUM ? 4 * CT * 2 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.y * CT;
  // per-lane sums over the workgroup's R <= 16 row blocks in fp32 (one value per block and accumulator: the error of such a partial sum,
  // <= 16 fp32 roundings, averages out over the ~10^5 partials of a tensor), everything behind it in fp64 -- 32 double accumulators cost the
  // kernel three of its five waves per SIMD (151 VGPRs: 79.9 us against 51.9 for the plain launch at the top level, gpurun_out/r06_s10)
  float p1[ACCUM ? NT : 1][4], p2[ACCUM ? NT : 1][4];
  float amax_o = 0.f;
  if constexpr (ACCUM) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { p1[nt][r] = 0.f; p2[nt][r] = 0.f; }
  }

  // staging role: thread -> (row, 16-B part of a 16-k half stage)
  const int srow = threadIdx.x >> 2, spart = threadIdx.x & 3;
  float4 pa[2], pb[NB4];
  const bool b_static = K <= KC;                     // one K stage: the B slab in the LDS serves every row block of the walk
  // pre: this tile's first stage is already in (pa, pb) -- requested under the previous tile (pipe); m_next >= 0: request the next tile's
  auto tile = [&](const int m0, const bool pre, const int m_next) __attribute__((always_inline)) {
  const bool srow_ok = (m0 + srow) < M;
  const float* arow = A.p + (srow_ok ? A.base(m0 + srow) : 0) + spart * 4;

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto fetch = [&](const float* ar, const bool ok, const int kc, const bool with_b) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      pa[h] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok && kc + 16 * h < K) pa[h] = ld4(ar + A.off16(kc + 16 * h));   // off16 argument is block-uniform
    }
    if (with_b) {
#pragma unroll
      for (int u = 0; u < NB4; ++u) {
        const int q = threadIdx.x + u * 256;
        pb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < 8 * CT) {
          const int co = q % CT, kq = q / CT;
          if (kc + kq * 4 < K) pb[u] = ld4(Bp + (((long long)(kc >> 2) + kq) * N + n0 + co) * 4);
        }
      }
    }
  };
  auto stash = [&](const bool with_b) {
#pragma unroll
    for (int h = 0; h < 2; ++h) st4(As + srow * AS + h * 16 + spart * 4, pa[h]);
    if (with_b) {
#pragma unroll
      for (int u = 0; u < NB4; ++u) {
        const int q = threadIdx.x + u * 256;
        if (q < 8 * CT) st4(Bs + q * 4, pb[u]);
      }
    }
  };

  if (!pre) fetch(arow, srow_ok, 0, true);
  stash(!(pre && b_static));
  __syncthreads();
  for (int kc = 0; kc < K; kc += KC) {
    const bool has_next = kc + KC < K;
    if (has_next) fetch(arow, srow_ok, kc + KC, true);
    else if (m_next >= 0) {                          // the next row block's first stage, in flight under these MFMAs and the epilogue below
      const bool ok_n = (m_next + srow) < M;
      fetch(A.p + (ok_n ? A.base(m_next + srow) : 0) + spart * 4, ok_n, 0, !b_static);
    }
#pragma unroll
    for (int kq = 0; kq < KC / 16; ++kq) {           // 16 k per (a, b) fragment pair
      const float4 a = ld4(As + (wave * 16 + li) * AS + kq * 16 + lg * 4);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float4 b = ld4(Bs + (((kq * 4 + lg) * CT) + nt * 16 + li) * 4);
        // D^T = B^T x A^T: rows of the MFMA = output columns (channels), columns = GEMM rows (voxels), so that a lane ends with FOUR
        // CONSECUTIVE channels of one voxel -- 16-byte stores / accumulate loads in the epilogue (the transposed-conv scatter wrote
        // 64-byte runs with dword stores: 2.8 TB/s on a 160 MB stream)
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, a.x, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, a.y, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, a.z, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, a.w, acc[nt], 0, 0, 0);
      }
    }
    if (!has_next) break;
    __syncthreads();
    stash(true);
    __syncthreads();
  }
  // epilogue: lane (li, lg) holds row m0 + wave*16 + li, columns n0 + nt*16 + lg*4 .. + 3
  const int mr = m0 + wave * 16 + li;
  if (mr < M) {
    float* crow = C.p + C.base(mr) + lg * 4;                            // one row decomposition per lane
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float* o = crow + C.off16(n0 + nt * 16);                          // block-uniform offset (Cs % 16 == 0)
      float4 v = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
      if (bias) {
        const float4 bv = ld4(bias + (n0 + nt * 16) % bias_mod + lg * 4);   // bias_mod % 16 == 0
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      }
      if (accumulate) { const float4 p = ld4(o); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
      if constexpr (STATS <= 2 || STATS == 7) st4(o, v);      // (7: the plain launch as a row-block walk, nothing else in the epilogue)
      if constexpr (STATS == 3) {
        p1[nt][0] += v.x; p2[nt][0] = fmaf(v.x, v.x, p2[nt][0]);
        p1[nt][1] += v.y; p2[nt][1] = fmaf(v.y, v.y, p2[nt][1]);
        p1[nt][2] += v.z; p2[nt][2] = fmaf(v.z, v.z, p2[nt][2]);
        p1[nt][3] += v.w; p2[nt][3] = fmaf(v.w, v.w, p2[nt][3]);
      }
      if constexpr (STATS >= 4) {
        // the arithmetic of k_norm_apply / k_col_partial<1> / k_norm_bwd_apply on the value just recomputed (k_conv3_c1 EPI 2 .. 4)
        const int cb = (n0 + nt * 16) % st.C + lg * 4;
        const long long pi = (long long)(blockIdx.x / st.wpg) * st.C + cb, GC = (long long)st.G * st.C;
        const float4 mu = ld4(st.bstats + pi), rs = ld4(st.bstats + GC + pi), sc = ld4(st.bstats + 2 * GC + pi), sh = ld4(st.bstats + 3 * GC + pi);
        const float yy[4] = {v.x, v.y, v.z, v.w};
        const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, r4[4] = {rs.x, rs.y, rs.z, rs.w}, s4[4] = {sc.x, sc.y, sc.z, sc.w}, h4[4] = {sh.x, sh.y, sh.z, sh.w};
        float ov[4];
        if constexpr (STATS == 4) {
          float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (st.aux) rv = ld4(st.aux + (o - C.p));
          const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = (yy[r] - m4[r]) * s4[r] + h4[r];
            ov[r] = act_fwd(z, st.act);
            if (st.aux) ov[r] += rr[r];
            const float t = fabsf(ov[r]);
            amax_o = (t > amax_o || t != t) ? t : amax_o;
          }
          st4(o, make_float4(ov[0], ov[1], ov[2], ov[3]));
        } else {
          const float4 dv = ld4(st.aux + (o - C.p));
          const float dd[4] = {dv.x, dv.y, dv.z, dv.w};
          float4 k1 = make_float4(0.f, 0.f, 0.f, 0.f), k2 = k1;
          if constexpr (STATS == 6) { k1 = ld4(st.c1c2 + pi); k2 = ld4(st.c1c2 + GC + pi); }
          const float a1[4] = {k1.x, k1.y, k1.z, k1.w}, a2[4] = {k2.x, k2.y, k2.z, k2.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = (yy[r] - m4[r]) * s4[r] + h4[r];
            const float dz = dd[r] * act_grad(z, st.act);
            const float xh = (yy[r] - m4[r]) * r4[r];
            if constexpr (STATS == 5) { p1[nt][r] += dz; p2[nt][r] = fmaf(dz, xh, p2[nt][r]); }
            else ov[r] = s4[r] * (dz - a1[r] - xh * a2[r]);
          }
          if constexpr (STATS == 6) st4(o, make_float4(ov[0], ov[1], ov[2], ov[3]));
        }
      }
      if constexpr (STATS == 1) {
        p1[nt][0] += v.x; p2[nt][0] = fmaf(v.x, v.x, p2[nt][0]);
        p1[nt][1] += v.y; p2[nt][1] = fmaf(v.y, v.y, p2[nt][1]);
        p1[nt][2] += v.z; p2[nt][2] = fmaf(v.z, v.z, p2[nt][2]);
        p1[nt][3] += v.w; p2[nt][3] = fmaf(v.w, v.w, p2[nt][3]);
      }
      if constexpr (STATS == 2) {
        // this lane's four channels of the consumer's statistics rows (group of the workgroup), its y at the element just written
        const int cb = (n0 + nt * 16) % st.C + lg * 4;
        const long long pi = (long long)(blockIdx.x / st.wpg) * st.C + cb, GC = (long long)st.G * st.C;
        const float4 mu = ld4(st.bstats + pi), rs = ld4(st.bstats + GC + pi), sc = ld4(st.bstats + 2 * GC + pi), sh = ld4(st.bstats + 3 * GC + pi);
        const float4 yv = ld4(st.by + (o - C.p));
        const float vv[4] = {v.x, v.y, v.z, v.w}, yy[4] = {yv.x, yv.y, yv.z, yv.w};
        const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, r4[4] = {rs.x, rs.y, rs.z, rs.w}, s4[4] = {sc.x, sc.y, sc.z, sc.w}, h4[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float z = (yy[r] - m4[r]) * s4[r] + h4[r];
          const float g1 = vv[r] * act_grad(z, st.act);
          const float xh = (yy[r] - m4[r]) * r4[r];
          p1[nt][r] += g1;
          p2[nt][r] = fmaf(g1, xh, p2[nt][r]);
        }
      }
    }
  }
  };

  if constexpr (STATS == 0) {
    tile(blockIdx.x * 64, false, -1);
  } else {
    const int rb0 = blockIdx.x * st.R, nblk = (M + 63) >> 6;
    const int rb1 = rb0 + st.R < nblk ? rb0 + st.R : nblk;
    const bool pipe = st.pipe != 0;
    for (int rb = rb0; rb < rb1; ++rb) {
      // every wave is done with the previous block's LDS stages.  LDS-only barrier: __syncthreads() would also drain this wave's epilogue
      // stores and the next block's operands already in flight
      if (rb > rb0) BCP_LDS_BARRIER();
      tile(rb * 64, pipe && rb > rb0, (pipe && rb + 1 < rb1) ? (rb + 1) * 64 : -1);
    }
    if constexpr (STATS == 4) {
      if (st.amax) { __syncthreads(); block_amax_publish(amax_o, st.amax); }
    }
    if constexpr (ACCUM) {
    // columns -> channels: n-tile nt holds channels (n0 + nt*16) % C ..; with C < CT the n-tiles of a workgroup repeat the channel set
    const int ntc = st.C < CT ? st.C >> 4 : NT;      // distinct 16-channel tiles of this workgroup (uniform)
    // (static indices only: ntc is 1, 2 or NT)
    if (NT > 1 && ntc == 1) {
#pragma unroll
      for (int nt = 1; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { p1[0][r] += p1[nt][r]; p2[0][r] += p2[nt][r]; }
    } else if (NT == 4 && ntc == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p1[0][r] += p1[NT > 2 ? 2 : 0][r]; p2[0][r] += p2[NT > 2 ? 2 : 0][r];
        p1[NT > 1 ? 1 : 0][r] += p1[NT > 3 ? 3 : 0][r]; p2[NT > 1 ? 1 : 0][r] += p2[NT > 3 ? 3 : 0][r];
      }
    }
    // the 16 row lanes of a wave as an fp32 tree (with the R blocks and the folded sub-positions <= 256 values per partial: <= 8 roundings
    // deep), the four waves and everything downstream in fp64
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = p1[nt][r], b = p2[nt][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (li == 0 && nt < ntc) { Ss[(wave * CT + nt * 16 + lg * 4 + r) * 2] = (double)a; Ss[(wave * CT + nt * 16 + lg * 4 + r) * 2 + 1] = (double)b; }
      }
    __syncthreads();
    const int ychan = st.C > CT ? st.C / CT : 1;                      // column slabs per channel set
    const int g = blockIdx.x / st.wpg, row = (blockIdx.x % st.wpg) * st.ysets + blockIdx.y / ychan;
    const int c0 = (blockIdx.y % ychan) * CT;
    if ((int)threadIdx.x < ntc * 16) {
      const int c = threadIdx.x;
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) { a += Ss[(w * CT + c) * 2]; b += Ss[(w * CT + c) * 2 + 1]; }
      double* dst = st.partial + (((long long)g * st.nb + row) * st.C + c0 + c) * 2;
      dst[0] = a;
      dst[1] = b;
    }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// TN: P[grp][k][n] = sum over the group's rows of A(m,k) * B(m,n).  Block = (row group, 64 k, 16*NT n);
// wave w owns k-subtile w.  Per k-step of 4 rows: lane (i, g) supplies A(row g, k i) and B(row g, n i).
// Row bases of a 64-row chunk are computed once (64 + 64 threads) into an LDS table; the next chunk is fetched
// into registers underneath the MFMAs of the current one.
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k_gemm_tn(RowMap A, RowMap B, float* __restrict__ partial, int M, int K, int N,
                                                 int rows_per_group) {
  constexpr int CT = NT * 16;
  constexpr int AS2 = 64 + 16;                          // [row][64 k] + bank spread
  constexpr int BS2 = (CT % 32 == 0) ? CT + 16 : CT;    // [row][CT n]
  constexpr int NB4 = (64 * (CT / 4) + 255) / 256;      // B float4s per thread per chunk
  __shared__ __attribute__((aligned(16))) float As[64 * AS2];
  __shared__ __attribute__((aligned(16))) float Bs[64 * BS2];
  __shared__ long long rbase[2][2][64];                 // [parity][A/B][row]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int grp = blockIdx.x, k0 = blockIdx.y * 64, n0 = blockIdx.z * CT;
  const int r_begin = grp * rows_per_group;
  int r_end = r_begin + rows_per_group;
  if (r_end > M) r_end = M;

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // launch-invariant column offsets of this thread's float4s
  const int apart = threadIdx.x & 15, arow0 = threadIdx.x >> 4;          // A: rows arow0 + 16u, k = k0 + apart*4
  const bool ak_ok = k0 + apart * 4 < K;
  const long long acol = A.off16(k0 + (apart >> 2) * 16) + (apart & 3) * 4;
  long long bcol[NB4];
  int brow[NB4];
#pragma unroll
  for (int u = 0; u < NB4; ++u) {
    const int q = threadIdx.x + u * 256;
    const int part = q % (CT / 4);
    brow[u] = q / (CT / 4);
    bcol[u] = B.off16(n0 + (part >> 2) * 16) + (part & 3) * 4;
  }
  auto bases = [&](int rc, int par) {     // threads 0..63: A rows, 64..127: B rows
    if (threadIdx.x < 128) {
      const int row = threadIdx.x & 63;
      const int m = rc + row < r_end ? rc + row : r_begin;
      rbase[par][threadIdx.x >> 6][row] = (threadIdx.x < 64) ? A.base(m) : B.base(m);
    }
  };
  float4 pa[4], pb[NB4];
  auto fetch = [&](int rc, int par) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = arow0 + 16 * u;
      pa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rc + row < r_end && ak_ok) pa[u] = ld4(A.p + rbase[par][0][row] + acol);
    }
#pragma unroll
    for (int u = 0; u < NB4; ++u) {
      pb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (brow[u] < 64 && rc + brow[u] < r_end) pb[u] = ld4(B.p + rbase[par][1][brow[u]] + bcol[u]);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) st4(As + (arow0 + 16 * u) * AS2 + apart * 4, pa[u]);
#pragma unroll
    for (int u = 0; u < NB4; ++u) {
      const int q = threadIdx.x + u * 256;
      if (brow[u] < 64) st4(Bs + brow[u] * BS2 + (q % (CT / 4)) * 4, pb[u]);
    }
  };

  if (r_begin >= r_end) return;
  bases(r_begin, 0);
  __syncthreads();
  fetch(r_begin, 0);
  stash();
  if (r_begin + 64 < r_end) bases(r_begin + 64, 1);
  __syncthreads();
  int par = 1;
  for (int rc = r_begin; rc < r_end; rc += 64) {
    const bool has_next = rc + 64 < r_end;
    if (has_next) fetch(rc + 64, par);                 // table `par` was filled before the last barrier
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
      const int row = kk * 4 + lg;
      const float a = As[row * AS2 + wave * 16 + li];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bs[row * BS2 + nt * 16 + li], acc[nt], 0, 0, 0);
    }
    if (!has_next) break;
    __syncthreads();
    stash();
    if (rc + 128 < r_end) bases(rc + 128, par ^ 1);
    __syncthreads();
    par ^= 1;
  }
  float* P = partial + (long long)grp * K * N;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = k0 + wave * 16 + lg * 4 + r;
      if (k < K) P[(long long)k * N + n0 + nt * 16 + li] = acc[nt][r];
    }
}

// out[ (k1*ok1 + k2*ok2 + n1*on1 + n2*on2) ] (+)= sum_g partial[g][k][n],  k = k1*K2 + k2, n = n1*N2 + n2
struct Idx4 { int K2, N2; long long sk1, sk2, sn1, sn2; };

// block = 16 consecutive outputs x 16 group-slots: the G partial slabs are summed by 16 threads per output (fixed order:
// deterministic) instead of one thread walking all of them (G is 256 at the top V-Net level: 256 dependent loads).
__global__ __launch_bounds__(256) void k_tn_reduce(const float* __restrict__ partial, float* __restrict__ out, int G, int K, int N,
                                                   Idx4 ix, int accumulate) {
  __shared__ float red[16][17];
  const long long total = (long long)K * N;
  const int col = threadIdx.x & 15, slot = threadIdx.x >> 4;
  for (long long i0 = (long long)blockIdx.x * 16; i0 < total; i0 += (long long)gridDim.x * 16) {
    const long long i = i0 + col;
    float s = 0.f;
    if (i < total)
      for (int g = slot; g < G; g += 16) s += partial[(long long)g * total + i];
    red[slot][col] = s;
    __syncthreads();
    if (slot == 0 && i < total) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[k][col];
      const int n = (int)(i % N), k = (int)(i / N);
      float* o = out + (k / ix.K2) * ix.sk1 + (k % ix.K2) * ix.sk2 + (n / ix.N2) * ix.sn1 + (n % ix.N2) * ix.sn2;
      *o = accumulate ? (*o + t) : t;
    }
    __syncthreads();
  }
}

// Bp[(k/4)][n][k%4] = w[(k/K2)*sk1 + (k%K2)*sk2 + (n/N2)*sn1 + (n%N2)*sn2]
__global__ __launch_bounds__(256) void k_pack_gemm_b(const float* __restrict__ w, float* __restrict__ bp, int K, int N, Idx4 ix) {
  const long long total = (long long)K * N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k4 = (int)(i & 3);
    const int n = (int)((i >> 2) % N);
    const int k = (int)(i / (4LL * N)) * 4 + k4;
    bp[i] = w[(k / ix.K2) * ix.sk1 + (k % ix.K2) * ix.sk2 + (n / ix.N2) * ix.sn1 + (n % ix.N2) * ix.sn2];
  }
}

struct GemmPackDesc { const float* w; float* bp; int K, N; Idx4 ix; };   // 64 bytes (ABI: bcp_k2_pack_desc fills it)
__global__ __launch_bounds__(256) void k_pack_gemm_b_many(const GemmPackDesc* __restrict__ descs) {
  const GemmPackDesc d = descs[blockIdx.y];
  const long long total = (long long)d.K * d.N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k4 = (int)(i & 3);
    const int n = (int)((i >> 2) % d.N);
    const int k = (int)(i / (4LL * d.N)) * 4 + k4;
    d.bp[i] = d.w[(k / d.ix.K2) * d.ix.sk1 + (k % d.ix.K2) * d.ix.sk2 + (n / d.ix.N2) * d.ix.sn1 + (n % d.ix.N2) * d.ix.sn2];
  }
}

// ------------------------------------------------------------------------------------------------
// 16 -> CO (<= 4) pointwise output conv: y[v][co] = b[co] + sum_ci x[v][ci] w[co][ci]
// ------------------------------------------------------------------------------------------------
// Optional prologue of the two kernels below: x is the RAW output of the last 3x3x3 conv and the head applies that layer's
// normalisation + activation + Dropout3d channel scale itself -- a = act((x - mean) * scale + shift) * cs, the arithmetic of
// k_norm_apply (norm.hip) -- so the 16-channel activation at full resolution is never written or re-read (128 MB each way at the LA
// size, twice per step).  stats = [5][G][16] as bcp_norm_fwd leaves them.
struct PwNorm {
  const float* stats;          // nullptr: plain 1x1 conv on x
  const float* chan_scale;     // nullable [N][16]
  long long vps;               // voxels per sample
  int G, spg;                  // normalisation groups, samples per group
  int act;
};
static constexpr int kPwMaxN = 32;

template <int CO, bool NORM>
__global__ __launch_bounds__(256) void k_pw16_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float* __restrict__ y, long long nvox, PwNorm pn) {
  __shared__ float Ws[CO * 16 + CO];
  __shared__ float Ns[NORM ? kPwMaxN * 64 : 1];      // per sample: mean, scale, shift, chan_scale x 16 channels
  if ((int)threadIdx.x < CO * 16) Ws[threadIdx.x] = w[threadIdx.x];
  if ((int)threadIdx.x < CO) Ws[CO * 16 + threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
  if (NORM) {
    const int N = pn.G * pn.spg;
    for (int i = threadIdx.x; i < N * 64; i += 256) {
      const int n = i >> 6, k = (i >> 4) & 3, c = i & 15, g = n / pn.spg;
      const int plane = k == 0 ? 0 : (k == 1 ? 2 : 3);                       // stats planes: mean, rstd, scale, shift, var
      Ns[i] = k < 3 ? pn.stats[((long long)plane * pn.G + g) * 16 + c] : (pn.chan_scale ? pn.chan_scale[n * 16 + c] : 1.f);
    }
  }
  __syncthreads();
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (long long)gridDim.x * blockDim.x) {
    float xv[16];
#pragma unroll
    for (int c = 0; c < 16; c += 4) {
      const float4 t = ld4(x + v * 16 + c);
      xv[c] = t.x; xv[c + 1] = t.y; xv[c + 2] = t.z; xv[c + 3] = t.w;
    }
    if (NORM) {
      const float* q = Ns + (int)((unsigned long long)v / (unsigned long long)pn.vps) * 64;
#pragma unroll
      for (int c = 0; c < 16; ++c) xv[c] = act_fwd((xv[c] - q[c]) * q[16 + c] + q[32 + c], pn.act) * q[48 + c];
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      float s = Ws[CO * 16 + co];
#pragma unroll
      for (int c = 0; c < 16; ++c) s = fmaf(xv[c], Ws[co * 16 + c], s);
      y[v * CO + co] = s;
    }
  }
}

// backward: dx[v][ci] = sum_co dy[v][co] w[co][ci];  acc (fp64 atomics): dw[co][ci], db[co]
template <int CO, bool NORM>
__global__ __launch_bounds__(256) void k_pw16_bwd(const float* __restrict__ x, const float* __restrict__ dy,
                                                  const float* __restrict__ w, float* __restrict__ dx,
                                                  double* __restrict__ accum /* [CO*16 + CO] */, long long nvox, PwNorm pn) {
  __shared__ float Ws[CO * 16];
  __shared__ double red[4][CO * 16 + CO];
  __shared__ float Ns[NORM ? kPwMaxN * 64 : 1];
  if ((int)threadIdx.x < CO * 16) Ws[threadIdx.x] = w[threadIdx.x];
  if (NORM) {
    const int N = pn.G * pn.spg;
    for (int i = threadIdx.x; i < N * 64; i += 256) {
      const int n = i >> 6, k = (i >> 4) & 3, c = i & 15, g = n / pn.spg;
      const int plane = k == 0 ? 0 : (k == 1 ? 2 : 3);
      Ns[i] = k < 3 ? pn.stats[((long long)plane * pn.G + g) * 16 + c] : (pn.chan_scale ? pn.chan_scale[n * 16 + c] : 1.f);
    }
  }
  __syncthreads();
  float gw[CO][16], gb[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    gb[co] = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) gw[co][c] = 0.f;
  }
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (long long)gridDim.x * blockDim.x) {
    float xv[16], dv[CO], o[16];
#pragma unroll
    for (int c = 0; c < 16; c += 4) {
      const float4 t = ld4(x + v * 16 + c);
      xv[c] = t.x; xv[c + 1] = t.y; xv[c + 2] = t.z; xv[c + 3] = t.w;
    }
    if (NORM) {      // the head's input activation, recomputed from the raw conv output (it was never stored)
      const float* q = Ns + (int)((unsigned long long)v / (unsigned long long)pn.vps) * 64;
#pragma unroll
      for (int c = 0; c < 16; ++c) xv[c] = act_fwd((xv[c] - q[c]) * q[16 + c] + q[32 + c], pn.act) * q[48 + c];
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) dv[co] = dy[v * CO + co];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      float s = 0.f;
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        s = fmaf(dv[co], Ws[co * 16 + c], s);
        gw[co][c] = fmaf(dv[co], xv[c], gw[co][c]);
      }
      o[c] = s;
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) gb[co] += dv[co];
#pragma unroll
    for (int c = 0; c < 16; c += 4) st4(dx + v * 16 + c, make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]));
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int co = 0; co < CO; ++co) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const double r = wave_sum((double)gw[co][c]);
      if (lane == 0) red[wid][co * 16 + c] = r;
    }
    const double rb = wave_sum((double)gb[co]);
    if (lane == 0) red[wid][CO * 16 + co] = rb;
  }
  __syncthreads();
  if ((int)threadIdx.x < CO * 16 + CO)
    atomicAdd(&accum[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void k_pw16_finalize(const double* __restrict__ accum, float* __restrict__ dw, float* __restrict__ db, int CO,
                                int accumulate) {
  const int i = threadIdx.x;
  if (i < CO * 16) dw[i] = (accumulate ? dw[i] : 0.f) + (float)accum[i];
  else if (i < CO * 16 + CO) db[i - CO * 16] = (accumulate ? db[i - CO * 16] : 0.f) + (float)accum[i];
}

// ---- round 5: the head's backward THROUGH the norm it applied on its way in (bcp_pw16_bwd_norm_bwd).  The gradient of the 16-channel
// activation, da[v][c] = sum_co dy[v][co] w[co][c], is two fmas per element from an 8-byte read: cheaper to recompute than to write and
// read back twice (128 MB each way at the LA size).  Pass 1 (k_pw16_bwd_stats) = k_pw16_bwd<CO, true> without the da store, plus the norm
// layer's backward statistics (sum dz, sum dz * xhat; dz = da * chan_scale * act'(z)) in k_col_partial<1>'s arithmetic and partial-row
// layout; the finalize kernels of norm.hip run in between; pass 2 (k_pw16_bwd_apply) recomputes da with the same fma chain and writes
// dy = scale * (dz - c1 - xhat * c2), k_norm_bwd_apply's arithmetic.  Blocks are sample-uniform (blockIdx.y = sample), so any G | N works.
// Thread = (voxel, four-channel group): da[c] and the head's dw[co][c] need nothing from the other channels of the voxel, so a thread keeps
// one float4 column for the whole loop (per-channel parameters in registers, a wave reads 1 KiB contiguous per instruction, ~60 VGPRs);
// the first version -- one thread per voxel, all 16 channels, k_pw16_bwd's shape -- needed 287.
struct PwCol { float mu[4], sc[4], sh[4], cs[4], rs[4]; };
__device__ __forceinline__ PwCol pw_col_load(const PwNorm& pn, int n, int col) {
  PwCol q;
  const int g = n / pn.spg;
  const float4 mu = ld4(pn.stats + ((long long)0 * pn.G + g) * 16 + col * 4), rs = ld4(pn.stats + ((long long)1 * pn.G + g) * 16 + col * 4);
  const float4 sc = ld4(pn.stats + ((long long)2 * pn.G + g) * 16 + col * 4), sh = ld4(pn.stats + ((long long)3 * pn.G + g) * 16 + col * 4);
  float4 cs = make_float4(1.f, 1.f, 1.f, 1.f);
  if (pn.chan_scale) cs = ld4(pn.chan_scale + n * 16 + col * 4);
  q.mu[0] = mu.x; q.mu[1] = mu.y; q.mu[2] = mu.z; q.mu[3] = mu.w;
  q.rs[0] = rs.x; q.rs[1] = rs.y; q.rs[2] = rs.z; q.rs[3] = rs.w;
  q.sc[0] = sc.x; q.sc[1] = sc.y; q.sc[2] = sc.z; q.sc[3] = sc.w;
  q.sh[0] = sh.x; q.sh[1] = sh.y; q.sh[2] = sh.z; q.sh[3] = sh.w;
  q.cs[0] = cs.x; q.cs[1] = cs.y; q.cs[2] = cs.z; q.cs[3] = cs.w;
  return q;
}

template <int CO>
__global__ __launch_bounds__(256) void k_pw16_bwd_stats(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ w,
                                                        double* __restrict__ hpart /* [N * gridDim.x][CO*16 + CO]: dw, db of the head per block */,
                                                        double* __restrict__ partial /* [G][spg * gridDim.x][16][2] */, PwNorm pn) {
  constexpr int U = 4, NV = CO * 5 + 8;            // per thread: dw[co][4], db[co], (sum dz)[4], (sum dz * xhat)[4]
  __shared__ double red[4][4][NV];
  const int n = blockIdx.y, g = n / pn.spg, col = threadIdx.x & 3;
  const PwCol q = pw_col_load(pn, n, col);
  float wv[CO][4];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
#pragma unroll
    for (int k = 0; k < 4; ++k) wv[co][k] = w[co * 16 + col * 4 + k];      // (a parameter view: no alignment promise)
  }
  double acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.0;
  const float* xs = x + (long long)n * pn.vps * 16;
  const float* ds = dy + (long long)n * pn.vps * CO;
  // the U elements of a trip are summed in fp32 (three additions per value), the trip's sums go into the fp64 accumulators: a quarter of
  // the conversions and fp64 operations of an element-wise fp64 sum (no time gained by it -- 43.4 us either way at the LA size, the
  // kernel's tail was the limit, see below -- but 16 VGPRs)
  float tr[NV];
  auto one = [&](const float4& xv4, const float (&dv)[CO]) {
    const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float z = (xv[k] - q.mu[k]) * q.sc[k] + q.sh[k];
      const float a = act_fwd(z, pn.act) * q.cs[k];                        // the head's input activation (never stored)
      float s = 0.f;
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        s = fmaf(dv[co], wv[co][k], s);
        tr[co * 4 + k] = fmaf(dv[co], a, tr[co * 4 + k]);
      }
      const float g1 = s * q.cs[k] * act_grad(z, pn.act);
      const float xh = (xv[k] - q.mu[k]) * q.rs[k];
      tr[CO * 5 + k] += g1;
      tr[CO * 5 + 4 + k] = fmaf(g1, xh, tr[CO * 5 + 4 + k]);
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) tr[CO * 4 + co] += dv[co];
  };
  const long long nv = pn.vps * 4, stride = (long long)gridDim.x * 256;
  long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; p + (U - 1) * stride < nv; p += U * stride) {
    float4 xv[U];
    float dv[U][CO];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long pp = p + u * stride;
      xv[u] = ld4(xs + pp * 4);
#pragma unroll
      for (int co = 0; co < CO; ++co) dv[u][co] = ds[(pp >> 2) * CO + co];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) tr[i] = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) one(xv[u], dv[u]);
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] += (double)tr[i];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) tr[i] = 0.f;
  for (; p < nv; p += stride) {           // at most U - 1 elements
    const float4 xv = ld4(xs + p * 4);
    float dv[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) dv[co] = ds[(p >> 2) * CO + co];
    one(xv, dv);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] += (double)tr[i];
  // lanes with equal column are 4 apart: xor-shuffles over 4, 8, 16, 32, then one LDS hop over the four waves
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) acc[i] += __shfl_xor(acc[i], o);
  }
  if (lane < 4) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wid][lane][i] = acc[i];
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < CO * 17) {
    const int co = t < CO * 16 ? t >> 4 : t - CO * 16, c = t & 15;
    const int cl = t < CO * 16 ? c >> 2 : 0, i = t < CO * 16 ? co * 4 + (c & 3) : CO * 4 + co;     // db: every column summed all voxels, take column 0's
    // per-block rows, summed by k_pw16_head_finalize in a fixed order: bitwise reproducible, and 32.4 us instead of 43.4 for 144 MB at the
    // LA size (gpurun_out/r05_s24) -- the first version added them with fp64 atomics, 768 workgroups finishing together onto 34 addresses
    hpart[((long long)n * gridDim.x + blockIdx.x) * (CO * 17) + t] = red[0][cl][i] + red[1][cl][i] + red[2][cl][i] + red[3][cl][i];
  } else if (t >= 128 && t < 160) {
    const int e = t - 128, ch = e >> 1, i = CO * 5 + (e & 1) * 4 + (ch & 3), cl = ch >> 2;       // e = channel * 2 + {sum dz, sum dz * xhat}
    const long long prow = (long long)g * pn.spg * gridDim.x + (long long)(n - g * pn.spg) * gridDim.x + blockIdx.x;
    partial[prow * 32 + e] = red[0][cl][i] + red[1][cl][i] + red[2][cl][i] + red[3][cl][i];
  }
}

// dw / db of the head from the per-block rows: block = one value, 256 threads walk the rows, fixed summation order
__global__ __launch_bounds__(256) void k_pw16_head_finalize(const double* __restrict__ hpart, int nrows, int CO, float* __restrict__ dw,
                                                            float* __restrict__ db, int accumulate) {
  __shared__ double red[4];
  const int e = blockIdx.x, HP = CO * 17;
  double a = 0.0;
  for (int r = threadIdx.x; r < nrows; r += 256) a += hpart[(long long)r * HP + e];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (red[0] + red[1]) + (red[2] + red[3]);
    float* o = e < CO * 16 ? dw + e : db + (e - CO * 16);
    *o = (accumulate ? *o : 0.f) + (float)t;
  }
}

template <int CO>
__global__ __launch_bounds__(256) void k_pw16_bwd_apply(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ w,
                                                        const float* __restrict__ c1c2 /* [2][G][16] */, float* __restrict__ dxraw, PwNorm pn,
                                                        float* __restrict__ amax_out) {
  constexpr int U = 4;
  const int n = blockIdx.y, g = n / pn.spg, col = threadIdx.x & 3;
  const PwCol q = pw_col_load(pn, n, col);
  const float4 k1 = ld4(c1c2 + (long long)g * 16 + col * 4), k2 = ld4(c1c2 + ((long long)pn.G + g) * 16 + col * 4);
  const float k1v[4] = {k1.x, k1.y, k1.z, k1.w}, k2v[4] = {k2.x, k2.y, k2.z, k2.w};
  float wv[CO][4];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
#pragma unroll
    for (int k = 0; k < 4; ++k) wv[co][k] = w[co * 16 + col * 4 + k];      // (a parameter view: no alignment promise)
  }
  float amax = 0.f;
  const float* xs = x + (long long)n * pn.vps * 16;
  const float* ds = dy + (long long)n * pn.vps * CO;
  float* os = dxraw + (long long)n * pn.vps * 16;
  auto one = [&](long long p, const float4& xv4, const float (&dv)[CO]) {
    const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float s = 0.f;
#pragma unroll
      for (int co = 0; co < CO; ++co) s = fmaf(dv[co], wv[co][k], s);
      const float z = (xv[k] - q.mu[k]) * q.sc[k] + q.sh[k];
      const float dz = s * q.cs[k] * act_grad(z, pn.act);
      const float xh = (xv[k] - q.mu[k]) * q.rs[k];
      o[k] = q.sc[k] * (dz - k1v[k] - xh * k2v[k]);
      const float t = fabsf(o[k]);
      amax = (t > amax || t != t) ? t : amax;
    }
    st4(os + p * 4, make_float4(o[0], o[1], o[2], o[3]));
  };
  const long long nv = pn.vps * 4, stride = (long long)gridDim.x * 256;
  long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; p + (U - 1) * stride < nv; p += U * stride) {
    float4 xv[U];
    float dv[U][CO];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long pp = p + u * stride;
      xv[u] = ld4(xs + pp * 4);
#pragma unroll
      for (int co = 0; co < CO; ++co) dv[u][co] = ds[(pp >> 2) * CO + co];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one(p + u * stride, xv[u], dv[u]);
  }
  for (; p < nv; p += stride) {
    const float4 xv = ld4(xs + p * 4);
    float dv[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) dv[co] = ds[(p >> 2) * CO + co];
    one(p, xv, dv);
  }
  if (amax_out) block_amax_publish(amax, amax_out);
}

// column sums of a [rows][C] matrix (bias gradients of convs that are NOT followed by a norm)
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ x, long long rows, int C, double* __restrict__ accum) {
  // thread t owns column t % C (C <= 256, 256 % C == 0); block-level sum first, then ONE atomic per column per block
  // (every thread used to issue its own fp64 atomic: 131k atomics onto C words)
  __shared__ double red[256];
  const int col = threadIdx.x % C, slot = threadIdx.x / C, slots = 256 / C;
  double s = 0.0;
  for (long long r = (long long)blockIdx.x * slots + slot; r < rows; r += (long long)gridDim.x * slots) s += (double)x[r * C + col];
  red[threadIdx.x] = s;
  __syncthreads();
  if ((int)threadIdx.x < C) {
    double t = 0.0;
    for (int k = 0; k < slots; ++k) t += red[k * C + col];
    atomicAdd(&accum[col], t);
  }
}
__global__ void k_colsum_finalize(const double* __restrict__ accum, float* __restrict__ out, int C, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) out[i] = (accumulate ? out[i] : 0.f) + (float)accum[i];
}

static RowMap make_map(const float* p, int mode, int L, int Cs, int D, int H, int W) {
  RowMap m;
  m.p = const_cast<float*>(p);
  m.mode = mode; m.L = L; m.Cs = Cs; m.D = D; m.H = H; m.W = W;
  return m;
}

static int pick_nt(int N, long long row_blocks) {
  int nt = 4;
  while (nt > 1 && (N % (nt * 16) != 0)) nt >>= 1;
  while (nt > 1 && row_blocks * (N / (nt * 16)) < 256) nt >>= 1;
  return nt;
}

static int launch_nn(RowMap A, const float* Bp, const float* bias, RowMap C, int M, int K, int N, int bias_mod, int accumulate,
                     hipStream_t s) {
  const int rb = cdiv(M, 64);
  const int nt = pick_nt(N, rb);
  const dim3 grid(rb, N / (nt * 16));
  const GemmStats none{nullptr, 0, 0, 0, 0, 0, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 0};
  // (round 6, option gemm_walk = W > 0) launches of >= 2 W workgroups as ~W workgroups that walk R <= 16 row blocks each with the next block's
  // operands in flight (k_gemm_nn<.., 7>: the statistics instances' walk without their epilogue)
  const int walk = options().gemm_walk;
  if (walk > 0 && (long long)rb * grid.y >= 2LL * walk) {
    int R = cdiv((long long)rb * grid.y, walk);
    if (R > 16) R = 16;
    GemmStats st = none;
    st.R = R;
    st.pipe = options().gemm_pipe;
    const dim3 g2(cdiv(rb, R), grid.y);
    if (nt == 4) hipLaunchKernelGGL((k_gemm_nn<4, 7>), g2, dim3(256), 0, s, A, Bp, bias, C, M, K, N, bias_mod, accumulate, st);
    else if (nt == 2) hipLaunchKernelGGL((k_gemm_nn<2, 7>), g2, dim3(256), 0, s, A, Bp, bias, C, M, K, N, bias_mod, accumulate, st);
    else hipLaunchKernelGGL((k_gemm_nn<1, 7>), g2, dim3(256), 0, s, A, Bp, bias, C, M, K, N, bias_mod, accumulate, st);
    return 0;
  }
  if (nt == 4) hipLaunchKernelGGL((k_gemm_nn<4>), grid, dim3(256), 0, s, A, Bp, bias, C, M, K, N, bias_mod, accumulate, none);
  else if (nt == 2) hipLaunchKernelGGL((k_gemm_nn<2>), grid, dim3(256), 0, s, A, Bp, bias, C, M, K, N, bias_mod, accumulate, none);
  else hipLaunchKernelGGL((k_gemm_nn<1>), grid, dim3(256), 0, s, A, Bp, bias, C, M, K, N, bias_mod, accumulate, none);
  return 0;
}

// the statistics variant's geometry: 64-row blocks per workgroup R, partial rows per group; false: not served (the norm's own pass runs)
struct StatPlan { int nt, R, wpg, ysets, nb; };
static bool stat_plan(StatPlan& p, int M, int N, int Cout, int groups) {
  if (groups < 1 || M % groups) return false;
  const int mpg = M / groups;
  if (mpg % 64) return false;                       // a 64-row block must not straddle two normalisation groups
  const int bpg = mpg / 64;
  p.nt = pick_nt(N, (long long)bpg * groups);
  const int CT = p.nt * 16;
  if (Cout % 16 || (Cout < CT && CT % Cout) || (Cout > CT && Cout % CT) || N % Cout) return false;
  const int gy = N / CT, ychan = Cout > CT ? Cout / CT : 1;
  p.ysets = gy / ychan;
  // as many row blocks per workgroup as keep <= ~1024 partial rows per group and >= 512 workgroups in the launch
  int R = 1;
  while (R < 16 && bpg % (R * 2) == 0 && ((long long)(bpg / R) * p.ysets > 1024) && (long long)(bpg / (R * 2)) * groups * gy >= 512) R *= 2;
  { const int v = options().gemm_stat_r; if (v > 0 && v <= 16 && bpg % v == 0) R = v; }      // measurement switch
  p.R = R;
  p.wpg = bpg / R;
  p.nb = p.wpg * p.ysets;
  return true;
}

template <int MODE>
static int launch_nn_mode(RowMap A, const float* Bp, const float* bias, RowMap C, int M, int K, int N, int Cch, int bias_mod, int accumulate,
                          double* partial, int groups, const float* by, const float* bstats, int act, const float* aux, const float* c1c2, float* amax,
                          hipStream_t s) {
  StatPlan p;
  if (!stat_plan(p, M, N, Cch, groups)) return 0;
  const dim3 grid(p.wpg * groups, N / (p.nt * 16));
  const GemmStats st{partial, p.nb, Cch, p.wpg, p.R, p.ysets, by, bstats, groups, act, aux, c1c2, amax, options().gemm_pipe};
  if (p.nt == 4) hipLaunchKernelGGL((k_gemm_nn<4, MODE>), grid, dim3(256), 0, s, A, Bp, bias, C, M, K, N, bias_mod, accumulate, st);
  else if (p.nt == 2) hipLaunchKernelGGL((k_gemm_nn<2, MODE>), grid, dim3(256), 0, s, A, Bp, bias, C, M, K, N, bias_mod, accumulate, st);
  else hipLaunchKernelGGL((k_gemm_nn<1, MODE>), grid, dim3(256), 0, s, A, Bp, bias, C, M, K, N, bias_mod, accumulate, st);
  return p.nb;
}

static int launch_nn_stats(RowMap A, const float* Bp, const float* bias, RowMap C, int M, int K, int N, int Cch, int bias_mod, int accumulate,
                           double* partial, int groups, const float* by, const float* bstats, int act, hipStream_t s) {
  if (by) return launch_nn_mode<2>(A, Bp, bias, C, M, K, N, Cch, bias_mod, accumulate, partial, groups, by, bstats, act, nullptr, nullptr, nullptr, s);
  return launch_nn_mode<1>(A, Bp, bias, C, M, K, N, Cch, bias_mod, accumulate, partial, groups, nullptr, nullptr, 0, nullptr, nullptr, nullptr, s);
}

static int tn_groups(int M, int K, int N, int nt) {
  const int chan_blocks = cdiv(K, 64) * (N / (nt * 16));
  int g = cdiv(512, chan_blocks);
  const int max_g = cdiv(M, 64);
  { const int v = options().tn_groups; if (v > 0 && v < g) g = v; }   // tests: force multi-chunk groups
  if (g > max_g) g = max_g;
  if (g < 1) g = 1;
  return g;
}
static int tn_nt(int N) {
  int nt = 4;
  while (nt > 1 && (N % (nt * 16) != 0)) nt >>= 1;
  return nt;
}

static int launch_tn(RowMap A, RowMap B, float* partial, float* out, int M, int K, int N, Idx4 ix, int accumulate, hipStream_t s) {
  const int nt = tn_nt(N);
  const int g0 = tn_groups(M, K, N, nt);
  const int rpg = cdiv(cdiv(M, g0), 64) * 64;
  const int G = cdiv(M, rpg);
  const dim3 grid(G, cdiv(K, 64), N / (nt * 16));
  if (nt == 4) hipLaunchKernelGGL((k_gemm_tn<4>), grid, dim3(256), 0, s, A, B, partial, M, K, N, rpg);
  else if (nt == 2) hipLaunchKernelGGL((k_gemm_tn<2>), grid, dim3(256), 0, s, A, B, partial, M, K, N, rpg);
  else hipLaunchKernelGGL((k_gemm_tn<1>), grid, dim3(256), 0, s, A, B, partial, M, K, N, rpg);
  const long long total = (long long)K * N;
  hipLaunchKernelGGL(k_tn_reduce, dim3((int)((total + 15) / 16 > 4096 ? 4096 : (total + 15) / 16)), dim3(256), 0, s, partial,
                     out, G, K, N, ix, accumulate);
  return 0;
}

}  // namespace bcp

using namespace bcp;

// kind: which GEMM the packed matrix feeds (see include/bcp_hip.h)
static int k2_pack_geometry(int Cin, int Cout, int kind, int& K, int& N, Idx4& ix) {
  switch (kind) {
    case BCP_PACK_DOWN_FWD:    // w[co][ci][s]: B[k = s*Cin + ci][n = co]
      K = 8 * Cin; N = Cout; ix = {Cin, Cout, 1, 8, 0, (long long)Cin * 8}; break;
    case BCP_PACK_DOWN_DGRAD:  // B[k = co][n = s*Cin + ci] = w[co][ci][s]
      K = Cout; N = 8 * Cin; ix = {Cout, Cin, 0, (long long)Cin * 8, 1, 8}; break;
    case BCP_PACK_UP_FWD:      // w[ci][co][s]: B[k = ci][n = s*Cout + co]
      K = Cin; N = 8 * Cout; ix = {Cin, Cout, 0, (long long)Cout * 8, 1, 8}; break;
    case BCP_PACK_UP_DGRAD:    // B[k = s*Cout + co][n = ci] = w[ci][co][s]
      K = 8 * Cout; N = Cin; ix = {Cout, Cin, 1, 8, 0, (long long)Cout * 8}; break;
    case BCP_PACK_PW_FWD:      // w[co][ci]: B[k = ci][n = co]
      K = Cin; N = Cout; ix = {Cin, Cout, 0, 1, 0, (long long)Cin}; break;
    case BCP_PACK_PW_DGRAD:    // B[k = co][n = ci] = w[co][ci]
      K = Cout; N = Cin; ix = {Cout, Cin, 0, (long long)Cin, 0, 1}; break;
    default: return -1;
  }
  return 0;
}

// kind: which GEMM the packed matrix feeds (see include/bcp_hip.h)
extern "C" int bcp_k2_pack_weight(const float* w, float* bp, int Cin, int Cout, int kind, void* stream) {
  BCP_REQUIRE(w && bp && Cin > 0 && Cout > 0, "bcp_k2_pack_weight: bad argument");
  BCP_REQUIRE(Cin % 16 == 0 && Cout % 16 == 0, "bcp_k2_pack_weight: channels must be multiples of 16");
  int K, N;
  Idx4 ix;
  BCP_REQUIRE(k2_pack_geometry(Cin, Cout, kind, K, N, ix) == 0, "bcp_k2_pack_weight: unknown kind %d", kind);
  const long long total = (long long)K * N;
  hipLaunchKernelGGL(k_pack_gemm_b, dim3((int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, bp, K, N, ix);
  BCP_CHECK_LAUNCH("bcp_k2_pack_weight");
  return BCP_OK;
}

// All k2 / 1x1 layers of a network in one launch: the host fills one 64-byte descriptor per (layer, kind) with
// bcp_k2_pack_desc, keeps the array in device memory and calls bcp_k2_pack_many whenever the weights changed.
extern "C" int bcp_k2_pack_desc(const float* w, float* bp, int Cin, int Cout, int kind, void* desc_out /* 64 bytes, host */) {
  BCP_REQUIRE(w && bp && desc_out && Cin > 0 && Cout > 0, "bcp_k2_pack_desc: bad argument");
  BCP_REQUIRE(Cin % 16 == 0 && Cout % 16 == 0, "bcp_k2_pack_desc: channels must be multiples of 16");
  static_assert(sizeof(GemmPackDesc) == 64, "descriptor layout is part of the ABI");
  GemmPackDesc d;
  d.w = w; d.bp = bp;
  BCP_REQUIRE(k2_pack_geometry(Cin, Cout, kind, d.K, d.N, d.ix) == 0, "bcp_k2_pack_desc: unknown kind %d", kind);
  memcpy(desc_out, &d, sizeof(d));
  return BCP_OK;
}

extern "C" int bcp_k2_pack_many(const void* descs_dev, int n, void* stream) {
  BCP_REQUIRE(descs_dev && n > 0, "bcp_k2_pack_many: bad argument");
  // (round 6: 256 blocks per descriptor instead of 32 -- a thread's elements are dependent gathers, 32 trips for the 8 x 128 x 256 matrices: 21.8 us
  // at the head of both networks' forward passes)
  hipLaunchKernelGGL(k_pack_gemm_b_many, dim3(256, n), dim3(256), 0, (hipStream_t)stream, (const GemmPackDesc*)descs_dev);
  BCP_CHECK_LAUNCH("bcp_k2_pack_many");
  return BCP_OK;
}

// Down conv forward:  y[coarse][Cout] = gather(x fine [N][D][H][W][Cin]) * B + bias
extern "C" int bcp_down_fwd(const float* x, const float* bp, const float* bias, float* y, int N, int D, int H, int W, int Cin,
                            int Cout, void* stream) {
  BCP_REQUIRE(x && bp && y, "bcp_down_fwd: null pointer");
  BCP_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && Cin % 16 == 0 && Cout % 16 == 0, "bcp_down_fwd: bad shape");
  const int M = N * (D / 2) * (H / 2) * (W / 2);
  launch_nn(make_map(x, MAP_PATCH, 8 * Cin, Cin, D, H, W), bp, bias, make_map(y, MAP_PLAIN, Cout, 0, 0, 0, 0), M, 8 * Cin, Cout,
            Cout, 0, (hipStream_t)stream);
  BCP_CHECK_LAUNCH("bcp_down_fwd");
  return BCP_OK;
}

// The k2s2 / transposed conv forward that also leaves the norm statistics of its output (round 6): stat_partial[groups][rows][Cout][2]
// doubles, rows = bcp_k2_stat_rows(kind, ...) (kind 0 = down conv, 1 = transposed conv; (D, H, W) the FINE dims as in bcp_down_fwd /
// bcp_up_fwd); rows == 0: not available for this shape (run the plain entry point and let bcp_norm_fwd make its own pass).
extern "C" int bcp_k2_stat_rows(int kind, int N, int D, int H, int W, int Cin, int Cout, int groups) {
  if (N < 1 || D < 2 || H < 2 || W < 2 || (D | H | W) & 1 || Cin % 16 || Cout % 16 || Cin < 16 || Cout < 16 || groups < 1) return 0;
  if (options().k2_stats == 0) return 0;
  const int M = N * (D / 2) * (H / 2) * (W / 2);
  // (k2_stats = 2: only where the output is >= 2^24 elements -- the statistics pass saved there is 19-23 us, the epilogue costs ~11 at every level)
  if (options().k2_stats == 2 && (long long)(kind == 0 ? M : 8LL * M) * Cout < (1LL << 24)) return 0;
  StatPlan p;
  return stat_plan(p, M, kind == 0 ? Cout : 8 * Cout, Cout, groups) ? p.nb : 0;
}

extern "C" int bcp_down_fwd_stats(const float* x, const float* bp, const float* bias, float* y, int N, int D, int H, int W, int Cin,
                                  int Cout, double* stat_partial, int groups, void* stream) {
  BCP_REQUIRE(x && bp && y && stat_partial, "bcp_down_fwd_stats: null pointer");
  BCP_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && Cin % 16 == 0 && Cout % 16 == 0, "bcp_down_fwd_stats: bad shape");
  const int M = N * (D / 2) * (H / 2) * (W / 2);
  const int nb = launch_nn_stats(make_map(x, MAP_PATCH, 8 * Cin, Cin, D, H, W), bp, bias, make_map(y, MAP_PLAIN, Cout, 0, 0, 0, 0), M, 8 * Cin,
                                 Cout, Cout, Cout, 0, stat_partial, groups, nullptr, nullptr, 0, (hipStream_t)stream);
  BCP_REQUIRE(nb > 0, "bcp_down_fwd_stats: fused statistics unavailable for this shape (check bcp_k2_stat_rows first)");
  BCP_CHECK_LAUNCH("bcp_down_fwd_stats");
  return BCP_OK;
}

extern "C" int bcp_up_fwd_stats(const float* x, const float* bp, const float* bias, float* y, int N, int D, int H, int W, int Cin,
                                int Cout, double* stat_partial, int groups, void* stream) {
  BCP_REQUIRE(x && bp && y && stat_partial, "bcp_up_fwd_stats: null pointer");
  BCP_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && Cin % 16 == 0 && Cout % 16 == 0, "bcp_up_fwd_stats: bad shape");
  const int M = N * (D / 2) * (H / 2) * (W / 2);
  const int nb = launch_nn_stats(make_map(x, MAP_PLAIN, Cin, 0, 0, 0, 0), bp, bias, make_map(y, MAP_PATCH, 8 * Cout, Cout, D, H, W), M, Cin,
                                 8 * Cout, Cout, Cout, 0, stat_partial, groups, nullptr, nullptr, 0, (hipStream_t)stream);
  BCP_REQUIRE(nb > 0, "bcp_up_fwd_stats: fused statistics unavailable for this shape (check bcp_k2_stat_rows first)");
  BCP_CHECK_LAUNCH("bcp_up_fwd_stats");
  return BCP_OK;
}

// Down conv dgrad: dx fine (+)= scatter(dy[coarse][Cout] * B');  (D,H,W) are the FINE dims
extern "C" int bcp_down_dgrad(const float* dy, const float* bp, float* dx, int N, int D, int H, int W, int Cin, int Cout,
                              int accumulate, void* stream) {
  BCP_REQUIRE(dy && bp && dx, "bcp_down_dgrad: null pointer");
  BCP_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && Cin % 16 == 0 && Cout % 16 == 0, "bcp_down_dgrad: bad shape");
  const int M = N * (D / 2) * (H / 2) * (W / 2);
  launch_nn(make_map(dy, MAP_PLAIN, Cout, 0, 0, 0, 0), bp, nullptr, make_map(dx, MAP_PATCH, 8 * Cin, Cin, D, H, W), M, Cout,
            8 * Cin, 1, accumulate, (hipStream_t)stream);
  BCP_CHECK_LAUNCH("bcp_down_dgrad");
  return BCP_OK;
}

// Up (transposed) conv forward: y fine [N][D][H][W][Cout] = scatter(x[coarse][Cin] * B) + bias; (D,H,W) FINE dims
extern "C" int bcp_up_fwd(const float* x, const float* bp, const float* bias, float* y, int N, int D, int H, int W, int Cin,
                          int Cout, void* stream) {
  BCP_REQUIRE(x && bp && y, "bcp_up_fwd: null pointer");
  BCP_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && Cin % 16 == 0 && Cout % 16 == 0, "bcp_up_fwd: bad shape");
  const int M = N * (D / 2) * (H / 2) * (W / 2);
  launch_nn(make_map(x, MAP_PLAIN, Cin, 0, 0, 0, 0), bp, bias, make_map(y, MAP_PATCH, 8 * Cout, Cout, D, H, W), M, Cin, 8 * Cout,
            Cout, 0, (hipStream_t)stream);
  BCP_CHECK_LAUNCH("bcp_up_fwd");
  return BCP_OK;
}

// The two dgrads whose epilogue leaves the backward statistics of the norm layer that CONSUMES their output (round 6; as
// bcp_conv3_dgrad_bwdstats for the 3x3x3 dgrads): the output (after the optional += of a skip gradient) is da of the layer in front of the
// k2s2 / transposed conv, y_prev / stats_prev that layer's pre-norm tensor and statistics table; stat_partial[groups][rows][Cin][2]
// receives (sum dz, sum dz * xhat) for bcp_norm_bwd(partial_in, nb_in = rows), whose statistics pass over (y, da) is then skipped.
// rows = bcp_k2_bwdstat_rows(kind: 0 down-conv dgrad / 1 transposed-conv dgrad, ...); 0: not available (plain entry points).
extern "C" int bcp_k2_bwdstat_rows(int kind, int N, int D, int H, int W, int Cin, int Cout, int groups) {
  if (N < 1 || D < 2 || H < 2 || W < 2 || (D | H | W) & 1 || Cin % 16 || Cout % 16 || Cin < 16 || Cout < 16 || groups < 1) return 0;
  if (options().k2_bwd_stats == 0 || options().fuse_bwd_stats == 0) return 0;
  const int M = N * (D / 2) * (H / 2) * (W / 2);
  // (k2_bwd_stats = 2: only where the output is >= 2^22 elements -- the pass saved there is 23-33 us in the step, the epilogue costs ~11 at every level)
  if (options().k2_bwd_stats == 2 && (long long)(kind == 0 ? 8LL * M : M) * Cin < (1LL << 22)) return 0;
  StatPlan p;
  return stat_plan(p, M, kind == 0 ? 8 * Cin : Cin, Cin, groups) ? p.nb : 0;
}

extern "C" int bcp_down_dgrad_bwdstats(const float* dy, const float* bp, float* dx, int N, int D, int H, int W, int Cin, int Cout, int accumulate,
                                       const float* y_prev, const float* stats_prev, int act, double* stat_partial, int groups, void* stream) {
  BCP_REQUIRE(dy && bp && dx && y_prev && stats_prev && stat_partial, "bcp_down_dgrad_bwdstats: null pointer");
  BCP_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && Cin % 16 == 0 && Cout % 16 == 0, "bcp_down_dgrad_bwdstats: bad shape");
  const int M = N * (D / 2) * (H / 2) * (W / 2);
  const int nb = launch_nn_stats(make_map(dy, MAP_PLAIN, Cout, 0, 0, 0, 0), bp, nullptr, make_map(dx, MAP_PATCH, 8 * Cin, Cin, D, H, W), M, Cout,
                                 8 * Cin, Cin, 1, accumulate, stat_partial, groups, y_prev, stats_prev, act, (hipStream_t)stream);
  BCP_REQUIRE(nb > 0, "bcp_down_dgrad_bwdstats: fused statistics unavailable for this shape (check bcp_k2_bwdstat_rows first)");
  BCP_CHECK_LAUNCH("bcp_down_dgrad_bwdstats");
  return BCP_OK;
}

extern "C" int bcp_up_dgrad_bwdstats(const float* dy, const float* bp, float* dx, int N, int D, int H, int W, int Cin, int Cout, int accumulate,
                                     const float* y_prev, const float* stats_prev, int act, double* stat_partial, int groups, void* stream) {
  BCP_REQUIRE(dy && bp && dx && y_prev && stats_prev && stat_partial, "bcp_up_dgrad_bwdstats: null pointer");
  BCP_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && Cin % 16 == 0 && Cout % 16 == 0, "bcp_up_dgrad_bwdstats: bad shape");
  const int M = N * (D / 2) * (H / 2) * (W / 2);
  const int nb = launch_nn_stats(make_map(dy, MAP_PATCH, 8 * Cout, Cout, D, H, W), bp, nullptr, make_map(dx, MAP_PLAIN, Cin, 0, 0, 0, 0), M, 8 * Cout,
                                 Cin, Cin, 1, accumulate, stat_partial, groups, y_prev, stats_prev, act, (hipStream_t)stream);
  BCP_REQUIRE(nb > 0, "bcp_up_dgrad_bwdstats: fused statistics unavailable for this shape (check bcp_k2_bwdstat_rows first)");
  BCP_CHECK_LAUNCH("bcp_up_dgrad_bwdstats");
  return BCP_OK;
}

// Round 6: the transposed conv + its norm layer with the conv output RECOMPUTED instead of stored (networks/VNet.py:101-113 UpsamplingDeconvBlock:
// ConvTranspose3d -> BatchNorm3d / InstanceNorm3d -> ReLU, then the decoder's skip add :268-283).  The layer is HBM-bound (32 MACs per output
// from a tensor an eighth its size), so y = up(x) never exists in HBM: forward = statistics pass (GEMM, nothing stored) -> finalize -> apply pass
// (GEMM again: a = act(norm(y)) + residual); backward = statistics pass (GEMM + da) -> finalize -> apply pass (dy).  At the top V-Net level
// that is 224 MB less per forward and 192 MB less per backward pass (y written once and read two / two times before).  Same arithmetic per
// element as bcp_up_fwd + bcp_norm_fwd / bcp_norm_bwd (k_norm_apply / k_col_partial<1> / k_norm_bwd_apply); the statistics are summed as in
// bcp_up_fwd_stats (fp32 lane partials, fp64 behind them).  rows = bcp_up_norm_rows(...): 0 = shape not served (or option up_recompute off).
namespace bcp {      // csrc/norm.hip
void norm_fwd_finalize_launch(const double* partial, int nb, int G, int C, long long rows_per_group, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, float* stats, hipStream_t s, float* amax_clear_or_null);
void norm_bwd_finalize_launch(const double* partial, int nb, int G, int C, long long rows_per_group, float* dgamma, float* dbeta, int accumulate,
                              float* c1c2raw, hipStream_t s, float* amax_clear_or_null);
}

extern "C" int bcp_up_norm_rows(int N, int D, int H, int W, int Cin, int Cout, int groups) {
  if (N < 1 || D < 2 || H < 2 || W < 2 || (D | H | W) & 1 || Cin % 16 || Cout % 16 || Cin < 16 || Cout < 16 || groups < 1 || N % groups) return 0;
  if (options().up_recompute == 0) return 0;
  const int M = N * (D / 2) * (H / 2) * (W / 2);
  if (options().up_recompute == 2 && 8LL * M * Cout < (1LL << 24)) return 0;      // (2: the top level only)
  StatPlan p;
  return stat_plan(p, M, 8 * Cout, Cout, groups) ? p.nb : 0;
}

extern "C" size_t bcp_up_norm_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int groups) {
  const int rows = bcp_up_norm_rows(N, D, H, W, Cin, Cout, groups);
  return rows > 0 ? (size_t)groups * rows * Cout * 2 * sizeof(double) + (size_t)4 * groups * Cout * sizeof(float) : 0;
}

extern "C" int bcp_up_fwd_norm(const float* x, const float* bp, const float* bias, int N, int D, int H, int W, int Cin, int Cout, int groups,
                               const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps, int act,
                               const float* residual, float* stats, void* workspace, float* out, float* amax_out_or_null, void* stream) {
  BCP_REQUIRE(x && bp && stats && workspace && out && groups >= 1, "bcp_up_fwd_norm: null pointer / bad groups");
  BCP_REQUIRE(aligned16(out) && aligned16(stats) && (!residual || aligned16(residual)), "bcp_up_fwd_norm: alignment");
  const int rows0 = bcp_up_norm_rows(N, D, H, W, Cin, Cout, groups);
  BCP_REQUIRE(rows0 > 0, "bcp_up_fwd_norm: shape not served (check bcp_up_norm_rows first)");
  hipStream_t s = (hipStream_t)stream;
  const int M = N * (D / 2) * (H / 2) * (W / 2);
  double* partial = reinterpret_cast<double*>(workspace);
  const RowMap A = make_map(x, MAP_PLAIN, Cin, 0, 0, 0, 0), Cm = make_map(out, MAP_PATCH, 8 * Cout, Cout, D, H, W);
  const int rows = launch_nn_mode<3>(A, bp, bias, Cm, M, Cin, 8 * Cout, Cout, Cout, 0, partial, groups, nullptr, nullptr, 0, nullptr, nullptr, nullptr, s);
  BCP_REQUIRE(rows == rows0, "bcp_up_fwd_norm: internal row count mismatch");
  norm_fwd_finalize_launch(partial, rows, groups, Cout, (long long)N / groups * D * H * W, gamma, beta, running_mean, running_var, momentum, eps, stats, s,
                           amax_out_or_null);
  launch_nn_mode<4>(A, bp, bias, Cm, M, Cin, 8 * Cout, Cout, Cout, 0, nullptr, groups, nullptr, stats, act, residual, nullptr, amax_out_or_null, s);
  BCP_CHECK_LAUNCH("bcp_up_fwd_norm");
  return BCP_OK;
}

extern "C" int bcp_up_norm_bwd(const float* x, const float* bp, const float* bias, const float* da, int N, int D, int H, int W, int Cin, int Cout,
                               int groups, const float* stats, int act, float* dgamma, float* dbeta, int accumulate, void* workspace, float* dy,
                               void* stream) {
  BCP_REQUIRE(x && bp && da && stats && workspace && dy && groups >= 1, "bcp_up_norm_bwd: null pointer / bad groups");
  BCP_REQUIRE(aligned16(da) && aligned16(dy), "bcp_up_norm_bwd: alignment");
  BCP_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bcp_up_norm_bwd: dgamma and dbeta come together");
  const int rows0 = bcp_up_norm_rows(N, D, H, W, Cin, Cout, groups);
  BCP_REQUIRE(rows0 > 0, "bcp_up_norm_bwd: shape not served (check bcp_up_norm_rows first)");
  hipStream_t s = (hipStream_t)stream;
  const int M = N * (D / 2) * (H / 2) * (W / 2);
  double* partial = reinterpret_cast<double*>(workspace);
  float* c1c2raw = reinterpret_cast<float*>(partial + (size_t)groups * rows0 * Cout * 2);
  const RowMap A = make_map(x, MAP_PLAIN, Cin, 0, 0, 0, 0), Cm = make_map(dy, MAP_PATCH, 8 * Cout, Cout, D, H, W);
  const int rows = launch_nn_mode<5>(A, bp, bias, Cm, M, Cin, 8 * Cout, Cout, Cout, 0, partial, groups, nullptr, stats, act, da, nullptr, nullptr, s);
  BCP_REQUIRE(rows == rows0, "bcp_up_norm_bwd: internal row count mismatch");
  norm_bwd_finalize_launch(partial, rows, groups, Cout, (long long)N / groups * D * H * W, dgamma, dbeta, accumulate, c1c2raw, s, nullptr);
  launch_nn_mode<6>(A, bp, bias, Cm, M, Cin, 8 * Cout, Cout, Cout, 0, nullptr, groups, nullptr, stats, act, da, c1c2raw, nullptr, s);
  BCP_CHECK_LAUNCH("bcp_up_norm_bwd");
  return BCP_OK;
}

// Up conv dgrad: dx[coarse][Cin] = gather(dy fine) * B'
extern "C" int bcp_up_dgrad(const float* dy, const float* bp, float* dx, int N, int D, int H, int W, int Cin, int Cout,
                            int accumulate, void* stream) {
  BCP_REQUIRE(dy && bp && dx, "bcp_up_dgrad: null pointer");
  BCP_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && Cin % 16 == 0 && Cout % 16 == 0, "bcp_up_dgrad: bad shape");
  const int M = N * (D / 2) * (H / 2) * (W / 2);
  launch_nn(make_map(dy, MAP_PATCH, 8 * Cout, Cout, D, H, W), bp, nullptr, make_map(dx, MAP_PLAIN, Cin, 0, 0, 0, 0), M, 8 * Cout,
            Cin, 1, accumulate, (hipStream_t)stream);
  BCP_CHECK_LAUNCH("bcp_up_dgrad");
  return BCP_OK;
}

extern "C" int bcp_pw_fwd(const float* x, const float* bp, const float* bias, float* y, long long rows, int Cin, int Cout,
                          void* stream) {
  BCP_REQUIRE(x && bp && y && rows > 0 && rows < (1LL << 31), "bcp_pw_fwd: bad argument");
  BCP_REQUIRE(Cin % 16 == 0 && Cout % 16 == 0, "bcp_pw_fwd: channels must be multiples of 16");
  launch_nn(make_map(x, MAP_PLAIN, Cin, 0, 0, 0, 0), bp, bias, make_map(y, MAP_PLAIN, Cout, 0, 0, 0, 0), (int)rows, Cin, Cout, Cout,
            0, (hipStream_t)stream);
  BCP_CHECK_LAUNCH("bcp_pw_fwd");
  return BCP_OK;
}

extern "C" size_t bcp_tn_workspace_bytes(long long M, int K, int N) {
  const int nt = tn_nt(N);
  const int g0 = tn_groups((int)M, K, N, nt);
  const int rpg = cdiv(cdiv(M, g0), 64) * 64;
  return (size_t)cdiv(M, rpg) * K * N * sizeof(float);
}

// Weight gradients.  kind selects the layer type; (D,H,W) are the FINE dims for the k2 kinds.
//   BCP_WG_DOWN: x fine [.,Cin], dy coarse [.,Cout] -> dw[Cout][Cin][8]
//   BCP_WG_UP:   x coarse [.,Cin], dy fine [.,Cout] -> dw[Cin][Cout][8]
//   BCP_WG_PW:   x [rows][Cin], dy [rows][Cout]      -> dw[Cout][Cin]      (rows = N*D*H*W)
extern "C" int bcp_k2_wgrad(const float* x, const float* dy, float* dw, int N, int D, int H, int W, int Cin, int Cout, int kind,
                            int accumulate, void* workspace, void* stream) {
  if (bcp::options().whatif & 8) return BCP_OK;      // MEASUREMENT ONLY (common.h Options::whatif)

  BCP_REQUIRE(x && dy && dw && workspace, "bcp_k2_wgrad: null pointer");
  BCP_REQUIRE(Cin % 16 == 0 && Cout % 16 == 0, "bcp_k2_wgrad: channels must be multiples of 16");
  float* ws = reinterpret_cast<float*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  if (kind == BCP_WG_DOWN) {
    const int M = N * (D / 2) * (H / 2) * (W / 2);
    Idx4 ix = {Cin, Cout, 1, 8, 0, (long long)Cin * 8};  // k = s*Cin + ci, n = co -> dw[co][ci][s]
    launch_tn(make_map(x, MAP_PATCH, 8 * Cin, Cin, D, H, W), make_map(dy, MAP_PLAIN, Cout, 0, 0, 0, 0), ws, dw, M, 8 * Cin, Cout, ix,
              accumulate, s);
  } else if (kind == BCP_WG_UP) {
    const int M = N * (D / 2) * (H / 2) * (W / 2);
    Idx4 ix = {Cin, Cout, 0, (long long)Cout * 8, 1, 8};  // k = ci, n = s*Cout + co -> dw[ci][co][s]
    launch_tn(make_map(x, MAP_PLAIN, Cin, 0, 0, 0, 0), make_map(dy, MAP_PATCH, 8 * Cout, Cout, D, H, W), ws, dw, M, Cin, 8 * Cout, ix,
              accumulate, s);
  } else if (kind == BCP_WG_PW) {
    const long long M = (long long)N * D * H * W;
    BCP_REQUIRE(M < (1LL << 31), "bcp_k2_wgrad: too many rows");
    Idx4 ix = {Cin, Cout, 0, 1, 0, (long long)Cin};  // k = ci, n = co -> dw[co][ci]
    launch_tn(make_map(x, MAP_PLAIN, Cin, 0, 0, 0, 0), make_map(dy, MAP_PLAIN, Cout, 0, 0, 0, 0), ws, dw, (int)M, Cin, Cout, ix,
              accumulate, s);
  } else {
    BCP_REQUIRE(false, "bcp_k2_wgrad: unknown kind %d", kind);
  }
  BCP_CHECK_LAUNCH("bcp_k2_wgrad");
  return BCP_OK;
}

static int pw16_fwd_impl(const char* who, const float* x, const float* w, const float* bias, float* y, long long nvox, int Cout, const PwNorm& pn,
                         hipStream_t s) {
  const int grid = (int)((nvox + 255) / 256 > 2048 ? 2048 : (nvox + 255) / 256);
  if (pn.stats) {
    if (Cout == 2) hipLaunchKernelGGL((k_pw16_fwd<2, true>), dim3(grid), dim3(256), 0, s, x, w, bias, y, nvox, pn);
    else if (Cout == 4) hipLaunchKernelGGL((k_pw16_fwd<4, true>), dim3(grid), dim3(256), 0, s, x, w, bias, y, nvox, pn);
    else BCP_REQUIRE(false, "%s: Cout=%d unsupported (2 or 4)", who, Cout);
  } else {
    if (Cout == 2) hipLaunchKernelGGL((k_pw16_fwd<2, false>), dim3(grid), dim3(256), 0, s, x, w, bias, y, nvox, pn);
    else if (Cout == 4) hipLaunchKernelGGL((k_pw16_fwd<4, false>), dim3(grid), dim3(256), 0, s, x, w, bias, y, nvox, pn);
    else BCP_REQUIRE(false, "%s: Cout=%d unsupported (2 or 4)", who, Cout);
  }
  return BCP_OK;
}

extern "C" int bcp_pw16_fwd(const float* x, const float* w, const float* bias, float* y, long long nvox, int Cout, void* stream) {
  BCP_REQUIRE(x && w && y && nvox > 0, "bcp_pw16_fwd: bad argument");
  if (int rc = pw16_fwd_impl("bcp_pw16_fwd", x, w, bias, y, nvox, Cout, PwNorm{nullptr, nullptr, 1, 1, 1, 0}, (hipStream_t)stream)) return rc;
  BCP_CHECK_LAUNCH("bcp_pw16_fwd");
  return BCP_OK;
}

static bool pw_norm_args(PwNorm& pn, const float* stats, const float* chan_scale, int N, int G, long long nvox, int act) {
  if (!stats || N < 1 || G < 1 || N % G || N > kPwMaxN || nvox % N) return false;
  pn = PwNorm{stats, chan_scale, nvox / N, G, N / G, act};
  return true;
}

extern "C" int bcp_pw16_fwd_norm(const float* x_raw, const float* stats, const float* chan_scale, int N, int G, int act, const float* w,
                                 const float* bias, float* y, long long nvox, int Cout, void* stream) {
  BCP_REQUIRE(x_raw && w && y && nvox > 0, "bcp_pw16_fwd_norm: bad argument");
  PwNorm pn;
  BCP_REQUIRE(pw_norm_args(pn, stats, chan_scale, N, G, nvox, act), "bcp_pw16_fwd_norm: needs stats, 1 <= N <= %d samples in G | N groups", kPwMaxN);
  if (int rc = pw16_fwd_impl("bcp_pw16_fwd_norm", x_raw, w, bias, y, nvox, Cout, pn, (hipStream_t)stream)) return rc;
  BCP_CHECK_LAUNCH("bcp_pw16_fwd_norm");
  return BCP_OK;
}

// workspace: (Cout*16 + Cout) doubles
static int pw16_bwd_impl(const char* who, const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, long long nvox,
                         int Cout, int accumulate, void* workspace, const PwNorm& pn, hipStream_t s) {
  double* acc = reinterpret_cast<double*>(workspace);
  hipMemsetAsync(acc, 0, (size_t)(Cout * 17) * sizeof(double), s);
  const int grid = (int)((nvox + 255) / 256 > 1024 ? 1024 : (nvox + 255) / 256);
  if (pn.stats) {
    if (Cout == 2) hipLaunchKernelGGL((k_pw16_bwd<2, true>), dim3(grid), dim3(256), 0, s, x, dy, w, dx, acc, nvox, pn);
    else if (Cout == 4) hipLaunchKernelGGL((k_pw16_bwd<4, true>), dim3(grid), dim3(256), 0, s, x, dy, w, dx, acc, nvox, pn);
    else BCP_REQUIRE(false, "%s: Cout=%d unsupported (2 or 4)", who, Cout);
  } else {
    if (Cout == 2) hipLaunchKernelGGL((k_pw16_bwd<2, false>), dim3(grid), dim3(256), 0, s, x, dy, w, dx, acc, nvox, pn);
    else if (Cout == 4) hipLaunchKernelGGL((k_pw16_bwd<4, false>), dim3(grid), dim3(256), 0, s, x, dy, w, dx, acc, nvox, pn);
    else BCP_REQUIRE(false, "%s: Cout=%d unsupported (2 or 4)", who, Cout);
  }
  hipLaunchKernelGGL(k_pw16_finalize, dim3(1), dim3(128), 0, s, acc, dw, db, Cout, accumulate);
  return BCP_OK;
}

extern "C" int bcp_pw16_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, long long nvox,
                            int Cout, int accumulate, void* workspace, void* stream) {
  BCP_REQUIRE(x && dy && w && dx && dw && db && workspace && nvox > 0, "bcp_pw16_bwd: bad argument");
  if (int rc = pw16_bwd_impl("bcp_pw16_bwd", x, dy, w, dx, dw, db, nvox, Cout, accumulate, workspace, PwNorm{nullptr, nullptr, 1, 1, 1, 0},
                             (hipStream_t)stream)) return rc;
  BCP_CHECK_LAUNCH("bcp_pw16_bwd");
  return BCP_OK;
}

// x_raw = the raw conv output the forward consumed (bcp_pw16_fwd_norm); dx = gradient w.r.t. the ACTIVATION (hand it to bcp_norm_bwd)
extern "C" int bcp_pw16_bwd_norm(const float* x_raw, const float* stats, const float* chan_scale, int N, int G, int act, const float* dy,
                                 const float* w, float* dx, float* dw, float* db, long long nvox, int Cout, int accumulate, void* workspace,
                                 void* stream) {
  BCP_REQUIRE(x_raw && dy && w && dx && dw && db && workspace && nvox > 0, "bcp_pw16_bwd_norm: bad argument");
  PwNorm pn;
  BCP_REQUIRE(pw_norm_args(pn, stats, chan_scale, N, G, nvox, act), "bcp_pw16_bwd_norm: needs stats, 1 <= N <= %d samples in G | N groups", kPwMaxN);
  if (int rc = pw16_bwd_impl("bcp_pw16_bwd_norm", x_raw, dy, w, dx, dw, db, nvox, Cout, accumulate, workspace, pn, (hipStream_t)stream)) return rc;
  BCP_CHECK_LAUNCH("bcp_pw16_bwd_norm");
  return BCP_OK;
}

// ---- the head's backward through the norm (kernels above).  workspace: bcp_pw16_bwd_norm_bwd_workspace_bytes
namespace bcp {      // csrc/norm.hip
void norm_bwd_finalize_launch(const double* partial, int nb, int G, int C, long long rows_per_group, float* dgamma, float* dbeta, int accumulate,
                              float* c1c2raw, hipStream_t s, float* amax_clear_or_null);
}
static int pw16_nbps(long long vps, int N) {
  long long nb = vps / 1024;                       // >= 16 float4 per thread: the block reduction (18-28 fp64 values) is the kernel's tail
  const long long cap = 768 / N < 1 ? 1 : 768 / N;  // k_pw16_bwd_stats holds 130 VGPRs (128 by force spills): three workgroups per CU are resident, one round of them
  if (nb > cap) nb = cap;
  return (int)(nb < 1 ? 1 : nb);
}
extern "C" size_t bcp_pw16_bwd_norm_bwd_workspace_bytes(int N, int G, long long nvox) {
  if (N < 1 || G < 1 || N % G || nvox < N || nvox % N) return 0;
  return (size_t)N * pw16_nbps(nvox / N, N) * (32 + 68) * sizeof(double) + (size_t)4 * G * 16 * sizeof(float);
}

// dy_raw = gradient w.r.t. the RAW conv output x_raw (what bcp_pw16_bwd_norm + bcp_norm_bwd leave), dw / db of the head, dgamma / dbeta of the
// norm (nullable pair); amax_out_or_null: |max| slots of dy_raw (cleared and max-reduced here, as bcp_norm_bwd does)
extern "C" int bcp_pw16_bwd_norm_bwd(const float* x_raw, const float* stats, const float* chan_scale, int N, int G, int act, const float* dy,
                                     const float* w, float* dy_raw, float* dw, float* db, float* dgamma, float* dbeta, int norm_accumulate,
                                     long long nvox, int Cout, int accumulate, void* workspace, float* amax_out_or_null, void* stream) {
  BCP_REQUIRE(x_raw && dy && w && dy_raw && dw && db && workspace && nvox > 0, "bcp_pw16_bwd_norm_bwd: bad argument");
  BCP_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bcp_pw16_bwd_norm_bwd: dgamma and dbeta come together");
  BCP_REQUIRE(Cout == 2 || Cout == 4, "bcp_pw16_bwd_norm_bwd: Cout=%d unsupported (2 or 4)", Cout);
  BCP_REQUIRE(aligned16(x_raw) && aligned16(dy_raw) && aligned16(stats) && (!chan_scale || aligned16(chan_scale)), "bcp_pw16_bwd_norm_bwd: alignment");
  PwNorm pn;
  BCP_REQUIRE(pw_norm_args(pn, stats, chan_scale, N, G, nvox, act), "bcp_pw16_bwd_norm_bwd: needs stats, 1 <= N <= %d samples in G | N groups", kPwMaxN);
  hipStream_t s = (hipStream_t)stream;
  const int nbps = pw16_nbps(pn.vps, N);
  double* partial = reinterpret_cast<double*>(workspace);
  double* hpart = partial + (size_t)N * nbps * 32;
  float* c1c2raw = reinterpret_cast<float*>(hpart + (size_t)N * nbps * 68);
  if (Cout == 2) hipLaunchKernelGGL((k_pw16_bwd_stats<2>), dim3(nbps, N), dim3(256), 0, s, x_raw, dy, w, hpart, partial, pn);
  else hipLaunchKernelGGL((k_pw16_bwd_stats<4>), dim3(nbps, N), dim3(256), 0, s, x_raw, dy, w, hpart, partial, pn);
  hipLaunchKernelGGL(k_pw16_head_finalize, dim3(Cout * 17), dim3(256), 0, s, hpart, N * nbps, Cout, dw, db, accumulate);
  norm_bwd_finalize_launch(partial, pn.spg * nbps, G, 16, pn.vps * pn.spg, dgamma, dbeta, norm_accumulate, c1c2raw, s, amax_out_or_null);
  long long gx = (pn.vps * 4 + 1023) / 1024;       // 4 float4 per thread and trip
  const long long cap = 2048 / N < 1 ? 1 : 2048 / N;
  if (gx > cap) gx = cap;
  if (Cout == 2) hipLaunchKernelGGL((k_pw16_bwd_apply<2>), dim3((int)gx, N), dim3(256), 0, s, x_raw, dy, w, c1c2raw, dy_raw, pn, amax_out_or_null);
  else hipLaunchKernelGGL((k_pw16_bwd_apply<4>), dim3((int)gx, N), dim3(256), 0, s, x_raw, dy, w, c1c2raw, dy_raw, pn, amax_out_or_null);
  BCP_CHECK_LAUNCH("bcp_pw16_bwd_norm_bwd");
  return BCP_OK;
}

// workspace: C doubles
extern "C" int bcp_colsum(const float* x, long long rows, int C, float* out, int accumulate, void* workspace, void* stream) {
  BCP_REQUIRE(x && out && workspace && rows > 0, "bcp_colsum: bad argument");
  BCP_REQUIRE(C >= 1 && C <= 256 && 256 % C == 0, "bcp_colsum: C must divide 256");
  double* acc = reinterpret_cast<double*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  hipMemsetAsync(acc, 0, (size_t)C * sizeof(double), s);
  const int slots = 256 / C;
  long long g = (rows + slots * 64 - 1) / (slots * 64);
  if (g > 512) g = 512;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_colsum, dim3((int)g), dim3(256), 0, s, x, rows, C, acc);
  hipLaunchKernelGGL(k_colsum_finalize, dim3(cdiv(C, 256)), dim3(256), 0, s, acc, out, C, accumulate);
  BCP_CHECK_LAUNCH("bcp_colsum");
  return BCP_OK;
}
