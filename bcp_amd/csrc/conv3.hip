// bcp_amd/csrc/conv3.hip -- 3x3x3 (V-Net) and 3x3 (U-Net) convolution, pad 1, stride 1, channels-last
// fp32, as an implicit GEMM on the CDNA4 fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact f32
// fmaf-chain numerics, 157 TF/s peak) -- SURVEY.md A1.1 / A2.   fwd, dgrad (same kernel, flipped
// + transposed weight pack) and wgrad.
//
// Reference ops: nn.Conv3d(k=3,pad=1) networks/VNet.py:17, nn.Conv2d(k=3,pad=1) networks/unet.py:19-25
// and their autograd backward.
//
// Data layout: X [N][D][H][W][Cin], Y [N][D][H][W][Cout] (NDHWC; 2D = D 1, KD 1).
// Packed weights Wp[tap][Cin16/4][Cout16][4] (cin%4 innermost) so that one ds_read_b128 feeds four
// MFMA k-steps: lane l = (i = l&15, g = l>>4) holds A[vox i][cin 4g..4g+3] and B[cin 4g..4g+3][cout i];
// k-step j uses element j of both, i.e. the k index of the instruction is the cin permutation
// {4g+j} -- the same permutation on both operands, so the dot product is unchanged.
//
// Block = 256 threads = 4 waves, output tile TD x TH x TW voxels (M = 64*MT) x (16*NT) channels.
// Per 16-channel cin chunk the (TD+2)(TH+2)(TW+2) input halo is staged ONCE in LDS ([vox][16+4 pad]),
// and all 27 taps read it at shifted offsets (2.5x halo over-read from L2 instead of 27x);
// weights stream through a double-buffered LDS stage of WT taps.  fp32 MFMA is slow enough
// (32 cycles / instruction / SIMD) that LDS bandwidth is not the limiter; the structure is chosen
// so every wave issues MT*NT*4 back-to-back MFMAs per tap per (MT+NT) ds_read_b128.
#include "conv3_defs.h"
#include "../../include/bcp_hip.h"
#include <cstdlib>
#include <cstdio>

namespace bcp {

// ------------------------------------------------------------------------------------------------
// forward / dgrad
// ------------------------------------------------------------------------------------------------
template <int KD, int TD, int TH, int TW, int NT, int WT>
__global__ __launch_bounds__(256) void k_conv3_mfma(const float* __restrict__ X, const float* __restrict__ Wp,
                                                    const float* __restrict__ bias, float* __restrict__ Y, ConvDims cd,
                                                    int accumulate, StatsArg st) {
  using TL = Tile<KD, TD, TH, TW>;
  constexpr int MT = TL::MT, T = TL::T, CT = NT * 16;
  constexpr int S = T / WT;                     // weight stages per cin chunk
  static_assert(T % WT == 0, "WT must divide the tap count");
  constexpr int NBUF = (S > 1) ? 2 : 1;
  constexpr int WSTAGE4 = WT * 4 * CT;          // float4s per weight stage
  constexpr int NW4 = (WSTAGE4 + 255) / 256;    // per-thread prefetch registers
  constexpr int NACC = (MT * NT == 1) ? 2 : 1;  // a lone accumulator would serialise on the 40-cycle MFMA latency

  HIP_DYNAMIC_SHARED(float4, smem4)   // float4 element type => 16-B aligned base, so ld4/st4 become ds_read/write_b128
  float* smem = reinterpret_cast<float*>(smem4);
  float* Xs = smem;                             // [HV][XS]
  float* Ws = smem + TL::HV * XS;               // [NBUF][WT][4][CT][4]
  double* Ss = reinterpret_cast<double*>(Ws + NBUF * WSTAGE4 * 4);   // [4][CT][2] statistics scratch (16-B aligned)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  int n, d0, h0, w0;
  tile_origin(cd, blockIdx.x, TD, TH, TW, n, d0, h0, w0);
  const int cout0 = blockIdx.y * CT;
  const int cin4 = cd.Cin16 >> 2;

  int voff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) voff[mt] = TL::voff((wave * MT + mt) * 16 + li) * XS + lg * 4;
  HaloFetch<TL> hf;
  hf.init(cd, Xs);

  f32x4 acc[NACC][MT][NT];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[a][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // split-K: blockIdx.z owns a contiguous range of cin chunks and writes its own partial slab (summed by k_sum_slabs)
  const int nchunks = cd.Cin16 >> 4;
  const int c_begin = (int)((long long)nchunks * blockIdx.z / gridDim.z), c_end = (int)((long long)nchunks * (blockIdx.z + 1) / gridDim.z);
  Y += (long long)blockIdx.z * cd.N * cd.D * cd.H * cd.W * cd.Cout;
  // One cin chunk at a time.  With a single weight stage per chunk (S == 1: the deep 64-voxel-tile configurations, 8-16
  // chunks per block) the NEXT chunk's halo and weights travel global -> registers underneath the MFMAs of the current
  // chunk and are dropped into the SAME LDS buffers between two barriers -- no extra LDS, so the 3 workgroups per CU stay.
  // (A first attempt that double-buffered the weight stage in LDS lost one resident workgroup per CU and was slower at the
  // in-step batch of 2: 83 vs 77 us at C=64, 65 vs 57 us at C=128.)
  if (S == 1) {
    float4 hpre[HaloFetch<TL>::NP], wpre[NW4];
    auto wfetch = [&](int cc) {
#pragma unroll
      for (int u = 0; u < NW4; ++u) {
        const int q = threadIdx.x + u * 256;
        if (q < WSTAGE4) {
          const int co = q % CT, cig = (q / CT) & 3, tl = q / (4 * CT);
          wpre[u] = ld4(Wp + ((((long long)tl * cin4 + cc * 4 + cig) * cd.Cout16) + cout0 + co) * 4);
        }
      }
    };
    auto wstash = [&]() {
#pragma unroll
      for (int u = 0; u < NW4; ++u) {
        const int q = threadIdx.x + u * 256;
        if (q < WSTAGE4) st4(Ws + q * 4, wpre[u]);
      }
    };
    hf.fetch(X, cd, n, d0, h0, w0, c_begin, hpre);
    wfetch(c_begin);
    hf.stash(hpre);
    wstash();
    __syncthreads();
    for (int cc = c_begin; cc < c_end; ++cc) {
      const bool has_next = cc + 1 < c_end;
      if (has_next) {
        hf.fetch(X, cd, n, d0, h0, w0, cc + 1, hpre);
        wfetch(cc + 1);
      }
#pragma unroll (WT > 9 ? 9 : WT)
      for (int tl = 0; tl < WT; ++tl) {
        const int toff = TL::tapoff(tl) * XS;
        float4 a[MT], b[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = ld4(Xs + voff[mt] + toff);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = ld4(Ws + ((tl * 4 + lg) * CT + nt * 16 + li) * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b[nt].x, acc[0][mt][nt], 0, 0, 0);
            acc[NACC - 1][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b[nt].y, acc[NACC - 1][mt][nt], 0, 0, 0);
            acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, b[nt].z, acc[0][mt][nt], 0, 0, 0);
            acc[NACC - 1][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, b[nt].w, acc[NACC - 1][mt][nt], 0, 0, 0);
          }
      }
      if (has_next) {
        __syncthreads();
        hf.stash(hpre);
        wstash();
        __syncthreads();
      }
    }
  } else
  for (int cc = c_begin; cc < c_end; ++cc) {
    __syncthreads();  // everyone is done with the previous chunk's LDS contents
    {
      float4 pre[HaloFetch<TL>::NP];
      hf.fetch(X, cd, n, d0, h0, w0, cc, pre);
      hf.stash(pre);
    }
    // weight stage 0 of this chunk  (an explicit loads-then-stores version was measured slower at C=64: 75.7 vs 70 us; so was
    // carrying the NEXT chunk's halo + stage-0 weights in registers under the last stage's MFMAs -- 75.8 vs 71.4 us at C=64,
    // 9.00 vs 8.96 ms per step: with 2-3 workgroups per CU the other workgroups' MFMAs already cover this round trip, and the
    // extra live registers cost more than the latency they hide)
    for (int q = threadIdx.x; q < WSTAGE4; q += 256) {
      const int co = q % CT, cig = (q / CT) & 3, tl = q / (4 * CT);
      st4(Ws + q * 4, ld4(Wp + ((((long long)tl * cin4 + cc * 4 + cig) * cd.Cout16) + cout0 + co) * 4));
    }
    __syncthreads();
#pragma unroll 1
    for (int st = 0; st < S; ++st) {
      float4 pre[NW4];
      if (S > 1 && st + 1 < S) {
#pragma unroll
        for (int u = 0; u < NW4; ++u) {
          const int q = threadIdx.x + u * 256;
          if (q < WSTAGE4) {
            const int co = q % CT, cig = (q / CT) & 3, tl = q / (4 * CT);
            const int tap = (st + 1) * WT + tl;
            pre[u] = ld4(Wp + ((((long long)tap * cin4 + cc * 4 + cig) * cd.Cout16) + cout0 + co) * 4);
          }
        }
      }
      const float* Wb = Ws + (st & (NBUF - 1)) * WSTAGE4 * 4;
#pragma unroll (WT > 9 ? 9 : WT)
      for (int tl = 0; tl < WT; ++tl) {
        const int toff = TL::tapoff(st * WT + tl) * XS;
        float4 a[MT], b[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = ld4(Xs + voff[mt] + toff);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = ld4(Wb + ((tl * 4 + lg) * CT + nt * 16 + li) * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b[nt].x, acc[0][mt][nt], 0, 0, 0);
            acc[NACC - 1][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b[nt].y, acc[NACC - 1][mt][nt], 0, 0, 0);
            acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, b[nt].z, acc[0][mt][nt], 0, 0, 0);
            acc[NACC - 1][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, b[nt].w, acc[NACC - 1][mt][nt], 0, 0, 0);
          }
      }
      if (S > 1) {
        if (st + 1 < S) {
          float* Wn = Ws + ((st + 1) & 1) * WSTAGE4 * 4;
#pragma unroll
          for (int u = 0; u < NW4; ++u) {
            const int q = threadIdx.x + u * 256;
            if (q < WSTAGE4) st4(Wn + q * 4, pre[u]);
          }
        }
        __syncthreads();
      }
    }
  }

  // epilogue: lane (li, lg) holds rows (voxels) lg*4+r, column (cout) li of each 16x16 tile
  double s1[NT], s2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { s1[nt] = 0.0; s2[nt] = 0.0; }
  const bool full = (TW % 4 == 0) && cout0 + CT <= cd.Cout && d0 + TD <= cd.D && h0 + TH <= cd.H && w0 + TW <= cd.W;   // uniform
  const long long tile_base = ((((long long)n * cd.D + d0) * cd.H + h0) * cd.W + w0) * cd.Cout;
  auto rows = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = (wave * MT + mt) * 16 + lg * 4 + r;
        const int tw = m % TW, th = (m / TW) % TH, td = m / (TW * TH);
        const int d = d0 + td, h = h0 + th, w = w0 + tw;
        if (full || (d < cd.D && h < cd.H && w < cd.W)) {
          const long long ro = tile_base + (unsigned)(((td * cd.H + th) * cd.W + tw) * cd.Cout);   // uniform base + 32-bit row offset
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int co = cout0 + nt * 16 + li;
            if (full || co < cd.Cout) {
              float v = acc[0][mt][nt][r];
              if (NACC == 2) v += acc[NACC - 1][mt][nt][r];
              if (bias) v += bias[co];
              if (accumulate) v += Y[ro + co];
              Y[ro + co] = v;
              stat_add<MODE>(s1[nt], s2[nt], v);
            }
          }
        }
      }
    }
  };
  if (!st.partial) rows(std::integral_constant<int, 0>{});
  else rows(std::integral_constant<int, 1>{});
  if (st.partial) {
    const int g = blockIdx.x / st.tiles_per_group, row = blockIdx.x % st.tiles_per_group;
    stats_flush<NT>(s1, s2, Ss, st.partial + ((long long)g * st.rows + row) * st.C * 2, cout0, cd.Cout);
  }
}

// y (+)= bias + sum_k part[k]   (split-K epilogue of the streaming kernel; deep V-Net levels only: <= 1 MB)
__global__ __launch_bounds__(256) void k_sum_slabs(const float* __restrict__ part, int SK, long long n, int C,
                                                   const float* __restrict__ bias, float* __restrict__ y, int accumulate) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float s = bias ? bias[i % C] : 0.f;
    for (int k = 0; k < SK; ++k) s += part[k * n + i];
    y[i] = accumulate ? y[i] + s : s;
  }
}

// ------------------------------------------------------------------------------------------------
// forward / dgrad, "resident" variant: persistent workgroups keep the WHOLE packed weight slab of their
// 16*NT output channels (all taps, all cin chunks) in LDS and walk a strided list of spatial tiles; the
// input halo of the NEXT (tile, cin-chunk) work item is fetched into registers while the MFMAs of the
// current one run, then dropped into the single LDS halo buffer between two barriers.  Versus the
// streaming kernel above this removes the per-tile weight reload (27.6 KB x 3920 tiles for the 16->16
// layer) and overlaps HBM/L2 latency with the matrix pipe instead of serialising load -> compute.
// Used whenever T*Cin16*16*NT*4 B of weights + one halo fit in the 160 KB LDS.
// ------------------------------------------------------------------------------------------------
template <int KD, int TD, int TH, int TW, int NT>
__global__ __launch_bounds__(256) void k_conv3_res(const float* __restrict__ X, const float* __restrict__ Wp,
                                                   const float* __restrict__ bias, float* __restrict__ Y, ConvDims cd,
                                                   int n_tiles, int accumulate, StatsArg st, int csplit) {
  using TL = Tile<KD, TD, TH, TW>;
  constexpr int MT = TL::MT, T = TL::T, CT = NT * 16;
  constexpr int NACC = (MT * NT == 1) ? 2 : 1;       // a lone accumulator would serialise on the 40-cycle MFMA latency
  using HF = HaloFetch<TL>;
  constexpr int NP = HF::NP;

  HIP_DYNAMIC_SHARED(float4, smem4)   // float4 element type => 16-B aligned base, so ld4/st4 become ds_read/write_b128
  float* smem = reinterpret_cast<float*>(smem4);
  // split-K over cin chunks (deep levels, whose full weight slab does not fit the LDS): blockIdx.z owns chunks
  // [c_begin, c_begin + nch) and writes its own partial output slab (summed -- with bias / += -- by k_sum_slabs)
  const int nch_all = cd.Cin16 >> 4;
  const int c_begin = (int)((long long)nch_all * blockIdx.z / csplit);
  const int nch = (int)((long long)nch_all * (blockIdx.z + 1) / csplit) - c_begin;
  Y += (long long)blockIdx.z * cd.N * cd.D * cd.H * cd.W * cd.Cout;
  float* Ws = smem;                                  // [nch][T][4][CT][4]
  float* Xs = smem + (size_t)nch * T * 4 * CT * 4;   // [HV][XS]
  double* Ss = reinterpret_cast<double*>(Xs + TL::HV * XS);   // [4][CT][2] statistics scratch

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int cout0 = blockIdx.y * CT;
  const int cin4 = cd.Cin16 >> 2;

  int voff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) voff[mt] = TL::voff((wave * MT + mt) * 16 + li) * XS + lg * 4;

  HF hf;
  hf.init(cd, Xs);

  // The MFMA is issued as D = W^T (16 cout x 4 cin) * X (4 cin x 16 voxels): lane (li, lg) then holds voxel li of the M tile and
  // the FOUR CONSECUTIVE output channels lg*4 .. lg*4+3, so the epilogue is one 16-byte store per (M tile, slab) instead of
  // four 4-byte stores.  Measured (tools/probe/issue_cost_probe.hip): next to fp32 MFMAs a global store costs the issuing
  // wave ~160-320 matrix-pipe cycles WHATEVER its width -- sixteen dword stores per tile were ~18 % of the 16 -> 16 layer.
  // launch-invariant epilogue constants: output offset of this lane's voxel in every M tile, bias of its four channels
  unsigned yoff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m0 = (wave * MT + mt) * 16 + li;
    const int tw = m0 % TW, th = (m0 / TW) % TH, td = m0 / (TW * TH);
    yoff[mt] = (unsigned)(((td * cd.H + th) * cd.W + tw) * cd.Cout + lg * 4);
  }
  float bv[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[nt][r] = (bias && cout0 + nt * 16 + lg * 4 + r < cd.Cout) ? bias[cout0 + nt * 16 + lg * 4 + r] : 0.f;
  const bool slab_full = cout0 + CT <= cd.Cout;

  f32x4 acc[NACC][MT][NT];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[a][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto fetch = [&](int t, int c, float4 (&pre)[NP]) {
    int n2, d2, h2, w2;
    tile_origin(cd, t, TD, TH, TW, n2, d2, h2, w2);
    hf.fetch(X, cd, n2, d2, h2, w2, c_begin + c, pre);
  };
  auto stash = [&](const float4 (&pre)[NP]) { hf.stash(pre); };

  // Work items of this block: (tile, chunk).  XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (each with
  // its own 4 MB L2), so workgroup b runs on XCD b % 8.  Each XCD owns ONE contiguous eighth of the (d-major) tile list and
  // its workgroups walk it side by side -- the tiles in flight on an XCD are spatial neighbours and their halo overlap
  // (2.5x over-read per tile) hits that XCD's L2 instead of going back to the fabric.
  int tile, t_end, t_step;
  if (gridDim.x % 8 == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    t_step = gridDim.x >> 3;
    tile = (int)((long long)n_tiles * xcd / 8) + j;
    t_end = (int)((long long)n_tiles * (xcd + 1) / 8);
  } else {
    tile = blockIdx.x; t_end = n_tiles; t_step = gridDim.x;
  }
  if (st.partial && (int)threadIdx.x < CT && cout0 + (int)threadIdx.x < cd.Cout) {
    // a persistent workgroup only flushes the groups it visits: its rows of the other groups must read as zero.  Written
    // here by the same threads that later flush (program order), instead of a hipMemsetAsync per conv launch (36 ten-
    // microsecond fills per ACDC step on the critical path)
    for (int g = 0; g < st.G; ++g) {
      double* z = st.partial + (((long long)g * st.rows + blockIdx.x) * st.C + cout0 + threadIdx.x) * 2;
      z[0] = 0.0; z[1] = 0.0;
    }
  }
  int ch = 0;
  if (tile >= t_end) return;
  double s1[NT][4], s2[NT][4];                       // fused norm statistics of the current group: this lane's 4 channels per slab
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.0; s2[nt][r] = 0.0; }
  int cur_g = st.partial ? tile / st.tiles_per_group : 0;
  {
    float4 pre[NP];
    fetch(tile, 0, pre);     // first halo in flight while the weights below are fetched: one exposed round trip, not two
    // resident weights: Wp[tap][cin4][Cout16][4] -> Ws[chunk][tap][cig][co][4].  Eight loads in flight per thread before the
    // first store: a load -> store loop pays one L2 round trip per iteration (7 for the 16->16 layer, 27 for 32->32 x 2 slabs)
    // on every launch, with the matrix pipe idle
    {
      constexpr int WB = 8;
      const int total = nch * T * 4 * CT;
      for (int q0 = threadIdx.x; q0 < total; q0 += 256 * WB) {
        float4 wv[WB];
  #pragma unroll
        for (int u = 0; u < WB; ++u) {
          const int q = q0 + u * 256;
          if (q < total) {
            const int co = q % CT, cig = (q / CT) & 3, tap = (q / (4 * CT)) % T, ch = q / (4 * CT * T);
            wv[u] = ld4(Wp + ((((long long)tap * cin4 + (c_begin + ch) * 4 + cig) * cd.Cout16) + cout0 + co) * 4);
          }
        }
  #pragma unroll
        for (int u = 0; u < WB; ++u) {
          const int q = q0 + u * 256;
          if (q < total) st4(Ws + (size_t)q * 4, wv[u]);
        }
      }
    }
    stash(pre);
  }
  __syncthreads();
  for (;;) {
    // next work item
    int ntile = tile, nchk = ch + 1;
    if (nchk == nch) { nchk = 0; ntile = tile + t_step; }
    const bool has_next = ntile < t_end;
    float4 pre[NP];
    if (has_next) fetch(ntile, nchk, pre);
    // MFMAs of the current item
    const float* Wc = Ws + (size_t)ch * T * 4 * CT * 4;
    // partial unroll: a full 27-tap unroll makes hipcc split the ds_read_b128 fragments into read2_b32/b64 pairs
#pragma unroll 9
    for (int tap = 0; tap < T; ++tap) {
      const int toff = TL::tapoff(tap) * XS;
      float4 a[MT], b[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = ld4(Xs + voff[mt] + toff);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = ld4(Wc + ((tap * 4 + lg) * CT + nt * 16 + li) * 4);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[nt].x, a[mt].x, acc[0][mt][nt], 0, 0, 0);
          acc[NACC - 1][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[nt].y, a[mt].y, acc[NACC - 1][mt][nt], 0, 0, 0);
          acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[nt].z, a[mt].z, acc[0][mt][nt], 0, 0, 0);
          acc[NACC - 1][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[nt].w, a[mt].w, acc[NACC - 1][mt][nt], 0, 0, 0);
        }
    }
    if (ch == nch - 1) {
      int n, d0, h0, w0;
      tile_origin(cd, tile, TD, TH, TW, n, d0, h0, w0);
      if (st.partial && tile / st.tiles_per_group != cur_g) {   // tiles are visited in increasing order: groups never come back
        stats_flush_t<NT>(s1, s2, Ss, st.partial + ((long long)cur_g * st.rows + blockIdx.x) * st.C * 2, cout0, cd.Cout);
        cur_g = tile / st.tiles_per_group;
      }
      const bool full = slab_full && (cd.Cout & 3) == 0 && d0 + TD <= cd.D && h0 + TH <= cd.H && w0 + TW <= cd.W;   // uniform
      const long long tile_base = ((((long long)n * cd.D + d0) * cd.H + h0) * cd.W + w0) * cd.Cout;
      auto rows = [&](auto mode_tag, auto acc_tag) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool ACCUM = decltype(acc_tag)::value;   // compile-time: a conditional read-modify-write serialises every store behind a vmcnt(0)
        if (full) {
          // whole tile inside the volume: uniform base + launch-invariant lane offsets, one 16-byte store per (M tile, slab)
          float* yb = Y + tile_base + cout0;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              float* p = yb + yoff[mt] + nt * 16;
              float4 v;
              v.x = acc[0][mt][nt][0] + (NACC == 2 ? acc[NACC - 1][mt][nt][0] : 0.f) + bv[nt][0];
              v.y = acc[0][mt][nt][1] + (NACC == 2 ? acc[NACC - 1][mt][nt][1] : 0.f) + bv[nt][1];
              v.z = acc[0][mt][nt][2] + (NACC == 2 ? acc[NACC - 1][mt][nt][2] : 0.f) + bv[nt][2];
              v.w = acc[0][mt][nt][3] + (NACC == 2 ? acc[NACC - 1][mt][nt][3] : 0.f) + bv[nt][3];
              if (ACCUM) { const float4 o = ld4(p); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
              st4(p, v);
              stat_add<MODE>(s1[nt][0], s2[nt][0], v.x); stat_add<MODE>(s1[nt][1], s2[nt][1], v.y);
              stat_add<MODE>(s1[nt][2], s2[nt][2], v.z); stat_add<MODE>(s1[nt][3], s2[nt][3], v.w);
            }
        } else {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int m = (wave * MT + mt) * 16 + li;
            const int tw = m % TW, th = (m / TW) % TH, td = m / (TW * TH);
            const int d = d0 + td, h = h0 + th, w = w0 + tw;
            if (d < cd.D && h < cd.H && w < cd.W) {
              const long long ro = tile_base + (unsigned)(((td * cd.H + th) * cd.W + tw) * cd.Cout);
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int co = cout0 + nt * 16 + lg * 4 + r;
                  if (co < cd.Cout) {
                    float v = acc[0][mt][nt][r];
                    if (NACC == 2) v += acc[NACC - 1][mt][nt][r];
                    v += bv[nt][r];
                    if (ACCUM) v += Y[ro + co];
                    Y[ro + co] = v;
                    stat_add<MODE>(s1[nt][r], s2[nt][r], v);
                  }
                }
            }
          }
        }
      };
      if (accumulate) {
        if (!st.partial) rows(std::integral_constant<int, 0>{}, std::true_type{});
        else rows(std::integral_constant<int, 1>{}, std::true_type{});
      } else {
        if (!st.partial) rows(std::integral_constant<int, 0>{}, std::false_type{});
        else rows(std::integral_constant<int, 1>{}, std::false_type{});
      }
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[a][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (!has_next) {
      if (st.partial) stats_flush_t<NT>(s1, s2, Ss, st.partial + ((long long)cur_g * st.rows + blockIdx.x) * st.C * 2, cout0, cd.Cout);
      break;
    }
    __syncthreads();   // every wave is done reading the halo buffer
    stash(pre);
    __syncthreads();
    tile = ntile;
    ch = nchk;
  }
}

// ------------------------------------------------------------------------------------------------
// wgrad: dW[tap][ci][co] = sum_v X[v + off(tap)][ci] * dY[v][co]
// GEMM per tap with M = 16 ci (one chunk), N = 16*NT co, K = voxels.  Lane (i, g) supplies
// A[ci i][vox g] and B[vox g][co i] per k-step of 4 voxels -- plain ds_read_b32 from the
// [vox][chan] tiles.  The 4 waves split the taps (wave w owns taps w, w+4, ...), every wave walks
// all M voxels of the tile; a block loops over a group of spatial tiles and writes ONE partial
// [T][16][CT] slab, which k_wgrad_reduce sums (deterministic, no atomics) straight into the torch
// weight-gradient layout [Cout][Cin][T].
// ------------------------------------------------------------------------------------------------
template <int KD, int TD, int TH, int TW, int NT>
__global__ __launch_bounds__(256) void k_conv3_wgrad(const float* __restrict__ X, const float* __restrict__ dY,
                                                     float* __restrict__ partial, ConvDims cd, int tiles_total,
                                                     int tiles_per_group) {
  using TL = Tile<KD, TD, TH, TW>;
  constexpr int T = TL::T, CT = NT * 16, M = TL::M;
  constexpr int TPW = (T + 3) / 4;                      // taps per wave
  constexpr int YS = (CT % 32 == 0) ? CT + 16 : CT;     // dY tile row stride (bank spread for the 4 k-groups)
  constexpr int NY4 = (M * (CT / 4) + 255) / 256;       // dY-tile float4s per thread
  static_assert(TW % 4 == 0, "a k-step is 4 consecutive voxels of one tile row");

  HIP_DYNAMIC_SHARED(float4, smem4)   // float4 element type => 16-B aligned base, so ld4/st4 become ds_read/write_b128
  float* smem = reinterpret_cast<float*>(smem4);
  float* Xs = smem;                 // [HV][XSW]
  float* Ys = smem + TL::HV * XSW;  // [M][YS]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int cc = blockIdx.y;                 // cin chunk
  const int cout0 = blockIdx.z * CT;
  const int grp = blockIdx.x;

  f32x4 acc[TPW][NT];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane parts of the LDS addresses; the (row, kw) parts are wave-uniform scalars
  int xoff[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = wave + 4 * t;
    xoff[t] = ((tap < T ? TL::tapoff(tap) : 0) + lg) * XSW + li;
  }
  const int yoff = lg * YS + li;

  int t_end = (grp + 1) * tiles_per_group;
  if (t_end > tiles_total) t_end = tiles_total;
  int tile = grp * tiles_per_group;
  if (tile >= t_end) return;

  using HF = HaloFetch<TL, XSW>;
  HF hf;
  hf.init(cd, Xs);
  // dY tile: float4 q of the [M][CT] tile, launch-invariant offsets for tiles that lie wholly inside the volume
  unsigned yrel[NY4];
#pragma unroll
  for (int u = 0; u < NY4; ++u) {
    const int q = threadIdx.x + u * 256;
    const int m = q / (CT / 4), c4 = q % (CT / 4);
    const int tw = m % TW, th = (m / TW) % TH, td = m / (TW * TH);
    yrel[u] = (unsigned)(((td * cd.H + th) * cd.W + tw) * cd.Cout + c4 * 4);
  }
  const bool slab_full = cout0 + CT <= cd.Cout;
  float4 px[HF::NP], py[NY4];
  auto fetch = [&](int tl) {   // global -> registers for tile tl
    int n, d0, h0, w0;
    tile_origin(cd, tl, TD, TH, TW, n, d0, h0, w0);
    hf.fetch(X, cd, n, d0, h0, w0, cc, px);
    if (slab_full && d0 + TD <= cd.D && h0 + TH <= cd.H && w0 + TW <= cd.W) {   // uniform
      const float* yb = dY + ((((long long)n * cd.D + d0) * cd.H + h0) * cd.W + w0) * cd.Cout + cout0;
#pragma unroll
      for (int u = 0; u < NY4; ++u) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((int)threadIdx.x + u * 256 < M * (CT / 4)) v = ld4(yb + yrel[u]);
        py[u] = v;
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < NY4; ++u) {
      const int q = threadIdx.x + u * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < M * (CT / 4)) {
        const int m = q / (CT / 4), c4 = q % (CT / 4);
        const int tw = m % TW, th = (m / TW) % TH, td = m / (TW * TH);
        const int d = d0 + td, h = h0 + th, w = w0 + tw;
        const int co = cout0 + c4 * 4;
        if (d < cd.D && h < cd.H && w < cd.W && co < cd.Cout)
          v = ld4(dY + ((((long long)n * cd.D + d) * cd.H + h) * cd.W + w) * cd.Cout + co);
      }
      py[u] = v;
    }
  };
  auto stash = [&]() {         // registers -> LDS
    hf.stash(px);
#pragma unroll
    for (int u = 0; u < NY4; ++u) {
      const int q = threadIdx.x + u * 256;
      if (q < M * (CT / 4)) st4(Ys + (q / (CT / 4)) * YS + (q % (CT / 4)) * 4, py[u]);
    }
  };

  fetch(tile);
  stash();
  __syncthreads();
  for (;;) {
    const bool has_next = tile + 1 < t_end;
    if (has_next) fetch(tile + 1);           // loads stay in flight under the MFMAs below
#pragma unroll 1
    for (int row = 0; row < M / TW; ++row) {
      const int th = row % TH, td = row / TH;
      const int xrow = ((td * TL::HH + th) * TL::HW) * XSW;   // wave-uniform
#pragma unroll
      for (int kw = 0; kw < TW / 4; ++kw) {
        const int m0 = row * TW + kw * 4;
        float b[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = Ys[m0 * YS + yoff + nt * 16];
        // no per-tap guard here: a wave whose last tap slot is past T (27 = 4*7 - 1) recomputes tap 0 into an accumulator
        // that is never stored -- 1/28 wasted MFMAs instead of predicated MFMAs and accumulator shuffles
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const float a = Xs[xrow + kw * 4 * XSW + xoff[t]];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nt], acc[t][nt], 0, 0, 0);
        }
      }
    }
    if (!has_next) break;
    __syncthreads();
    stash();
    __syncthreads();
    ++tile;
  }
  // partial[grp][tap][ci][co]: lane (li, lg) holds ci = lg*4 + r (rows), co = li (cols)
  float* P = partial + (long long)grp * T * cd.Cin16 * cd.Cout16;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = wave + 4 * t;
    if (tap < T) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          P[((long long)tap * cd.Cin16 + cc * 16 + lg * 4 + r) * cd.Cout16 + cout0 + nt * 16 + li] = acc[t][nt][r];
    }
  }
}

// dW_torch[co][ci][tap] (+)= sum_g partial[g][tap][ci][co].  One block per (ci, 64-wide co slab): the [T][64] tile is
// read with co fastest (coalesced), transposed through LDS and written as T-contiguous runs per (co, ci).
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dW, int G, int T,
                                                      int Cin, int Cout, int Cin16, int Cout16, int accumulate) {
  __shared__ float tile[27 * 65];
  const int ci = blockIdx.x, co0 = blockIdx.y * 64;
  const long long slab = (long long)T * Cin16 * Cout16;
  for (int q = threadIdx.x; q < T * 64; q += 256) {
    const int col = q & 63, tap = q >> 6;
    float s = 0.f;
    if (co0 + col < Cout) {
      const float* p = partial + ((long long)tap * Cin16 + ci) * Cout16 + co0 + col;
      // four independent partial sums: one dependent add chain over the G slabs serialised the loads (17.7 us for 14 MB at the
      // 128-channel level, a quarter of the weight-gradient kernel it follows)
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int g = 0;
      for (; g + 3 < G; g += 4) {
        const float a = p[g * slab], b = p[(g + 1) * slab], c = p[(g + 2) * slab], d = p[(g + 3) * slab];
        s0 += a; s1 += b; s2 += c; s3 += d;
      }
      for (; g < G; ++g) s0 += p[g * slab];
      s = (s0 + s1) + (s2 + s3);
    }
    tile[tap * 65 + col] = s;
  }
  __syncthreads();
  for (int q = threadIdx.x; q < T * 64; q += 256) {
    const int tap = q % T, col = q / T;
    if (co0 + col < Cout) {
      float* o = dW + ((long long)(co0 + col) * Cin + ci) * T + tap;
      const float v = tile[tap * 65 + col];
      *o = accumulate ? (*o + v) : v;
    }
  }
}

// Many-group variant (shallow layers: one cin chunk x one slab => hundreds of spatial groups): one block per
// (ci, CW-wide co slab, tap); 256 threads = CW co x (256 / CW) group-slots, 4 independent loads in flight per thread,
// LDS sum over the slots in a fixed order (deterministic).  CW = 16 for the 16-channel layers: with a fixed 64-wide
// slab three quarters of the block idled and each thread walked 128 slabs one load at a time (31 us per reduce,
// ~18 % of the 16->16 weight gradient).
template <int CW>
__global__ __launch_bounds__(256) void k_wgrad_reduce_deep(const float* __restrict__ partial, float* __restrict__ dW, int G, int T,
                                                           int Cin, int Cout, int Cin16, int Cout16, int accumulate) {
  constexpr int SL = 256 / CW;
  __shared__ float red[SL][CW];
  const int ci = blockIdx.x, co0 = blockIdx.y * CW, tap = blockIdx.z;
  const int col = threadIdx.x % CW, slot = threadIdx.x / CW;
  const long long slab = (long long)T * Cin16 * Cout16;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (co0 + col < Cout) {
    const float* p = partial + ((long long)tap * Cin16 + ci) * Cout16 + co0 + col;
    int g = slot;
    for (; g + 3 * SL < G; g += 4 * SL) {
      const float a = p[(long long)g * slab], b = p[(long long)(g + SL) * slab];
      const float c = p[(long long)(g + 2 * SL) * slab], d = p[(long long)(g + 3 * SL) * slab];
      s0 += a; s1 += b; s2 += c; s3 += d;
    }
    for (; g < G; g += SL) s0 += p[(long long)g * slab];
  }
  red[slot][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (slot == 0 && co0 + col < Cout) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < SL; ++k) v += red[k][col];
    float* o = dW + ((long long)(co0 + col) * Cin + ci) * T + tap;
    *o = accumulate ? (*o + v) : v;
  }
}

// Round 6: the same sum with the SLAB as the fastest dimension of the read: a workgroup owns 64 float4 columns of the [T][Cin16][Cout16] slab
// (1 KB contiguous per wave and slab) and 16 group slots, 4 independent float4 loads in flight per thread; the 16 slots are added in order
// through the LDS (deterministic), the finished float4 (four consecutive co of one (tap, ci)) is scattered into dW[co][ci][tap].  The kernel
// above reads 4 bytes per lane at a slab stride: 53 us for the 506 slabs (14 MB) of the 16 -> 16 layer, on the weight-gradient stream.
__global__ __launch_bounds__(1024) void k_wgrad_reduce_flat(const float* __restrict__ partial, float* __restrict__ dW, int G, int T, int Cin,
                                                            int Cout, int Cin16, int Cout16, int accumulate) {
  __shared__ float4 red[16][64];
  const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const long long sq = (long long)T * Cin16 * Cout16 / 4;      // float4 per slab
  const long long q = (long long)blockIdx.x * 64 + lane;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  if (q < sq) {
    const float4* p = reinterpret_cast<const float4*>(partial) + q;
    int g = slot;
    for (; g + 48 < G; g += 64) {
      const float4 a = p[(long long)g * sq], b = p[(long long)(g + 16) * sq], c = p[(long long)(g + 32) * sq], d = p[(long long)(g + 48) * sq];
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
      s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
      s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
      s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
    }
    for (; g < G; g += 16) { const float4 a = p[(long long)g * sq]; s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; }
  }
  red[slot][lane] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
  __syncthreads();
  if (slot == 0 && q < sq) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) { const float4 t = red[k][lane]; v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
    const long long e = q * 4;
    const int co = (int)(e % Cout16), ci = (int)((e / Cout16) % Cin16), tap = (int)(e / ((long long)Cout16 * Cin16));
    if (ci < Cin) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (co + k < Cout) {
          float* o = dW + ((long long)(co + k) * Cin + ci) * T + tap;
          *o = accumulate ? (*o + v[k]) : v[k];
        }
    }
  }
}

static void launch_reduce_deep(const float* ws, float* dw, int G, int T, int Cin, int Cout, int Cin16, int Cout16, int accumulate,
                               hipStream_t s) {
  // (slabs of at least 24 x 64 float4: the 2-D 16 -> 16 layers' 576-float4 slabs are 9 workgroups of this kernel, 17 us where the one below takes 9)
  if (options().wgrad_reduce_flat != 0 && (Cout16 & 3) == 0 && aligned16(ws) &&
      (options().wgrad_reduce_flat == 2 || (long long)T * Cin16 * Cout16 / 4 >= 24 * 64)) {
    const long long sq = (long long)T * Cin16 * Cout16 / 4;
    hipLaunchKernelGGL(k_wgrad_reduce_flat, dim3((unsigned)((sq + 63) / 64)), dim3(1024), 0, s, ws, dw, G, T, Cin, Cout, Cin16, Cout16, accumulate);
    return;
  }
  if (Cout <= 16)
    hipLaunchKernelGGL((k_wgrad_reduce_deep<16>), dim3(Cin, cdiv(Cout, 16), T), dim3(256), 0, s, ws, dw, G, T, Cin, Cout, Cin16, Cout16, accumulate);
  else if (Cout <= 32)
    hipLaunchKernelGGL((k_wgrad_reduce_deep<32>), dim3(Cin, cdiv(Cout, 32), T), dim3(256), 0, s, ws, dw, G, T, Cin, Cout, Cin16, Cout16, accumulate);
  else
    hipLaunchKernelGGL((k_wgrad_reduce_deep<64>), dim3(Cin, cdiv(Cout, 64), T), dim3(256), 0, s, ws, dw, G, T, Cin, Cout, Cin16, Cout16, accumulate);
}

// ------------------------------------------------------------------------------------------------
// weight packing (torch [Cout][Cin][T] -> Wp[tap][Cin16/4][Cout16][4]), forward and dgrad flavours
// ------------------------------------------------------------------------------------------------
// fwd:   Wp[t][ci/4][co][ci%4] = w[co][ci][t]                       (GEMM K = cin, N = cout)
// dgrad: Wp[t][co/4][ci][co%4] = w[co][ci][T-1-t]                   (K = cout, N = cin; flipped taps)
__global__ __launch_bounds__(256) void k_pack_conv3(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                                    int T, int K16, int N16, int dgrad) {
  const long long total = (long long)T * K16 * N16;
  float* hdr = wp + pack_off_hdr(T, K16, N16);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k4 = (int)(i & 3);
    const int nn = (int)((i >> 2) % N16);
    const int kq = (int)((i / (4LL * N16)) % (K16 / 4));
    const int t = (int)(i / ((long long)K16 * N16));
    const int kk = kq * 4 + k4;
    float v = 0.f;
    if (!dgrad) {
      if (kk < Cin && nn < Cout) v = w[((long long)nn * Cin + kk) * T + t];
    } else {
      if (kk < Cout && nn < Cin) v = w[((long long)kk * Cin + nn) * T + (T - 1 - t)];
    }
    wp[i] = v;
  }
  // three-piece bf16 section behind the fp32 pack (conv3b.hip): Wb[k / 16][pair][piece][n (N16)][32: (tap & 1) * 16 + k % 16]
  const int TP = (T + 1) / 2;
  unsigned short* wb = reinterpret_cast<unsigned short*>(wp + total);
  const long long tb16 = (long long)TP * K16 * N16 * 2;          // (value, piece) triples: one per (chunk, pair, n, j)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tb16; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 31);
    const int nn = (int)((i >> 5) % N16);
    const int tp = (int)((i / (32LL * N16)) % TP);
    const int ch = (int)(i / (32LL * N16 * TP));
    const int t = 2 * tp + (j >> 4), kk = ch * 16 + (j & 15);
    float v = 0.f;
    if (t < T) {
      if (!dgrad) { if (kk < Cin && nn < Cout) v = w[((long long)nn * Cin + kk) * T + t]; }
      else if (kk < Cout && nn < Cin) v = w[((long long)kk * Cin + nn) * T + (T - 1 - t)];
    }
    unsigned short pc[3];
    split3_bf16(v, pc);
#pragma unroll
    for (int q = 0; q < 3; ++q) wb[((((long long)ch * TP + tp) * 3 + q) * N16 + nn) * 32 + j] = pc[q];
    // two-plane fp16 section (round 4, conv3b.hip PL = 2), pre-scaled by the layer's power-of-two scale (header: k_wamax)
    {
      float amax = 0.f;
#pragma unroll
      for (int b = 0; b < 16; ++b) amax = fmaxf(amax, hdr[1 + b]);
      const float sc = f16_scale(amax);
      unsigned short* wh = reinterpret_cast<unsigned short*>(wp + pack_off_f16(T, K16, N16));
      const float vs = v * sc;
      const _Float16 h0 = (_Float16)vs;
      const _Float16 h1 = (_Float16)(vs - (float)h0);
      wh[((((long long)ch * TP + tp) * 2 + 0) * N16 + nn) * 32 + j] = __builtin_bit_cast(unsigned short, h0);
      wh[((((long long)ch * TP + tp) * 2 + 1) * N16 + nn) * 32 + j] = __builtin_bit_cast(unsigned short, h1);
      if (i == 0) hdr[0] = amax;
    }
  }
}

// max |w| over count floats, this block's share (blocks of a 16-wide grid row interleave 1024-float chunks): 16-byte loads, four
// independent chains per thread (round 4, first version: one scalar load per iteration = a dependent-latency loop, 100 us for the V-Net)
// (round 6: any block size up to 1024 -- the many-layer launch runs 1024 threads per block: with 256 the 256 x 256 x 27 layers were 27 dependent
// trips per thread, 14.8 us at the head of both networks' forward passes)
__device__ __forceinline__ float wamax_block(const float* __restrict__ w, long long count, float* red /* [16] shared */) {
  float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
  const int nt = blockDim.x;
  if ((reinterpret_cast<uintptr_t>(w) & 15u) == 0) {
    const long long nv = count >> 2, stride = 16LL * nt;
    long long i = (long long)blockIdx.x * nt + threadIdx.x;
    for (; i + 3 * stride < nv; i += 4 * stride) {
      const float4 a = ld4(w + i * 4), b = ld4(w + (i + stride) * 4), c = ld4(w + (i + 2 * stride) * 4), d = ld4(w + (i + 3 * stride) * 4);
      m0 = fmaxf(m0, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
      m1 = fmaxf(m1, fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
      m2 = fmaxf(m2, fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))));
      m3 = fmaxf(m3, fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w))));
    }
    for (; i < nv; i += stride) {
      const float4 a = ld4(w + i * 4);
      m0 = fmaxf(m0, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < (int)(count & 3)) m1 = fmaxf(m1, fabsf(w[(nv << 2) + threadIdx.x]));
  } else {
    for (long long i = (long long)blockIdx.x * nt + threadIdx.x; i < count; i += 16LL * nt) m0 = fmaxf(m0, fabsf(w[i]));
  }
  float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  float r = red[0];
  for (int k = 1; k < (nt >> 6); ++k) r = fmaxf(r, red[k]);
  return r;
}

// max |w| of a layer in 16 per-block partials -> header[1 .. 16] of its pack (no atomics, nothing to zero): the packers read them
__global__ __launch_bounds__(1024) void k_wamax(const float* __restrict__ w, long long count, float* __restrict__ hdr) {
  __shared__ float red[16];
  const float m = wamax_block(w, count, red);
  if (threadIdx.x == 0) hdr[1 + blockIdx.x] = m;
  if (blockIdx.x == 0 && threadIdx.x >= 17 && threadIdx.x < kPackHeaderFloats) hdr[threadIdx.x] = 0.f;      // reserved words: defined contents
}

// all conv layers of a network in ONE launch (blockIdx.y = descriptor): 80 five-microsecond pack launches per step otherwise
struct PackDesc { const float* w; float* wp; int Cout, Cin, T, K16, N16, dgrad; };
static constexpr int kMaxPackDescs = 512;

// max |w| of every layer of the descriptor list: block (b, di) leaves the b-th of 16 partial maxima in header[1 + b] of desc di's pack
__global__ __launch_bounds__(1024) void k_wamax_many(const PackDesc* __restrict__ descs) {
  __shared__ float red[16];
  const PackDesc d = descs[blockIdx.y];
  float* hdr = d.wp + pack_off_hdr(d.T, d.K16, d.N16);
  // bit 11 of the descriptor's last word (round 6): "the previous descriptor packs the SAME weight tensor" (the dgrad twin behind its forward
  // descriptor): its partial maxima are the twin's -- k_pack_conv3_many reads them there -- and the weights are not read a second time
  if (d.dgrad & 0x800) {
    if (blockIdx.x == 0 && threadIdx.x >= 17 && threadIdx.x < kPackHeaderFloats) hdr[threadIdx.x] = 0.f;
    return;
  }
  const float m = wamax_block(d.w, (long long)d.Cout * d.Cin * d.T, red);
  if (threadIdx.x == 0) hdr[1 + blockIdx.x] = m;
  if (blockIdx.x == 0 && threadIdx.x >= 17 && threadIdx.x < kPackHeaderFloats) hdr[threadIdx.x] = 0.f;      // reserved words: defined contents
}
// Work unit = one 16 x 16 (k, n) block of one layer, all taps: the 16 source rows are runs of 16*T contiguous floats
// (coalesced reads), transposed through LDS, written as 256-B runs.  Units are dealt round-robin to the blocks, so the
// 256-channel layers (256 units each) no longer serialise on a fixed 64 blocks of scattered 4-byte gathers.
__global__ __launch_bounds__(256) void k_pack_conv3_many(const PackDesc* __restrict__ descs, int n, int sections) {
  // sections (measurement switch, option pack_sections; 7 = everything): bit 0 the fp32 pack, bit 1 the three bf16 planes, bit 2 the two fp16 planes
  __shared__ float tile[16 * 16 * 27];
  __shared__ int ubeg[kMaxPackDescs + 1];           // exclusive prefix sums of the per-layer unit counts
  // (scanning the descriptor array in global memory per unit cost ~80 dependent scalar loads = 15+ us per unit)
  // counts loaded in parallel (one descriptor per thread), prefix-summed out of LDS: a single thread walking the
  // descriptors paid one dependent global-load latency per layer (~80 us for 84 descriptors, in EVERY block)
  for (int i = threadIdx.x; i < n; i += blockDim.x) ubeg[i + 1] = (descs[i].K16 >> 4) * (descs[i].N16 >> 4);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    ubeg[0] = 0;
    for (int i = 1; i <= n; ++i) { t += ubeg[i]; ubeg[i] = t; }
  }
  __syncthreads();
  const int total = ubeg[n];
  for (int unit = blockIdx.x; unit < total; unit += gridDim.x) {
    int lo = 0, hi = n;                             // largest di with ubeg[di] <= unit
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ubeg[mid] <= unit) lo = mid; else hi = mid; }
    const int di = lo, u = unit - ubeg[lo];
    PackDesc d = descs[di];
    // the descriptor's last word: bit 0 = dgrad flavour, bits 8-10 = the sections this layer's launches read (0: all) -- round 6: a network
    // repacks every weight every step, and most layers only ever read ONE of the three sections (the two fp16 planes)
    int sec = (d.dgrad >> 8) & 7;
    sec = (sec ? sec : 7) & sections;
    const bool twin = (d.dgrad & 0x800) != 0 && di > 0;      // (k_wamax_many: the weight's partial maxima sit in the previous descriptor's header)
    d.dgrad &= 1;
    const int nb = d.N16 >> 4, k0 = (u / nb) * 16, n0 = (u % nb) * 16, T = d.T;
    // source rows: fwd w[nn][kk..][t] (row = nn, inner = kk = cin); dgrad w[kk][nn..][t] (row = kk = cout, inner = nn = cin)
    const int row0 = d.dgrad ? k0 : n0, in0 = d.dgrad ? n0 : k0;
    for (int q = threadIdx.x; q < 256 * T; q += 256) {
      const int r = q / (16 * T), rem = q - r * 16 * T;
      const int in = rem / T, t = rem - in * T;
      float v = 0.f;
      if (row0 + r < d.Cout && in0 + in < d.Cin) v = d.w[((long long)(row0 + r) * d.Cin + in0 + in) * T + t];
      tile[q] = v;                                   // tile[row][inner][t]
    }
    __syncthreads();
    if (sec & 1)
    for (int q = threadIdx.x; q < 256 * T; q += 256) {
      const int k4 = q & 3, nn = (q >> 2) & 15, kq = (q >> 6) & 3, t = q >> 8;
      const int kk = kq * 4 + k4;
      const float v = d.dgrad ? tile[(kk * 16 + nn) * T + (T - 1 - t)] : tile[(nn * 16 + kk) * T + t];
      d.wp[(((long long)t * (d.K16 >> 2) + (k0 >> 2) + kq) * d.N16 + n0 + nn) * 4 + k4] = v;
    }
    {  // three-piece bf16 section (see k_pack_conv3): this unit = chunk k0 / 16, columns n0 .. n0 + 15; two k per thread and step:
       // exactly one packed pair per piece (v_cvt_pk_bf16_f32; the integer split of the first version cost 80 of the launch's 150 us)
      const int TP = (T + 1) / 2, ch = k0 >> 4;
      unsigned* wb = reinterpret_cast<unsigned*>(d.wp + (long long)T * d.K16 * d.N16);
      if (sec & 2)
      for (int q = threadIdx.x; q < TP * 16 * 16; q += 256) {
        const int j2 = q & 15, nn = (q >> 4) & 15, tp = q >> 8;
        const int t = 2 * tp + (j2 >> 3), kk = (j2 * 2) & 15;
        float v0 = 0.f, v1 = 0.f;
        if (t < T) {
          v0 = d.dgrad ? tile[(kk * 16 + nn) * T + (T - 1 - t)] : tile[(nn * 16 + kk) * T + t];
          v1 = d.dgrad ? tile[((kk + 1) * 16 + nn) * T + (T - 1 - t)] : tile[(nn * 16 + kk + 1) * T + t];
        }
        unsigned* o = wb + ((((((long long)ch * TP + tp) * 3) * d.N16 + n0 + nn) * 32 + j2 * 2) >> 1);
        const long long ps = ((long long)d.N16 * 32) >> 1;             // piece stride in packed pairs
        unsigned h = cvt_pk_bf16(v0, v1);
        o[0] = h;
        v0 -= __uint_as_float(h << 16); v1 -= __uint_as_float(h & 0xffff0000u);
        h = cvt_pk_bf16(v0, v1);
        o[ps] = h;
        v0 -= __uint_as_float(h << 16); v1 -= __uint_as_float(h & 0xffff0000u);
        o[2 * ps] = cvt_pk_bf16(v0, v1);
      }
      // two-plane fp16 section (round 4): same (chunk, pair, n, k) order with two pieces, pre-scaled by the layer's power of two
      float* hdr = d.wp + pack_off_hdr(T, d.K16, d.N16);
      const float* hsrc = hdr;
      if (twin) { const PackDesc t = descs[di - 1]; hsrc = t.wp + pack_off_hdr(t.T, t.K16, t.N16); }
      float amax = 0.f;
#pragma unroll
      for (int b = 0; b < 16; ++b) amax = fmaxf(amax, hsrc[1 + b]);
      const float sc = f16_scale(amax);
      if (u == 0 && threadIdx.x == 0) hdr[0] = amax;
      unsigned* wh = reinterpret_cast<unsigned*>(d.wp + pack_off_f16(T, d.K16, d.N16));
      if (sec & 4)
      for (int q = threadIdx.x; q < TP * 16 * 16; q += 256) {
        const int j2 = q & 15, nn = (q >> 4) & 15, tp = q >> 8;
        const int t = 2 * tp + (j2 >> 3), kk = (j2 * 2) & 15;
        float v0 = 0.f, v1 = 0.f;
        if (t < T) {
          v0 = d.dgrad ? tile[(kk * 16 + nn) * T + (T - 1 - t)] : tile[(nn * 16 + kk) * T + t];
          v1 = d.dgrad ? tile[((kk + 1) * 16 + nn) * T + (T - 1 - t)] : tile[(nn * 16 + kk + 1) * T + t];
        }
        v0 *= sc; v1 *= sc;
        unsigned* o = wh + ((((((long long)ch * TP + tp) * 2) * d.N16 + n0 + nn) * 32 + j2 * 2) >> 1);
        const long long ps = ((long long)d.N16 * 32) >> 1;
        const unsigned h = cvt_pk_f16(v0, v1);
        o[0] = h;
        o[ps] = cvt_pk_f16(v0 - f16lo_to_f32(h), v1 - f16hi_to_f32(h));
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// first layer: Cin = 1 -> Cout = 16 (HBM-bound: 4 B in, 64 B out per voxel).  Still a matrix product -- voxels x taps
// times taps x 16 channels -- so it runs on the matrix cores as well: A[vox][tap] is read straight out of the
// single-channel LDS halo at shifted offsets, B[tap][co] sits in registers, K = T taps padded to a multiple of 4 with
// zero weights.  (The first version did 432 VALU FMAs per voxel with one LDS broadcast read per FMA: 5x off the HBM
// roofline.)
// ------------------------------------------------------------------------------------------------
// EPI: what happens to y = conv + bias (round 3: the layer is HBM-bound -- 4 B in, 64 B out per voxel -- and the conv is 27 MACs per
// output, so the norm around it RECOMPUTES y instead of storing and re-reading it; bcp_conv3_c1_norm_fwd / _bwd):
//   0  store y (+ forward statistics when st.partial)            bcp_conv3_c1_fwd / _fwd_stats
//   1  forward statistics only, nothing stored                    pass 1 of the fused forward
//   2  a = act((y - mean) * scale + shift) [* mask * s] -> out    pass 2 of the fused forward: y never exists in HBM
//   3  backward statistics (sum dz, sum dz * xhat) from da        pass 1 of the fused backward
//   4  dy = scale * (dz - c1 - xhat * c2) -> out                  pass 2 of the fused backward
//   5  (round 5) dy as in 4, but into LDS, and THIS layer's weight gradient from it: dW[tap][co] += sum_v x[v + off(tap)] * dy[v][co] on
//      the matrix cores (k_conv3_c1_wgrad's product; the x halo is in LDS already, a wave multiplies the 64 voxels it just differentiated)
//      -> one partial [T][16] slab per workgroup at Y.  The first layer has no dgrad, so dy had no other reader: 128 MB written and read
//      back at the LA size, and the last launch of the backward pass (k_conv3_c1_wgrad, alone on the step's tail), are gone
//      (bcp_conv3_c1_norm_bwd_wgrad)
struct C1Norm {
  const float* stats;          // float[5][G][16]: mean, rstd, scale, shift, ...
  const float* da;             // EPI 3 / 4: gradient w.r.t. the activation, [voxel][16]
  const float* c1c2;           // EPI 4: float[2][G][16]
  const uint8_t* elem_mask;    // nullable [voxel][16]: elementwise Dropout keep mask (U-Net)
  float elem_scale;
  int act, G;
  float* amax;                 // EPI 2, nullable: |max| slots of the activation written (round 4: the fp16 pre-scale of the conv that reads it)
  const unsigned long long* mask_seed;   // nullable: evaluate the Dropout keep bits from this device seed instead of reading elem_mask
  float p_keep;                          //   (bern_keep, common.h; see norm.hip NormEpilogue)
};

template <int KD, int TD, int TH, int TW, int EPI = 0>
__global__ __launch_bounds__(256) void k_conv3_c1(const float* __restrict__ X, const float* __restrict__ w /*[16][1][T]*/,
                                                  const float* __restrict__ bias, float* __restrict__ Y, ConvDims cd, int n_tiles,
                                                  int tiles_per_block, StatsArg st, C1Norm nm) {
  using TL = Tile<KD, TD, TH, TW>;
  constexpr int T = TL::T, MT = TL::MT, KS = (T + 3) / 4;
  __shared__ __attribute__((aligned(16))) float Xs[TL::HV];
  __shared__ double Ss[4 * 16 * 2];
  __shared__ __attribute__((aligned(16))) float Ys[EPI == 5 ? TL::M * 16 : 4];      // EPI 5: dy of the current tile, [voxel][16]
  static_assert(EPI != 5 || (TL::M == 256 && T <= 32 && MT == 4), "EPI 5: four waves x 64 voxels, two 16-tap MFMA row tiles");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  int aoff[2] = {0, 0};                                    // EPI 5: lane's tap offset for the two row tiles of the weight-gradient product (tap = tt*16 + li)
  f32x4 wacc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  if (EPI == 5) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) aoff[tt] = (tt * 16 + li < T) ? TL::tapoff(tt * 16 + li) : 0;
  }
  // lane (li, lg): A[co = li][k = lg] of k-step ks is the weight of tap 4*ks + lg.  D = W x X^T (rows = channels, columns = voxels):
  // a lane ends with FOUR CONSECUTIVE channels (lg*4 ..) of voxel li -- one 16-byte store per m-tile (round 3; the dword stores of
  // the [vox][co] orientation ran this 128 MB stream at 3.2 TB/s) and the layout the fused norm statistics want.
  float wv[KS];
  int toff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int tap = ks * 4 + lg;
    wv[ks] = tap < T ? w[li * T + tap] : 0.f;
    toff[ks] = tap < T ? TL::tapoff(tap) : 0;
  }
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bv = ld4(bias + lg * 4);
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};       // fused norm statistics of this lane's four channels (bcp_conv3_c1_fwd_stats)
  float amax_o = 0.f;                                      // EPI 2: max |a| of what this thread writes
  unsigned mseed_lo = 0, mseed_hi = 0;
  if (EPI >= 2 && nm.mask_seed) { const unsigned long long s_ = *nm.mask_seed; mseed_lo = (unsigned)s_; mseed_hi = (unsigned)(s_ >> 32); }
  const bool has_mask = EPI >= 2 && (nm.elem_mask != nullptr || nm.mask_seed != nullptr);
  const int t_begin = blockIdx.x * tiles_per_block;
  int t_end = t_begin + tiles_per_block;
  if (t_end > n_tiles) t_end = n_tiles;
  // EPI >= 2: this lane's four channels of the statistics rows of the workgroup's group (its tiles lie in ONE group)
  float mu[4] = {0, 0, 0, 0}, rs[4] = {1, 1, 1, 1}, sc[4] = {1, 1, 1, 1}, sh[4] = {0, 0, 0, 0}, k1[4] = {0, 0, 0, 0}, k2[4] = {0, 0, 0, 0};
  if (EPI >= 2) {
    const int g = t_begin / st.tiles_per_group;
    const long long GC = (long long)nm.G * 16, o = (long long)g * 16 + lg * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mu[r] = nm.stats[o + r]; rs[r] = nm.stats[GC + o + r]; sc[r] = nm.stats[2 * GC + o + r]; sh[r] = nm.stats[3 * GC + o + r];
      if (EPI >= 4) { k1[r] = nm.c1c2[o + r]; k2[r] = nm.c1c2[GC + o + r]; }
    }
  }
  // 3-D: the halo of the NEXT tile travels in registers under the current tile's MFMAs (round 4, same box, three interleaved pairs: LA
  // step 5.29-5.32 vs 5.30-5.36 ms; the kernel ALONE stays at 34 us for 11 us of matrix work -- its fp64 statistics and one dependent
  // 7-MFMA chain per m-tile bound it, not the round trip).  2-D (PRE = false: fetched at the top of its own tile): the U-Net launches
  // one or two 16x16 tiles per workgroup and the ACDC step was 3.43-3.44 vs 3.41-3.43 ms with the prefetch (tools/sessions/r04_s23.sh)
  constexpr bool PRE = KD == 3;
  constexpr int NPRE = (TL::HV + 255) / 256;
  float pre[NPRE];
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    int n, d0, h0, w0;
    tile_origin(cd, tile, TD, TH, TW, n, d0, h0, w0);
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int q = threadIdx.x + i * 256;
      const int hw = q % TL::HW, hh = (q / TL::HW) % TL::HH, hd = q / (TL::HW * TL::HH);
      const int d = d0 - TL::PD + hd, h = h0 - 1 + hh, wq = w0 - 1 + hw;
      float v = 0.f;
      if (q < TL::HV && (unsigned)d < (unsigned)cd.D && (unsigned)h < (unsigned)cd.H && (unsigned)wq < (unsigned)cd.W)
        v = X[(((long long)n * cd.D + d) * cd.H + h) * cd.W + wq];
      pre[i] = v;
    }
  };
  if (PRE && t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    int n, d0, h0, w0;
    tile_origin(cd, tile, TD, TH, TW, n, d0, h0, w0);
    if (tile > t_begin) __syncthreads();                  // every wave is done with the previous tile's halo
    if (!PRE) fetch(tile);
#pragma unroll
    for (int i = 0; i < NPRE; ++i)
      if (threadIdx.x + i * 256 < TL::HV) Xs[threadIdx.x + i * 256] = pre[i];
    __syncthreads();
    if (PRE && tile + 1 < t_end) fetch(tile + 1);
    const bool full = d0 + TD <= cd.D && h0 + TH <= cd.H && w0 + TW <= cd.W;   // uniform
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = (wave * MT + mt) * 16 + li;             // B[k = lg][vox = li]
      const int vo = TL::voff(m);
      const int tw = m % TW, th = (m / TW) % TH, td = m / (TW * TH);
      const int d = d0 + td, h = h0 + th, wq = w0 + tw;
      const bool ok = full || (d < cd.D && h < cd.H && wq < cd.W);
      const long long e = ((((long long)n * cd.D + d) * cd.H + h) * cd.W + wq) * 16 + lg * 4;
      float4 dav = make_float4(0.f, 0.f, 0.f, 0.f);
      uchar4 m4 = make_uchar4(1, 1, 1, 1);
      if (EPI >= 3 && ok) dav = ld4(nm.da + e);              // (requested before the MFMAs: the round trip hides under them)
      if (has_mask && ok) m4 = nm.mask_seed ? bern_keep4(e, mseed_lo, mseed_hi, nm.p_keep) : *reinterpret_cast<const uchar4*>(nm.elem_mask + e);
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ks], Xs[vo + toff[ks]], acc, 0, 0, 0);
      float4 dyl = make_float4(0.f, 0.f, 0.f, 0.f);          // EPI 5: this lane's four dy values (zero outside the volume)
      if (ok) {
        const float yv[4] = {acc[0] + bv.x, acc[1] + bv.y, acc[2] + bv.z, acc[3] + bv.w};
        if (EPI == 0) st4(Y + e, make_float4(yv[0], yv[1], yv[2], yv[3]));
        if (EPI <= 1) {
          if (st.partial) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[r] += (double)yv[r]; s2[r] += (double)yv[r] * (double)yv[r]; }
          }
        } else {
          const float ms[4] = {m4.x ? nm.elem_scale : 0.f, m4.y ? nm.elem_scale : 0.f, m4.z ? nm.elem_scale : 0.f, m4.w ? nm.elem_scale : 0.f};
          const float dd[4] = {dav.x, dav.y, dav.z, dav.w};
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {                     // the arithmetic of k_norm_apply / k_col_partial<1> / k_norm_bwd_apply
            const float z = (yv[r] - mu[r]) * sc[r] + sh[r];
            if (EPI == 2) {
              o[r] = act_fwd(z, nm.act);
              if (has_mask) o[r] *= ms[r];
              const float t = fabsf(o[r]);
              amax_o = (t > amax_o || t != t) ? t : amax_o;
            } else {
              const float cs = has_mask ? ms[r] : 1.f;
              const float dz = dd[r] * cs * act_grad(z, nm.act);
              const float xh = (yv[r] - mu[r]) * rs[r];
              if (EPI == 3) { s1[r] += (double)dz; s2[r] += (double)dz * (double)xh; }
              else o[r] = sc[r] * (dz - k1[r] - xh * k2[r]);
            }
          }
          if (EPI == 2 || EPI == 4) st4(Y + e, make_float4(o[0], o[1], o[2], o[3]));
          if (EPI == 5) dyl = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
      if (EPI == 5) st4(Ys + m * 16 + lg * 4, dyl);
    }
    if (EPI == 5) {
      // D[tap = tt*16 + lg*4 + r][co = li] += sum over this wave's 64 voxels (the ones it just wrote: rows wave*64 .. +63 of Ys)
      __syncthreads();
#pragma unroll 4
      for (int ks = 0; ks < 16; ++ks) {
        const int v = wave * 64 + ks * 4 + lg;               // A[tap = li][k = lg], B[k = lg][co = li]
        const int vo = TL::voff(v);
        const float b = Ys[v * 16 + li];
        wacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Xs[vo + aoff[0]], b, wacc[0], 0, 0, 0);
        wacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Xs[vo + aoff[1]], b, wacc[1], 0, 0, 0);
      }
    }
  }
  if (EPI == 5) {      // sum the four waves, write this workgroup's partial [tap][co] slab (k_conv3_c1_wgrad's ending)
    __syncthreads();
    float* red = Ys;                                         // [4 waves][32 taps][16]
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 32 + tt * 16 + lg * 4 + r) * 16 + li] = wacc[tt][r];
    __syncthreads();
    for (int o = threadIdx.x; o < T * 16; o += 256)
      Y[(long long)blockIdx.x * T * 16 + o] = (red[o] + red[512 + o]) + (red[1024 + o] + red[1536 + o]);
  }
  if (st.partial) {
    // one row per workgroup (its tiles lie in ONE normalisation group: the launcher picks tiles_per_block | tiles_per_group)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double a = s1[r], b = s2[r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
      if (li == 0) { Ss[(wave * 16 + lg * 4 + r) * 2] = a; Ss[(wave * 16 + lg * 4 + r) * 2 + 1] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 16) {
      const int c = threadIdx.x;
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int wv_ = 0; wv_ < 4; ++wv_) { a += Ss[(wv_ * 16 + c) * 2]; b += Ss[(wv_ * 16 + c) * 2 + 1]; }
      const int gg = t_begin / st.tiles_per_group, row = (t_begin % st.tiles_per_group) / tiles_per_block;
      double* dst = st.partial + (((long long)gg * st.rows + row) * 16 + c) * 2;
      dst[0] = a; dst[1] = b;
    }
  }
  if (EPI == 2 && nm.amax) { __syncthreads(); block_amax_publish(amax_o, nm.amax); }
}

// wgrad of the Cin = 1 layer: dW[co][0][tap] = sum_v x[v + off(tap)] * dY[v][co] -- a (taps x voxels) x (voxels x 16)
// product: M = taps (two 16-row MFMA tiles, rows >= T unused), N = 16 channels, K = voxels.  The four waves split the
// voxels of a tile; a block walks a group of tiles with the next tile's x halo and dY rows prefetched into registers,
// sums its four wave accumulators through LDS at the end and writes one partial [T][16] slab (reduced deterministically
// by k_wgrad_reduce(_deep), Cin16 = 1 there).  This layer's weight gradient is the LAST kernel of the backward pass --
// nothing is left to overlap it with, so its 0.2 ms (VALU version) sat on the step's critical path.
template <int KD, int TD, int TH, int TW>
__global__ __launch_bounds__(256) void k_conv3_c1_wgrad(const float* __restrict__ X, const float* __restrict__ dY,
                                                        float* __restrict__ partial, ConvDims cd, int tiles_total,
                                                        int tiles_per_group) {
  using TL = Tile<KD, TD, TH, TW>;
  constexpr int T = TL::T, M = TL::M;
  static_assert(M == 256 && T <= 32, "four waves x 64 voxels, two 16-tap MFMA row tiles");
  constexpr int NXS = (TL::HV + 255) / 256;            // halo floats per thread
  __shared__ __attribute__((aligned(16))) float Xs[TL::HV];
  __shared__ __attribute__((aligned(16))) float Ys[M * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int grp = blockIdx.x;
  int t_end = (grp + 1) * tiles_per_group;
  if (t_end > tiles_total) t_end = tiles_total;
  int tile = grp * tiles_per_group;
  if (tile >= t_end) return;

  int aoff[2];                                         // lane's tap offset for the two row tiles (tap = tt*16 + li)
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) aoff[tt] = (tt * 16 + li < T) ? TL::tapoff(tt * 16 + li) : 0;
  f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};

  float px[NXS];
  float4 py[4];
  auto fetch = [&](int tl) {
    int n, d0, h0, w0;
    tile_origin(cd, tl, TD, TH, TW, n, d0, h0, w0);
#pragma unroll
    for (int u = 0; u < NXS; ++u) {
      const int q = threadIdx.x + u * 256;
      float v = 0.f;
      if (q < TL::HV) {
        const int hw = q % TL::HW, hh = (q / TL::HW) % TL::HH, hd = q / (TL::HW * TL::HH);
        const int d = d0 - TL::PD + hd, h = h0 - 1 + hh, wq = w0 - 1 + hw;
        if ((unsigned)d < (unsigned)cd.D && (unsigned)h < (unsigned)cd.H && (unsigned)wq < (unsigned)cd.W)
          v = X[(((long long)n * cd.D + d) * cd.H + h) * cd.W + wq];
      }
      px[u] = v;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = threadIdx.x + u * 256;            // M * 4 float4s
      const int m = q >> 2, c4 = q & 3;
      const int tw = m % TW, th = (m / TW) % TH, td = m / (TW * TH);
      const int d = d0 + td, h = h0 + th, wq = w0 + tw;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (d < cd.D && h < cd.H && wq < cd.W) v = ld4(dY + ((((long long)n * cd.D + d) * cd.H + h) * cd.W + wq) * 16 + c4 * 4);
      py[u] = v;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int u = 0; u < NXS; ++u) {
      const int q = threadIdx.x + u * 256;
      if (q < TL::HV) Xs[q] = px[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) st4(Ys + (threadIdx.x + u * 256) * 4, py[u]);
  };

  fetch(tile);
  stash();
  __syncthreads();
  for (;;) {
    const bool has_next = tile + 1 < t_end;
    if (has_next) fetch(tile + 1);
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {                  // this wave's 64 voxels, 4 per k-step
      const int v = wave * 64 + ks * 4 + lg;           // A[tap = li][k = lg], B[k = lg][co = li]
      const int vo = TL::voff(v);
      const float b = Ys[v * 16 + li];
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Xs[vo + aoff[0]], b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Xs[vo + aoff[1]], b, acc[1], 0, 0, 0);
    }
    if (!has_next) break;
    __syncthreads();
    stash();
    __syncthreads();
    ++tile;
  }
  // D[tap = tt*16 + lg*4 + r][co = li]: sum the four waves, write partial[grp][tap][ci = 0][co]
  __syncthreads();
  float* red = Ys;                                     // [4 waves][32 taps][16]
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 32 + tt * 16 + lg * 4 + r) * 16 + li] = acc[tt][r];
  __syncthreads();
  for (int o = threadIdx.x; o < T * 16; o += 256)
    partial[(long long)grp * T * 16 + o] = (red[o] + red[512 + o]) + (red[1024 + o] + red[1536 + o]);
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------

static int split_k(long long blocks, int nch) {   // deep levels: too few tiles to fill 256 CUs -> split the cin chunks
  const int f = options().splitk;
  if (f >= 1 && f <= 4 && f <= nch) return f;
  if (blocks > 256 || nch < 4) return 1;
  int sk = nch / 2;
  if (sk > 4) sk = 4;
  return sk;
}

template <int KD, int TD, int TH, int TW, int NT, int WT>
static int launch_fwd(const float* X, const float* Wp, const float* bias, float* Y, ConvDims cd, int accumulate, float* ws,
                      double* stat_partial, int G, bool dry, hipStream_t s) {
  using TL = Tile<KD, TD, TH, TW>;
  constexpr int S = TL::T / WT, NBUF = S > 1 ? 2 : 1;
  const size_t lds = (size_t)(TL::HV * XS + NBUF * WT * 4 * NT * 16 * 4) * sizeof(float) + 4 * NT * 16 * 2 * sizeof(double);
  cd.tiles_d = cdiv(cd.D, TD); cd.tiles_h = cdiv(cd.H, TH); cd.tiles_w = cdiv(cd.W, TW);
  auto kfn = k_conv3_mfma<KD, TD, TH, TW, NT, WT>;
  if (lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int gx = cd.N * cd.tiles_d * cd.tiles_h * cd.tiles_w, gy = cd.Cout16 / (NT * 16);
  const int sk = ws ? split_k((long long)gx * gy, cd.Cin16 / 16) : 1;
  StatsArg st{nullptr, 0, 1, cd.Cout, G > 0 ? G : 1};
  if (sk == 1 && G > 0 && gx % G == 0) { st.rows = gx / G; st.tiles_per_group = gx / G; st.partial = stat_partial; }
  if (dry) return sk == 1 && G > 0 && gx % G == 0 ? gx / G : 0;
  if (sk == 1) {
    hipLaunchKernelGGL(kfn, dim3(gx, gy, 1), dim3(256), lds, s, X, Wp, bias, Y, cd, accumulate, st);
  } else {
    const long long n = (long long)cd.N * cd.D * cd.H * cd.W * cd.Cout;
    StatsArg none{nullptr, 0, 1, cd.Cout, 1};
    hipLaunchKernelGGL(kfn, dim3(gx, gy, sk), dim3(256), lds, s, X, Wp, (const float*)nullptr, ws, cd, 0, none);
    hipLaunchKernelGGL(k_sum_slabs, dim3((int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, s, ws, sk, n, cd.Cout, bias, Y, accumulate);
  }
  return st.partial ? st.rows : 0;
}

template <int KD, int TD, int TH, int TW, int NT>
static int launch_res(const float* X, const float* Wp, const float* bias, float* Y, ConvDims cd, int accumulate,
                      double* stat_partial, int G, bool dry, hipStream_t s) {
  using TL = Tile<KD, TD, TH, TW>;
  const int nch = cd.Cin16 / 16;                        // resident chunks per workgroup
  const size_t lds = ((size_t)nch * TL::T * 4 * NT * 16 * 4 + (size_t)TL::HV * XS) * sizeof(float) + 4 * NT * 16 * 2 * sizeof(double);
  cd.tiles_d = cdiv(cd.D, TD); cd.tiles_h = cdiv(cd.H, TH); cd.tiles_w = cdiv(cd.W, TW);
  const int tiles = cd.N * cd.tiles_d * cd.tiles_h * cd.tiles_w;
  const int slabs = cd.Cout16 / (NT * 16);
  const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
  int P = (256 * (per_cu > 3 ? 3 : per_cu)) / slabs;    // persistent workgroups per slab
  if (per_cu > 2) P = 512 / slabs;
  const Options& o = options();
  if (o.res_pcu >= 1 && o.res_pcu <= 4 && o.res_pcu <= per_cu) P = 256 * o.res_pcu / slabs;
  if (o.conv3_p > 0 && o.conv3_p < P) P = o.conv3_p;    // tests: force multi-tile loops
  if (P < 1) P = 1;
  if (P > tiles) P = tiles;
  StatsArg st{nullptr, 0, 1, cd.Cout, G > 0 ? G : 1};
  const bool stats_ok = G > 0 && tiles % G == 0;
  if (dry) return stats_ok ? P : 0;
  if (stats_ok && stat_partial) { st.partial = stat_partial; st.rows = P; st.tiles_per_group = tiles / G; }   // (rows of unvisited groups are zeroed by the kernel itself)
  auto kfn = k_conv3_res<KD, TD, TH, TW, NT>;
  if (lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kfn, dim3(P, slabs, 1), dim3(256), lds, s, X, Wp, bias, Y, cd, tiles, accumulate, st, 1);
  return st.partial ? P : 0;
}

template <int KD, int TD, int TH, int TW, int NT>
static int launch_wgrad(const float* X, const float* dY, float* partial, ConvDims cd, int groups, hipStream_t s) {
  using TL = Tile<KD, TD, TH, TW>;
  constexpr int CT = NT * 16, YS = (CT % 32 == 0) ? CT + 16 : CT;
  const size_t lds = (size_t)(TL::HV * XSW + TL::M * YS) * sizeof(float);
  cd.tiles_d = cdiv(cd.D, TD); cd.tiles_h = cdiv(cd.H, TH); cd.tiles_w = cdiv(cd.W, TW);
  const int tiles = cd.N * cd.tiles_d * cd.tiles_h * cd.tiles_w;
  const int tpg = cdiv(tiles, groups);
  auto kfn = k_conv3_wgrad<KD, TD, TH, TW, NT>;
  if (lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const dim3 grid(cdiv(tiles, tpg), cd.Cin16 / 16, cd.Cout16 / CT);
  hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, X, dY, partial, cd, tiles, tpg);
  return cdiv(tiles, tpg);
}

// tile / blocking choice: big tiles while the grid still fills 256 CUs, otherwise 64-voxel tiles
// and narrower channel slabs (deep V-Net levels are tiny GEMMs).
static Cfg choose_cfg(int KD, int N, int D, int H, int W, int Cout16, bool for_wgrad = false) {
  Cfg c;
  c.KD = KD;
  const long long vox = (long long)N * D * H * W;
  if (KD == 3) {
    if (vox >= 256LL * 1024) { c.TD = 4; c.TH = 4; c.TW = 16; }
    else if (vox >= 64LL * 1024) { c.TD = 4; c.TH = 8; c.TW = 8; }
    else {
      // deep V-Net levels (14x14x10, 7x7x5, ...): partial tiles waste up to half of the MFMA work, so pick the 64-voxel
      // tile shape that covers the volume with the fewest tiles (wgrad needs TW % 4 == 0)
      static const int cand[4][3] = {{4, 4, 4}, {2, 8, 4}, {2, 16, 2}, {8, 8, 1}};
      long long best = -1;
      for (int k = 0; k < (for_wgrad ? 2 : 4); ++k) {
        const long long t = (long long)cdiv(D, cand[k][0]) * cdiv(H, cand[k][1]) * cdiv(W, cand[k][2]);
        if (best < 0 || t < best) { best = t; c.TD = cand[k][0]; c.TH = cand[k][1]; c.TW = cand[k][2]; }
      }
    }
  } else {
    c.TD = 1;
    if (vox >= 128LL * 1024) { c.TH = 16; c.TW = 16; } else { c.TH = 8; c.TW = 8; }
  }
  const int M = c.TD * c.TH * c.TW;
  const long long tiles = (long long)N * cdiv(D, c.TD) * cdiv(H, c.TH) * cdiv(W, c.TW);
  int nt = 4;
  while (nt > 1 && (Cout16 % (nt * 16) != 0)) nt >>= 1;
  const long long want = tiles <= 128 ? 512 : 256;                // tiny spatial extents: >= 2 blocks per CU
  while (nt > 1 && tiles * (Cout16 / (nt * 16)) < want) nt >>= 1;  // more blocks for small problems
  (void)M;
  c.NT = nt;
  c.WT = 0;
  if (!for_wgrad) {
    const int* f = options().conv3_cfg;   // measurements: forces the streaming kernel's tile / slab width
    if (f[0] > 0 && Cout16 % (f[3] * 16) == 0) { c.TD = f[0]; c.TH = f[1]; c.TW = f[2]; c.NT = f[3]; }
  }
  return c;
}

// conv3b.hip / conv3bw.hip: fp32 numerics on the bf16 matrix pipe (three-piece operands)
int b6_wgrad(const float* x, const float* dy, float* partial, float* dw, int accumulate, const ConvDims& cd, int KD, hipStream_t s);
size_t b6_wgrad_workspace_bytes(const ConvDims& cd, int KD);
int b6_fwd(const float* x, const float* wp, const float* bias, float* y, const ConvDims& cd, int KD, int accumulate, void* workspace,
           double* stat_partial, int G, bool dry, hipStream_t s, bool* handled, int* raw_sk = nullptr, const BwdStatsIn* bw = nullptr);

}  // namespace bcp

using namespace bcp;

static int fill_dims(ConvDims& cd, int N, int D, int H, int W, int Cin, int Cout) {
  cd.N = N; cd.D = D; cd.H = H; cd.W = W; cd.Cin = Cin; cd.Cout = Cout;
  cd.Cin16 = (Cin + 15) / 16 * 16;
  cd.Cout16 = (Cout + 15) / 16 * 16;
  cd.tiles_d = cd.tiles_h = cd.tiles_w = 0;
  cd.xcd = options().conv3_xcd;
  cd.xamax = nullptr;
  cd.yamax = nullptr;
  return 0;
}

extern "C" size_t bcp_conv3_packed_weight_floats(int Cin, int Cout, int KD) {
  // fp32 pack [T][K16 / 4][N16][4], the three-piece bf16 pack [K16 / 16][TP][3][N16][32] (2 B elements) and the two-piece fp16 pack
  // [K16 / 16][TP][2][N16][32] of conv3b.hip, and a 32-float header (max |w|: the fp16 planes' power-of-two scale)
  const int K16 = (Cin + 15) / 16 * 16, N16 = (Cout + 15) / 16 * 16, T = KD * 9;
  return (size_t)pack_off_hdr(T, K16, N16) + kPackHeaderFloats;
}

extern "C" int bcp_conv3_pack_weight(const float* w, float* wp_fwd, float* wp_dgrad, int Cin, int Cout, int KD, void* stream) {
  BCP_REQUIRE(w && (wp_fwd || wp_dgrad), "bcp_conv3_pack_weight: null pointer");
  BCP_REQUIRE((KD == 1 || KD == 3) && Cin > 0 && Cout > 0, "bcp_conv3_pack_weight: bad arguments");
  const int T = KD * 9, Ci16 = (Cin + 15) / 16 * 16, Co16 = (Cout + 15) / 16 * 16;
  const long long total = (long long)T * Ci16 * Co16;
  const int grid = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
  const long long count = (long long)Cout * Cin * T;
  if (wp_fwd) {
    hipLaunchKernelGGL(k_wamax, dim3(16), dim3(1024), 0, (hipStream_t)stream, w, count, wp_fwd + pack_off_hdr(T, Ci16, Co16));
    hipLaunchKernelGGL(k_pack_conv3, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wp_fwd, Cout, Cin, T, Ci16, Co16, 0);
  }
  if (wp_dgrad) {
    hipLaunchKernelGGL(k_wamax, dim3(16), dim3(1024), 0, (hipStream_t)stream, w, count, wp_dgrad + pack_off_hdr(T, Co16, Ci16));
    hipLaunchKernelGGL(k_pack_conv3, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wp_dgrad, Cout, Cin, T, Co16, Ci16, 1);
  }
  BCP_CHECK_LAUNCH("bcp_conv3_pack_weight");
  return BCP_OK;
}

// descs: device array of n {const float* w; float* wp; int Cout, Cin, T, K16, N16, dgrad} (40 B each, see bcp_hip.h)
extern "C" int bcp_conv3_pack_many(const void* descs_dev, int n, void* stream) {
  BCP_REQUIRE(descs_dev && n > 0 && n <= kMaxPackDescs, "bcp_conv3_pack_many: need 1..%d descriptors", kMaxPackDescs);
  static_assert(sizeof(PackDesc) == 40, "descriptor layout is part of the ABI");
  hipLaunchKernelGGL(k_wamax_many, dim3(16, n), dim3(1024), 0, (hipStream_t)stream, (const PackDesc*)descs_dev);
  // (2048 workgroups: the V-Net's 84 layers are 1368 work units of one 16 x 16 x taps block each -- with 1024 workgroups a third of them
  //  did two units and the launch lasted as long as those)
  hipLaunchKernelGGL(k_pack_conv3_many, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)descs_dev, n, options().pack_sections);
  BCP_CHECK_LAUNCH("bcp_conv3_pack_many");
  return BCP_OK;
}

#define BCP_FWD_CASE(KD_, TD_, TH_, TW_, NT_, WT_)                                                             \
  if (c.KD == KD_ && c.TD == TD_ && c.TH == TH_ && c.TW == TW_ && c.NT == NT_) {                               \
    rows = launch_fwd<KD_, TD_, TH_, TW_, NT_, WT_>(x, wp, bias, y, cd, accumulate, (float*)workspace, stat_partial, G, dry, (hipStream_t)stream); \
    done = true;                                                                                               \
  }
#define BCP_RES_CASE(KD_, TD_, TH_, TW_, NT_)                                                                  \
  if (r.KD == KD_ && r.TD == TD_ && r.TH == TH_ && r.TW == TW_ && r.NT == NT_) {                               \
    rows = launch_res<KD_, TD_, TH_, TW_, NT_>(x, wp, bias, y, cd, accumulate, stat_partial, G, dry, (hipStream_t)stream); \
    done = true;                                                                                               \
  }

// resident-weight variant: tile by problem size, widest channel slab whose weights + one halo fit in LDS
static bool choose_res(Cfg& r, int KD, int N, int D, int H, int W, int Cin16, int Cout16) {
  r.KD = KD;
  const long long vox = (long long)N * D * H * W;
  if (KD == 3) {
    r.TD = 4; r.TH = 4;
    r.TW = vox >= 256LL * 1024 ? 16 : (vox >= 64LL * 1024 ? 8 : 4);
  } else {
    r.TD = 1;
    // 16x16 tiles from 10 K pixels per launch on (the 64^2 and 32^2 U-Net levels at a grouped batch of 12): ALONE the 8x8-tile
    // kernels with their wider slabs are faster (48 vs 67 us at 64 channels), inside the ACDC step the 16x16 ones win
    // (5.35 vs 5.43 ms, interleaved A/B; thresholds 128 K / 40 K / 10 K / 1 K pixels: 5.43 / 5.39 / 5.35 / 5.36 ms)
    const long long thr = options().res_tile2d_vox;
    if (vox >= thr && H >= 16 && W >= 16) { r.TH = 16; r.TW = 16; } else { r.TH = 8; r.TW = 8; }
  }
  const int PD = KD == 3 ? 1 : 0;
  const long long hv = (long long)(r.TD + 2 * PD) * (r.TH + 2) * (r.TW + 2);
  const long long tiles = (long long)N * cdiv(D, r.TD) * cdiv(H, r.TH) * cdiv(W, r.TW);
  int nt_max = 4;
  { const int v = options().res_nt; if (v == 1 || v == 2 || v == 4) nt_max = v; }
  for (int nt = nt_max; nt >= 1; nt >>= 1) {
    if (Cout16 % (nt * 16)) continue;
    const long long lds = ((long long)KD * 9 * Cin16 * nt * 16 + hv * XS) * 4 + 4 * nt * 16 * 2 * 8;
    if (lds > 158 * 1024) continue;
    if (nt > 1 && tiles * (Cout16 / (nt * 16)) < 512) continue;   // keep >= 2 work items per CU
    r.NT = nt;
    const int mt = (r.TD * r.TH * r.TW) / 64;
    if (mt * nt < 2) return false;                                // one MFMA tile per wave per tap: the streaming kernel's 2-slab blocks are faster
    return tiles * (Cout16 / (nt * 16)) >= 192;                   // too few items: the streaming kernel's finer grid wins
  }
  return false;
}

extern "C" size_t bcp_conv3_fwd_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int KD) {
  // split-K partial slabs (deep levels only): at most 8 copies of the output
  const long long n = (long long)N * D * H * W * Cout;
  return n <= options().conv3_sk_elems ? (size_t)(8 * n * sizeof(float)) : 0;
}

// shared by the launch and by the statistics-rows query: returns the number of partial rows per group the chosen kernel
// writes (0: this shape does not support fused statistics, e.g. split-K), or a negative error
namespace bcp { int b6_last_planes(); }
// which SECTION of the packed weight the last real (non-dry) forward / dgrad launch of this thread read: 1 = the fp32 pack (conv3.hip kernels),
// 2 = the three bf16 planes, 4 = the two fp16 planes (conv3b.hip).  The host learns from it which sections a layer's packs must hold
// (bcp_conv3_last_section; networks/_hipnet.py)
static thread_local int g_last_section = 0;

static int conv3_fwd_impl(const float* x, const float* wp, const float* bias, float* y, int N, int D, int H, int W, int Cin, int Cout,
                          int KD, int accumulate, void* workspace, double* stat_partial, int G, bool dry, void* stream,
                          const float* x_amax = nullptr) {
  ConvDims cd;
  fill_dims(cd, N, D, H, W, Cin, Cout);
  cd.xamax = x_amax;
  bool done = false;
  int rows = 0;
  Cfg r;
  rows = b6_fwd(x, wp, bias, y, cd, KD, accumulate, workspace, stat_partial, G, dry, (hipStream_t)stream, &done);
  if (!dry) g_last_section = done ? (bcp::b6_last_planes() == 2 ? 4 : 2) : 1;
  if (done) return rows;
  if (choose_res(r, KD, N, D, H, W, cd.Cin16, cd.Cout16)) {
    BCP_RES_CASE(3, 4, 4, 16, 1) BCP_RES_CASE(3, 4, 4, 16, 2)
    BCP_RES_CASE(3, 4, 4, 8, 1) BCP_RES_CASE(3, 4, 4, 8, 2) BCP_RES_CASE(3, 4, 4, 8, 4)
    BCP_RES_CASE(3, 4, 4, 4, 1) BCP_RES_CASE(3, 4, 4, 4, 2) BCP_RES_CASE(3, 4, 4, 4, 4)
    BCP_RES_CASE(1, 1, 16, 16, 1) BCP_RES_CASE(1, 1, 16, 16, 2) BCP_RES_CASE(1, 1, 16, 16, 4)
    BCP_RES_CASE(1, 1, 8, 8, 1) BCP_RES_CASE(1, 1, 8, 8, 2) BCP_RES_CASE(1, 1, 8, 8, 4)
  }
  if (!done) {
    const Cfg c = choose_cfg(KD, N, D, H, W, cd.Cout16);
    BCP_FWD_CASE(3, 4, 4, 16, 1, 27) BCP_FWD_CASE(3, 4, 4, 16, 2, 9) BCP_FWD_CASE(3, 4, 4, 16, 4, 3)
    BCP_FWD_CASE(3, 4, 8, 8, 1, 27) BCP_FWD_CASE(3, 4, 8, 8, 2, 9) BCP_FWD_CASE(3, 4, 8, 8, 4, 3)
    BCP_FWD_CASE(3, 4, 4, 4, 1, 27) BCP_FWD_CASE(3, 4, 4, 4, 2, 9) BCP_FWD_CASE(3, 4, 4, 4, 4, 3)
    BCP_FWD_CASE(3, 2, 8, 4, 1, 27) BCP_FWD_CASE(3, 2, 8, 4, 2, 9) BCP_FWD_CASE(3, 2, 8, 4, 4, 3)
    BCP_FWD_CASE(3, 2, 16, 2, 1, 27) BCP_FWD_CASE(3, 2, 16, 2, 2, 9) BCP_FWD_CASE(3, 2, 16, 2, 4, 3)
    BCP_FWD_CASE(3, 8, 8, 1, 1, 27) BCP_FWD_CASE(3, 8, 8, 1, 2, 9) BCP_FWD_CASE(3, 8, 8, 1, 4, 3)
    BCP_FWD_CASE(1, 1, 16, 16, 1, 9) BCP_FWD_CASE(1, 1, 16, 16, 2, 9) BCP_FWD_CASE(1, 1, 16, 16, 4, 3)
    BCP_FWD_CASE(1, 1, 8, 8, 1, 9) BCP_FWD_CASE(1, 1, 8, 8, 2, 9) BCP_FWD_CASE(1, 1, 8, 8, 4, 3)
    BCP_REQUIRE(done, "bcp_conv3_fwd: no kernel instance for KD=%d tile=%dx%dx%d NT=%d", c.KD, c.TD, c.TH, c.TW, c.NT);
  }
  return rows;
}

extern "C" int bcp_conv3_fwd(const float* x, const float* wp, const float* bias, float* y, int N, int D, int H, int W, int Cin,
                             int Cout, int KD, int accumulate, void* workspace, const float* x_amax_or_null, void* stream) {
  BCP_REQUIRE(x && wp && y, "bcp_conv3_fwd: null pointer");
  BCP_REQUIRE((KD == 1 || KD == 3) && N > 0 && D > 0 && H > 0 && W > 0, "bcp_conv3_fwd: bad extents");
  BCP_REQUIRE(KD == 3 || D == 1, "bcp_conv3_fwd: KD=1 needs D=1");
  BCP_REQUIRE(Cin % 4 == 0 && Cin >= 4, "bcp_conv3_fwd: Cin=%d must be a multiple of 4 (Cin=1 has its own entry point)", Cin);
  BCP_REQUIRE(aligned16(x) && aligned16(wp), "bcp_conv3_fwd: x / wp must be 16-B aligned");
  const int rc = conv3_fwd_impl(x, wp, bias, y, N, D, H, W, Cin, Cout, KD, accumulate, workspace, nullptr, 0, false, stream, x_amax_or_null);
  if (rc < 0) return rc;
  BCP_CHECK_LAUNCH("bcp_conv3_fwd");
  return BCP_OK;
}

// Fused variant: also emits the per-channel (sum, sum of squares) partials of y for `groups` normalisation groups
// (groups consecutive sample ranges).  rows = bcp_conv3_stat_rows(...) partial rows per group are written into
// stat_partial[groups][rows][Cout][2] doubles; rows == 0 means "not available for this shape": run bcp_conv3_fwd and
// let bcp_norm_fwd compute its own statistics.
extern "C" int bcp_conv3_stat_rows(int N, int D, int H, int W, int Cin, int Cout, int KD, int groups, int has_workspace) {
  if (Cin % 4 || Cin < 4 || groups < 1) return 0;
  static float dummy;
  const int rc = conv3_fwd_impl(nullptr, nullptr, nullptr, nullptr, N, D, H, W, Cin, Cout, KD, 0, has_workspace ? &dummy : nullptr, nullptr,
                                groups, true, nullptr);
  return rc < 0 ? 0 : rc;
}

// which matrix pipe serves bcp_conv3_fwd / _fwd_stats for this shape under the current options: 0 = v_mfma_f32_16x16x4_f32 (fp32
// operands), 1 = v_mfma_f32_16x16x32_bf16 with three-piece operands (conv3b.hip; fp32-equivalent results, six MFMAs per K = 32 block).
// For the measurement record (bench.py prices the two against different peaks); no launch.
extern "C" size_t bcp_conv3_fwd_path(int N, int D, int H, int W, int Cin, int Cout, int KD) {
  if (Cin % 4 || Cin < 4) return 0;
  ConvDims cd;
  fill_dims(cd, N, D, H, W, Cin, Cout);
  bool handled = false;
  static float dummy;
  b6_fwd(nullptr, nullptr, nullptr, nullptr, cd, KD, 0, &dummy, nullptr, 0, true, nullptr, &handled);
  return handled ? 1 : 0;
}

namespace bcp { int b6_last_planes(); }
// operand planes of the kernel that serves this shape when the launch carries the tensors' maxima (x_amax ...): 3 = three bf16 planes
// (six MFMAs per K block), 2 = two fp16 planes (three), 0 = not on the 16-bit matrix pipe at all.  Measurement record (bench.py prices
// the op against bf16 peak / 6 or / 3 accordingly); no launch.  wgrad != 0: the weight gradient of the same layer.
extern "C" int bcp_conv3_planes(int N, int D, int H, int W, int Cin, int Cout, int KD, int wgrad) {
  if (Cin % 4 || Cin < 4) return 0;
  ConvDims cd;
  fill_dims(cd, N, D, H, W, Cin, Cout);
  static float dummy;
  if (wgrad) return b6_wgrad_workspace_bytes(cd, KD) > 0 ? (options().conv3_f16 != 0 ? 2 : 3) : 0;
  cd.xamax = &dummy;
  bool handled = false;
  b6_fwd(nullptr, nullptr, nullptr, nullptr, cd, KD, 0, &dummy, nullptr, 0, true, nullptr, &handled);
  return handled ? bcp::b6_last_planes() : 0;
}

extern "C" int bcp_conv3_last_section(void) { return g_last_section; }

extern "C" size_t bcp_conv3_wgrad_path(int N, int D, int H, int W, int Cin, int Cout, int KD) {
  ConvDims cd;
  fill_dims(cd, N, D, H, W, Cin, Cout);
  return b6_wgrad_workspace_bytes(cd, KD) > 0 ? 1 : 0;       // conv3bw.hip takes the shape
}

extern "C" int bcp_conv3_fwd_stats(const float* x, const float* wp, const float* bias, float* y, int N, int D, int H, int W, int Cin,
                                   int Cout, int KD, void* workspace, double* stat_partial, int groups, const float* x_amax_or_null,
                                   void* stream) {
  BCP_REQUIRE(x && wp && y && stat_partial, "bcp_conv3_fwd_stats: null pointer");
  BCP_REQUIRE((KD == 1 || KD == 3) && N > 0 && D > 0 && H > 0 && W > 0 && groups >= 1, "bcp_conv3_fwd_stats: bad extents");
  BCP_REQUIRE(Cin % 4 == 0 && Cin >= 4, "bcp_conv3_fwd_stats: Cin=%d must be a multiple of 4", Cin);
  BCP_REQUIRE(aligned16(x) && aligned16(wp), "bcp_conv3_fwd_stats: x / wp must be 16-B aligned");
  const int rc = conv3_fwd_impl(x, wp, bias, y, N, D, H, W, Cin, Cout, KD, 0, workspace, stat_partial, groups, false, stream, x_amax_or_null);
  if (rc < 0) return rc;
  BCP_REQUIRE(rc > 0, "bcp_conv3_fwd_stats: fused statistics unavailable for this shape (check bcp_conv3_stat_rows first)");
  BCP_CHECK_LAUNCH("bcp_conv3_fwd_stats");
  return BCP_OK;
}

// dgrad with the consumer's norm-backward statistics in its epilogue (bf16-pipe kernels only): the conv output is da of the norm
// layer whose pre-norm tensor is y_prev / statistics stats_prev; stat_partial receives (sum dz, sum dz * xhat) partials
// [groups][rows][Cout][2], rows = bcp_conv3_bwdstat_rows(...) (0: not available for this shape), to be handed to bcp_norm_bwd as
// partial_in -- its statistics pass over (y, da) is then skipped.  No dropout epilogues (chan_scale / elem_mask) on that layer.
extern "C" int bcp_conv3_bwdstat_rows(int N, int D, int H, int W, int Cin, int Cout, int KD, int groups) {
  if (Cin % 4 || Cin < 4 || groups < 1 || options().fuse_bwd_stats == 0) return 0;
  ConvDims cd;
  fill_dims(cd, N, D, H, W, Cin, Cout);
  bool handled = false;
  static float dummy;
  const BwdStatsIn bw{&dummy, &dummy, 0};
  const int rows = b6_fwd(nullptr, nullptr, nullptr, nullptr, cd, KD, 0, &dummy, nullptr, groups, true, nullptr, &handled, nullptr, &bw);
  return handled && rows > 0 ? rows : 0;
}

extern "C" int bcp_conv3_dgrad_bwdstats(const float* dy, const float* wp_dgrad, float* da, int N, int D, int H, int W, int Cin, int Cout,
                                        int KD, const float* y_prev, const float* stats_prev, int act, void* workspace,
                                        double* stat_partial, int groups, const float* dy_amax_or_null, void* stream) {
  BCP_REQUIRE(dy && wp_dgrad && da && y_prev && stats_prev && stat_partial, "bcp_conv3_dgrad_bwdstats: null pointer");
  BCP_REQUIRE((KD == 1 || KD == 3) && N > 0 && D > 0 && H > 0 && W > 0 && groups >= 1, "bcp_conv3_dgrad_bwdstats: bad extents");
  BCP_REQUIRE(Cin % 4 == 0 && Cin >= 4, "bcp_conv3_dgrad_bwdstats: Cin=%d must be a multiple of 4", Cin);
  BCP_REQUIRE(aligned16(dy) && aligned16(wp_dgrad) && aligned16(y_prev), "bcp_conv3_dgrad_bwdstats: dy / wp / y_prev must be 16-B aligned");
  ConvDims cd;
  fill_dims(cd, N, D, H, W, Cin, Cout);
  cd.xamax = dy_amax_or_null;
  bool handled = false;
  const BwdStatsIn bw{y_prev, stats_prev, act};
  const int rows = b6_fwd(dy, wp_dgrad, nullptr, da, cd, KD, 0, workspace, stat_partial, groups, false, (hipStream_t)stream, &handled, nullptr, &bw);
  BCP_REQUIRE(handled && rows > 0, "bcp_conv3_dgrad_bwdstats: fused statistics unavailable for this shape (check bcp_conv3_bwdstat_rows first)");
  g_last_section = bcp::b6_last_planes() == 2 ? 4 : 2;
  BCP_CHECK_LAUNCH("bcp_conv3_dgrad_bwdstats");
  return BCP_OK;
}

// Raw variant for the deep levels: the kernel's split-K partial slabs are the RESULT -- float[nslabs][N*D*H*W*Cout], no bias, no
// slab-sum launch; bcp_norm_fwd_slabs / bcp_norm_bwd_slabs sum them (and add the bias) on their way in.  bcp_conv3_fwd_nslabs tells
// how many slabs the launch will write for this shape under the current options (1..8), or 0: shape not served in raw mode (use
// bcp_conv3_fwd).  Forward and dgrad alike (dgrad = the flipped pack).
extern "C" int bcp_conv3_fwd_nslabs(int N, int D, int H, int W, int Cin, int Cout, int KD) {
  if (Cin % 4 || Cin < 4 || (KD != 1 && KD != 3) || N < 1 || D < 1 || H < 1 || W < 1) return 0;
  if ((long long)N * D * H * W * Cout > options().conv3_sk_elems) return 0;      // the slab buffer is sized like bcp_conv3_fwd_workspace_bytes: deep levels only
  ConvDims cd;
  fill_dims(cd, N, D, H, W, Cin, Cout);
  bool handled = false;
  static float dummy;
  int sk = 0;
  b6_fwd(nullptr, nullptr, nullptr, &dummy, cd, KD, 0, &dummy, nullptr, 0, true, nullptr, &handled, &sk);
  return handled ? sk : 0;
}

extern "C" int bcp_conv3_fwd_raw(const float* x, const float* wp, float* slabs, int nslab, int N, int D, int H, int W, int Cin, int Cout,
                                 int KD, const float* x_amax_or_null, void* stream) {
  BCP_REQUIRE(x && wp && slabs, "bcp_conv3_fwd_raw: null pointer");
  BCP_REQUIRE((KD == 1 || KD == 3) && N > 0 && D > 0 && H > 0 && W > 0, "bcp_conv3_fwd_raw: bad extents");
  BCP_REQUIRE(KD == 3 || D == 1, "bcp_conv3_fwd_raw: KD=1 needs D=1");
  BCP_REQUIRE(Cin % 4 == 0 && Cin >= 4, "bcp_conv3_fwd_raw: Cin=%d must be a multiple of 4", Cin);
  BCP_REQUIRE(aligned16(x) && aligned16(wp) && aligned16(slabs), "bcp_conv3_fwd_raw: x / wp / slabs must be 16-B aligned");
  BCP_REQUIRE((long long)N * D * H * W * Cout <= options().conv3_sk_elems, "bcp_conv3_fwd_raw: output too large for raw slabs (check bcp_conv3_fwd_nslabs)");
  ConvDims cd;
  fill_dims(cd, N, D, H, W, Cin, Cout);
  bool handled = false;
  int sk = 0;
  // the caller sized `slabs` for nslab slabs: the count the launch will write under the CURRENT options must be that one (a count
  // cached under other options would be a device buffer overflow) -- dry run first, nothing is launched on a mismatch
  b6_fwd(nullptr, nullptr, nullptr, slabs, cd, KD, 0, slabs, nullptr, 0, true, nullptr, &handled, &sk);
  BCP_REQUIRE(handled && sk > 0, "bcp_conv3_fwd_raw: shape not served in raw mode (check bcp_conv3_fwd_nslabs first)");
  BCP_REQUIRE(sk == nslab, "bcp_conv3_fwd_raw: the launch writes %d slabs under the current options, the caller allocated %d (stale bcp_conv3_fwd_nslabs answer?)", sk, nslab);
  handled = false;
  cd.xamax = x_amax_or_null;
  b6_fwd(x, wp, nullptr, slabs, cd, KD, 0, slabs, nullptr, 0, false, (hipStream_t)stream, &handled, &sk);
  BCP_REQUIRE(handled && sk == nslab, "bcp_conv3_fwd_raw: shape not served in raw mode (check bcp_conv3_fwd_nslabs first)");
  g_last_section = bcp::b6_last_planes() == 2 ? 4 : 2;
  BCP_CHECK_LAUNCH("bcp_conv3_fwd_raw");
  return BCP_OK;
}

// groups so that the wgrad grid has a few hundred blocks
static int wgrad_groups(const Cfg& c, int N, int D, int H, int W, int Cin16, int Cout16) {
  const int tiles = N * cdiv(D, c.TD) * cdiv(H, c.TH) * cdiv(W, c.TW);
  const int chan_blocks = (Cin16 / 16) * (Cout16 / (c.NT * 16));
  int g = cdiv(512, chan_blocks);
  if (g > tiles) g = tiles;
  if (g < 1) g = 1;
  const int tpg = cdiv(tiles, g);
  return cdiv(tiles, tpg);
}

static Cfg choose_wgrad_cfg(int KD, int N, int D, int H, int W, int Cout16) {
  Cfg c = choose_cfg(KD, N, D, H, W, Cout16, true);
  // accumulators: TPW * NT * 4 regs (TPW = 7 for 27 taps) -> cap NT at 4; small problems keep NT from choose_cfg.
  // (Measured: forcing the widest slab at the deep levels -- more MFMAs per staged tile but fewer, longer blocks -- is
  // slower: C=128 wgrad 82 / 85 / 106 us and C=256 57 / 58 / 65 us for NT = 1 / 2 / 4.)
  int nt = c.NT > 2 ? 2 : c.NT;     // C=64: 75 us with 2-slab blocks vs 79 us with 4 (two workgroups per CU instead of one)
  // Tile sweep (option wgrad_tile): the mid level (2 x 56x56x40, C=32) runs 140 us ALONE with 4x4x8 tiles against 158 us with
  // 4x8x8 -- but inside the step, next to the dgrad / norm-backward kernels of the main stream, the 256-voxel tile wins
  // (8.95 vs 9.02 ms per step, interleaved A/B of the two libraries): 4x8x8 stays.  The 16-channel level keeps 4x4x16
  // (259 us vs 286 / 283 / 314 us for 4x4x8 / 4x8x8 / 4x4x4).
  // 2-D (ACDC) levels below 128 K pixels: choose_cfg hands out 8x8 tiles for the forward grid's sake, but a weight-gradient
  // block gets its parallelism from the (cin chunk, cout slab) grid, and with 9 taps an 8x8 tile is only 288 MFMAs per staged
  // halo + dY tile: 16x16 tiles run 52 / 61 / 54 us instead of 58 / 66 / 66 us at 64 / 128 / 256 channels, and the ACDC step
  // 5.45 instead of 5.68 ms.
  if (KD == 1 && H >= 16 && W >= 16) { c.TH = 16; c.TW = 16; }
  const Options& o = options();
  if (o.wgrad_tile[0] > 0) {   // measurements: TD,TH,TW[,min_voxels,max_voxels] (range: one level only)
    const long long vox = (long long)N * D * H * W;
    if (vox >= o.wgrad_tile[3] && vox <= o.wgrad_tile[4]) { c.TD = (int)o.wgrad_tile[0]; c.TH = (int)o.wgrad_tile[1]; c.TW = (int)o.wgrad_tile[2]; }
  }
  { const int v = o.wgrad_nt; if ((v == 1 || v == 2 || v == 4) && Cout16 % (v * 16) == 0) nt = v; }
  c.NT = nt;
  return c;
}

extern "C" size_t bcp_conv3_wgrad_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int KD) {
  const int Ci16 = (Cin + 15) / 16 * 16, Co16 = (Cout + 15) / 16 * 16;
  if (Cin == 1) {
    const int tiles = N * cdiv(D, KD == 3 ? 4 : 1) * cdiv(H, KD == 3 ? 4 : 16) * cdiv(W, 16);
    int g = tiles < 512 ? tiles : 512;
    return (size_t)g * KD * 9 * 16 * sizeof(float);
  }
  ConvDims cd;
  fill_dims(cd, N, D, H, W, Cin, Cout);
  const Cfg c = choose_wgrad_cfg(KD, N, D, H, W, Co16);
  const int g = wgrad_groups(c, N, D, H, W, Ci16, Co16);
  const size_t a = (size_t)g * KD * 9 * Ci16 * Co16 * sizeof(float), c6 = b6_wgrad_workspace_bytes(cd, KD);
  return a > c6 ? a : c6;
}

#define BCP_WG_CASE(KD_, TD_, TH_, TW_, NT_)                                                     \
  if (c.KD == KD_ && c.TD == TD_ && c.TH == TH_ && c.TW == TW_ && c.NT == NT_) {                 \
    G = launch_wgrad<KD_, TD_, TH_, TW_, NT_>(x, dy, ws, cd, groups, (hipStream_t)stream);       \
    done = true;                                                                                 \
  }

extern "C" int bcp_conv3_wgrad(const float* x, const float* dy, float* dw, int N, int D, int H, int W, int Cin, int Cout, int KD,
                               int accumulate, void* workspace, const float* x_amax_or_null, const float* dy_amax_or_null, void* stream) {
  if (bcp::options().whatif & 8) return BCP_OK;      // MEASUREMENT ONLY (common.h Options::whatif)

  BCP_REQUIRE(x && dy && dw && workspace, "bcp_conv3_wgrad: null pointer");
  BCP_REQUIRE((KD == 1 || KD == 3) && N > 0 && D > 0 && H > 0 && W > 0, "bcp_conv3_wgrad: bad extents");
  BCP_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "bcp_conv3_wgrad: Cin/Cout must be multiples of 4");
  ConvDims cd;
  fill_dims(cd, N, D, H, W, Cin, Cout);
  bool done = false;
  const Cfg c = choose_wgrad_cfg(KD, N, D, H, W, cd.Cout16);
  const int groups = wgrad_groups(c, N, D, H, W, cd.Cin16, cd.Cout16);
  float* ws = reinterpret_cast<float*>(workspace);
  if (x_amax_or_null && dy_amax_or_null) { cd.xamax = x_amax_or_null; cd.yamax = dy_amax_or_null; }      // (conv3bw.hip: two fp16 planes per operand)
  int G = b6_wgrad(x, dy, ws, dw, accumulate, cd, KD, (hipStream_t)stream);      // partial slabs from the bf16-pipe kernel (conv3bw.hip), when it takes the shape
  if (G < 0) {                                   // deep levels (round 6): the kernel wrote the gradient itself, no slabs, no reduce
    BCP_CHECK_LAUNCH("bcp_conv3_wgrad");
    return BCP_OK;
  }
  if (G > 0) done = true;
  if (!done) {
  BCP_WG_CASE(3, 4, 4, 16, 1) BCP_WG_CASE(3, 4, 4, 16, 2) BCP_WG_CASE(3, 4, 4, 16, 4)
  BCP_WG_CASE(3, 4, 8, 8, 1) BCP_WG_CASE(3, 4, 8, 8, 2) BCP_WG_CASE(3, 4, 8, 8, 4)
  BCP_WG_CASE(3, 4, 4, 8, 1) BCP_WG_CASE(3, 4, 4, 8, 2) BCP_WG_CASE(3, 4, 4, 8, 4)
  BCP_WG_CASE(3, 4, 4, 4, 1) BCP_WG_CASE(3, 4, 4, 4, 2) BCP_WG_CASE(3, 4, 4, 4, 4)
  BCP_WG_CASE(3, 2, 8, 4, 1) BCP_WG_CASE(3, 2, 8, 4, 2) BCP_WG_CASE(3, 2, 8, 4, 4)
  BCP_WG_CASE(1, 1, 16, 16, 1) BCP_WG_CASE(1, 1, 16, 16, 2) BCP_WG_CASE(1, 1, 16, 16, 4)
  BCP_WG_CASE(1, 1, 8, 8, 1) BCP_WG_CASE(1, 1, 8, 8, 2) BCP_WG_CASE(1, 1, 8, 8, 4)
  }
  BCP_REQUIRE(done, "bcp_conv3_wgrad: no kernel instance for KD=%d tile=%dx%dx%d NT=%d", c.KD, c.TD, c.TH, c.TW, c.NT);
  const int T = KD * 9;
  if (G >= 32 || options().wgrad_reduce_flat == 2)      // (2: tests force the flat sum for every group count)
    launch_reduce_deep(ws, dw, G, T, Cin, Cout, cd.Cin16, cd.Cout16, accumulate, (hipStream_t)stream);
  else
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(Cin, cdiv(Cout, 64)), dim3(256), 0, (hipStream_t)stream, ws, dw, G, T, Cin, Cout, cd.Cin16,
                       cd.Cout16, accumulate);
  BCP_CHECK_LAUNCH("bcp_conv3_wgrad");
  return BCP_OK;
}

// ---- Cin = 1 -> Cout = 16 first layer
// tiles per workgroup of the first-layer kernel when it also takes the statistics: the largest of 8 / 4 / 2 / 1 that divides the tiles of
// a normalisation group (a workgroup's statistics row belongs to ONE group) -- 980 rows per group at the LA size instead of 3920
static int c1_tiles_per_block(int tiles_per_group) {
  for (int t = 8; t > 1; t >>= 1)
    if (tiles_per_group % t == 0 && tiles_per_group / t >= 256) return t;
  return 1;
}

template <int EPI>
static void c1_launch(int KD, dim3 grid, hipStream_t s, const float* x, const float* w, const float* bias, float* y, const ConvDims& cd, int tiles,
                      int tpb, const StatsArg& st, const C1Norm& nm) {
  if (KD == 3) hipLaunchKernelGGL((k_conv3_c1<3, 4, 4, 16, EPI>), grid, dim3(256), 0, s, x, w, bias, y, cd, tiles, tpb, st, nm);
  else hipLaunchKernelGGL((k_conv3_c1<1, 1, 16, 16, EPI>), grid, dim3(256), 0, s, x, w, bias, y, cd, tiles, tpb, st, nm);
}

// epi: the kernel's EPI mode; returns the statistics rows per group (0: groups do not divide the batch)
static int c1_fwd_impl(const float* x, const float* w, const float* bias, float* y, int N, int D, int H, int W, int KD, double* stat_partial,
                       int groups, bool dry, hipStream_t s, int epi = 0, const C1Norm* nmp = nullptr) {
  ConvDims cd;
  fill_dims(cd, N, D, H, W, 1, 16);
  if (KD == 3) { cd.tiles_d = cdiv(D, 4); cd.tiles_h = cdiv(H, 4); cd.tiles_w = cdiv(W, 16); }
  else { cd.tiles_d = 1; cd.tiles_h = cdiv(H, 16); cd.tiles_w = cdiv(W, 16); }
  const int tiles = N * cd.tiles_d * cd.tiles_h * cd.tiles_w;
  StatsArg st{nullptr, 0, 1, 16, groups > 0 ? groups : 1};
  int tpb = 1;
  if (groups > 0) {
    if (N % groups) return 0;                             // tiles are sample-major: a group = whole samples
    const int tpg = tiles / groups;
    tpb = c1_tiles_per_block(tpg);
    st.tiles_per_group = tpg; st.rows = tpg / tpb; st.partial = stat_partial;
    if (dry) return st.rows;
  }
  const dim3 grid(cdiv(tiles, tpb));
  const C1Norm none{nullptr, nullptr, nullptr, nullptr, 1.f, 0, 1, nullptr};
  const C1Norm& nm = nmp ? *nmp : none;
  switch (epi) {
    case 1: c1_launch<1>(KD, grid, s, x, w, bias, y, cd, tiles, tpb, st, nm); break;
    case 2: c1_launch<2>(KD, grid, s, x, w, bias, y, cd, tiles, tpb, st, nm); break;
    case 3: c1_launch<3>(KD, grid, s, x, w, bias, y, cd, tiles, tpb, st, nm); break;
    case 4: c1_launch<4>(KD, grid, s, x, w, bias, y, cd, tiles, tpb, st, nm); break;
    case 5: c1_launch<5>(KD, grid, s, x, w, bias, y, cd, tiles, tpb, st, nm); break;
    default: c1_launch<0>(KD, grid, s, x, w, bias, y, cd, tiles, tpb, st, nm); break;
  }
  return st.rows;
}

extern "C" int bcp_conv3_c1_fwd(const float* x, const float* w, const float* bias, float* y, int N, int D, int H, int W, int KD,
                                void* stream) {
  BCP_REQUIRE(x && w && y, "bcp_conv3_c1_fwd: null pointer");
  BCP_REQUIRE((KD == 1 && D == 1) || KD == 3, "bcp_conv3_c1_fwd: bad KD/D");
  BCP_REQUIRE(aligned16(y) && (!bias || aligned16(bias)), "bcp_conv3_c1_fwd: y / bias must be 16-B aligned");
  c1_fwd_impl(x, w, bias, y, N, D, H, W, KD, nullptr, 0, false, (hipStream_t)stream);
  BCP_CHECK_LAUNCH("bcp_conv3_c1_fwd");
  return BCP_OK;
}

// fused variant (as bcp_conv3_fwd_stats): the epilogue also leaves the (sum, sum^2) partials of y for `groups` consecutive sample
// ranges -- stat_partial = double[groups][rows][16][2], rows = bcp_conv3_c1_stat_rows(...) -- so the norm behind the first layer
// needs no statistics pass over its 64-byte-per-voxel output
extern "C" int bcp_conv3_c1_stat_rows(int N, int D, int H, int W, int KD, int groups) {
  if (groups < 1 || N < 1 || D < 1 || H < 1 || W < 1 || !((KD == 1 && D == 1) || KD == 3)) return 0;
  return c1_fwd_impl(nullptr, nullptr, nullptr, nullptr, N, D, H, W, KD, nullptr, groups, true, nullptr);
}

extern "C" int bcp_conv3_c1_fwd_stats(const float* x, const float* w, const float* bias, float* y, int N, int D, int H, int W, int KD,
                                      double* stat_partial, int groups, void* stream) {
  BCP_REQUIRE(x && w && y && stat_partial && groups >= 1, "bcp_conv3_c1_fwd_stats: null pointer / bad groups");
  BCP_REQUIRE((KD == 1 && D == 1) || KD == 3, "bcp_conv3_c1_fwd_stats: bad KD/D");
  BCP_REQUIRE(aligned16(y) && (!bias || aligned16(bias)), "bcp_conv3_c1_fwd_stats: y / bias must be 16-B aligned");
  const int rows = c1_fwd_impl(x, w, bias, y, N, D, H, W, KD, stat_partial, groups, false, (hipStream_t)stream);
  BCP_REQUIRE(rows > 0, "bcp_conv3_c1_fwd_stats: fused statistics unavailable (N %% groups != 0): check bcp_conv3_c1_stat_rows first");
  BCP_CHECK_LAUNCH("bcp_conv3_c1_fwd_stats");
  return BCP_OK;
}

// ---- first layer + its norm with RECOMPUTE (round 3): Conv3d/2d(1 -> 16, k = 3) + BatchNorm / InstanceNorm + activation (+ elementwise
// dropout) of networks/VNet.py:17-26 block_one / networks/unet.py:19-28 in_conv -- y = conv + bias is never written: pass 1 takes the
// statistics from the accumulators, pass 2 runs the same MFMAs again and stores the activation; the backward recomputes y the same way
// next to da.  128 MB of y written + read back twice per direction at the LA size become two 8 MB reads of x.
// workspace: bcp_conv3_c1_norm_workspace_bytes.  Results are bit-identical to bcp_conv3_c1_fwd_stats + bcp_norm_fwd / bcp_norm_bwd.
namespace bcp {      // csrc/norm.hip
void norm_fwd_finalize_launch(const double* partial, int nb, int G, int C, long long rows_per_group, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, float* stats, hipStream_t s, float* amax_clear_or_null);
void norm_bwd_finalize_launch(const double* partial, int nb, int G, int C, long long rows_per_group, float* dgamma, float* dbeta, int accumulate,
                              float* c1c2raw, hipStream_t s, float* amax_clear_or_null);
}

extern "C" size_t bcp_conv3_c1_norm_workspace_bytes(int N, int D, int H, int W, int KD, int groups) {
  const int rows = bcp_conv3_c1_stat_rows(N, D, H, W, KD, groups);
  return rows > 0 ? (size_t)groups * rows * 16 * 2 * sizeof(double) + (size_t)4 * groups * 16 * sizeof(float) : 0;
}

extern "C" int bcp_conv3_c1_norm_fwd(const float* x, const float* w, const float* bias, int N, int D, int H, int W, int KD, int groups,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                     int act, const uint8_t* elem_mask, float elem_scale, const unsigned long long* mask_seed, float mask_p_keep,
                                     float* stats, void* workspace, float* out, float* amax_out_or_null, void* stream) {
  BCP_REQUIRE(x && w && stats && workspace && out && groups >= 1, "bcp_conv3_c1_norm_fwd: null pointer / bad groups");
  BCP_REQUIRE((KD == 1 && D == 1) || KD == 3, "bcp_conv3_c1_norm_fwd: bad KD/D");
  BCP_REQUIRE(aligned16(out) && aligned16(stats) && (!bias || aligned16(bias)), "bcp_conv3_c1_norm_fwd: alignment");
  hipStream_t s = (hipStream_t)stream;
  double* partial = reinterpret_cast<double*>(workspace);
  const int rows = c1_fwd_impl(x, w, bias, nullptr, N, D, H, W, KD, partial, groups, false, s, 1);
  BCP_REQUIRE(rows > 0, "bcp_conv3_c1_norm_fwd: the groups must be whole samples (N %% groups == 0)");
  norm_fwd_finalize_launch(partial, rows, groups, 16, (long long)N / groups * D * H * W, gamma, beta, running_mean, running_var, momentum, eps, stats, s,
                           amax_out_or_null);
  const C1Norm nm{stats, nullptr, nullptr, elem_mask, elem_scale, act, groups, amax_out_or_null, mask_seed, mask_p_keep};
  c1_fwd_impl(x, w, bias, out, N, D, H, W, KD, nullptr, groups, false, s, 2, &nm);
  BCP_CHECK_LAUNCH("bcp_conv3_c1_norm_fwd");
  return BCP_OK;
}

extern "C" int bcp_conv3_c1_norm_bwd(const float* x, const float* w, const float* bias, const float* da, int N, int D, int H, int W, int KD,
                                     int groups, const float* stats, int act, const uint8_t* elem_mask, float elem_scale,
                                     const unsigned long long* mask_seed, float mask_p_keep, float* dgamma,
                                     float* dbeta, int accumulate, void* workspace, float* dy, void* stream) {
  BCP_REQUIRE(x && w && da && stats && workspace && dy && groups >= 1, "bcp_conv3_c1_norm_bwd: null pointer / bad groups");
  BCP_REQUIRE((KD == 1 && D == 1) || KD == 3, "bcp_conv3_c1_norm_bwd: bad KD/D");
  BCP_REQUIRE(aligned16(da) && aligned16(dy) && (!bias || aligned16(bias)), "bcp_conv3_c1_norm_bwd: alignment");
  BCP_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bcp_conv3_c1_norm_bwd: dgamma and dbeta come together");
  hipStream_t s = (hipStream_t)stream;
  double* partial = reinterpret_cast<double*>(workspace);
  const int rows0 = bcp_conv3_c1_stat_rows(N, D, H, W, KD, groups);
  BCP_REQUIRE(rows0 > 0, "bcp_conv3_c1_norm_bwd: the groups must be whole samples (N %% groups == 0)");
  float* c1c2raw = reinterpret_cast<float*>(partial + (size_t)groups * rows0 * 16 * 2);
  C1Norm nm{stats, da, c1c2raw, elem_mask, elem_scale, act, groups, nullptr, mask_seed, mask_p_keep};
  const int rows = c1_fwd_impl(x, w, bias, nullptr, N, D, H, W, KD, partial, groups, false, s, 3, &nm);
  norm_bwd_finalize_launch(partial, rows, groups, 16, (long long)N / groups * D * H * W, dgamma, dbeta, accumulate, c1c2raw, s, nullptr);
  c1_fwd_impl(x, w, bias, dy, N, D, H, W, KD, nullptr, groups, false, s, 4, &nm);
  BCP_CHECK_LAUNCH("bcp_conv3_c1_norm_bwd");
  return BCP_OK;
}

// Round 5: the same backward with the layer's own weight gradient folded into pass 2 (kernel EPI 5): dy is never written -- the first
// layer has no dgrad, its weight gradient was dy's only reader -- and bcp_conv3_c1_wgrad, the last launch of the backward pass, is not
// needed.  dw[16][1][T] (+= when dw_accumulate) is what bcp_conv3_c1_norm_bwd + bcp_conv3_c1_wgrad leave, up to fp32 summation order
// (one partial slab per workgroup of pass 2 instead of per tile group).  workspace: bcp_conv3_c1_norm_bwd_wgrad_workspace_bytes.
extern "C" size_t bcp_conv3_c1_norm_bwd_wgrad_workspace_bytes(int N, int D, int H, int W, int KD, int groups) {
  const int rows = bcp_conv3_c1_stat_rows(N, D, H, W, KD, groups);
  if (rows <= 0) return 0;
  return bcp_conv3_c1_norm_workspace_bytes(N, D, H, W, KD, groups) + (size_t)groups * rows * (KD * 9) * 16 * sizeof(float);
}

extern "C" int bcp_conv3_c1_norm_bwd_wgrad(const float* x, const float* w, const float* bias, const float* da, int N, int D, int H, int W, int KD,
                                           int groups, const float* stats, int act, const uint8_t* elem_mask, float elem_scale,
                                           const unsigned long long* mask_seed, float mask_p_keep, float* dgamma, float* dbeta,
                                           int accumulate, void* workspace, float* dw, int dw_accumulate, void* stream) {
  BCP_REQUIRE(x && w && da && stats && workspace && dw && groups >= 1, "bcp_conv3_c1_norm_bwd_wgrad: null pointer / bad groups");
  BCP_REQUIRE((KD == 1 && D == 1) || KD == 3, "bcp_conv3_c1_norm_bwd_wgrad: bad KD/D");
  BCP_REQUIRE(aligned16(da) && (!bias || aligned16(bias)), "bcp_conv3_c1_norm_bwd_wgrad: alignment");
  BCP_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bcp_conv3_c1_norm_bwd_wgrad: dgamma and dbeta come together");
  hipStream_t s = (hipStream_t)stream;
  double* partial = reinterpret_cast<double*>(workspace);
  const int rows0 = bcp_conv3_c1_stat_rows(N, D, H, W, KD, groups);
  BCP_REQUIRE(rows0 > 0, "bcp_conv3_c1_norm_bwd_wgrad: the groups must be whole samples (N %% groups == 0)");
  float* c1c2raw = reinterpret_cast<float*>(partial + (size_t)groups * rows0 * 16 * 2);
  float* wpart = c1c2raw + (size_t)4 * groups * 16;            // [groups * rows0 workgroups][T][16]
  C1Norm nm{stats, da, c1c2raw, elem_mask, elem_scale, act, groups, nullptr, mask_seed, mask_p_keep};
  const int rows = c1_fwd_impl(x, w, bias, nullptr, N, D, H, W, KD, partial, groups, false, s, 3, &nm);
  norm_bwd_finalize_launch(partial, rows, groups, 16, (long long)N / groups * D * H * W, dgamma, dbeta, accumulate, c1c2raw, s, nullptr);
  c1_fwd_impl(x, w, bias, wpart, N, D, H, W, KD, nullptr, groups, false, s, 5, &nm);
  const int G = groups * rows, T = KD * 9;                     // workgroups of pass 2 = statistics rows of pass 1 (same tiling)
  if (G >= 32 || options().wgrad_reduce_flat == 2)      // (2: tests force the flat sum for every group count)
    launch_reduce_deep(wpart, dw, G, T, 1, 16, 1, 16, dw_accumulate, s);
  else
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(1, 1), dim3(256), 0, s, wpart, dw, G, T, 1, 16, 1, 16, dw_accumulate);
  BCP_CHECK_LAUNCH("bcp_conv3_c1_norm_bwd_wgrad");
  return BCP_OK;
}

extern "C" int bcp_conv3_c1_wgrad(const float* x, const float* dy, float* dw, int N, int D, int H, int W, int KD, int accumulate,
                                  void* workspace, void* stream) {
  BCP_REQUIRE(x && dy && dw && workspace, "bcp_conv3_c1_wgrad: null pointer");
  BCP_REQUIRE((KD == 1 && D == 1) || KD == 3, "bcp_conv3_c1_wgrad: bad KD/D");
  ConvDims cd;
  fill_dims(cd, N, D, H, W, 1, 16);
  float* ws = reinterpret_cast<float*>(workspace);
  int G;
  if (KD == 3) {
    cd.tiles_d = cdiv(D, 4); cd.tiles_h = cdiv(H, 4); cd.tiles_w = cdiv(W, 16);
    const int tiles = N * cd.tiles_d * cd.tiles_h * cd.tiles_w;
    const int g0 = tiles < 512 ? tiles : 512, tpg = cdiv(tiles, g0);
    G = cdiv(tiles, tpg);
    hipLaunchKernelGGL((k_conv3_c1_wgrad<3, 4, 4, 16>), dim3(G), dim3(256), 0, (hipStream_t)stream, x, dy, ws, cd, tiles, tpg);
  } else {
    cd.tiles_d = 1; cd.tiles_h = cdiv(H, 16); cd.tiles_w = cdiv(W, 16);
    const int tiles = N * cd.tiles_h * cd.tiles_w;
    const int g0 = tiles < 512 ? tiles : 512, tpg = cdiv(tiles, g0);
    G = cdiv(tiles, tpg);
    hipLaunchKernelGGL((k_conv3_c1_wgrad<1, 1, 16, 16>), dim3(G), dim3(256), 0, (hipStream_t)stream, x, dy, ws, cd, tiles, tpg);
  }
  const int T = KD * 9;
  if (G >= 32 || options().wgrad_reduce_flat == 2)      // (2: tests force the flat sum for every group count)
    launch_reduce_deep(ws, dw, G, T, 1, 16, 1, 16, accumulate, (hipStream_t)stream);
  else
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(1, 1), dim3(256), 0, (hipStream_t)stream, ws, dw, G, T, 1, 16, 1, 16, accumulate);
  BCP_CHECK_LAUNCH("bcp_conv3_c1_wgrad");
  return BCP_OK;
}
