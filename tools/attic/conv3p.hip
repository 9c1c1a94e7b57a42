// bcp_amd/csrc/conv3p.hip -- 3x3x3 / 3x3 convolution forward + dgrad as a PERSISTENT 8-WAVE PIPELINE (round 2).
//
// Reference ops: nn.Conv3d(k=3,pad=1) networks/VNet.py:17 (and its autograd dgrad: the same kernel on the flipped /
// transposed weight pack), SURVEY.md A1.1.
//
// Why a second kernel family (measured in round 1): the 4-wave kernels of conv3.hip run the mid / deep V-Net
// levels at 32-62 % of the fp32-MFMA peak -- one wave per SIMD for the 32-channel level (110 KB of resident weights per
// workgroup), 2.19 / 2.5 equal workgroups per CU at the 128 / 256-channel levels (the CU that gets three sets the kernel
// time: 73 % / 83 % before anything else), 12-48 % of the MFMA rows spent on padding voxels of box tiles that do not divide
// 14x14x10 / 7x7x5, two barriers with an exposed LDS refill between them per work item.
//
// Structure here:
//   * ONE 512-thread workgroup per CU (8 waves = 2 per SIMD, WM x WN wave grid over the output tile BM voxels x BN channels):
//     the two waves of a SIMD share one copy of the staged operands and cover each other's LDS latency.
//   * The work of a workgroup is a flat sequence of STAGES (item, cin chunk, tap group).  Every stage runs the same way:
//     issue the global loads of the NEXT stage's operands (weights: WT taps x 16 cin x BN cout; input halo: once per chunk)
//     into registers, run this stage's MFMAs out of LDS buffer [cur], drop the registers into LDS buffer [next], ONE
//     LDS-only barrier.  Nothing is ever waited for except at the end of a stage that is >= 4600 cycles of MFMA long.
//   * Weights are streamed through LDS (double-buffered stage) instead of kept resident: at fp32-MFMA rates the stream is
//     ~5 B/clk/CU, and it frees the LDS for a second halo buffer and for BN = 32 / 64 channel slabs.
//   * M tiles without padding waste.  BRICK mode: an item is NB bricks of 4x4x4 voxels, each with its own 6x6x6 halo (any
//     volume whose extents are multiples of 4 is covered exactly: 28x28x20, 56x56x40, 112x112x80, 24^3 ...); a 16-row MFMA
//     tile is one 4x4 d-slice of a brick, so the three kd taps of a (kh, kw) pair re-use d-planes held in registers.
//     FLAT mode (small volumes: 14x14x10, 7x7x5, 12^3, 6^3): an item is BM CONSECUTIVE voxels of one sample in raster order,
//     the staged input is the contiguous range [m0 - R, m0 + BM + R) with R = H*W + W + 1, a tap is a constant offset into
//     it and a 27-bit per-lane mask zeroes the taps that would wrap around a row / plane / sample end.
//   * Balanced grids by construction: the host picks (slab width, split-K) so that items x slabs x splits <= 256 = one
//     workgroup per CU, all co-resident, equal work (the 128-channel level: 62 x 4 = 248 workgroups; the 256-channel level:
//     8 x 8 x 4 = 256).
//
// Numerics: v_mfma_f32_16x16x4_f32 (exact fp32 fmaf chain); only the summation ORDER differs from conv3.hip
// (chunk -> kh -> kw -> kd -> cin).  Fused per-channel (sum, sum of squares) statistics as in conv3.hip (fp64).
#include "conv3_defs.h"
#include "../../include/bcp_hip.h"

namespace bcp {

enum { P8_BRICK = 0, P8_FLAT = 1 };

// Measurement-only switches (tools/ablate_p8.sh builds variants of the library; the product build has 0):
//   1: no epilogue (stores / statistics)   2: no halo prefetch + stash inside the loop   4: no weight prefetch + stash
//   8: no MFMA stage body
#ifndef P8_ABLATE
#define P8_ABLATE 0
#endif

struct P8Args {
  const float* X;
  const float* Wp;
  const float* bias;
  float* Y;
  ConvDims cd;
  int n_items;          // M tiles over the whole batch
  int items_per_group;  // statistics groups never share an item
  int bricks_per_group; // BRICK: bricks of one group (samples_per_group * bricks per sample)
  int bd, bh, bw;       // BRICK: bricks per sample along d / h / w
  int spg;              // samples per group
  int V, R;             // FLAT: voxels per sample, halo radius of the staged range
  int tiles_per_sample; // FLAT
  int accumulate;
  long long slab_stride; // split-K: elements between the partial output slabs
  StatsArg st;
};

template <int KD, int MODE, int WM, int WN, int MT, int NT, int NB, int WT, int NWB, int AVMAX>
struct P8 {
  static constexpr int T = KD * 9, PD = (KD == 3) ? 1 : 0;
  static constexpr int S = T / WT;                 // stages per cin chunk
  static constexpr int KHS = WT / (3 * KD);        // kh values per stage
  static constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
  static constexpr int HVB = (4 + 2 * PD) * 36;    // halo voxels of one brick (6x6x6)
  static constexpr int AV = (MODE == P8_BRICK) ? NB * HVB : AVMAX;   // staged input voxels (upper bound in FLAT mode)
  static constexpr int NA4 = (MODE == P8_BRICK) ? NB * ((HVB * 4 + 511) / 512) : (AV * 4 + 511) / 512;
  static constexpr int WST4 = WT * 4 * BN;         // float4s per weight stage
  static constexpr int NW4 = (WST4 + 511) / 512;
  static_assert(WM * WN == 8, "8 waves");
  static_assert(T % WT == 0 && WT % (3 * KD) == 0, "a stage is a whole number of kh rows");
  static_assert(MODE == P8_FLAT || (KD == 3 && NB * 4 == WM * MT), "BRICK: an M tile is one d-slice of a brick");
  static_assert(MODE == P8_FLAT || (4 % MT == 0), "BRICK: the M tiles of a wave lie in one brick");
};

// ------------------------------------------------------------------------------------------------
template <int KD, int MODE, int WM, int WN, int MT, int NT, int NB, int WT, int NWB, int AVMAX>
__global__ __launch_bounds__(512) void k_c3p(P8Args a) {
  using C = P8<KD, MODE, WM, WN, MT, NT, NB, WT, NWB, AVMAX>;
  constexpr int T = C::T, PD = C::PD, S = C::S, KHS = C::KHS, BM = C::BM, BN = C::BN, HVB = C::HVB;
  constexpr int NA4 = C::NA4, WST4 = C::WST4, NW4 = C::NW4;
  const ConvDims& cd = a.cd;

  HIP_DYNAMIC_SHARED(float4, smem4)
  float* smem = reinterpret_cast<float*>(smem4);
  const int av = (MODE == P8_BRICK) ? C::AV : (BM + 2 * a.R);     // staged voxels
  float* Abuf = smem;                                            // [2][av][16]
  float* Wbuf = smem + 2 * av * 16;                              // [NWB][WT][4][BN][4]
  double* Ss = reinterpret_cast<double*>(Wbuf + NWB * WST4 * 4); // [WM][BN][2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the tile arithmetic below stays on the scalar unit
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave % WM, wn = wave / WM;
  const int cout0 = blockIdx.y * BN;
  const int cin4 = cd.Cin16 >> 2;
  const int nch_all = cd.Cin16 >> 4;
  const int c0 = (int)((long long)nch_all * blockIdx.z / gridDim.z), c1 = (int)((long long)nch_all * (blockIdx.z + 1) / gridDim.z);
  const int nch = c1 - c0;
  (void)nch;
  float* Y = a.Y + (long long)blockIdx.z * a.slab_stride;

  // ---- items of this workgroup (XCD-aware: each XCD walks one contiguous eighth of the list, its workgroups side by side)
  int tile, t_end, t_step;
  if (gridDim.x % 8 == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    t_step = gridDim.x >> 3;
    tile = (int)((long long)a.n_items * xcd / 8) + j;
    t_end = (int)((long long)a.n_items * (xcd + 1) / 8);
  } else {
    tile = blockIdx.x; t_end = a.n_items; t_step = gridDim.x;
  }
  const StatsArg& st = a.st;
  if (st.partial && tid < BN && cout0 + tid < cd.Cout) {   // rows of groups this workgroup never visits must read as zero
    for (int g = 0; g < st.G; ++g) {
      double* z = st.partial + (((long long)g * st.rows + blockIdx.x) * st.C + cout0 + tid) * 2;
      z[0] = 0.0; z[1] = 0.0;
    }
  }
  if (tile >= t_end) return;

  // ---- launch-invariant staging maps
  // weights: float4 q of a stage = (tl, cig, co); tl = (khl * 3 + kw) * KD + kd
  unsigned wrel[NW4];
#pragma unroll
  for (int u = 0; u < NW4; ++u) {
    const int q = (tid + u * 512 < WST4) ? tid + u * 512 : tid;      // tail threads repeat their first float4: no predicates
    const int co = q % BN, cig = (q / BN) & 3, tl = q / (4 * BN);
    const int kd = tl % KD, kw = (tl / KD) % 3, khl = tl / (3 * KD);
    const int tap0 = (KD == 3 ? kd * 9 : 0) + khl * 3 + kw;
    wrel[u] = (unsigned)((((long long)tap0 * cin4 + cig) * cd.Cout16 + co) * 4);
  }
  const long long w_stage_step = (long long)KHS * 3 * cin4 * cd.Cout16 * 4;   // next kh group
  const long long w_chunk_step = (long long)4 * cd.Cout16 * 4;                // next cin chunk
  // input halo, BRICK: per brick two float4 slots per thread: r = tid + u2 * 512 < 864 -> (halo voxel, 16-B part)
  constexpr int AU = (MODE == P8_BRICK) ? (HVB * 4 + 511) / 512 : NA4;
  unsigned arel[AU];
  int ahd[AU], ahh[AU], ahw[AU];
  if constexpr (MODE == P8_BRICK) {
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      const int r = (tid + u * 512 < HVB * 4) ? tid + u * 512 : tid;   // tail threads repeat their first float4
      const int hv = r >> 2, part = r & 3;
      ahd[u] = hv / 36; ahh[u] = (hv / 6) % 6; ahw[u] = hv % 6;
      arel[u] = (unsigned)(((ahd[u] * cd.H + ahh[u]) * cd.W + ahw[u]) * cd.Cin + part * 4);
    }
  }
  const int apart = tid & 3;

  // ---- per-lane operand addressing
  // A fragment: lane (li = voxel row of the M tile, lg = k group) reads 16 B at [voxel][4 lg .. 4 lg + 3]
  int abase;                       // floats, relative to the halo buffer; same for every M tile of the wave up to a stride
  constexpr int GM0_STRIDE = (MODE == P8_BRICK) ? 36 * 16 : 16 * 16;   // distance between consecutive M tiles of a wave
  if constexpr (MODE == P8_BRICK) {
    const int gm0 = wm * MT, b = gm0 >> 2, td0 = gm0 & 3;
    abase = (b * HVB + (td0 * 6 + (li >> 2)) * 6 + (li & 3)) * 16 + lg * 4;
  } else {
    abase = (a.R + wm * MT * 16 + li) * 16 + lg * 4;
  }
  const int bbase = (lg * BN + wn * NT * 16 + li) * 4;   // B fragment of local tap 0; + tl * 4 * BN * 4 per tap, + nt * 64

  f32x4 acc[MT][NT];
  f32x4 acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};   // MT * NT == 1: a lone accumulator would serialise on the 40-cycle MFMA latency
  constexpr bool TWOACC = (MT * NT == 1);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = cout0 + wn * NT * 16 + nt * 16 + li;
    bv[nt] = (a.bias && co < cd.Cout) ? a.bias[co] : 0.f;
  }

  // ---- staging helpers
  // Every load is issued unconditionally (invalid lanes read the tile-independent safe address X) and the validity bits are
  // applied when the registers are finally stored to LDS one stage later: `v = 0; if (ok) v = load` makes hipcc wait for each
  // load at the merge point of its branch, i.e. one exposed L2 / HBM round trip per float4 at the top of every stage.
  float4 wreg[NW4], areg[NA4];
  unsigned aok = 0;
  auto wfetch = [&](int c, int s) __attribute__((always_inline)) {
    const float* wb = a.Wp + (long long)s * w_stage_step + (long long)c * w_chunk_step + (long long)cout0 * 4;
#pragma unroll
    for (int u = 0; u < NW4; ++u) wreg[u] = ld4(wb + wrel[u]);
  };
  auto wstash = [&](float* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NW4; ++u) st4(dst + ((tid + u * 512 < WST4) ? tid + u * 512 : tid) * 4, wreg[u]);
  };
  auto afetch = [&](int t, int c) __attribute__((always_inline)) {
    const bool cok = c * 16 + apart * 4 < cd.Cin;
    aok = 0;
    if constexpr (MODE == P8_BRICK) {
      const int g = t / a.items_per_group, lb0 = (t - g * a.items_per_group) * NB;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int lb = lb0 + b;                      // brick within the group
        const bool bok = lb < a.bricks_per_group;
        const int bps = a.bd * a.bh * a.bw;
        const int n = g * a.spg + lb / bps, rb = lb % bps;
        const int d0 = (rb / (a.bh * a.bw)) * 4, h0 = ((rb / a.bw) % a.bh) * 4, w0 = (rb % a.bw) * 4;
        const float* xb = a.X + ((((long long)n * cd.D + (d0 - 1)) * cd.H + (h0 - 1)) * cd.W + (w0 - 1)) * cd.Cin + c * 16;
#pragma unroll
        for (int u = 0; u < AU; ++u) {
          const bool ok = bok && cok && (unsigned)(d0 - 1 + ahd[u]) < (unsigned)cd.D &&
                          (unsigned)(h0 - 1 + ahh[u]) < (unsigned)cd.H && (unsigned)(w0 - 1 + ahw[u]) < (unsigned)cd.W;
          areg[b * AU + u] = ld4(ok ? xb + arel[u] : a.X);
          aok |= (ok ? 1u : 0u) << (b * AU + u);
        }
      }
    } else {
      const int n = t / a.tiles_per_sample, m0 = (t - n * a.tiles_per_sample) * BM;
      const float* xb = a.X + ((long long)n * a.V + (m0 - a.R)) * cd.Cin + c * 16 + apart * 4;
#pragma unroll
      for (int u = 0; u < NA4; ++u) {
        const int f = (tid + u * 512 < av * 4) ? tid + u * 512 : (av - 1) * 4 + apart;   // tail threads repeat the last voxel
        const int rv = f >> 2;
        const bool ok = cok && (unsigned)(m0 - a.R + rv) < (unsigned)a.V;
        areg[u] = ld4(ok ? xb + (long long)rv * cd.Cin : a.X);
        aok |= (ok ? 1u : 0u) << u;
      }
    }
  };
  auto astash = [&](float* dst) __attribute__((always_inline)) {
    if constexpr (MODE == P8_BRICK) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int u = 0; u < AU; ++u)
          st4(dst + b * HVB * 16 + ((tid + u * 512 < HVB * 4) ? tid + u * 512 : tid) * 4,
              ((aok >> (b * AU + u)) & 1u) ? areg[b * AU + u] : make_float4(0.f, 0.f, 0.f, 0.f));
    } else {
#pragma unroll
      for (int u = 0; u < NA4; ++u)
        st4(dst + ((tid + u * 512 < av * 4) ? tid + u * 512 : (av - 1) * 4 + apart) * 4, ((aok >> u) & 1u) ? areg[u] : make_float4(0.f, 0.f, 0.f, 0.f));
    }
  };

  // FLAT: per M tile of this lane, bit tl of amask = "tap tl stays inside the sample" (and the voxel itself exists)
  unsigned amask[MT];
  auto flat_masks = [&](int t) __attribute__((always_inline)) {
    const int n = t / a.tiles_per_sample, m0 = (t - n * a.tiles_per_sample) * BM;
    (void)n;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + (wm * MT + mt) * 16 + li;
      unsigned mk = 0;
      if (m < a.V) {
        const int hw = cd.H * cd.W;
        const int d = m / hw, r2 = m - d * hw, h = r2 / cd.W, w = r2 - h * cd.W;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) {
              const bool ok = (unsigned)(d + kd - PD) < (unsigned)cd.D && (unsigned)(h + kh - 1) < (unsigned)cd.H &&
                              (unsigned)(w + kw - 1) < (unsigned)cd.W;
              mk |= (ok ? 1u : 0u) << ((kh * 3 + kw) * KD + kd);
            }
      }
      amask[mt] = mk;
    }
  };

  // ---- fused statistics
  double s1[NT], s2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { s1[nt] = 0.0; s2[nt] = 0.0; }
  int cur_g = st.partial ? tile / a.items_per_group : 0;
  auto stats_flush8 = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      double x = s1[nt], y = s2[nt];
      x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
      y += __shfl_xor(y, 16); y += __shfl_xor(y, 32);
      if (lg == 0) {
        const int c = wn * NT * 16 + nt * 16 + li;
        Ss[(wm * BN + c) * 2] = x; Ss[(wm * BN + c) * 2 + 1] = y;
      }
      s1[nt] = 0.0; s2[nt] = 0.0;
    }
    __syncthreads();
    if (tid < BN && cout0 + tid < cd.Cout) {
      double x = 0.0, y = 0.0;
#pragma unroll
      for (int w = 0; w < WM; ++w) { x += Ss[(w * BN + tid) * 2]; y += Ss[(w * BN + tid) * 2 + 1]; }
      double* dst = st.partial + (((long long)g * st.rows + blockIdx.x) * st.C + cout0 + tid) * 2;
      dst[0] = x; dst[1] = y;
    }
    __syncthreads();
  };

  // ---- epilogue of one item: accumulators -> Y (+ bias, += when asked), statistics, accumulators = 0
  auto epilogue = [&](int t) __attribute__((always_inline)) {
    const bool want_stats = st.partial != nullptr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int gm = wm * MT + mt;
      long long row0;          // element offset of accumulator row r = 0 of this lane
      int rstep;               // elements between rows r and r + 1
      bool rok[4];
      if constexpr (MODE == P8_BRICK) {
        const int g = t / a.items_per_group, lb = (t - g * a.items_per_group) * NB + (gm >> 2);
        const int bps = a.bd * a.bh * a.bw;
        const int n = g * a.spg + lb / bps, rb = lb % bps;
        const int d = (rb / (a.bh * a.bw)) * 4 + (gm & 3), h = ((rb / a.bw) % a.bh) * 4 + lg, w0 = (rb % a.bw) * 4;
        const bool ok = lb < a.bricks_per_group && d < cd.D && h < cd.H;
        row0 = ((((long long)n * cd.D + d) * cd.H + h) * cd.W + w0) * cd.Cout;
        rstep = cd.Cout;
#pragma unroll
        for (int r = 0; r < 4; ++r) rok[r] = ok && w0 + r < cd.W;
      } else {
        const int n = t / a.tiles_per_sample, m = (t - n * a.tiles_per_sample) * BM + gm * 16 + lg * 4;
        row0 = ((long long)n * a.V + m) * cd.Cout;
        rstep = cd.Cout;
#pragma unroll
        for (int r = 0; r < 4; ++r) rok[r] = m + r < a.V;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = cout0 + wn * NT * 16 + nt * 16 + li;
        if (co < cd.Cout) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (rok[r]) {
              float* p = Y + row0 + (long long)r * rstep + co;
              float v = acc[mt][nt][r];
              if (TWOACC) v += acc1[r];
              v += bv[nt];
              *p = v;
              if (want_stats) { s1[nt] += (double)v; s2[nt] += (double)v * (double)v; }
            }
          }
        }
        acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  // ---- one stage of MFMAs out of LDS.  Explicit software pipeline: the operand fragments of micro-step j + 1 (one tap) are
  // read from LDS BEFORE the MFMAs of micro-step j are issued (two register sets; sched_barrier keeps hipcc from sinking the
  // reads back to their first use).  A tap is only 4 * MT * NT MFMAs = 128 .. 512 matrix-pipe cycles: with reads issued
  // just-in-time the two waves of a SIMD cannot cover the LDS round trip (measured, first version of this kernel: the
  // 128-channel level ran its stages at 60 % of the MFMA rate, the 64-channel level at 84 %).
  auto compute = [&](const float* Ab, const float* Wb, int s) __attribute__((always_inline)) {
    if constexpr (MODE == P8_BRICK) {
      float4 ap[2][MT + 2], bb[2][NT];
      auto ldA = [&](int grp, int set) __attribute__((always_inline)) {      // the MT + 2 d-planes this wave's d-slices touch through the three kd taps of (kh, kw)
        const int kh = s * KHS + grp / 3, kw = grp % 3;
#pragma unroll
        for (int p = 0; p < MT + 2; ++p) ap[set][p] = ld4(Ab + abase + ((p * 6 + kh) * 6 + kw) * 16);
      };
      auto ldB = [&](int tl, int set) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bb[set][nt] = ld4(Wb + tl * 4 * BN * 4 + bbase + nt * 64);
      };
      ldA(0, 0);
      ldB(0, 0);
#pragma unroll
      for (int tl = 0; tl < WT; ++tl) {       // tl = (khl * 3 + kw) * 3 + kd
        const int grp = tl / 3, kd = tl % 3;
        if (tl + 1 < WT) {
          if ((tl + 1) % 3 == 0) ldA(grp + 1, (grp + 1) & 1);
          ldB(tl + 1, (tl + 1) & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float4 av4 = ap[grp & 1][mt + kd], bv4 = bb[tl & 1][nt];
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av4.x, bv4.x, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av4.y, bv4.y, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av4.z, bv4.z, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av4.w, bv4.w, acc[mt][nt], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      const int hw = cd.H * cd.W;
      float4 aa[2][MT], bb[2][NT];
      auto ldAB = [&](int tl, int set) __attribute__((always_inline)) {      // tl = (khl * 3 + kw) * KD + kd
        const int kd = tl % KD, kw = (tl / KD) % 3, kh = s * KHS + tl / (3 * KD);
        const int off = ((kd - PD) * hw + (kh - 1) * cd.W + (kw - 1)) * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aa[set][mt] = ld4(Ab + abase + mt * GM0_STRIDE + off);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bb[set][nt] = ld4(Wb + tl * 4 * BN * 4 + bbase + nt * 64);
      };
      ldAB(0, 0);
#pragma unroll
      for (int tl = 0; tl < WT; ++tl) {
        if (tl + 1 < WT) ldAB(tl + 1, (tl + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const int tg = s * WT + tl;           // global tap index in mask order ((kh * 3 + kw) * KD + kd)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          float4 av4 = aa[tl & 1][mt];
          if (!((amask[mt] >> tg) & 1u)) av4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float4 bv4 = bb[tl & 1][nt];
            if constexpr (TWOACC) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av4.x, bv4.x, acc[mt][nt], 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av4.y, bv4.y, acc1, 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av4.z, bv4.z, acc[mt][nt], 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av4.w, bv4.w, acc1, 0, 0, 0);
            } else {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av4.x, bv4.x, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av4.y, bv4.y, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av4.z, bv4.z, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av4.w, bv4.w, acc[mt][nt], 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- the stage sequence.  Control flow is arranged so that every prefetch (issue -> stash) is STRAIGHT-LINE code: a chunk
  // is S stages unrolled at compile time, all chunks but the last of the workgroup run through chunk_body<false>, the last one
  // through chunk_body<true> (no prefetch).  With `if (has_next) fetch ... if (has_next) stash` hipcc's wait-count pass cannot
  // prove on the loop back edge that the prefetch registers have landed and drains vmcnt to 0 -- i.e. the epilogue's stores --
  // at the top of every stage (measured: the 16-channel layer at 64 % instead of 79 %).
  constexpr bool RES = (NWB == 1);                 // single weight stage in total (checked by the host): loaded once
  int c = c0, k = 0;                               // chunk counter k: halo buffer k & 1, weight buffer (k * S + s) & 1
  afetch(tile, c);
  wfetch(c, 0);
  wstash(Wbuf);
  astash(Abuf);
  if (MODE == P8_FLAT) flat_masks(tile);
  BCP_LDS_BARRIER();

  auto chunk_body = [&](auto last_tag, int nt_, int nc) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_tag)::value;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int q = k * S + s;
      const bool more = !LAST || s + 1 < S;        // a stage follows inside this workgroup (compile time per unrolled s)
      if (more && !RES && !(P8_ABLATE & 4)) {
        if (s + 1 < S) wfetch(c, s + 1); else wfetch(nc, 0);
      }
      if (!LAST && s == S - 1 && !(P8_ABLATE & 2)) afetch(nt_, nc);
      if (!(P8_ABLATE & 8)) compute(Abuf + (k & 1) * av * 16, Wbuf + (RES ? 0 : (q & 1)) * WST4 * 4, s);
      if (more && !RES && !(P8_ABLATE & 4)) wstash(Wbuf + ((q + 1) & 1) * WST4 * 4);
      if (!LAST && s == S - 1 && !(P8_ABLATE & 2)) astash(Abuf + ((k + 1) & 1) * av * 16);
      if (s == S - 1 && c == c1 - 1) {             // item complete
        if (st.partial) {
          const int g = tile / a.items_per_group;
          if (g != cur_g) { stats_flush8(cur_g); cur_g = g; }
        }
        if (!(P8_ABLATE & 1)) epilogue(tile);
        else if (acc[0][0][0] == 1.2345e-30f) Y[tid] = acc[0][0][1];   // keep the accumulators alive
        if (MODE == P8_FLAT && !LAST) flat_masks(nt_);
      }
      if (more) BCP_LDS_BARRIER();
    }
  };
  for (;;) {
    int nc = c + 1, nt_ = tile;
    if (nc == c1) { nc = c0; nt_ = tile + t_step; }
    if (nt_ >= t_end) break;
    chunk_body(std::integral_constant<bool, false>{}, nt_, nc);
    ++k;
    c = nc; tile = nt_;
  }
  chunk_body(std::integral_constant<bool, true>{}, tile, c);
  if (st.partial) stats_flush8(cur_g);
}

// y (+)= bias + sum_k part[k]  (split-K epilogue; deep levels only: <= 1 MB)
__global__ __launch_bounds__(256) void k_p8_sum_slabs(const float* __restrict__ part, int SK, long long n, int C,
                                                      const float* __restrict__ bias, float* __restrict__ y, int accumulate) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float s = bias ? bias[i % C] : 0.f;
    for (int k = 0; k < SK; ++k) s += part[k * n + i];
    y[i] = accumulate ? y[i] + s : s;
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct P8Plan {
  int mode, cfg;          // cfg: index into the instance table below
  int P, slabs, ksplit;
  size_t lds;
  P8Args args;
  int stat_rows;          // partial rows per group (0: fused statistics not available)
};

template <int KD, int MODE, int WM, int WN, int MT, int NT, int NB, int WT, int NWB, int AVMAX>
static void p8_launch(const P8Plan& pl, hipStream_t s) {
  auto kfn = k_c3p<KD, MODE, WM, WN, MT, NT, NB, WT, NWB, AVMAX>;
  if (pl.lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
  hipLaunchKernelGGL(kfn, dim3(pl.P, pl.slabs, pl.ksplit), dim3(512), pl.lds, s, pl.args);
}

// instance table: {mode, WM, WN, MT, NT, NB, WT, NWB}
//  0: BRICK 8x1 MT2 NT1 NB4 WT27 NWB1   16 -> 16 channels (BN 16): weights of the single chunk resident, 138 KB
//  1: BRICK 4x2 MT4 NT1 NB4 WT9  NWB2   BN 32 (the 32-channel level), 147 KB
//  2: BRICK 4x2 MT2 NT2 NB2 WT9  NWB2   BN 64 (the 64-channel level), 129 KB
//  3: FLAT  4x2 MT1 NT1     WT27 NWB2   BM 64 x BN 32, R <= 160 (the 128 / 256-channel levels), <= 160 KB
//  4: FLAT  4x2 MT1 NT1     WT9  NWB2   2-D (KD = 1): BM 64 x BN 32
static constexpr int kFlatAvMax = 384;   // BM + 2 R <= 384
static bool p8_plan(P8Plan& pl, const ConvDims& cd, int KD, int G, bool want_stats, bool has_ws) {
  const Options& o = options();
  if (o.conv3_p8 == 0) return false;
  const bool force = o.conv3_p8 >= 2;
  const long long vps = (long long)cd.D * cd.H * cd.W, vox = vps * cd.N;
  const int PD = KD == 3 ? 1 : 0;
  const int R = PD * cd.H * cd.W + cd.W + 1;
  P8Args& a = pl.args;
  a.cd = cd;
  int BM, BN, NB = 0, WT, NWB;
  const int nch = cd.Cin16 / 16;
  if (KD == 3 && o.conv3_p8 == 2 && cd.Cout16 == 16 && cd.Cin16 == 16) { pl.cfg = 0; pl.mode = P8_BRICK; BM = 256; BN = 16; NB = 4; WT = 27; NWB = 1; }
  else if (KD == 3 && cd.Cout16 % 32 == 0 && cd.Cout16 % 64 != 0 && (force ? o.conv3_p8 == 2 : vox >= 64LL * 1024)) { pl.cfg = 1; pl.mode = P8_BRICK; BM = 256; BN = 32; NB = 4; WT = 9; NWB = 2; }
  else if (KD == 3 && cd.Cout16 % 64 == 0 && (force ? o.conv3_p8 == 2 : (vox >= 16LL * 1024 && 64 + 2 * R > kFlatAvMax))) { pl.cfg = 2; pl.mode = P8_BRICK; BM = 128; BN = 64; NB = 2; WT = 9; NWB = 2; }
  else if (cd.Cout16 % 32 == 0 && 64 + 2 * R <= kFlatAvMax && (force || KD == 3)) { pl.cfg = KD == 3 ? 3 : 4; pl.mode = P8_FLAT; BM = 64; BN = 32; WT = KD == 3 ? 27 : 9; NWB = 2; }
  else return false;
  if (!force && !((o.conv3_p8_cfgs >> (pl.cfg == 1 ? 0 : pl.cfg == 2 ? 1 : 2)) & 1)) return false;
  pl.slabs = cd.Cout16 / BN;
  pl.ksplit = 1;
  const int spg = G > 0 ? cd.N / G : cd.N;           // samples per statistics group (G = 0: no statistics, one group)
  const int groups = G > 0 ? G : 1;
  if (G > 0 && cd.N % G) return false;
  a.spg = spg;
  if (pl.mode == P8_BRICK) {
    a.bd = cdiv(cd.D, 4); a.bh = cdiv(cd.H, 4); a.bw = cdiv(cd.W, 4);
    // padding waste of partial bricks: leave shapes that waste more than a quarter to the other kernels unless forced
    const long long covered = (long long)a.bd * a.bh * a.bw * 64;
    if (!force && covered * 4 > vps * 5) return false;
    a.bricks_per_group = spg * a.bd * a.bh * a.bw;
    a.items_per_group = cdiv(a.bricks_per_group, NB);
    a.V = 0; a.R = 0; a.tiles_per_sample = 0;
  } else {
    a.V = (int)vps; a.R = R;
    a.tiles_per_sample = cdiv(vps, BM);
    a.items_per_group = spg * a.tiles_per_sample;
    a.bd = a.bh = a.bw = 0; a.bricks_per_group = 0;
  }
  a.n_items = groups * a.items_per_group;
  // split-K (deep levels): fill the CUs when (items x slabs) leaves at least half of them idle.  A split launch cannot fuse
  // the norm statistics (partial sums): stat_rows = 0 tells the caller to run the plain forward + a statistics pass.
  (void)want_stats;
  if (has_ws && nch >= 2) {
    int ks = 256 / (a.n_items * pl.slabs);
    if (ks > nch) ks = nch;
    if (ks > 8) ks = 8;
    if (ks >= 2) pl.ksplit = ks;
    if (o.splitk >= 1 && o.splitk <= 8 && o.splitk <= nch) pl.ksplit = o.splitk;
  }
  if (!force && pl.mode == P8_FLAT && pl.ksplit > 1 && !((o.conv3_p8_cfgs >> 3) & 1)) return false;   // bit 3: flat tiles WITH split-K (slab sum + a separate statistics pass)
  if (NWB == 1 && cdiv(nch, pl.ksplit) * (KD * 9 / WT) > 1) return false;
  int maxP = 256 / (pl.slabs * pl.ksplit);
  if (maxP < 1) maxP = 1;
  int P = a.n_items < maxP ? a.n_items : maxP;
  if (o.conv3_p > 0 && o.conv3_p < P) P = o.conv3_p;
  pl.P = P;
  const int av = pl.mode == P8_BRICK ? NB * (4 + 2 * PD) * 36 : BM + 2 * R;
  pl.lds = ((size_t)2 * av * 16 + (size_t)NWB * WT * 4 * BN * 4) * sizeof(float) + (size_t)8 * BN * 2 * sizeof(double);
  if (pl.lds > 160 * 1024) return false;
  pl.stat_rows = (pl.ksplit == 1) ? P : 0;
  return true;
}

int p8_fwd(const float* x, const float* wp, const float* bias, float* y, const ConvDims& cd, int KD, int accumulate, void* workspace,
           double* stat_partial, int G, bool dry, hipStream_t s, bool* handled) {
  P8Plan pl;
  *handled = false;
  const bool want_stats = G > 0;
  // "+=" into y is left to the kernels of conv3.hip: a conditional read-modify-write in the epilogue makes hipcc put a
  // vmcnt(0) in front of EVERY store (eight serialised store round trips per item, measured 2.3 us of a 3 us phase)
  if (accumulate) return 0;
  if (!p8_plan(pl, cd, KD, G, want_stats, workspace != nullptr)) return 0;
  *handled = true;
  if (dry) return want_stats ? pl.stat_rows : 0;
  P8Args& a = pl.args;
  a.X = x; a.Wp = wp; a.Y = y; a.bias = bias; a.accumulate = accumulate; a.slab_stride = 0;
  a.st = StatsArg{nullptr, 0, 1, cd.Cout, G > 0 ? G : 1};
  if (want_stats && stat_partial && pl.stat_rows > 0) { a.st.partial = stat_partial; a.st.rows = pl.stat_rows; a.st.tiles_per_group = a.items_per_group; }
  const long long n = (long long)cd.N * cd.D * cd.H * cd.W * cd.Cout;
  if (pl.ksplit > 1) { a.Y = (float*)workspace; a.bias = nullptr; a.accumulate = 0; a.slab_stride = n; }
  switch (pl.cfg) {
    case 0: p8_launch<3, P8_BRICK, 8, 1, 2, 1, 4, 27, 1, 0>(pl, s); break;
    case 1: p8_launch<3, P8_BRICK, 4, 2, 4, 1, 4, 9, 2, 0>(pl, s); break;
    case 2: p8_launch<3, P8_BRICK, 4, 2, 2, 2, 2, 9, 2, 0>(pl, s); break;
    case 3: p8_launch<3, P8_FLAT, 4, 2, 1, 1, 0, 27, 2, kFlatAvMax>(pl, s); break;
    case 4: p8_launch<1, P8_FLAT, 4, 2, 1, 1, 0, 9, 2, kFlatAvMax>(pl, s); break;
    default: set_error("p8_fwd: bad plan"); return BCP_EUNSUP;
  }
  if (pl.ksplit > 1)
    hipLaunchKernelGGL(k_p8_sum_slabs, dim3((int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, s, (const float*)workspace,
                       pl.ksplit, n, cd.Cout, bias, y, accumulate);
  return a.st.partial ? pl.stat_rows : 0;
}

int p8_wgrad(const float*, const float*, float*, const ConvDims&, int, int, void*, hipStream_t, bool* handled) {
  *handled = false;
  return 0;
}
size_t p8_wgrad_workspace_bytes(const ConvDims&, int) { return 0; }

}  // namespace bcp
