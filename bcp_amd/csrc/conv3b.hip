// bcp_amd/csrc/conv3b.hip -- 3x3x3 / 3x3 convolution (forward and dgrad) with fp32 numerics on the BF16 matrix pipe.
//
// v_mfma_f32_16x16x4_f32 issues once per 32 cycles per SIMD and -- measured on gfx950 (tools/probe/overlap_probe.hip) -- shares
// its issue time with every VALU / LDS / VMEM instruction of the SIMD, so the fp32 kernels of conv3.hip stop near
// 100-110 TFLOP/s.  Here both operands are split into THREE bf16 pieces when they enter the LDS (8 + 8 + 8 mantissa bits:
// x = p0 + p1 + p2 up to 2^-26 |x|) and the tap loop issues six v_mfma_f32_16x16x32_bf16 per K = 32 block
//     a0 b0 + a0 b1 + a1 b0 + a0 b2 + a1 b1 + a2 b0        (dropped cross terms < 2^-24 of the product)
// with fp32 accumulation inside the matrix core: bf16 x bf16 products are exact in fp32, so the result differs from the fp32-MFMA
// kernels only by the summation order and the dropped 2^-24 terms -- fp32-equivalent (tests/kernel_checks.py check_conv3_b6:
// error vs fp64 within 3x of the fp32 kernel's).  Six such MFMAs take ~102 cycles against 256 for the eight 16x16x4_f32 they
// replace, and the bf16 MFMA leaves issue slots for the LDS reads / VALU around it.
//
// One MFMA covers TWO taps x 16 input channels (lanes lg 0-1: tap A, lanes lg 2-3: tap B); an odd tap count is padded with a
// zero-weight tap.  Workgroup = 256 threads = 4 waves along M, tile TD x TH x TW voxels (M = 64 * MT) x 16 * NT channels;
// LDS: the halo of one 16-channel chunk as three bf16 planes [HV][16] (32-byte rows; the MFMA rows of an m-tile are permuted so
// that the 16-byte fragment reads are conflict-free, see b6_row) and a double-buffered weight stage of SP tap pairs
// [piece][pair][cout][32 k] copied from the pre-split bf16 pack the weight packer writes behind the fp32 pack.  Two workgroups per
// CU: one computes while the other refills (the halo split is ~22 VALU per float4: v_cvt_pk_bf16_f32).
//
// Kernel families in this file (which one serves a shape: b6_fwd at the end; DESIGN.md sections 3a / 3b):
//   k_c3d  weight fragments straight from global memory, no barrier in the tap loop, fragment stream; 32-channel slabs on 256-voxel
//          tiles, and -- persistent, next tile's halo under the current tile's MFMAs -- the 16 -> 16 layers (3-D and 2-D)
//   k_c3p  64-voxel x 64-channel tiles as an LDS-DMA software pipeline (round 3): weight ring of three slots filled by
//          global_load_lds three stages ahead, next stage's fragments into a second register set between the MFMAs
//   k_c3q  the same pipeline on FLAT 64-voxel tiles for the deep levels (14x14x10, 7x7x5, 12^3, 6^3 ...), split-K
//   k_c3b / k_c3h / k_c3f  the register-staged predecessors (2-D instances of k_c3b are still the U-Net's 64-channel-slab kernel; the
//          others are kept behind conv3_b6_pipe = 0 and as the bit-identity twins of the pipelines in the tests)
// Reference ops: nn.Conv3d(k=3,pad=1) networks/VNet.py:17, nn.Conv2d(k=3,pad=1) networks/unet.py:19-25 and their backward.
#include "conv3_defs.h"
#include "../../include/bcp_hip.h"

// measurement only (tools/ablate_b6.sh): compile parts of k_c3b out -- 1 no epilogue stores, 2 no halo fetch / split, 4 no weight
// stages, 8 no MFMAs, 16 no stage barriers, 32 fragment reads hoisted out of the stage loop.  The product is built with 0.
#ifndef B6_ABLATE
#define B6_ABLATE 0
#endif
// measurement only (tools/sessions/r04_s14.sh): parts of the backward-statistics epilogue (MODE 2) off -- 1 no re-read of y (the value just
// computed stands in), 2 no statistics arithmetic, 4 fp32 instead of fp64 accumulation.  The product is built with 0.
#ifndef BCP_BW_ABLATE
#define BCP_BW_ABLATE 0
#endif
// k_c3d: A fragments of a tap pair as a prefetched stream in a fixed order (1, the product) or all requested at the top of the pair (0: round 2)
#ifndef BCP_C3D_STREAM
#define BCP_C3D_STREAM 1
#endif
#ifndef BCP_C3D_FD
#define BCP_C3D_FD 3        // prefetch distance of the stream in fragments (4 costs the 32-slab instance its second workgroup per CU)
#endif
#ifndef BCP_C3D_MP
#define BCP_C3D_MP 2        // m-tiles per MFMA group: consecutive MFMAs into one accumulator are MP * NT apart
#endif

// measurement only (-DBCP_TS_DEBUG=1, tools/ts_probe.py): wave 0 of the first 64 workgroups of k_c3f stamps s_memtime at its phase
// boundaries into a device array read back with bcp_debug_ts -- where does a deep-level workgroup's life go?
#ifndef BCP_TS_DEBUG
#define BCP_TS_DEBUG 0
#endif
#if BCP_TS_DEBUG
__device__ unsigned long long bcp_ts_buf[64 * 64];
__device__ unsigned long long bcp_ts_span[8192 * 2];     // every workgroup of the launch: constant-clock start / end (slots 0 / 61)
#define BCP_TS(i) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 64 && (i) < 64) bcp_ts_buf[(blockIdx.x >> 3) * 64 + (i)] = __builtin_amdgcn_s_memtime(); \
    if (__builtin_constant_p(i) && threadIdx.x == 0 && ((i) == 0 || (i) == 61)) { const unsigned l_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); if (l_ < 8192) bcp_ts_span[l_ * 2 + ((i) == 61)] = wall_clock64(); } } while (0)
// (slots 59 / 62: the 100 MHz constant clock at the same two points as slots 0 / 61 -> the shader clock the kernel actually ran at)
#define BCP_TSR(i) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 64 && (i) < 64) bcp_ts_buf[(blockIdx.x >> 3) * 64 + (i)] = wall_clock64(); } while (0)
extern "C" int bcp_debug_ts(unsigned long long* out_host) { return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(bcp_ts_buf), sizeof(bcp_ts_buf)); }
extern "C" int bcp_debug_ts_span(unsigned long long* out_host) { return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(bcp_ts_span), sizeof(bcp_ts_span)); }
extern "C" int bcp_debug_ts_clear() { static unsigned long long z[8192 * 2]; hipMemcpyToSymbol(HIP_SYMBOL(bcp_ts_buf), z, sizeof(bcp_ts_buf)); return (int)hipMemcpyToSymbol(HIP_SYMBOL(bcp_ts_span), z, sizeof(bcp_ts_span)); }
#else
#define BCP_TS(i) ((void)0)
#define BCP_TSR(i) ((void)0)
#endif

namespace bcp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// operand planes: PL = 3 -> three bf16 pieces, six MFMAs per K block; PL = 2 -> two pre-scaled fp16 pieces, three MFMAs (conv3_defs.h)
template <int PL> struct Pipe;
template <> struct Pipe<3> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ void split(const float4& v, float, unsigned short* base, int plane_stride) { split_store4(v, base, plane_stride); }
  static __device__ __forceinline__ long long pack_off(int T, int K16, int N16) { return pack_off_bf16(T, K16, N16); }
};
template <> struct Pipe<2> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ void split(const float4& v, float s, unsigned short* base, int plane_stride) { split_store4_f16(v, s, base, plane_stride); }
  static __device__ __forceinline__ long long pack_off(int T, int K16, int N16) { return pack_off_f16(T, K16, N16); }
};
static constexpr int XSB = 16;   // bf16 elements per halo voxel row: 32-byte rows, no padding (see b6_row)

// ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
// (/opt/skills/guides/MI355X_MICROARCH.md, LDS table) -- i.e. MFMA rows {0-3, 12-15} of one k half together with rows {4-11} of the
// other.  With 32-byte voxel rows a group is conflict-free when the halo row indices of each of the two row sets are distinct
// mod 8: b6_row maps MFMA row i of a 16-voxel m-tile to the tile voxel (w fastest) so that each set is ONE run of 8 voxels along
// W (TW = 8), or two runs of 4 two H-rows apart (TW = 4; halo rows 6 apart: 0-3 and 12-15 mod 8); TW = 16 needs no permutation.
// (The natural order costs 8-12 LDS cycles per read instead of 4: measured, the tap loop was LDS-bound at 2x its MFMA time.)
template <int TW>
__device__ __forceinline__ constexpr int b6_row(int i) {
  return TW >= 16 ? i : TW == 8 ? (i < 4 ? i : (i >= 12 ? i - 8 : i + 4)) : (i < 8 ? i : (i < 12 ? i + 4 : i - 4));
}

// epilogue shared by the bf16-pipe forward kernels: lane (li, lg) holds voxel b6_row(li) of each m-tile, channels lg*4 .. lg*4+3 of
// each 16-channel n-tile (D = W^T-tile x X-tile), so a lane stores 16 bytes per (m-tile, n-tile)
// (MTv / wv: m-tiles per wave and the wave's position along m when the waves of a workgroup also split the channels -- k_c3h)
// BW: the backward-statistics epilogue (MODE 2) exists only in the instances that carry it -- its parameter registers cost the persistent
// 16-channel kernel and the 2-D 64-channel-slab kernel their second workgroup per CU when every instance compiled it in (round 3,
// measured: 171 -> 310 us in the step)
template <class TL, int TD, int TH, int TW, int NT, int MTv = TL::MT, bool BW = false>
__device__ __forceinline__ void b6_store_tile(f32x4 (&acc)[MTv][NT], float* __restrict__ Y, const float* __restrict__ bias, const ConvDims& cd,
                                              int n, int d0, int h0, int w0, int cout0, int accumulate, bool want_stats,
                                              double (&s1)[NT][4], double (&s2)[NT][4], int wv = -1, const StatsArg* sb = nullptr, int gg = 0,
                                              float osc = 1.f /* power of two undoing the fp16 pre-scales (PL = 2); 1: exact no-op */) {
  constexpr int MT = MTv, CT = NT * 16;
  const int lane = threadIdx.x & 63, wave = wv >= 0 ? wv : (int)(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const bool full = cout0 + CT <= cd.Cout && (cd.Cout & 3) == 0 && d0 + TD <= cd.D && h0 + TH <= cd.H && w0 + TW <= cd.W;   // uniform
  const long long tile_base = ((((long long)n * cd.D + d0) * cd.H + h0) * cd.W + w0) * cd.Cout;
  float bv[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cout0 + nt * 16 + lg * 4 + r;
      bv[nt][r] = (bias && co < cd.Cout) ? bias[co] : 0.f;
    }
  // MODE 2 (backward statistics, see StatsArg): this lane's channels of the consumer norm layer's statistics rows of group gg
  const bool bwd = BW && want_stats && sb && sb->by;
  float pmu[BW ? NT : 1][4], prs[BW ? NT : 1][4], psc[BW ? NT : 1][4], psh[BW ? NT : 1][4];
  if (BW && bwd) {
    const long long GC = (long long)sb->G * sb->C;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cout0 + nt * 16 + lg * 4 + r;
        const long long i = (long long)gg * sb->C + (co < cd.Cout ? co : 0);
        pmu[nt][r] = sb->bstats[i]; prs[nt][r] = sb->bstats[GC + i]; psc[nt][r] = sb->bstats[2 * GC + i]; psh[nt][r] = sb->bstats[3 * GC + i];
      }
  }
  auto rows = [&](auto mode_tag, auto acc_tag) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool ACCUM = decltype(acc_tag)::value;
    // dz = da * act'(z), xhat = (y - mean) * rstd -- k_col_partial<1>'s arithmetic on the value just stored
    auto bstat = [&](int nt, int r, float v, float yv) __attribute__((always_inline)) {
      if (!BW) return;
      const int pn = BW ? nt : 0;
      if (BCP_BW_ABLATE & 2) return;
      const float z = (yv - pmu[pn][r]) * psc[pn][r] + psh[pn][r];
      const float g1 = v * act_grad(z, sb->bact);
      const float xh = (yv - pmu[pn][r]) * prs[pn][r];
      if (BCP_BW_ABLATE & 4) { s1[nt][r] = (double)((float)s1[nt][r] + g1); s2[nt][r] = (double)((float)s2[nt][r] + g1 * xh); return; }
      s1[nt][r] += (double)g1;
      s2[nt][r] += (double)g1 * (double)xh;
    };
    // (MODE 2, full tiles: ALL of the lane's y values are requested before the first store -- one exposed round trip per tile, not MT;
    //  the fragment registers of the tap loop are dead here, so the MT * NT float4 cost no occupancy)
    float4 yball[MODE == 2 ? MT : 1][NT];
    if (MODE == 2 && full) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = (wave * MT + mt) * 16 + b6_row<TW>(li);
        const int tw = m % TW, th = (m / TW) % TH, td = m / (TW * TH);
        const long long eoff = tile_base + (unsigned)(((td * cd.H + th) * cd.W + tw) * cd.Cout) + cout0 + lg * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) yball[MODE == 2 ? mt : 0][nt] = (BCP_BW_ABLATE & 1) ? make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]) : ld4(sb->by + eoff + nt * 16);
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = (wave * MT + mt) * 16 + b6_row<TW>(li);
      const int tw = m % TW, th = (m / TW) % TH, td = m / (TW * TH);
      const int d = d0 + td, h = h0 + th, w = w0 + tw;
      const long long eoff = tile_base + (unsigned)(((td * cd.H + th) * cd.W + tw) * cd.Cout) + cout0 + lg * 4;
      float* yrow = Y + eoff;
      if (full) {
        float4 (&yb)[NT] = yball[MODE == 2 ? mt : 0];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float4 v = make_float4(acc[mt][nt][0] * osc + bv[nt][0], acc[mt][nt][1] * osc + bv[nt][1], acc[mt][nt][2] * osc + bv[nt][2], acc[mt][nt][3] * osc + bv[nt][3]);
          if (ACCUM) { const float4 o = ld4(yrow + nt * 16); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
          if (!(B6_ABLATE & 1) || v.x == 1.2345e-30f) st4(yrow + nt * 16, v);
          if (MODE == 2) {
            bstat(nt, 0, v.x, yb[nt].x); bstat(nt, 1, v.y, yb[nt].y); bstat(nt, 2, v.z, yb[nt].z); bstat(nt, 3, v.w, yb[nt].w);
          } else {
            stat_add<MODE>(s1[nt][0], s2[nt][0], v.x); stat_add<MODE>(s1[nt][1], s2[nt][1], v.y);
            stat_add<MODE>(s1[nt][2], s2[nt][2], v.z); stat_add<MODE>(s1[nt][3], s2[nt][3], v.w);
          }
        }
      } else if (d < cd.D && h < cd.H && w < cd.W) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = cout0 + nt * 16 + lg * 4 + r;
            if (co < cd.Cout) {
              float v = acc[mt][nt][r] * osc + bv[nt][r];
              if (ACCUM) v += yrow[nt * 16 + r];
              if (!(B6_ABLATE & 1) || v == 1.2345e-30f) yrow[nt * 16 + r] = v;
              if (MODE == 2) bstat(nt, r, v, sb->by[eoff + nt * 16 + r]);
              else stat_add<MODE>(s1[nt][r], s2[nt][r], v);
            }
          }
      }
    }
  };
  if (!want_stats) {
    if (accumulate) rows(std::integral_constant<int, 0>{}, std::true_type{});
    else rows(std::integral_constant<int, 0>{}, std::false_type{});
  } else if (BW && bwd) rows(std::integral_constant<int, BW ? 2 : 1>{}, std::false_type{});
  else rows(std::integral_constant<int, 1>{}, std::false_type{});       // the statistics variants never accumulate (bcp_conv3_fwd_stats)
}

// one tile per workgroup (k_c3b): store + one statistics row per tile
template <class TL, int TD, int TH, int TW, int NT, bool BW = false>
__device__ __forceinline__ void b6_epilogue(f32x4 (&acc)[TL::MT][NT], float* __restrict__ Y, const float* __restrict__ bias, const ConvDims& cd,
                                            int n, int d0, int h0, int w0, int cout0, int accumulate, const StatsArg& st, double* Ss, int bx,
                                            float osc = 1.f) {
  double s1[NT][4], s2[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.0; s2[nt][r] = 0.0; }
  b6_store_tile<TL, TD, TH, TW, NT, TL::MT, BW>(acc, Y, bias, cd, n, d0, h0, w0, cout0, accumulate, st.partial != nullptr, s1, s2, -1, &st,
                                                st.partial ? bx / st.tiles_per_group : 0, osc);
  if (st.partial) {
    const int gg = bx / st.tiles_per_group, row = bx % st.tiles_per_group;
    BCP_LDS_BARRIER();                           // the scratch below aliases nothing, but waves may still be in the last stage
    stats_flush_t<NT>(s1, s2, Ss, st.partial + ((long long)gg * st.rows + row) * st.C * 2, cout0, cd.Cout);
  }
}

template <int KD, int TD, int TH, int TW, int NT, int SP, bool BW = false, int PL = 3>
__global__ __launch_bounds__(256) void k_c3b(const float* __restrict__ X, const float* __restrict__ Wp, const float* __restrict__ bias,
                                             float* __restrict__ Y, ConvDims cd, int accumulate, StatsArg st) {
  using TL = Tile<KD, TD, TH, TW>;
  constexpr int MT = TL::MT, T = TL::T, TP = (T + 1) / 2, CT = NT * 16;
  constexpr int S = ((TP + SP - 1) / SP + 1) & ~1;             // weight stages per cin chunk, even (2-D: one all-zero pad stage)
  static_assert(S >= 4, "the stage pipeline needs four weight stages per chunk");
  constexpr int XPLANE = TL::HV * XSB;                         // bf16 elements per halo piece plane
  constexpr int WPLANE = SP * CT * 32;                         // bf16 elements per weight piece plane of one stage
  constexpr int WSTAGE = PL * WPLANE;                          // one stage buffer
  constexpr int NW4 = (SP * 4 * PL * CT + 255) / 256;          // 16-byte pieces of a stage per thread
  using HF = HaloFetch<TL>;
  using PP = Pipe<PL>;
  using frag_t = typename PP::frag;

  HIP_DYNAMIC_SHARED(float4, smem4)
  unsigned short* Xb = reinterpret_cast<unsigned short*>(smem4);   // [PL][HV][XSB]
  unsigned short* Wb = Xb + PL * XPLANE;                           // [2][PL][SP][CT][32]
  double* Ss = reinterpret_cast<double*>(Wb + 2 * WSTAGE);         // [4][CT][2] statistics scratch

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bx = cd.xcd ? xcd_tile(blockIdx.x, gridDim.x, gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) : (int)blockIdx.x;   // tile of this workgroup
  int n, d0, h0, w0;
  tile_origin(cd, bx, TD, TH, TW, n, d0, h0, w0);
  const int cout0 = blockIdx.y * CT;
  const int cin4 = cd.Cin16 >> 2;

  int voff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) voff[mt] = TL::voff((wave * MT + mt) * 16 + b6_row<TW>(li)) * XSB + (lg & 1) * 8;
  // weight rows [cout][32 k] = 64 B: the 16-byte k quarter q of cout c sits at position q ^ (c & 8 ? 2 : 0) -- conflict-free fragment reads
  const int woff = li * 32 + ((lg ^ ((li & 8) ? 2 : 0)) * 8);
  HF hf;
  hf.init(cd, reinterpret_cast<float*>(smem4));      // (its fp32 LDS slot is not used: the stash below writes the bf16 planes)

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // split-K: blockIdx.z owns a contiguous range of cin chunks and writes its own partial slab (summed by k_sum_slabs)
  const int nchunks = cd.Cin16 >> 4;
  const int c_begin = (int)((long long)nchunks * blockIdx.z / gridDim.z), c_end = (int)((long long)nchunks * (blockIdx.z + 1) / gridDim.z);
  Y += (long long)blockIdx.z * cd.N * cd.D * cd.H * cd.W * cd.Cout;

  // stage `sg` of chunk `cc`: tap pairs sg*SP .. sg*SP+SP-1 of the pre-split pack Wb16[chunk][pair][piece][Cout16][32] (bf16, behind the
  // fp32 pack: bcp_conv3_packed_weight_floats); 16-byte piece q of the stage = (pair, piece, cout, k quarter)
  const unsigned short* Wb16 = reinterpret_cast<const unsigned short*>(Wp + PP::pack_off(T, cd.Cin16, cd.Cout16));
  float xsc = 1.f, osc = 1.f;          // PL = 2: power-of-two pre-scales (k_c3d)
  if (PL == 2) {
    const int ex = f16_scale_exp(amax_read(cd.xamax)), ew = f16_scale_exp(Wp[pack_off_hdr(T, cd.Cin16, cd.Cout16)]);
    xsc = ldexpf(1.f, ex);
    osc = ldexpf(1.f, -(ex + ew));
  }
  // (branch-free: every thread loads -- slots past the stage re-read its first pieces, pairs past TP re-read the last pair -- and
  //  wstash drops / zeroes what is not wanted; a predicated load would cost a vmcnt(0) per stage, see fetch_nb)
  // (one tap pair per stage: ONE uniform base per stage + a per-thread 32-bit offset fixed for the whole kernel; see k_c3h)
  unsigned wq[NW4];
#pragma unroll
  for (int u = 0; u < NW4; ++u) {
    const int q = (threadIdx.x + u * 256) % (SP * 4 * PL * CT);
    const int kq = q & 3, co = (q >> 2) % CT, sp = (q / (4 * CT)) % PL;
    wq[u] = (unsigned)((sp * cd.Cout16 + cout0 + co) * 32 + kq * 8);
  }
  auto wfetch = [&](int cc, int sg, float4 (&wpre)[NW4]) __attribute__((always_inline)) {
    if constexpr (SP == 1) {
      const unsigned short* wst = Wb16 + (long long)(cc * TP + (sg < TP ? sg : TP - 1)) * PL * cd.Cout16 * 32;      // uniform
#pragma unroll
      for (int u = 0; u < NW4; ++u) wpre[u] = *reinterpret_cast<const float4*>(wst + wq[u]);
    } else {
#pragma unroll
      for (int u = 0; u < NW4; ++u) {
        const int q = (threadIdx.x + u * 256) % (SP * 4 * PL * CT);
        const int pr = q / (4 * PL * CT);
        const int tp = sg * SP + pr < TP ? sg * SP + pr : TP - 1;
        wpre[u] = *reinterpret_cast<const float4*>(Wb16 + (long long)(cc * TP + tp) * PL * cd.Cout16 * 32 + wq[u]);
      }
    }
  };
  auto wstash = [&](unsigned short* Wbuf, int sg, const float4 (&wpre)[NW4]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NW4; ++u) {
      const int q = threadIdx.x + u * 256;
      const int kq = q & 3, co = (q >> 2) % CT, sp = (q / (4 * CT)) % PL, pr = q / (4 * PL * CT);
      const float4 v = (sg * SP + pr < TP) ? wpre[u] : make_float4(0.f, 0.f, 0.f, 0.f);       // pad pairs: zero weights
      if ((SP * 4 * PL * CT) % 256 == 0 || q < SP * 4 * PL * CT) *reinterpret_cast<float4*>(Wbuf + sp * WPLANE + (pr * CT + co) * 32 + ((kq ^ ((co & 8) ? 2 : 0)) * 8)) = v;
    }
  };
  unsigned hvm = 0;                                  // validity bits of the halo rows in flight (fetch_nb)
  auto hfetch = [&](int cc, float4 (&pre)[HF::NP]) __attribute__((always_inline)) {
    if (!(B6_ABLATE & 2)) hvm = hf.fetch_nb(X, cd, n, d0, h0, w0, cc, pre);
  };
  auto hstash = [&](const float4 (&pre)[HF::NP]) __attribute__((always_inline)) {
    if (B6_ABLATE & 2) return;
#pragma unroll
    for (int u = 0; u < HF::NP; ++u)
      if (hf.act && u * HF::RPP + hf.r0 < HF::HR) {
        const float4 v = ((hvm >> u) & 1u) ? pre[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        PP::split(v, xsc, Xb + ((u * HF::RPP + hf.r0) * TL::HW + hf.hw) * XSB + hf.part * 4, XPLANE);
      }
  };

  // Weight stage sg of a chunk lives in LDS buffer sg & 1 (S is even).  Two stages are in flight in registers (a stage is ~0.35 us
  // of MFMAs -- less than an L2 round trip, so a one-stage prefetch distance stalled every stage): register set W1 carries the odd
  // stages, W0 the even ones; a set is refilled with the stage three ahead right after its own stage went to the LDS, so no
  // register is ever copied.  NOTHING in the loop is a conditional load: hipcc's wait-count pass does not follow paths and answers
  // a predicated load (or a rotating register copy) with vmcnt(0) -- a full round trip per stage.  Stages past the end of the
  // block re-read its last stage (never stashed); the next chunk's halo is fetched between two copies of the stage loop.
  // Barriers inside the loop order LDS traffic only (BCP_LDS_BARRIER): the global loads stay in flight across them.
  auto wfetch_at = [&](int cc, int sg, float4 (&wpre)[NW4]) __attribute__((always_inline)) {     // stage sg (may run past S) of chunk cc
    while (sg >= S) { sg -= S; ++cc; }      // (S = 4 for the 2-D two-pair stages: a five-stage lookahead can cross TWO chunk ends)
    if (cc >= c_end) { cc = c_end - 1; sg = S - 1; }
    if (!(B6_ABLATE & 4)) wfetch(cc, sg, wpre);
  };
  constexpr int HPF = S - 4;                         // even stage of a chunk in front of which the next chunk's halo is fetched
  // (since late round 2: FOUR stages in flight, chunk body fully unrolled with static register sets -- see k_c3h)
  float4 hpre[HF::NP], W[4][NW4];
  hfetch(c_begin, hpre);
  wfetch_at(c_begin, 0, W[0]);
  wfetch_at(c_begin, 1, W[1]);
  wfetch_at(c_begin, 2, W[2]);
  wfetch_at(c_begin, 3, W[3]);
  hstash(hpre);
  if (!(B6_ABLATE & 4)) wstash(Wb, 0, W[0]);
  wfetch_at(c_begin, 4, W[0]);
  BCP_LDS_BARRIER();

  frag_t abl_a[MT][PL], abl_b[NT][PL];               // (measurement only, B6_ABLATE & 32)
  if (B6_ABLATE & 32) {
#pragma unroll
    for (int s = 0; s < PL; ++s) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) abl_a[mt][s] = *reinterpret_cast<const frag_t*>(Xb + s * XPLANE + voff[mt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) abl_b[nt][s] = *reinterpret_cast<const frag_t*>(Wb + s * WPLANE + nt * 16 * 32 + woff);
    }
  }
  // (order within a stage: fragment reads, the NEXT stage's weights to the other buffer -- last read one stage ago, every wave has passed
  //  the barrier that ended that stage -- and the refill of their registers, THEN the MFMAs: the wave reaches the barrier behind its
  //  last MFMA; see k_c3h)
  auto stage = [&](int cc, int sg, float4 (&Wn)[NW4]) __attribute__((always_inline)) {
    const unsigned short* Wc = Wb + (sg & 1) * WSTAGE;
    frag_t a[SP][MT][PL], b[SP][NT][PL];
#pragma unroll
    for (int pr = 0; pr < SP; ++pr) {
      const int tp = sg * SP + pr;
      if (tp < TP && !(B6_ABLATE & 8)) {     // uniform
        // lanes lg 0-1 (k 0..15) carry tap 2*tp, lanes lg 2-3 (k 16..31) tap 2*tp+1 (the pad tap reads tap T-1's voxels against zero weights)
        const int t0 = 2 * tp, t1 = 2 * tp + 1 < T ? 2 * tp + 1 : T - 1;
        const int tA = ((t0 / 9) * TL::HH + (t0 / 3) % 3) * TL::HW + t0 % 3, tB = ((t1 / 9) * TL::HH + (t1 / 3) % 3) * TL::HW + t1 % 3;
        const int toff = ((lg >> 1) ? tB : tA) * XSB;
#pragma unroll
        for (int s = 0; s < PL; ++s) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a[pr][mt][s] = (B6_ABLATE & 32) ? abl_a[mt][s] : *reinterpret_cast<const frag_t*>(Xb + s * XPLANE + voff[mt] + toff);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) b[pr][nt][s] = (B6_ABLATE & 32) ? abl_b[nt][s] : *reinterpret_cast<const frag_t*>(Wc + s * WPLANE + (pr * CT + nt * 16) * 32 + woff);
        }
      }
    }
    if (!(B6_ABLATE & 4) && (sg + 1 < S || cc + 1 < c_end)) wstash(Wb + ((sg + 1) & 1) * WSTAGE, sg + 1 < S ? sg + 1 : 0, Wn);
    wfetch_at(cc, sg + 5, Wn);
#pragma unroll
    for (int pr = 0; pr < SP; ++pr) {
      const int tp = sg * SP + pr;
      if (tp < TP && !(B6_ABLATE & 8)) {     // uniform
        // D = W^T-tile x X-tile: rows = output channels, columns = voxels, so that a lane ends up with FOUR CONSECUTIVE channels of
        // one voxel (one 16-byte store).  Smallest terms first; the six products of an accumulator are spread over the MT*NT accumulators.
#define BCP_B6(I, J)                                                                                            \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)            \
      acc[mt][nt] = PP::mfma(b[pr][nt][J], a[pr][mt][I], acc[mt][nt]);
        if constexpr (PL == 3) { BCP_B6(2, 0) BCP_B6(1, 1) BCP_B6(0, 2) BCP_B6(1, 0) BCP_B6(0, 1) BCP_B6(0, 0) }
        else { BCP_B6(1, 0) BCP_B6(0, 1) BCP_B6(0, 0) }
#undef BCP_B6
      }
    }
    if (sg + 1 < S && !(B6_ABLATE & 16)) BCP_LDS_BARRIER();               // (a chunk boundary brings its own)
  };
  auto chunk = [&](int cc, auto phase_tag) __attribute__((always_inline)) {
    constexpr int PH = decltype(phase_tag)::value;
    if (cc > c_begin) {
      BCP_LDS_BARRIER();                             // every wave is done with the previous chunk's halo planes
      hstash(hpre);
      BCP_LDS_BARRIER();
    }
#pragma unroll
    for (int sg = 0; sg < S; ++sg) {
      if (sg == HPF) hfetch(cc + 1 < c_end ? cc + 1 : cc, hpre);      // (compile-time position; the last chunk re-reads its own halo: no conditional load)
      stage(cc, sg, W[(PH + sg + 1) & 3]);
    }
  };
#pragma unroll 1
  for (int cc = c_begin; cc < c_end; cc += 2) {
    chunk(cc, std::integral_constant<int, 0>{});
    if (cc + 1 < c_end) chunk(cc + 1, std::integral_constant<int, (S & 3)>{});
  }

  b6_epilogue<TL, TD, TH, TW, NT, BW>(acc, Y, bias, cd, n, d0, h0, w0, cout0, accumulate, st, Ss, bx, osc);
}

// ------------------------------------------------------------------------------------------------
// k_c3b for the 64-voxel x 64-channel workgroup tile with the four waves arranged 2 x 2: wave (wm, wn) owns m-tiles 2 wm, 2 wm + 1 and
// n-tiles 2 wn, 2 wn + 1.  Same 24 MFMAs per wave and tap pair as k_c3b's 1 x 4 arrangement, but 6 + 6 instead of 3 + 12
// ds_read_b128 of fragments: the LDS pipe of these kernels is throughput-bound at the mid / deep levels (15 reads per wave and stage
// x 8 waves per CU = 960 LDS cycles against 768 cycles of MFMA), so reads per MFMA are what counts.
// One tap pair per stage, two weight buffers, everything else as k_c3b.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stats_flush_22(double (&s1)[2][4], double (&s2)[2][4], double* __restrict__ Ss /* [4 waves][32][2] */,
                                               double* __restrict__ dst_row, int cout0, int Cout) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double a = s1[nt][r], b = s2[nt][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
      if (li == 0) { Ss[(wave * 32 + nt * 16 + lg * 4 + r) * 2] = a; Ss[(wave * 32 + nt * 16 + lg * 4 + r) * 2 + 1] = b; }
    }
  __syncthreads();
  if ((int)threadIdx.x < 64 && cout0 + (int)threadIdx.x < Cout) {
    const int c = threadIdx.x, wn = c >> 5, cl = c & 31;          // channel c belongs to waves (wm = 0, wn) and (wm = 1, wn): wave = wn * 2 + wm
    const double a = Ss[((wn * 2) * 32 + cl) * 2] + Ss[((wn * 2 + 1) * 32 + cl) * 2];
    const double b = Ss[((wn * 2) * 32 + cl) * 2 + 1] + Ss[((wn * 2 + 1) * 32 + cl) * 2 + 1];
    dst_row[(cout0 + c) * 2] = a;
    dst_row[(cout0 + c) * 2 + 1] = b;
  }
  __syncthreads();
}

template <int KD, int TD, int TH, int TW, bool BW = false>
__global__ __launch_bounds__(256) void k_c3h(const float* __restrict__ X, const float* __restrict__ Wp, const float* __restrict__ bias,
                                             float* __restrict__ Y, ConvDims cd, int accumulate, StatsArg st) {
  using TL = Tile<KD, TD, TH, TW>;
  static_assert(TL::M == 64, "k_c3h: 64-voxel tiles (four m-tiles: two per wave)");
  constexpr int T = TL::T, TP = (T + 1) / 2, CT = 64;
  constexpr int S = (TP + 1) & ~1;                             // weight stages per cin chunk, even (2-D: one all-zero pad stage)
  static_assert(S >= 4, "the stage pipeline needs four weight stages per chunk");
  constexpr int XPLANE = TL::HV * XSB;
  constexpr int WPLANE = CT * 32;
  constexpr int WSTAGE = 3 * WPLANE;
  constexpr int NW4 = (12 * CT + 255) / 256;
  using HF = HaloFetch<TL>;

  BCP_TS(0);
  BCP_TSR(59);
  HIP_DYNAMIC_SHARED(float4, smem4)
  unsigned short* Xb = reinterpret_cast<unsigned short*>(smem4);   // [3][HV][XSB]
  unsigned short* Wb = Xb + 3 * XPLANE;                            // [2][3][CT][32]
  double* Ss = reinterpret_cast<double*>(Wb + 2 * WSTAGE);         // [4][32][2] statistics scratch

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  const int bx = cd.xcd ? xcd_tile(blockIdx.x, gridDim.x, gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) : (int)blockIdx.x;   // tile of this workgroup
  int n, d0, h0, w0;
  tile_origin(cd, bx, TD, TH, TW, n, d0, h0, w0);
  const int cout0 = blockIdx.y * CT;

  int voff[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) voff[mt] = TL::voff((wm * 2 + mt) * 16 + b6_row<TW>(li)) * XSB + (lg & 1) * 8;
  const int woff = (wn * 32 + li) * 32 + ((lg ^ ((li & 8) ? 2 : 0)) * 8);      // this wave's two n-tiles start at channel wn * 32 of the slab
  HF hf;
  hf.init(cd, reinterpret_cast<float*>(smem4));

  f32x4 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = cd.Cin16 >> 4;
  const int c_begin = (int)((long long)nchunks * blockIdx.z / gridDim.z), c_end = (int)((long long)nchunks * (blockIdx.z + 1) / gridDim.z);
  Y += (long long)blockIdx.z * cd.N * cd.D * cd.H * cd.W * cd.Cout;

  const unsigned short* Wb16 = reinterpret_cast<const unsigned short*>(Wp + (long long)T * cd.Cin16 * cd.Cout16);
  // weight fetch of stage (cc, sg): ONE uniform base per stage (scalar registers) + a per-thread 32-bit offset fixed for the whole
  // kernel -- the per-load 64-bit index arithmetic this replaced was ~25 VALU instructions per stage on the critical path of the wave
  unsigned wq[NW4];
#pragma unroll
  for (int u = 0; u < NW4; ++u) {
    const int q = (threadIdx.x + u * 256) % (12 * CT);
    const int kq = q & 3, co = (q >> 2) % CT, sp = q / (4 * CT);
    wq[u] = (unsigned)((sp * cd.Cout16 + cout0 + co) * 32 + kq * 8);
  }
  auto wfetch_at = [&](int cc, int sg, float4 (&wpre)[NW4]) __attribute__((always_inline)) {
    while (sg >= S) { sg -= S; ++cc; }      // (S = 4 for the 2-D two-pair stages: a five-stage lookahead can cross TWO chunk ends)
    if (cc >= c_end) { cc = c_end - 1; sg = S - 1; }
    int tp = sg < TP ? sg : TP - 1;
    if (B6_ABLATE & 64) { tp = 0; cc = c_begin; }        // (measurement: every stage re-reads the first stage's weights -- cache-resident fetches)
    const unsigned short* wst = Wb16 + (long long)(cc * TP + tp) * 3 * cd.Cout16 * 32;      // uniform
#pragma unroll
    for (int u = 0; u < NW4; ++u) wpre[u] = *reinterpret_cast<const float4*>(wst + wq[u]);
  };
  auto wstash = [&](unsigned short* Wbuf, int sg, const float4 (&wpre)[NW4]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NW4; ++u) {
      const int q = threadIdx.x + u * 256;
      const int kq = q & 3, co = (q >> 2) % CT, sp = q / (4 * CT);
      const float4 v = (sg < TP) ? wpre[u] : make_float4(0.f, 0.f, 0.f, 0.f);       // pad pair: zero weights
      if ((12 * CT) % 256 == 0 || q < 12 * CT) *reinterpret_cast<float4*>(Wbuf + sp * WPLANE + co * 32 + ((kq ^ ((co & 8) ? 2 : 0)) * 8)) = v;
    }
  };
  unsigned hvm = 0;
  auto hfetch = [&](int cc, float4 (&pre)[HF::NP]) __attribute__((always_inline)) { hvm = hf.fetch_nb(X, cd, n, d0, h0, w0, cc, pre); };
  auto hstash = [&](const float4 (&pre)[HF::NP]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < HF::NP; ++u)
      if (hf.act && u * HF::RPP + hf.r0 < HF::HR) {
        const float4 v = ((hvm >> u) & 1u) ? pre[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        split_store4(v, Xb + ((u * HF::RPP + hf.r0) * TL::HW + hf.hw) * XSB + hf.part * 4, XPLANE);
      }
  };

  // FOUR weight stages in flight in registers (k_c3b: two): a stage is ~0.9 us here and its 12 KB come from L2 under the traffic of
  // ~500 workgroups -- with every stage re-reading stage 0's (cache-resident) weights the kernel ran 11 % faster, i.e. a two-stage
  // window stalls.  Set (g + 1) & 3 holds stage g + 1's weights while stage g runs (g = stage counted over the workgroup's chunks); it
  // goes to LDS buffer (g + 1) & 1 during stage g and is refilled with stage g + 5.  The chunk body is fully unrolled (static register
  // sets, compile-time tap offsets); S = 14 is not a multiple of 4, so chunks alternate between the two phases of the rotation.
  constexpr int HPF = S - 4;
  BCP_TS(1);
  float4 hpre[HF::NP], W[4][NW4];
  hfetch(c_begin, hpre);
  wfetch_at(c_begin, 0, W[0]);
  wfetch_at(c_begin, 1, W[1]);
  wfetch_at(c_begin, 2, W[2]);
  wfetch_at(c_begin, 3, W[3]);
  hstash(hpre);
  wstash(Wb, 0, W[0]);
  wfetch_at(c_begin, 4, W[0]);
  BCP_LDS_BARRIER();
  BCP_TS(2);

  // One stage.  Order within it (round 3, from the s_memtime stamps of tools/ts_probe.py: fragments + MFMAs 664 ticks, weight stash +
  // fetch issued AFTER them 436, barrier 128 -- a serial chain per wave of which only the MFMA part can hide behind the co-resident
  // workgroup's wave): the next stage's weights go to the other LDS buffer and the refill of their registers is issued BEFORE the
  // MFMAs (that buffer was last read in stage sg - 1, which every wave left through the barrier), so the wave reaches the next
  // barrier straight behind its last MFMA.
  auto stage = [&](int cc, int sg, float4 (&Wn)[NW4]) __attribute__((always_inline)) {
    const unsigned short* Wc = Wb + (sg & 1) * WSTAGE;
    bf16x8 a[2][3], b[2][3];
    if (sg < TP) {     // uniform
      const int t0 = 2 * sg, t1 = 2 * sg + 1 < T ? 2 * sg + 1 : T - 1;
      const int tA = ((t0 / 9) * TL::HH + (t0 / 3) % 3) * TL::HW + t0 % 3, tB = ((t1 / 9) * TL::HH + (t1 / 3) % 3) * TL::HW + t1 % 3;
      const int toff = ((lg >> 1) ? tB : tA) * XSB;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a[mt][s] = *reinterpret_cast<const bf16x8*>(Xb + s * XPLANE + voff[mt] + toff);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b[nt][s] = *reinterpret_cast<const bf16x8*>(Wc + s * WPLANE + nt * 16 * 32 + woff);
      }
    }
    if (sg + 1 < S || cc + 1 < c_end) wstash(Wb + ((sg + 1) & 1) * WSTAGE, sg + 1 < S ? sg + 1 : 0, Wn);
    wfetch_at(cc, sg + 5, Wn);
    if (sg < TP) {
#define BCP_B6(I, J)                                                                                            \
  _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)              \
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nt][J], a[mt][I], acc[mt][nt], 0, 0, 0);
      BCP_B6(2, 0) BCP_B6(1, 1) BCP_B6(0, 2) BCP_B6(1, 0) BCP_B6(0, 1) BCP_B6(0, 0)
#undef BCP_B6
    }
    if (sg + 1 < S) BCP_LDS_BARRIER();
    BCP_TS(3 + (cc - c_begin) * S + sg);
  };
  auto chunk = [&](int cc, auto phase_tag) __attribute__((always_inline)) {
    constexpr int PH = decltype(phase_tag)::value;
    if (cc > c_begin) {
      BCP_LDS_BARRIER();                             // every wave is done with the previous chunk's halo planes
      hstash(hpre);
      BCP_LDS_BARRIER();
    }
#pragma unroll
    for (int sg = 0; sg < S; ++sg) {
      if (sg == HPF) hfetch(cc + 1 < c_end ? cc + 1 : cc, hpre);      // (compile-time position: not a conditional load)
      stage(cc, sg, W[(PH + sg + 1) & 3]);
    }
  };
#pragma unroll 1
  for (int cc = c_begin; cc < c_end; cc += 2) {
    chunk(cc, std::integral_constant<int, 0>{});
    if (cc + 1 < c_end) chunk(cc + 1, std::integral_constant<int, (S & 3)>{});
  }

  BCP_TS(60);
  double s1[2][4], s2[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.0; s2[nt][r] = 0.0; }
  b6_store_tile<TL, TD, TH, TW, 2, 2, BW>(acc, Y, bias, cd, n, d0, h0, w0, cout0 + wn * 32, accumulate, st.partial != nullptr, s1, s2, wm, &st,
                                      st.partial ? bx / st.tiles_per_group : 0);
  if (st.partial) {
    const int gg = bx / st.tiles_per_group, row = bx % st.tiles_per_group;
    BCP_LDS_BARRIER();
    stats_flush_22(s1, s2, Ss, st.partial + ((long long)gg * st.rows + row) * st.C * 2, cout0, cd.Cout);
  }
  BCP_TS(61);
  BCP_TSR(62);
}

// ------------------------------------------------------------------------------------------------
// k_c3h as a SOFTWARE PIPELINE (round 3).  What the s_memtime stamps of k_c3h showed (tools/ts_probe.py, 64 -> 64 channels at 28 x 28 x
// 20 x 2): a workgroup alone on its CU needs ~810 ticks per stage for 384 cycles of MFMA per wave, two co-resident ones ~1000 and ~1600
// (the older one wins the arbitration), and the launch lasts as long as the slower of the pair -- each wave runs the chain
// [fragment reads -> wait -> 24 MFMAs -> weight stash -> barrier] serially and two waves per SIMD cannot fill each other's gaps.  Here
// the chain is cut:
//   * the weight stages come by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a ring of THREE 12 KB
//     slots, three stages ahead: the DMA of stage g + 3 is issued at the top of stage g into the slot stage g's fragments were read
//     from a stage earlier, and is waited for (counted vmcnt) in front of the barrier that ends stage g + 1;
//   * the fragments of stage g + 1 (halo planes + slot (g + 1) % 3) are read into a second register set at the top of stage g, so the
//     MFMAs of a stage start right behind the barrier with their operands in registers;
//   * the halo has two buffers: the next chunk's planes are written during the chunk's second-last stage, so a chunk boundary is an
//     ordinary stage (k_c3h: two extra barriers around the stash, ~1300 ticks).
// The LDS image of a slot is lane-linear (piece q of the stage at byte 16 q); the XOR swizzle of the k quarters that makes the
// fragment reads conflict-free is applied on the SOURCE address of the DMA (the read side keeps k_c3h's addresses).
// LDS: 2 x 20.25 KB halo + 3 x 12 KB weights + 2 KB = 78.5 KB -> two workgroups per CU.  Same MFMA sequence per accumulator as
// k_c3h: bit-identical results.
// ------------------------------------------------------------------------------------------------
template <int KD, int TD, int TH, int TW, bool BW = false, int PL = 3>
__global__ __launch_bounds__(256) void k_c3p(const float* __restrict__ X, const float* __restrict__ Wp, const float* __restrict__ bias,
                                             float* __restrict__ Y, ConvDims cd, int accumulate, StatsArg st) {
  using TL = Tile<KD, TD, TH, TW>;
  static_assert(TL::M == 64, "k_c3p: 64-voxel tiles (four m-tiles: two per wave)");
  constexpr int T = TL::T, TP = (T + 1) / 2, CT = 64;
  constexpr int S = TP;                                        // stages per cin chunk: 14 (3-D) / 5 (2-D); no pad stage -- with an odd count the
                                                               // fragment register sets swap roles from chunk to chunk (PAR below), which the
                                                               // statically unrolled chunk PAIR absorbs
  static_assert(S >= 5 && S <= 14, "k_c3p: 5 (3x3) or 14 (3x3x3) tap pairs per chunk");
  constexpr int XPLANE = TL::HV * XSB, XBUF = PL * XPLANE;     // 16-bit elements per halo plane / buffer
  constexpr int WPLANE = CT * 32, WSLOT = PL * WPLANE;         // one ring slot: PL planes x 64 rows x 64 B = 12 KB (8 KB with two fp16 planes)
  using PP = Pipe<PL>;
  using frag_t = typename PP::frag;
#ifndef BCP_C3P_HFS
#define BCP_C3P_HFS 1
#endif
  constexpr int HFS = BCP_C3P_HFS >= 0 ? BCP_C3P_HFS : S - 6, HSS = S - 2;     // stage of a chunk that fetches / stashes the next chunk's halo (fetch early: see k_c3d's HPF)
  using HF = HaloFetch<TL>;

  BCP_TS(0);
  BCP_TSR(59);
  HIP_DYNAMIC_SHARED(float4, smem4)
  unsigned short* Xb = reinterpret_cast<unsigned short*>(smem4);   // [2][PL][HV][XSB]
  unsigned short* Wr = Xb + 2 * XBUF;                              // [3 slots][PL][CT][32]
  double* Ss = reinterpret_cast<double*>(Wr + 3 * WSLOT);          // [4][32][2] statistics scratch

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  const int bx = cd.xcd ? xcd_tile(blockIdx.x, gridDim.x, gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) : (int)blockIdx.x;   // tile of this workgroup
  int n, d0, h0, w0;
  tile_origin(cd, bx, TD, TH, TW, n, d0, h0, w0);
  const int cout0 = blockIdx.y * CT;

  int voff[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) voff[mt] = TL::voff((wm * 2 + mt) * 16 + b6_row<TW>(li)) * XSB + (lg & 1) * 8;
  const int woff = (wn * 32 + li) * 32 + ((lg ^ ((li & 8) ? 2 : 0)) * 8);
  HF hf;
  hf.init(cd, reinterpret_cast<float*>(smem4));

  f32x4 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = cd.Cin16 >> 4;
  const int c_begin = (int)((long long)nchunks * blockIdx.z / gridDim.z), c_end = (int)((long long)nchunks * (blockIdx.z + 1) / gridDim.z);
  Y += (long long)blockIdx.z * cd.N * cd.D * cd.H * cd.W * cd.Cout;

  // DMA piece q = u * 256 + thread of a stage lands at byte 16 q of the slot = (plane u, row q >> 2 & 63, k quarter POSITION q & 3);
  // that position holds k quarter (q & 3) ^ (row & 8 ? 2 : 0): per-thread source byte offsets, fixed for the whole kernel
  const char* Wb16 = reinterpret_cast<const char*>(Wp + PP::pack_off(T, cd.Cin16, cd.Cout16));
  unsigned wq[3];
  {
    const int co = (threadIdx.x >> 2) & 63, kq = (threadIdx.x & 3) ^ ((co & 8) ? 2 : 0);
#pragma unroll
    for (int u = 0; u < 3; ++u) wq[u] = (unsigned)((((u < PL ? u : 0) * cd.Cout16 + cout0 + co) * 32 + kq * 8) * 2);
  }
  float xsc = 1.f, osc = 1.f;          // PL = 2: power-of-two pre-scales (k_c3d)
  if (PL == 2) {
    const int ex = f16_scale_exp(amax_read(cd.xamax)), ew = f16_scale_exp(Wp[pack_off_hdr(T, cd.Cin16, cd.Cout16)]);
    xsc = ldexpf(1.f, ex);
    osc = ldexpf(1.f, -(ex + ew));
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  char* const Wr_wave = reinterpret_cast<char*>(Wr) + wave_u * 1024;        // this wave's 1 KB of every plane
  auto wdma = [&](int cc, int sg, unsigned slot_bytes) __attribute__((always_inline)) {
    while (sg >= S) { sg -= S; ++cc; }
    if (cc >= c_end) { cc = c_end - 1; sg = S - 1; }                        // (past the end: re-read the last stage; never used)
    const char* wst = Wb16 + (long long)(cc * TP + sg) * PL * cd.Cout16 * 64;      // uniform
#pragma unroll
    for (int u = 0; u < PL; ++u) BCP_GLDS16(wst + wq[u], Wr_wave + slot_bytes + u * 4096);
  };
  unsigned hvm = 0;
  auto hfetch = [&](int cc, float4 (&pre)[HF::NP]) __attribute__((always_inline)) { hvm = hf.fetch_nb(X, cd, n, d0, h0, w0, cc, pre); };
  auto hstash = [&](int hb, const float4 (&pre)[HF::NP]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < HF::NP; ++u)
      if (hf.act && u * HF::RPP + hf.r0 < HF::HR) {
        const float4 v = ((hvm >> u) & 1u) ? pre[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        PP::split(v, xsc, Xb + hb * XBUF + ((u * HF::RPP + hf.r0) * TL::HW + hf.hw) * XSB + hf.part * 4, XPLANE);
      }
  };
  // fragments of stage sg (halo buffer hb, ring slot at element offset slot_el) -> one register set
  auto frag_read = [&](int hb, int sg, unsigned slot_el, frag_t (&a)[2][PL], frag_t (&b)[2][PL]) __attribute__((always_inline)) {
    const int t0 = 2 * sg, t1 = 2 * sg + 1 < T ? 2 * sg + 1 : T - 1;
    const int tA = ((t0 / 9) * TL::HH + (t0 / 3) % 3) * TL::HW + t0 % 3, tB = ((t1 / 9) * TL::HH + (t1 / 3) % 3) * TL::HW + t1 % 3;
    const int toff = ((lg >> 1) ? tB : tA) * XSB;
    const unsigned short* Xc = Xb + hb * XBUF;
    const unsigned short* Wc = Wr + slot_el;
#pragma unroll
    for (int s = 0; s < PL; ++s) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) a[mt][s] = *reinterpret_cast<const frag_t*>(Xc + s * XPLANE + voff[mt] + toff);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) b[nt][s] = *reinterpret_cast<const frag_t*>(Wc + s * WPLANE + nt * 16 * 32 + woff);
    }
  };

  BCP_TS(1);
  // prologue: halo of the first chunk, weight stages 0 .. 2, fragments of stage 0
  float4 hpre[HF::NP];
  frag_t fa[2][2][PL], fb[2][2][PL];
  unsigned sl0 = 0, sl1 = WSLOT * 2, sl2 = 2 * WSLOT * 2;      // ring slots (byte offsets) of stages g, g + 1, g + 2
  hfetch(c_begin, hpre);
  wdma(c_begin, 0, sl0);
  wdma(c_begin, 1, sl1);
  wdma(c_begin, 2, sl2);
  hstash(0, hpre);
  BCP_VM_LDS_BARRIER(0);
  frag_read(0, 0, sl0 >> 1, fa[0], fb[0]);
  BCP_LDS_BARRIER();                                           // every wave has its stage-0 fragments: slot 0 may be refilled
  BCP_TS(2);

  // One stage = 24 MFMAs (PL = 2: 12) on the current register set with the 15 (10) memory instructions of the pipeline (PL DMAs of stage
  // g + 3, 4 PL fragment reads of stage g + 1) issued BETWEEN them, one behind each of the first 15 MFMAs: an MFMA keeps the matrix pipe busy for 16
  // cycles while the wave issues the next instruction.  Issued in front of the MFMAs instead they cost the wave ~300 cycles per stage
  // (684 ticks per stage for a workgroup alone on its CU against 384 cycles of MFMA); hipcc left alone sinks the reads next to THEIR
  // MFMAs -- pulled up across the barrier -- and the pipeline is gone: the order is pinned with sched_barrier.
  auto stage = [&](int cc, auto hb_tag, auto sg_tag) __attribute__((always_inline)) {
    constexpr int HB = decltype(hb_tag)::value, sg = decltype(sg_tag)::value, PAR = (HB * S + sg) & 1;
    constexpr int NHB = sg + 1 < S ? HB : HB ^ 1, NSG = sg + 1 < S ? sg + 1 : 0;     // next stage (behind the last chunk: stale planes, never used)
    if (sg == HFS) hfetch(cc + 1 < c_end ? cc + 1 : cc, hpre);      // (compile-time position; the last chunk re-reads its own halo)
    if (sg == HSS) hstash(HB ^ 1, hpre);
    // uniform addresses of the stage's memory instructions
    int dcc = cc, dsg = sg + 3;
    while (dsg >= S) { dsg -= S; ++dcc; }
    const bool dma_on = sg + 3 < S || cc + 1 < c_end;                              // uniform: nothing to fetch behind the workgroup's last stage
    const char* wst = Wb16 + (long long)(dcc * TP + dsg) * PL * cd.Cout16 * 64;
    char* const wdst = Wr_wave + sl0;
    constexpr int t0 = 2 * NSG, t1 = 2 * NSG + 1 < T ? 2 * NSG + 1 : T - 1;
    constexpr int tA = ((t0 / 9) * TL::HH + (t0 / 3) % 3) * TL::HW + t0 % 3, tB = ((t1 / 9) * TL::HH + (t1 / 3) % 3) * TL::HW + t1 % 3;
    const int toff = ((lg >> 1) ? tB : tA) * XSB;
    const unsigned short* Xc = Xb + NHB * XBUF + toff;
    const unsigned short* Wc = Wr + (sl1 >> 1) + woff;
    __builtin_amdgcn_sched_barrier(0);
    // piece products, smallest terms first (as k_c3h); two planes: a1 b0, a0 b1, a0 b0
    constexpr int NPR = PL * (PL + 1) / 2, NMEM = 5 * PL;
    constexpr int PI[6] = {PL == 3 ? 2 : 1, PL == 3 ? 1 : 0, 0, 1, 0, 0}, PJ[6] = {0, 1, PL == 3 ? 2 : 0, 0, 1, 0};
#pragma unroll
    for (int k = 0; k < 4 * NPR; ++k) {
      const int pr = k >> 2, mt = (k >> 1) & 1, nt = k & 1;
      acc[mt][nt] = PP::mfma(fb[PAR][nt][PJ[pr]], fa[PAR][mt][PI[pr]], acc[mt][nt]);
      if (k < PL) { if (dma_on) BCP_GLDS16(wst + wq[k < 3 ? k : 0], wdst + k * 4096); }
      else if (k < NMEM) {
        const int r = k - PL, sp = r >> 2, w = r & 3;
        if (w < 2) fa[PAR ^ 1][w][sp] = *reinterpret_cast<const frag_t*>(Xc + sp * XPLANE + voff[w]);
        else fb[PAR ^ 1][w - 2][sp] = *reinterpret_cast<const frag_t*>(Wc + sp * WPLANE + (w - 2) * 16 * 32);
      }
      if (k < NMEM) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // The DMA issued a stage ago must have landed before anyone reads its slot at the top of the next stage: at most this stage's
    // three DMAs may still be in flight.  The halo loads of the fetch stage are younger than that DMA too, but they must NOT be
    // added to the count: vmcnt is in order among the LDS-DMAs, not between an LDS-DMA and an ordinary load -- halo loads that hit
    // the L2 retire in front of an older DMA that went to HBM, the counter drops to 3 + NP with the old DMA still in flight, and the
    // next stage reads a slot that has not landed (measured: wrong results exactly where the weight stream misses the caches -- first
    // call on cold caches, the 256-channel level; tools/diag/pipe_diag.py).  Counting only DMAs, the wait also covers the halo
    // loads of this stage (once per chunk; they have had the stage to arrive).
    if (dma_on) BCP_VM_LDS_BARRIER(PL); else BCP_VM_LDS_BARRIER(0);      // (no DMAs issued in this stage: the previous stage's are the youngest)
    const unsigned t = sl0; sl0 = sl1; sl1 = sl2; sl2 = t;
    BCP_TS(3 + (cc - c_begin) * S + sg);
  };
  auto chunk = [&](int cc, auto hb_tag) __attribute__((always_inline)) {
#define BCP_ST(I) if constexpr (I < S) stage(cc, hb_tag, std::integral_constant<int, I>{});
    BCP_ST(0) BCP_ST(1) BCP_ST(2) BCP_ST(3) BCP_ST(4) BCP_ST(5) BCP_ST(6) BCP_ST(7) BCP_ST(8) BCP_ST(9) BCP_ST(10) BCP_ST(11) BCP_ST(12) BCP_ST(13)
#undef BCP_ST
  };
#pragma unroll 1
  for (int cc = c_begin; cc < c_end; cc += 2) {
    chunk(cc, std::integral_constant<int, 0>{});
    if (cc + 1 < c_end) chunk(cc + 1, std::integral_constant<int, 1>{});
  }
  BCP_TS(60);

  double s1[2][4], s2[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.0; s2[nt][r] = 0.0; }
  b6_store_tile<TL, TD, TH, TW, 2, 2, BW>(acc, Y, bias, cd, n, d0, h0, w0, cout0 + wn * 32, accumulate, st.partial != nullptr, s1, s2, wm, &st,
                                      st.partial ? bx / st.tiles_per_group : 0, osc);
  if (st.partial) {
    const int gg = bx / st.tiles_per_group, row = bx % st.tiles_per_group;
    BCP_LDS_BARRIER();
    stats_flush_22(s1, s2, Ss, st.partial + ((long long)gg * st.rows + row) * st.C * 2, cout0, cd.Cout);
  }
  BCP_TS(61);
  BCP_TSR(62);
}

// ------------------------------------------------------------------------------------------------
// Variant without a weight stage: every wave loads its weight fragments STRAIGHT from the pre-split pack into registers (one
// coalesced 1 KB load per fragment: 16 output channels x 64 B; the four waves of a workgroup and the co-resident workgroup hit
// the same lines in L1 / L2), one tap pair ahead.  The LDS holds the halo planes only, so the tap loop has NO barrier at all -- the
// waves run free and the matrix pipe always finds one with work; k_c3b synchronises its four waves after every tap pair (~0.35 us)
// to hand the weight buffers over.  The pair loop is fully unrolled (compile-time tap offsets and register sets).
// ------------------------------------------------------------------------------------------------
#ifndef BCP_C3D_ATTR
#define BCP_C3D_ATTR
#endif
template <int KD, int TD, int TH, int TW, int NT, bool PER, bool BW = false, int PL = 3>
__global__ __launch_bounds__(256) BCP_C3D_ATTR void k_c3d(const float* __restrict__ X, const float* __restrict__ Wp, const float* __restrict__ bias,
                                             float* __restrict__ Y, ConvDims cd, int n_tiles, int accumulate, StatsArg st) {
  using TL = Tile<KD, TD, TH, TW>;
  constexpr int MT = TL::MT, T = TL::T, TP = (T + 1) / 2, TPE = (TP + 1) & ~1, CT = NT * 16;
  constexpr int XPLANE = TL::HV * XSB;
  using HF = HaloFetch<TL>;
  using PP = Pipe<PL>;
  using frag_t = typename PP::frag;

  BCP_TS(0);
  BCP_TSR(59);
  HIP_DYNAMIC_SHARED(float4, smem4)
  unsigned short* Xb = reinterpret_cast<unsigned short*>(smem4);   // [PL][HV][XSB]
  double* Ss = reinterpret_cast<double*>(Xb + PL * XPLANE);        // [4][CT][2] statistics scratch

  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lg = lane >> 4, wave = threadIdx.x >> 6;
  const int cout0 = blockIdx.y * CT;

  int voff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) voff[mt] = TL::voff((wave * MT + mt) * 16 + b6_row<TW>(li)) * XSB + (lg & 1) * 8;
  HF hf;
  hf.init(cd, reinterpret_cast<float*>(smem4));

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = cd.Cin16 >> 4;
  const int c_begin = (int)((long long)nchunks * blockIdx.z / gridDim.z), c_end = (int)((long long)nchunks * (blockIdx.z + 1) / gridDim.z);
  const int nch = c_end - c_begin;
  Y += (long long)blockIdx.z * cd.N * cd.D * cd.H * cd.W * cd.Cout;

  // PERSISTENT: this workgroup owns tiles blockIdx.x, blockIdx.x + gridDim.x, ...; work item `it` = (tile it / nch, chunk it % nch) of
  // that list.  The halo of item it + 1 -- the NEXT TILE's first chunk after a tile's last -- travels under item it's tap pairs, so
  // the fetch latency of a tile's first halo and the epilogue stores of the previous tile overlap with MFMAs (the 16-channel
  // level has ONE chunk per tile: without this a workgroup's life is prologue + epilogue).
  // (PER = false: one tile per workgroup -- the 32-channel-slab instances, whose persistent form needs 355 VGPRs and would drop to
  //  one workgroup per CU: 102-109 vs 78-84 us)
  // XCD-aware (cd.xcd): one tile per workgroup -> xcd_tile(); persistent -> each XCD (= blockIdx.x % 8 when the grid row is a multiple
  // of 8) walks ONE contiguous eighth of the tile list with its gridDim.x / 8 workgroups side by side, as k_conv3_res does
  int t_first, t_step, my_tiles;
  if (!PER) {
    t_first = cd.xcd ? xcd_tile(blockIdx.x, gridDim.x, gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) : (int)blockIdx.x;
    t_step = 0; my_tiles = 1;
  } else if ((cd.xcd & 2) && (gridDim.x & 7) == 0) {
    // (measurement switch only: with every XCD walking its own eighth the eight tile streams start 16 MB apart and march in lockstep --
    //  fabric fetch halves (337 -> 176 MB at 2x112x112x80x16) but the kernel runs 242 instead of 186 us alone, 310 vs 171 in the step:
    //  the round-robin deal below keeps all workgroups inside one moving window of the volume)
    const int xc = blockIdx.x & 7, xs = (int)((long long)n_tiles * xc / 8), xe = (int)((long long)n_tiles * (xc + 1) / 8);
    t_step = gridDim.x >> 3;
    t_first = xs + (int)(blockIdx.x >> 3);
    my_tiles = t_first < xe ? (xe - 1 - t_first) / t_step + 1 : 0;
  } else {
    t_first = blockIdx.x; t_step = gridDim.x;
    my_tiles = (int)blockIdx.x < n_tiles ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  }
  const int n_items = my_tiles * nch;

  // statistics: one row per (group, workgroup); rows this workgroup never reaches must read as zero
  const bool want_stats = PER && st.partial != nullptr;
  if (want_stats && (int)threadIdx.x < CT && cout0 + (int)threadIdx.x < cd.Cout) {
    for (int g = 0; g < st.G; ++g) {
      double* z = st.partial + (((long long)g * st.rows + blockIdx.x) * st.C + cout0 + threadIdx.x) * 2;
      z[0] = 0.0; z[1] = 0.0;
    }
  }
  if (n_items == 0) return;
  double s1[NT][4], s2[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.0; s2[nt][r] = 0.0; }

  // PL = 2: power-of-two pre-scale of the activations (their |max| from the norm apply pass that wrote them) and of the weights
  // (pack header); osc undoes both on the accumulators
  float xsc = 1.f, osc = 1.f;
  if (PL == 2) {
    const int ex = f16_scale_exp(amax_read(cd.xamax)), ew = f16_scale_exp(Wp[pack_off_hdr(T, cd.Cin16, cd.Cout16)]);
    xsc = ldexpf(1.f, ex);
    osc = ldexpf(1.f, -(ex + ew));
  }
  // lane (li, lg): output channel li of n-tile nt, k quarter lg of pre-split pack row [chunk][pair][piece][cout][32 k]
  const unsigned short* Wl = reinterpret_cast<const unsigned short*>(Wp + PP::pack_off(T, cd.Cin16, cd.Cout16)) + (long long)(cout0 + li) * 32 + lg * 8;
  const long long piece_stride = (long long)cd.Cout16 * 32;
  auto bload = [&](int cc, int tp, frag_t (&b)[NT][PL]) __attribute__((always_inline)) {
    if (B6_ABLATE & 4) return;
    const unsigned short* p = Wl + ((long long)cc * TP + (tp < TP ? tp : TP - 1)) * PL * piece_stride;
#pragma unroll
    for (int s = 0; s < PL; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt][s] = *reinterpret_cast<const frag_t*>(p + s * piece_stride + nt * 16 * 32);
  };
  unsigned hvm = 0;
  float4 hpre[HF::NP];
  auto hfetch_item = [&](int it) __attribute__((always_inline)) {      // (past the end: re-read the last item, no conditional load)
    if (B6_ABLATE & 2) return;
    const int itc = it < n_items ? it : n_items - 1;
    const int tl = t_first + (itc / nch) * t_step;
    int n, d0, h0, w0;
    tile_origin(cd, tl, TD, TH, TW, n, d0, h0, w0);
    hvm = hf.fetch_nb(X, cd, n, d0, h0, w0, c_begin + itc % nch, hpre);
  };
  auto hstash = [&]() __attribute__((always_inline)) {
    if (B6_ABLATE & 2) return;
#pragma unroll
    for (int u = 0; u < HF::NP; ++u)
      if (hf.act && u * HF::RPP + hf.r0 < HF::HR) {
        const float4 v = ((hvm >> u) & 1u) ? hpre[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        PP::split(v, xsc, Xb + ((u * HF::RPP + hf.r0) * TL::HW + hf.hw) * XSB + hf.part * 4, XPLANE);
      }
  };

  frag_t B0[NT][PL], B1[NT][PL];
  if (B6_ABLATE & 4) {
#pragma unroll
    for (int s = 0; s < PL; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { B0[nt][s] = *reinterpret_cast<const frag_t*>(Xb + s * 64 + nt * 8 + lane); B1[nt][s] = B0[nt][s]; }
  }
  BCP_TS(1);
  hfetch_item(0);
  bload(c_begin, 0, B0);
  hstash();
  BCP_LDS_BARRIER();
  BCP_TS(2);
  // pair in front of which the next item's halo is fetched.  Round 3 (s_memtime stamps, tools/ts_probe.py, 32-channel level): with the
  // fetch four pairs ahead (TPE - 4) a chunk boundary cost ~9000 ticks against ~1000 per pair -- every workgroup of the launch
  // fetches at the same time and the loads take longer than four pairs; the registers are live across the peak of the register
  // pressure either way, so the fetch moves to the front of the chunk
  // (the instances with the backward-statistics epilogue sit at 248 registers: from pair 7 down the longer live range costs them the
  //  second workgroup per CU -- they fetch six pairs ahead)
#ifndef BCP_C3D_HPF
#define BCP_C3D_HPF 3
#endif
  constexpr int HPF = TPE >= 10 ? (BCP_C3D_HPF >= 0 && !BW ? BCP_C3D_HPF : TPE - 6) : (TPE >= 6 ? TPE - 4 : 0);
  int cur_g = want_stats ? t_first / st.tiles_per_group : 0;
#pragma unroll 1
  for (int it = 0; it < n_items; ++it) {
    if (it > 0) {
      BCP_LDS_BARRIER();                             // every wave is done with the previous item's halo planes
      hstash();
      BCP_LDS_BARRIER();
    }
    const int cc = c_begin + it % nch;
    const int ccn = c_begin + (it + 1 < n_items ? (it + 1) % nch : it % nch);
#if BCP_C3D_STREAM
    frag_t nx[BCP_C3D_FD];                           // the next pair's first fragments (fragment stream, below)
#endif
#pragma unroll
    for (int tp = 0; tp < TPE; ++tp) {
      // the next pair's weight fragments (the next item's pair 0 after the last one) into the other register set
      if (tp & 1) { if (tp + 1 < TPE) bload(cc, tp + 1, B0); else bload(ccn, 0, B0); }
      else bload(cc, tp + 1, B1);
      if (tp == HPF) hfetch_item(it + 1);
      if (tp < TP && !(B6_ABLATE & 8)) {
        const int t0 = 2 * tp, t1 = 2 * tp + 1 < T ? 2 * tp + 1 : T - 1;
        const int tA = ((t0 / 9) * TL::HH + (t0 / 3) % 3) * TL::HW + t0 % 3, tB = ((t1 / 9) * TL::HH + (t1 / 3) % 3) * TL::HW + t1 % 3;
        const int toff = ((lg >> 1) ? tB : tA) * XSB;
#if BCP_C3D_STREAM
        // FRAGMENT STREAM (round 3).  hipcc kept ONE register quad for the piece-2 fragments of a pair and re-loaded it four times --
        // read, s_waitcnt, two MFMAs, read again (ISA of round 2's loop): the LDS round trip was exposed ~4x per pair, the "LDS
        // fragment reads" share of the ablation (47.8 us for MFMAs + reads against a 36.4 us MFMA floor at the 32-channel level).
        // Here the MT * 3 A fragments of a pair are consumed in a FIXED order -- m-tile pairs, pieces 2, 1, 0 -- and each is requested
        // FD fragments ahead of its MFMAs; sched_barrier pins the read in front of the MFMA group it hides under, so the compiler's
        // wait counts become lgkmcnt(FD - 1) instead of lgkmcnt(0).  Per accumulator the products arrive as (2,0) (1,1) (1,0) (0,2)
        // (0,1) (0,0): the two ~2^-16 terms swap places against round 2's order, nothing else; consecutive MFMAs into one accumulator
        // are four apart.
        {
          constexpr int MP = (NT > 1 && MT % BCP_C3D_MP == 0) ? BCP_C3D_MP : 1;      // m-tiles per MFMA group (one n-tile per wave: 1 -- measured 161 vs 168 us at the 16-channel level)
          constexpr int NF = MT * PL, FD = BCP_C3D_FD < NF ? BCP_C3D_FD : NF;     // fragments per pair, prefetch distance
          static_assert(FD <= NF && MT % MP == 0, "fragment stream geometry");
          auto fidx = [&](int f, int& mt, int& I) __attribute__((always_inline)) { const int g = f / (PL * MP), r = f % (PL * MP); I = PL - 1 - r / MP; mt = g * MP + r % MP; };
          // the stream runs ACROSS the pairs of a chunk: the first FD fragments of pair tp + 1 are requested under the last MFMA groups
          // of pair tp (nx), so only a chunk's first pair waits for an LDS round trip
          const int t0n = 2 * (tp + 1), t1n = 2 * (tp + 1) + 1 < T ? 2 * (tp + 1) + 1 : T - 1;
          const int tAn = ((t0n / 9) * TL::HH + (t0n / 3) % 3) * TL::HW + t0n % 3, tBn = ((t1n / 9) * TL::HH + (t1n / 3) % 3) * TL::HW + t1n % 3;
          const int toffn = ((lg >> 1) ? tBn : tAn) * XSB;
          const bool have_nx = tp > 0;                         // (compile-time after unrolling: pair 0 of a chunk follows a barrier)
          const bool want_nx = tp + 1 < TP;
          frag_t fr[NF];
#pragma unroll
          for (int f = 0; f < FD; ++f) {
            int mt, I;
            fidx(f, mt, I);
            if (have_nx) fr[f] = nx[f];
            else fr[f] = *reinterpret_cast<const frag_t*>(Xb + I * XPLANE + voff[mt] + ((B6_ABLATE & 32) ? 0 : toff));
          }
#pragma unroll
          for (int f0 = 0; f0 < NF; f0 += MP) {
#pragma unroll
            for (int q = 0; q < MP; ++q) {
              const int g = f0 + q + FD;
              int mt, I;
              if (g < NF) { fidx(g, mt, I); fr[g] = *reinterpret_cast<const frag_t*>(Xb + I * XPLANE + voff[mt] + ((B6_ABLATE & 32) ? 0 : toff)); }
              else if (want_nx) { fidx(g - NF, mt, I); nx[g - NF] = *reinterpret_cast<const frag_t*>(Xb + I * XPLANE + voff[mt] + ((B6_ABLATE & 32) ? 0 : toffn)); }
            }
            __builtin_amdgcn_sched_barrier(0);
            int mt0, I;
            fidx(f0, mt0, I);
#pragma unroll
            for (int J = PL - 1 - I; J >= 0; --J)
#pragma unroll
              for (int q = 0; q < MP; ++q)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                  if (tp & 1) acc[mt0 + q][nt] = PP::mfma(B1[nt][J], fr[f0 + q], acc[mt0 + q][nt]);
                  else acc[mt0 + q][nt] = PP::mfma(B0[nt][J], fr[f0 + q], acc[mt0 + q][nt]);
                }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#else
        static_assert(PL == 3, "the unstreamed tap loop exists for the bf16 planes only");
        bf16x8 a[MT][3];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a[mt][s] = *reinterpret_cast<const bf16x8*>(Xb + s * XPLANE + voff[mt] + ((B6_ABLATE & 32) ? 0 : toff));
        // an odd tap count: the second half of the last pair multiplies tap T-1's voxels by the pack's zero weights
#define BCP_B6(BS, I, J)                                                                                        \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)            \
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BS[nt][J], a[mt][I], acc[mt][nt], 0, 0, 0);
        if (tp & 1) { BCP_B6(B1, 2, 0) BCP_B6(B1, 1, 1) BCP_B6(B1, 0, 2) BCP_B6(B1, 1, 0) BCP_B6(B1, 0, 1) BCP_B6(B1, 0, 0) }
        else { BCP_B6(B0, 2, 0) BCP_B6(B0, 1, 1) BCP_B6(B0, 0, 2) BCP_B6(B0, 1, 0) BCP_B6(B0, 0, 1) BCP_B6(B0, 0, 0) }
#undef BCP_B6
#endif
      }
      if (!PER) BCP_TS(3 + it * TPE + tp);
    }
    if (it % nch == nch - 1) {                       // the tile is complete: store it (uniform branch; stores only, no loads to wait for)
      const int tl = t_first + (it / nch) * t_step;
      int n, d0, h0, w0;
      tile_origin(cd, tl, TD, TH, TW, n, d0, h0, w0);
      if (!PER) {
        BCP_TS(60);
        BCP_LDS_BARRIER();
        b6_epilogue<TL, TD, TH, TW, NT, BW>(acc, Y, bias, cd, n, d0, h0, w0, cout0, accumulate, st, Ss, t_first, osc);      // one statistics row per tile
        BCP_TS(61);
        BCP_TSR(62);
        return;
      }
      if (want_stats && tl / st.tiles_per_group != cur_g) {
        BCP_LDS_BARRIER();
        stats_flush_t<NT>(s1, s2, Ss, st.partial + ((long long)cur_g * st.rows + blockIdx.x) * st.C * 2, cout0, cd.Cout);
        cur_g = tl / st.tiles_per_group;
      }
      b6_store_tile<TL, TD, TH, TW, NT, TL::MT, BW>(acc, Y, bias, cd, n, d0, h0, w0, cout0, accumulate, want_stats, s1, s2, -1, &st, cur_g, osc);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  if (want_stats) {
    BCP_LDS_BARRIER();
    stats_flush_t<NT>(s1, s2, Ss, st.partial + ((long long)cur_g * st.rows + blockIdx.x) * st.C * 2, cout0, cd.Cout);
  }
}

// ------------------------------------------------------------------------------------------------
// FLAT variant for the deep levels (14x14x10, 7x7x5, 12^3, 6^3 ...): the M dimension is BM = 64 CONSECUTIVE voxels of one sample
// (flat index over D*H*W) instead of a brick, so nothing is lost to tile padding (2x8x4 bricks waste 37 % of 14x14x10 and 109 % of
// 7x7x5).  The halo is the flat voxel range [m0 - R, m0 + BM + R), R = H*W + W + 1 (one row of 32 bytes per voxel and piece, as in
// k_c3b); a tap is a row offset kd*H*W + kh*W + kw.  Neighbours that wrap around a row / plane end -- or leave the volume -- are
// not zero in a flat range: each lane carries the 27 validity bits of ITS voxel and clears the A fragment of an invalid (voxel,
// tap) with v_cndmask (12 per fragment triple: bf16 MFMAs leave the issue slots for that; on the fp32 pipe the same masks cost more
// than the padding they saved).  Everything else is k_c3b: weight stages of SP tap pairs in LDS, the stage pipeline, split-K.
// ------------------------------------------------------------------------------------------------
template <int KD, int NT, int SP, int AVMAX>
__global__ __launch_bounds__(256) void k_c3f(const float* __restrict__ X, const float* __restrict__ Wp, const float* __restrict__ bias,
                                             float* __restrict__ Y, ConvDims cd, int tiles_per_sample, int accumulate, StatsArg st) {
  constexpr int BM = 64, T = KD * 9, TP = (T + 1) / 2, CT = NT * 16, PD = KD == 3 ? 1 : 0;
  constexpr int S = ((TP + SP - 1) / SP + 1) & ~1;
  static_assert(S >= 4, "the stage pipeline needs four weight stages per chunk");
  constexpr int XPLANE = AVMAX * XSB;
  constexpr int WPLANE = SP * CT * 32;
  constexpr int WSTAGE = 3 * WPLANE;
  constexpr int NW4 = (SP * 12 * CT + 255) / 256;
  constexpr int NP = (AVMAX * 4 + 255) / 256;                  // halo float4 per thread (row, 4-channel part)

  HIP_DYNAMIC_SHARED(float4, smem4)
  unsigned short* Xb = reinterpret_cast<unsigned short*>(smem4);   // [3][AVMAX][XSB]
  unsigned short* Wb = Xb + 3 * XPLANE;                            // [2][3][SP][CT][32]
  double* Ss = reinterpret_cast<double*>(Wb + 2 * WSTAGE);         // [4][CT][2] statistics scratch

  BCP_TS(0);
  BCP_TSR(59);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int V = cd.D * cd.H * cd.W, HW = cd.H * cd.W;
  const int R = PD * HW + cd.W + 1, AV = BM + 2 * R;               // AV <= AVMAX (checked by the launcher)
  const int bx = cd.xcd ? xcd_tile(blockIdx.x, gridDim.x, gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) : (int)blockIdx.x;   // tile of this workgroup
  const int n = bx / tiles_per_sample, m0 = (bx % tiles_per_sample) * BM;
  const int cout0 = blockIdx.y * CT;

  // this lane's voxel (one m-tile per wave), its halo row and the validity bits of its 27 neighbours
  const int ml = wave * 16 + li, mv = m0 + ml;
  const int vrow = (ml + R) * XSB + (lg & 1) * 8;
  unsigned vbits = 0;
  {
    const int w = mv % cd.W, h = (mv / cd.W) % cd.H, d = mv / HW;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int kw = t % 3, kh = (t / 3) % 3, kd = t / 9;
      const bool ok = (unsigned)(w + kw - 1) < (unsigned)cd.W && (unsigned)(h + kh - 1) < (unsigned)cd.H && (unsigned)(d + kd - PD) < (unsigned)cd.D;
      vbits |= (ok ? 1u : 0u) << t;
    }
    if (mv >= V) vbits = 0;
  }
  const int woff = li * 32 + ((lg ^ ((li & 8) ? 2 : 0)) * 8);

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = cd.Cin16 >> 4;
  const int c_begin = (int)((long long)nchunks * blockIdx.z / gridDim.z), c_end = (int)((long long)nchunks * (blockIdx.z + 1) / gridDim.z);
  Y += (long long)blockIdx.z * cd.N * V * cd.Cout;

  const unsigned short* Wb16 = reinterpret_cast<const unsigned short*>(Wp + (long long)T * cd.Cin16 * cd.Cout16);
  // (ONE uniform base per stage + a per-thread 32-bit offset fixed for the whole kernel; see k_c3h)
  static_assert(SP == 1, "k_c3f: one tap pair per stage");
  unsigned wq[NW4];
#pragma unroll
  for (int u = 0; u < NW4; ++u) {
    const int q = (threadIdx.x + u * 256) % (12 * CT);
    const int kq = q & 3, co = (q >> 2) % CT, sp = q / (4 * CT);
    wq[u] = (unsigned)((sp * cd.Cout16 + cout0 + co) * 32 + kq * 8);
  }
  auto wfetch = [&](int cc, int sg, float4 (&wpre)[NW4]) __attribute__((always_inline)) {
    const unsigned short* wst = Wb16 + (long long)(cc * TP + (sg < TP ? sg : TP - 1)) * 3 * cd.Cout16 * 32;      // uniform
#pragma unroll
    for (int u = 0; u < NW4; ++u) wpre[u] = *reinterpret_cast<const float4*>(wst + wq[u]);
  };
  auto wstash = [&](unsigned short* Wbuf, int sg, const float4 (&wpre)[NW4]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NW4; ++u) {
      const int q = threadIdx.x + u * 256;
      const int kq = q & 3, co = (q >> 2) % CT, sp = (q / (4 * CT)) % 3, pr = q / (12 * CT);
      const float4 v = (sg * SP + pr < TP) ? wpre[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      if ((SP * 12 * CT) % 256 == 0 || q < SP * 12 * CT) *reinterpret_cast<float4*>(Wbuf + sp * WPLANE + (pr * CT + co) * 32 + ((kq ^ ((co & 8) ? 2 : 0)) * 8)) = v;
    }
  };
  // halo: row r of the flat range = voxel m0 - R + r of sample n (zero outside [0, V) and beyond Cin); branch-free loads
  unsigned hvm = 0;
  const long long xbase = (long long)n * V * cd.Cin;
  auto hfetch = [&](int cc, float4 (&pre)[NP]) __attribute__((always_inline)) {
    hvm = 0;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int q = threadIdx.x + u * 256, r = q >> 2, part = q & 3;
      const int gm = m0 - R + r;
      const unsigned ok = (r < AV && (unsigned)gm < (unsigned)V && cc * 16 + part * 4 < cd.Cin) ? 1u : 0u;
      const unsigned off = ok ? (unsigned)(xbase + (long long)gm * cd.Cin + cc * 16 + part * 4) : 0u;
      pre[u] = ld4(X + off);
      hvm |= ok << u;
    }
  };
  auto hstash = [&](const float4 (&pre)[NP]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int q = threadIdx.x + u * 256, r = q >> 2, part = q & 3;
      if (r < AV) {
        const float4 v = ((hvm >> u) & 1u) ? pre[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        split_store4(v, Xb + r * XSB + part * 4, XPLANE);
      }
    }
  };
  auto wfetch_at = [&](int cc, int sg, float4 (&wpre)[NW4]) __attribute__((always_inline)) {
    while (sg >= S) { sg -= S; ++cc; }      // (S = 4 for the 2-D two-pair stages: a five-stage lookahead can cross TWO chunk ends)
    if (cc >= c_end) { cc = c_end - 1; sg = S - 1; }
    if (B6_ABLATE & 64) { cc = c_begin; sg = 0; }       // (measurement: every stage re-reads the first stage's weights -- cache-resident fetches)
    wfetch(cc, sg, wpre);
  };
  // four weight stages in flight in registers, chunk body fully unrolled (static register sets): see k_c3h
  constexpr int HPF = S - 4;
  BCP_TS(1);
  float4 hpre[NP], W[4][NW4];
  hfetch(c_begin, hpre);
  wfetch_at(c_begin, 0, W[0]);
  wfetch_at(c_begin, 1, W[1]);
  wfetch_at(c_begin, 2, W[2]);
  wfetch_at(c_begin, 3, W[3]);
  hstash(hpre);
  wstash(Wb, 0, W[0]);
  wfetch_at(c_begin, 4, W[0]);
  BCP_LDS_BARRIER();
  BCP_TS(2);

  // (order within a stage: fragment reads, the next stage's weights to the other buffer + the refill of their registers, THEN the MFMAs;
  //  see k_c3h)
  auto stage = [&](int cc, int sg, float4 (&Wn)[NW4]) __attribute__((always_inline)) {
    const unsigned short* Wc = Wb + (sg & 1) * WSTAGE;
    const int tp = sg;
    bf16x8 a[3], b[NT][3];
    if (tp < TP) {     // uniform
      const int t0 = 2 * tp, t1 = 2 * tp + 1 < T ? 2 * tp + 1 : T - 1;
      const int oA = ((t0 / 9) - PD) * HW + ((t0 / 3) % 3 - 1) * cd.W + (t0 % 3 - 1);      // wave-uniform row offsets of the two taps
      const int oB = ((t1 / 9) - PD) * HW + ((t1 / 3) % 3 - 1) * cd.W + (t1 % 3 - 1);
      const int toff = ((lg >> 1) ? oB : oA) * XSB;
      const bool ok = (((lg >> 1) ? (vbits >> t1) : (vbits >> t0)) & 1u) != 0;
      const bf16x8 zero = __builtin_bit_cast(bf16x8, make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(Xb + s * XPLANE + vrow + toff);
        a[s] = ok ? v : zero;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt][s] = *reinterpret_cast<const bf16x8*>(Wc + s * WPLANE + nt * 16 * 32 + woff);
      }
    }
    if (sg + 1 < S || cc + 1 < c_end) wstash(Wb + ((sg + 1) & 1) * WSTAGE, sg + 1 < S ? sg + 1 : 0, Wn);
    wfetch_at(cc, sg + 5, Wn);
    if (tp < TP) {
#define BCP_B6(I, J)                                                                                            \
  _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                              \
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nt][J], a[I], acc[nt], 0, 0, 0);
      BCP_B6(2, 0) BCP_B6(1, 1) BCP_B6(0, 2) BCP_B6(1, 0) BCP_B6(0, 1) BCP_B6(0, 0)
#undef BCP_B6
    }
    if (sg + 1 < S) BCP_LDS_BARRIER();
    BCP_TS(3 + (cc - c_begin) * S + sg);
  };
  auto chunk = [&](int cc, auto phase_tag) __attribute__((always_inline)) {
    constexpr int PH = decltype(phase_tag)::value;
    if (cc > c_begin) {
      BCP_LDS_BARRIER();
      hstash(hpre);
      BCP_LDS_BARRIER();
    }
#pragma unroll
    for (int sg = 0; sg < S; ++sg) {
      if (sg == HPF) hfetch(cc + 1 < c_end ? cc + 1 : cc, hpre);      // (compile-time position: not a conditional load)
      stage(cc, sg, W[(PH + sg + 1) & 3]);
    }
  };
#pragma unroll 1
  for (int cc = c_begin; cc < c_end; cc += 2) {
    chunk(cc, std::integral_constant<int, 0>{});
    if (cc + 1 < c_end) chunk(cc + 1, std::integral_constant<int, (S & 3)>{});
  }

  BCP_TS(60);
  // epilogue: lane (li, lg) holds voxel m0 + wave*16 + li, channels lg*4 .. lg*4+3 of each n-tile: 16-byte stores into flat rows
  double s1[NT][4], s2[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.0; s2[nt][r] = 0.0; }
  const bool vec = cout0 + CT <= cd.Cout && (cd.Cout & 3) == 0;     // uniform
  const bool want_stats = st.partial != nullptr;
  if (mv < V) {
    float* yrow = Y + ((long long)n * V + mv) * cd.Cout + cout0 + lg * 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cout0 + nt * 16 + lg * 4 + r;
        float v = acc[nt][r] + ((bias && co < cd.Cout) ? bias[co] : 0.f);
        if (accumulate && co < cd.Cout) v += yrow[nt * 16 + r];
        acc[nt][r] = v;
        if (want_stats && co < cd.Cout) { s1[nt][r] += (double)v; s2[nt][r] += (double)v * (double)v; }
      }
      if (vec) st4(yrow + nt * 16, make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]));
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (cout0 + nt * 16 + lg * 4 + r < cd.Cout) yrow[nt * 16 + r] = acc[nt][r];
      }
    }
  }
  if (want_stats) {
    const int gg = bx / st.tiles_per_group, row = bx % st.tiles_per_group;
    BCP_LDS_BARRIER();
    stats_flush_t<NT>(s1, s2, Ss, st.partial + ((long long)gg * st.rows + row) * st.C * 2, cout0, cd.Cout);
  }
  BCP_TS(61);
  BCP_TSR(62);
}

// ------------------------------------------------------------------------------------------------
// k_c3f as the software pipeline of k_c3p (round 3): weight stages by LDS-DMA into a ring of three slots, three stages ahead; the
// fragments of stage g + 1 read into a second register set between the MFMAs of stage g; one memory instruction behind each MFMA.
// What differs from k_c3p: one m-tile x NT n-tiles per wave (3 + 3 NT fragment reads per 6 NT MFMAs); ONE halo buffer (two of 36 KB
// would cost the second workgroup per CU), so a chunk boundary keeps its two barriers around the stash and the first stage of a chunk
// reads its voxel fragments late; a neighbour outside the volume is not masked after the read but reads the all-zero row AVMAX of the
// planes (a select on the address, one per stage, instead of on twelve data registers behind the wait for the read).
// LDS: 3 x 385 x 32 B halo + 3 slots x 3 x CT x 64 B + statistics scratch = 36 + 36 + 2 KB at 64-channel slabs.
// Same MFMA sequence per accumulator as k_c3f: bit-identical results.
// ------------------------------------------------------------------------------------------------
template <int NT, int AVMAX, int PL = 3>
__global__ __launch_bounds__(256) void k_c3q(const float* __restrict__ X, const float* __restrict__ Wp, const float* __restrict__ bias,
                                             float* __restrict__ Y, ConvDims cd, int tiles_per_sample, int accumulate, StatsArg st) {
  constexpr int BM = 64, T = 27, TP = 14, S = 14, CT = NT * 16;
  constexpr int XPLANE = (AVMAX + 1) * XSB;                      // (row AVMAX: zeros)
  constexpr int WPLANE = CT * 32, WSLOT = PL * WPLANE;           // elements
  constexpr int NPIECE = 4 * PL * CT, NDMA = (NPIECE + 255) / 256;   // 16-byte pieces of a stage; DMA instructions per wave and stage
  constexpr int NPR = PL * (PL + 1) / 2;                         // piece products per accumulator and stage
  using PP = Pipe<PL>;
  using frag_t = typename PP::frag;
  constexpr int NP = (AVMAX * 4 + 255) / 256;                    // halo float4 per thread (row, 4-channel part)
  constexpr int HPF = BCP_C3P_HFS >= 0 ? BCP_C3P_HFS : S - 4;   // stage that fetches the next chunk's halo (early: see k_c3d's HPF)
  constexpr int NMEM = NDMA + PL + PL * NT, NMMA = NPR * NT;

  HIP_DYNAMIC_SHARED(float4, smem4)
  unsigned short* Xb = reinterpret_cast<unsigned short*>(smem4);   // [PL][AVMAX + 1][XSB]
  unsigned short* Wr = Xb + PL * XPLANE;                           // [3 slots][PL][CT][32]
  double* Ss = reinterpret_cast<double*>(Wr + 3 * WSLOT);          // [4][CT][2] statistics scratch

  BCP_TS(0);
  BCP_TSR(59);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int V = cd.D * cd.H * cd.W, HW = cd.H * cd.W;
  const int R = HW + cd.W + 1, AV = BM + 2 * R;                    // AV <= AVMAX (checked by the launcher)
  // Workgroup -> (tile, channel slab, cin range).  The workgroups of one (slab, cin range) read the SAME weight stream (12 KB x 14 per
  // chunk); dealt out in launch order they land on all eight XCDs and every L2 pulls every weight of the layer from the fabric.  With
  // a multiple of eight weight streams, XCD k (= linear workgroup id % 8) can take streams k * Gw / 8 .. and all their tiles.  Fabric
  // fetch per launch (rocprofv3 FETCH_SIZE) and time alone:
  //   256 channels (64 streams): 94 -> 15.8 MB, 21.8 -> 19.4 us                                          => on
  //   128 channels (8 streams, one per XCD): 36.3 -> 6.8 MB (= 4 MB of activations + 2.65 MB of weights, the algorithmic minimum)
  //     but 28.4 -> 30.3 us alone (all 248 waves of an XCD ask its L2 for the same lines at the same time), and the op's in-step
  //     launch time 33 -> 37 us; the STEP does not move (LA 6.05 vs 6.02 ms over three interleaved pairs: what this kernel loses the
  //     other stream's kernels gain from the emptier fabric)                                              => off (conv3_xcd bit 3 turns it on)
  //   (bit 4: the tile order of xcd_tile() -- 29.2 us, 24.3 MB -- for measurements)
  int bx, by = blockIdx.y, bz = blockIdx.z;
  {
    const int Gw = gridDim.y * gridDim.z;
    if (cd.xcd && (Gw & 7) == 0 && (Gw >= 16 || (cd.xcd & 8))) {
      const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), j = L >> 3;
      const int grp = (L & 7) * (Gw >> 3) + j / (int)gridDim.x;
      bx = j % (int)gridDim.x; by = grp % (int)gridDim.y; bz = grp / (int)gridDim.y;
    } else {
      bx = (cd.xcd & 16) ? xcd_tile(blockIdx.x, gridDim.x, gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) : (int)blockIdx.x;
    }
  }
  const int n = bx / tiles_per_sample, m0 = (bx % tiles_per_sample) * BM;
  const int cout0 = by * CT;

  // this lane's voxel (one m-tile per wave), its halo row and the validity bits of its 27 neighbours
  const int ml = wave * 16 + li, mv = m0 + ml;
  const int vrow = (ml + R) * XSB + (lg & 1) * 8, zrow = AVMAX * XSB + (lg & 1) * 8;
  unsigned vbits = 0;
  if (mv < V) {
    const int w = mv % cd.W, h = (mv / cd.W) % cd.H, d = mv / HW;
    const unsigned mw = (w >= 1 ? 1u : 0u) | 2u | (w + 1 < cd.W ? 4u : 0u);
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
        if ((unsigned)(d + kd - 1) < (unsigned)cd.D && (unsigned)(h + kh - 1) < (unsigned)cd.H) vbits |= mw << (kd * 9 + kh * 3);
  }
  const int woff = li * 32 + ((lg ^ ((li & 8) ? 2 : 0)) * 8);

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = cd.Cin16 >> 4;
  const int c_begin = (int)((long long)nchunks * bz / gridDim.z), c_end = (int)((long long)nchunks * (bz + 1) / gridDim.z);
  Y += (long long)bz * cd.N * V * cd.Cout;

  // DMA instruction u of wave w carries pieces (u * 256 + w * 64) % NPIECE + lane (instructions past the stage repeat its first
  // pieces: same bytes to the same place); piece q = (plane q / (4 CT), row (q >> 2) % CT, k quarter POSITION q & 3) at byte 16 q
  const char* Wb16 = reinterpret_cast<const char*>(Wp + PP::pack_off(T, cd.Cin16, cd.Cout16));
  float xsc = 1.f, osc = 1.f;          // PL = 2: power-of-two pre-scales (k_c3d)
  if (PL == 2) {
    const int ex = f16_scale_exp(amax_read(cd.xamax)), ew = f16_scale_exp(Wp[pack_off_hdr(T, cd.Cin16, cd.Cout16)]);
    xsc = ldexpf(1.f, ex);
    osc = ldexpf(1.f, -(ex + ew));
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  static_assert(NDMA <= 3, "k_c3q: at most 64-channel slabs");
  unsigned wq[3];      // (literal bounds: with a bound that depends on the template parameter the DMA builtin's address argument is type-dependent and
  int wdst[3];         //  hipcc's HOST pass silently drops the kernel's stub -- the library then fails to load with an undefined symbol)
#pragma unroll
  for (int u = 0; u < NDMA; ++u) {
    const int qb = (u * 256 + wave_u * 64) % NPIECE, q = qb + lane;
    const int co = (q >> 2) % CT, sp = q / (4 * CT), kq = (q & 3) ^ ((co & 8) ? 2 : 0);
    wq[u] = (unsigned)(((sp * cd.Cout16 + cout0 + co) * 32 + kq * 8) * 2);
    wdst[u] = qb * 16;
  }
  char* const Wr_b = reinterpret_cast<char*>(Wr);
  // (results through reference parameters: a value-returning lambda in a kernel template makes hipcc's HOST pass drop the kernel's
  //  stub without a diagnostic -- the library then fails to load with an undefined symbol)
  auto wsrc = [&](int cc, int sg, const char*& src) __attribute__((always_inline)) {
    while (sg >= S) { sg -= S; ++cc; }
    src = Wb16 + (long long)(cc * TP + sg) * PL * cd.Cout16 * 64;            // uniform
  };
  // halo: row r of the flat range = voxel m0 - R + r of sample n (zero outside [0, V) and beyond Cin); branch-free loads
  unsigned hvm = 0;
  const long long xbase = (long long)n * V * cd.Cin;
  auto hfetch = [&](int cc, float4 (&pre)[NP]) __attribute__((always_inline)) {
    hvm = 0;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int q = threadIdx.x + u * 256, r = q >> 2, part = q & 3;
      const int gm = m0 - R + r;
      const unsigned ok = (r < AV && (unsigned)gm < (unsigned)V && cc * 16 + part * 4 < cd.Cin) ? 1u : 0u;
      const unsigned off = ok ? (unsigned)(xbase + (long long)gm * cd.Cin + cc * 16 + part * 4) : 0u;
      pre[u] = ld4(X + off);
      hvm |= ok << u;
    }
  };
  auto hstash = [&](const float4 (&pre)[NP]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int q = threadIdx.x + u * 256, r = q >> 2, part = q & 3;
      if (r < AV) {
        const float4 v = ((hvm >> u) & 1u) ? pre[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        PP::split(v, xsc, Xb + r * XSB + part * 4, XPLANE);
      }
    }
  };
  // address of this lane's voxel fragment of stage sg (the all-zero row where the neighbour lies outside the volume)
  auto a_addr = [&](int sg, const unsigned short*& xa) __attribute__((always_inline)) {
    const int t0 = 2 * sg, t1 = 2 * sg + 1 < T ? 2 * sg + 1 : T - 1;
    const int oA = ((t0 / 9) - 1) * HW + ((t0 / 3) % 3 - 1) * cd.W + (t0 % 3 - 1);      // wave-uniform row offsets of the two taps
    const int oB = ((t1 / 9) - 1) * HW + ((t1 / 3) % 3 - 1) * cd.W + (t1 % 3 - 1);
    const bool ok = (((lg >> 1) ? (vbits >> t1) : (vbits >> t0)) & 1u) != 0;
    xa = Xb + (ok ? vrow + ((lg >> 1) ? oB : oA) * XSB : zrow);
  };

  BCP_TS(1);
  float4 hpre[NP];
  frag_t fa[2][PL], fb[2][NT][PL];
  unsigned sl0 = 0, sl1 = WSLOT * 2, sl2 = 2 * WSLOT * 2;      // ring slots (byte offsets) of stages g, g + 1, g + 2
  hfetch(c_begin, hpre);
  auto wdma = [&](int sg, unsigned slot) __attribute__((always_inline)) {
    const char* w;
    wsrc(c_begin, sg, w);
#pragma unroll
    for (int u = 0; u < NDMA; ++u) BCP_GLDS16(w + wq[u], Wr_b + slot + wdst[u]);
  };
  wdma(0, sl0);
  wdma(1, sl1);
  wdma(2, sl2);
  if (threadIdx.x < 2 * PL) *reinterpret_cast<float4*>(Xb + (threadIdx.x >> 1) * XPLANE + AVMAX * XSB + (threadIdx.x & 1) * 8) = make_float4(0.f, 0.f, 0.f, 0.f);   // the zero rows
  hstash(hpre);
  BCP_VM_LDS_BARRIER(0);
  auto frag0 = [&]() __attribute__((always_inline)) {
    const unsigned short* Xc;
    a_addr(0, Xc);
    const unsigned short* Wc = Wr + (sl0 >> 1) + woff;
#pragma unroll
    for (int s = 0; s < PL; ++s) {
      fa[0][s] = *reinterpret_cast<const frag_t*>(Xc + s * XPLANE);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[0][nt][s] = *reinterpret_cast<const frag_t*>(Wc + s * WPLANE + nt * 16 * 32);
    }
  };
  frag0();
  BCP_LDS_BARRIER();                                           // every wave has its stage-0 fragments: slot 0 may be refilled
  BCP_TS(2);

  auto stage = [&](int cc, auto sg_tag) __attribute__((always_inline)) {
    constexpr int sg = decltype(sg_tag)::value, PAR = sg & 1, NSG = sg + 1 < S ? sg + 1 : 0;
    constexpr bool LAST = sg + 1 == S;                         // the next stage's voxel fragments wait for the next chunk's planes
    if (sg == HPF) hfetch(cc + 1 < c_end ? cc + 1 : cc, hpre);      // (compile-time position; the last chunk re-reads its own halo)
    const char* wst;
    wsrc(cc, sg + 3, wst);
    const bool dma_on = sg + 3 < S || cc + 1 < c_end;          // uniform: nothing to fetch behind the workgroup's last stage
    const unsigned short* Xc;
    a_addr(NSG, Xc);
    const unsigned short* Wc = Wr + (sl1 >> 1) + woff;
    __builtin_amdgcn_sched_barrier(0);
    constexpr int PI[6] = {PL == 3 ? 2 : 1, PL == 3 ? 1 : 0, 0, 1, 0, 0}, PJ[6] = {0, 1, PL == 3 ? 2 : 0, 0, 1, 0};       // piece products, smallest terms first (as k_c3f); two planes: a1 b0, a0 b1, a0 b0
    int m = 0;                                                 // memory instructions issued so far (compile-time after unrolling)
#pragma unroll
    for (int k = 0; k < NMMA; ++k) {
      const int pr = k / NT, nt = k % NT;
      acc[nt] = PP::mfma(fb[PAR][nt][PJ[pr]], fa[PAR][PI[pr]], acc[nt]);
#pragma unroll
      for (; m < ((k + 1) * NMEM + NMMA - 1) / NMMA; ++m) {
        if (m < NDMA) { if (dma_on) BCP_GLDS16(wst + wq[m < 3 ? m : 0], Wr_b + sl0 + wdst[m < 3 ? m : 0]); }
        else if (m < NDMA + PL * NT) {
          const int r = m - NDMA, sp = r / NT, nt2 = r % NT;
          fb[PAR ^ 1][nt2][sp] = *reinterpret_cast<const frag_t*>(Wc + sp * WPLANE + nt2 * 16 * 32);
        } else if (!LAST) {
          const int sp = m - NDMA - PL * NT;
          fa[PAR ^ 1][sp] = *reinterpret_cast<const frag_t*>(Xc + sp * XPLANE);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the DMA issued a stage ago must have landed before anyone reads its slot at the top of the next stage: at most this stage's NDMA
    // DMAs may still be in flight (see k_c3p: the halo loads of the fetch stage are NOT added to the count)
    if (dma_on) BCP_VM_LDS_BARRIER(NDMA); else BCP_VM_LDS_BARRIER(0);      // (no DMAs issued in this stage: the previous stage's are the youngest)
    const unsigned t = sl0; sl0 = sl1; sl1 = sl2; sl2 = t;
    BCP_TS(3 + (cc - c_begin) * S + sg);
  };
#pragma unroll 1
  for (int cc = c_begin; cc < c_end; ++cc) {
#define BCP_ST(I) stage(cc, std::integral_constant<int, I>{});
    BCP_ST(0) BCP_ST(1) BCP_ST(2) BCP_ST(3) BCP_ST(4) BCP_ST(5) BCP_ST(6) BCP_ST(7) BCP_ST(8) BCP_ST(9) BCP_ST(10) BCP_ST(11) BCP_ST(12) BCP_ST(13)
#undef BCP_ST
    // chunk boundary (every wave is done with the planes: the barrier that ended the last stage): the next chunk's planes, then the
    // voxel fragments of its first stage
    if (cc + 1 < c_end) {
      hstash(hpre);
      BCP_LDS_BARRIER();
      const unsigned short* Xc;
      a_addr(0, Xc);
#pragma unroll
      for (int s = 0; s < PL; ++s) fa[0][s] = *reinterpret_cast<const frag_t*>(Xc + s * XPLANE);
    }
  }
  BCP_TS(60);

  // epilogue: lane (li, lg) holds voxel m0 + wave*16 + li, channels lg*4 .. lg*4+3 of each n-tile: 16-byte stores into flat rows
  double s1[NT][4], s2[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.0; s2[nt][r] = 0.0; }
  const bool vec = cout0 + CT <= cd.Cout && (cd.Cout & 3) == 0;     // uniform
  const bool want_stats = st.partial != nullptr;
  if (mv < V) {
    float* yrow = Y + ((long long)n * V + mv) * cd.Cout + cout0 + lg * 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cout0 + nt * 16 + lg * 4 + r;
        float v = acc[nt][r] * osc + ((bias && co < cd.Cout) ? bias[co] : 0.f);
        if (accumulate && co < cd.Cout) v += yrow[nt * 16 + r];
        acc[nt][r] = v;
        if (want_stats && co < cd.Cout) { s1[nt][r] += (double)v; s2[nt][r] += (double)v * (double)v; }
      }
      if (vec) st4(yrow + nt * 16, make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]));
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (cout0 + nt * 16 + lg * 4 + r < cd.Cout) yrow[nt * 16 + r] = acc[nt][r];
      }
    }
  }
  if (want_stats) {
    const int gg = bx / st.tiles_per_group, row = bx % st.tiles_per_group;
    BCP_LDS_BARRIER();
    stats_flush_t<NT>(s1, s2, Ss, st.partial + ((long long)gg * st.rows + row) * st.C * 2, cout0, cd.Cout);
  }
  BCP_TS(61);
  BCP_TSR(62);
}

__global__ __launch_bounds__(256) void k_b6_sum_slabs(const float* __restrict__ part, int K, long long n, int Cout,
                                                      const float* __restrict__ bias, float* __restrict__ y, int accumulate) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = bias ? bias[i % Cout] : 0.f;
    for (int k = 0; k < K; ++k) v += part[(long long)k * n + i];
    y[i] = accumulate ? y[i] + v : v;
  }
}

// tile configurations whose dgrad launches carry the backward-statistics epilogue (bcp_conv3_dgrad_bwdstats): the conv -> conv edges of
// the V-Nets' 32- / 64-channel levels and of the U-Net's decoder blocks; everything else answers "not available" and takes the plain dgrad
template <int KD, int TD, int TH, int TW, int NT, int SP>
constexpr bool b6_has_bw() {
  return (KD == 3 && TD == 4 && TH == 8 && TW == 8 && NT == 2 && SP == 1) || (KD == 3 && TD == 4 && TH == 4 && TW == 8 && NT == 2 && SP == 1) ||
         (KD == 3 && TD == 4 && TH == 4 && TW == 4 && NT == 4 && SP == 1) || (KD == 1 && TD == 1 && TH == 16 && TW == 16 && NT == 2 && SP == 1) ||
         (KD == 1 && TD == 1 && TH == 8 && TW == 16 && NT == 2 && SP == 2);
  // (not the 2-D 8x16 x 64-channel-slab instance: with the epilogue it needs 256 + 48 registers = one workgroup per CU)
}

// tile configurations that have a two-plane fp16 instance (PL = 2; selected when the launch carries the input's |max| and option conv3_f16
// allows it): round 4 -- the direct-weight kernel k_c3d (16- / 32-channel slabs, 3-D and 2-D)
template <int KD, int TD, int TH, int TW, int NT, int SP>
constexpr bool b6_has_f16() {
  return SP == 1 && ((KD == 3 && TD == 4 && TH == 8 && TW == 8 && (NT == 1 || NT == 2)) || (KD == 3 && TD == 4 && TH == 4 && TW == 8 && NT == 2) ||
                     (KD == 1 && TD == 1 && TH == 16 && TW == 16 && (NT == 1 || NT == 2)) ||
                     (KD == 1 && TD == 1 && TH == 8 && TW == 16 && NT == 4) ||      // k_c3b, the U-Net's 64-channel-slab kernel
                     (KD == 3 && TD * TH * TW == 64 && NT == 4));        // k_c3p (64-voxel x 64-channel pipeline, 4x4x4 / 2x8x4 bricks; not its register-staged twins)
}

// dynamic LDS of k_c3p: two halo buffers, three weight slots, statistics scratch
template <class TL, int PL = 3>
static constexpr size_t kC3pLds = (size_t)2 * PL * TL::HV * XSB * 2 + (size_t)3 * PL * 64 * 32 * 2 + (size_t)4 * 32 * 2 * sizeof(double);

// measurement record (bcp_conv3_fwd_planes): how many operand planes the last launcher call on this thread chose (dry runs included)
thread_local int g_b6_last_planes = 3;

template <int KD, int TD, int TH, int TW, int NT, int SP>
static int b6_launch(const float* X, const float* Wp, const float* bias, float* Y, ConvDims cd, int accumulate, float* ws,
                     double* stat_partial, int G, bool dry, hipStream_t s, int* raw_sk, const BwdStatsIn* bw) {
  using TL = Tile<KD, TD, TH, TW>;
  constexpr int CT = NT * 16;
  // k_c3d only where a wave's weight traffic is small next to its MFMAs: 256-voxel tiles with a 32-channel slab (6 KB per 48 MFMAs;
  // the 64-voxel / 64-channel instances would pull 12 KB per 24 MFMAs through L1: 79-84 vs 47-51 us)
  const bool direct = options().conv3_b6_direct != 0 && (options().conv3_b6_direct >= 2 || (TL::MT == 4 && NT <= 2));
  // two fp16 planes instead of three bf16 ones: the launch must carry the input's |max| (cd.xamax) and the instance must exist
  const bool pipe64 = TL::M == 64 && NT == 4 && SP == 1 && options().conv3_b6_w22 != 0 && options().conv3_b6_pipe != 0;      // -> k_c3p
  const bool staged2d = KD == 1 && NT == 4;                                                       // -> k_c3b itself
  const bool use_f16 = b6_has_f16<KD, TD, TH, TW, NT, SP>() && (staged2d || (NT == 4 ? pipe64 : direct)) && cd.xamax != nullptr && options().conv3_f16 != 0;
  if (!use_f16) cd.xamax = nullptr;
  const int npl = use_f16 ? 2 : 3;
  g_b6_last_planes = npl;
  const size_t lds = (size_t)npl * TL::HV * XSB * 2 + (direct ? 0 : (size_t)2 * npl * SP * CT * 32 * 2) + (size_t)4 * CT * 2 * sizeof(double);
  cd.tiles_d = cdiv(cd.D, TD); cd.tiles_h = cdiv(cd.H, TH); cd.tiles_w = cdiv(cd.W, TW);
  auto kfn = k_c3b<KD, TD, TH, TW, NT, SP>;
  if constexpr (KD == 1 && NT == 4 && b6_has_f16<KD, TD, TH, TW, NT, SP>()) {
    if (use_f16) kfn = k_c3b<KD, TD, TH, TW, NT, SP, false, 2>;       // the U-Net's 64-channel slabs on two fp16 planes
  }
  if (lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int gx = cd.N * cd.tiles_d * cd.tiles_h * cd.tiles_w, gy = cd.Cout16 / CT;
  const int nch = cd.Cin16 / 16;
  int sk = 1;
  const bool ws_fits = ws && (long long)cd.N * cd.D * cd.H * cd.W * cd.Cout <= options().conv3_sk_elems;     // bcp_conv3_fwd_workspace_bytes
  if (ws_fits && (long long)gx * gy <= 256 && nch >= 4) { sk = nch / 2; if (sk > 4) sk = 4; }
  { const int f = options().splitk; if (ws_fits && f >= 1 && f <= 4 && f <= nch) sk = f; }
  if (raw_sk) {
    // raw mode (bcp_conv3_fwd_raw): the sk partial slabs go to Y = float[sk][N*D*H*W*Cout] and stay there -- no bias, no slab sum, no
    // statistics; the consumer (bcp_norm_fwd_slabs / bcp_norm_bwd_slabs) sums them on its way in
    *raw_sk = sk;
    if (dry) return 0;
    StatsArg none{nullptr, 0, 1, cd.Cout, 1};
    if (direct && NT != 1) {                           // (16-channel slabs: the persistent form of k_c3d is not a slab writer; staged kernel)
      auto kd = k_c3d<KD, TD, TH, TW, NT, false>;
      if constexpr (b6_has_f16<KD, TD, TH, TW, NT, SP>()) {
        if (use_f16) kd = k_c3d<KD, TD, TH, TW, NT, false, false, 2>;
      }
      hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kd, dim3(gx, gy, sk), dim3(256), lds, s, X, Wp, (const float*)nullptr, Y, cd, gx, 0, none);
    } else {
      // the STAGED kernel always needs its weight stage: `lds` left it out when `direct` chose k_c3d (16-channel slabs
      // come here with direct && NT == 1 and overran their allocation by the 6 KB stage)
      size_t lds_k = lds + (direct ? (size_t)2 * npl * SP * CT * 32 * 2 : 0);
      if constexpr (TL::M == 64 && NT == 4 && SP == 1) {
        if (options().conv3_b6_w22 != 0) kfn = k_c3h<KD, TD, TH, TW>;
        if (options().conv3_b6_w22 != 0 && options().conv3_b6_pipe != 0) {
          kfn = k_c3p<KD, TD, TH, TW>; lds_k = kC3pLds<TL>;
          if constexpr (b6_has_f16<KD, TD, TH, TW, NT, SP>()) {
            if (use_f16) { kfn = k_c3p<KD, TD, TH, TW, false, 2>; lds_k = kC3pLds<TL, 2>; }
          }
        }
      }
      if (lds_k > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);
      hipLaunchKernelGGL(kfn, dim3(gx, gy, sk), dim3(256), lds_k, s, X, Wp, (const float*)nullptr, Y, cd, 0, none);
    }
    return 0;
  }
  StatsArg st{nullptr, 0, 1, cd.Cout, G > 0 ? G : 1};
  if (sk == 1 && G > 0 && gx % G == 0) { st.rows = gx / G; st.tiles_per_group = gx / G; st.partial = stat_partial; }
  if (bw) {                                           // backward statistics in the epilogue (bcp_conv3_dgrad_bwdstats): one-pass launches only
    if (!b6_has_bw<KD, TD, TH, TW, NT, SP>() || sk != 1 || G <= 0 || gx % G || (direct && NT == 1)) return 0;
    if (options().fuse_bwd_stats == 2 && KD == 3 && NT == 2) return 0;      // measurement: not at the 3-D 32-channel slabs (k_c3d's epilogue: +39 us per launch)
    st.by = bw->y; st.bstats = bw->stats; st.bact = bw->act;
  }
  if (direct) {
    // persistent grid (16-channel slabs): tiles dealt round-robin to P workgroups (two per CU), balanced: every workgroup gets
    // cdiv(tiles, slots) tiles; 32-channel slabs: one tile per workgroup (register budget, see k_c3d)
    constexpr bool PER = NT == 1;
    const int slots = 512 / (gy * sk) > 0 ? 512 / (gy * sk) : 1;
    const int per = PER ? cdiv(gx, slots) : 1;
    int P = cdiv(gx, per);
    if (PER && (cd.xcd & 2) && P >= 8) P = (P + 7) & ~7;                                           // the XCD-aware walk needs whole rounds of the 8 XCDs
    if (PER && options().conv3_p > 0 && options().conv3_p < P) P = options().conv3_p;      // tests: few workgroups, many tiles each
    StatsArg sd{nullptr, 0, 1, cd.Cout, G > 0 ? G : 1};
    if (sk == 1 && G > 0 && gx % G == 0) { sd.rows = PER ? P : gx / G; sd.tiles_per_group = gx / G; sd.partial = stat_partial; }
    if (bw) { sd.by = bw->y; sd.bstats = bw->stats; sd.bact = bw->act; }
    if (dry) return (sk == 1 && G > 0 && gx % G == 0) ? (PER ? P : gx / G) : 0;
    auto kd = k_c3d<KD, TD, TH, TW, NT, PER>;
    if constexpr (b6_has_bw<KD, TD, TH, TW, NT, SP>() && !PER) {
      if (bw) kd = k_c3d<KD, TD, TH, TW, NT, PER, true>;
    }
    if constexpr (b6_has_f16<KD, TD, TH, TW, NT, SP>()) {
      if (use_f16) {
        kd = k_c3d<KD, TD, TH, TW, NT, PER, false, 2>;
        if constexpr (b6_has_bw<KD, TD, TH, TW, NT, SP>() && !PER) {
          if (bw) kd = k_c3d<KD, TD, TH, TW, NT, PER, true, 2>;
        }
      }
    }
    hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (sk == 1) {
      hipLaunchKernelGGL(kd, dim3(P, gy, 1), dim3(256), lds, s, X, Wp, bias, Y, cd, gx, accumulate, sd);
    } else {
      const long long n = (long long)cd.N * cd.D * cd.H * cd.W * cd.Cout;
      StatsArg none{nullptr, 0, 1, cd.Cout, 1};
      hipLaunchKernelGGL(kd, dim3(P, gy, sk), dim3(256), lds, s, X, Wp, (const float*)nullptr, ws, cd, gx, 0, none);
      hipLaunchKernelGGL(k_b6_sum_slabs, dim3((int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, s, ws, sk, n, cd.Cout, bias, Y,
                         accumulate);
    }
    return sd.partial ? sd.rows : 0;
  }
  if (dry) return sk == 1 && G > 0 && gx % G == 0 ? gx / G : 0;
  if constexpr (b6_has_bw<KD, TD, TH, TW, NT, SP>()) {
    if (bw) {
      kfn = k_c3b<KD, TD, TH, TW, NT, SP, true>;
      if (lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
  }
  size_t lds_k = lds;
  if constexpr (TL::M == 64 && NT == 4 && SP == 1) {
    if (options().conv3_b6_w22 != 0) {
      kfn = k_c3h<KD, TD, TH, TW>;       // 2 x 2 wave arrangement of the 64 x 64 tile (same LDS layout and size)
      if constexpr (b6_has_bw<KD, TD, TH, TW, NT, SP>()) {
        if (bw) kfn = k_c3h<KD, TD, TH, TW, true>;
      }
      if (options().conv3_b6_pipe != 0) {   // the same tile as an LDS-DMA software pipeline (its own LDS layout)
        kfn = k_c3p<KD, TD, TH, TW>;
        if constexpr (b6_has_bw<KD, TD, TH, TW, NT, SP>()) {
          if (bw) kfn = k_c3p<KD, TD, TH, TW, true>;
        }
        lds_k = kC3pLds<TL>;
        if constexpr (b6_has_f16<KD, TD, TH, TW, NT, SP>()) {
          if (use_f16) {
            kfn = k_c3p<KD, TD, TH, TW, false, 2>;
            if constexpr (b6_has_bw<KD, TD, TH, TW, NT, SP>()) {
              if (bw) kfn = k_c3p<KD, TD, TH, TW, true, 2>;
            }
            lds_k = kC3pLds<TL, 2>;
          }
        }
      }
      if (lds_k > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);
    }
  }
  if (sk == 1) {
    hipLaunchKernelGGL(kfn, dim3(gx, gy, 1), dim3(256), lds_k, s, X, Wp, bias, Y, cd, accumulate, st);
  } else {
    const long long n = (long long)cd.N * cd.D * cd.H * cd.W * cd.Cout;
    StatsArg none{nullptr, 0, 1, cd.Cout, 1};
    hipLaunchKernelGGL(kfn, dim3(gx, gy, sk), dim3(256), lds_k, s, X, Wp, (const float*)nullptr, ws, cd, 0, none);
    hipLaunchKernelGGL(k_b6_sum_slabs, dim3((int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, s, ws, sk, n, cd.Cout, bias, Y,
                       accumulate);
  }
  return st.partial ? st.rows : 0;
}

static constexpr int kB6FlatAvMax = 384;    // flat halo rows (BM + 2 R) the flat instances hold: 3 x 384 x 32 B = 36 KB

template <int KD, int NT, int SP>
static int b6_launch_flat(const float* X, const float* Wp, const float* bias, float* Y, ConvDims cd, int accumulate, float* ws,
                          double* stat_partial, int G, bool dry, hipStream_t s, int* raw_sk, const BwdStatsIn* bw) {
  constexpr int CT = NT * 16, BM = 64;
  const int V = cd.D * cd.H * cd.W, tps = cdiv(V, BM);
  size_t lds = (size_t)3 * kB6FlatAvMax * XSB * 2 + (size_t)2 * 3 * SP * CT * 32 * 2 + (size_t)4 * CT * 2 * sizeof(double);
  auto kfn = k_c3f<KD, NT, SP, kB6FlatAvMax>;
  bool use_f16 = false;
  if constexpr (KD == 3 && SP == 1) {
    if (options().conv3_b6_pipe != 0) {       // the LDS-DMA software pipeline (its own LDS layout: a zero row per plane, three weight slots)
      auto kq = k_c3q<NT, kB6FlatAvMax>;
      kfn = kq;
      lds = (size_t)3 * (kB6FlatAvMax + 1) * XSB * 2 + (size_t)3 * 3 * CT * 32 * 2 + (size_t)4 * CT * 2 * sizeof(double);
      if (cd.xamax != nullptr && options().conv3_f16 != 0) {      // two fp16 planes (round 4): the launch carries the input's |max|
        auto kq2 = k_c3q<NT, kB6FlatAvMax, 2>;
        kfn = kq2;
        lds = (size_t)2 * (kB6FlatAvMax + 1) * XSB * 2 + (size_t)3 * 2 * CT * 32 * 2 + (size_t)4 * CT * 2 * sizeof(double);
        use_f16 = true;
      }
    }
  }
  if (!use_f16) cd.xamax = nullptr;
  g_b6_last_planes = use_f16 ? 2 : 3;
  if (lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int gx = cd.N * tps, gy = cd.Cout16 / CT;
  const int nch = cd.Cin16 / 16;
  // split-K over the cin chunks: a deep-level workgroup's life is a serial chain of weight stages (one barrier per tap pair), so the
  // chain is cut as short as the grid allows -- the largest power of two (<= 8, >= 2 chunks per workgroup) that keeps the launch
  // within the 512 co-resident workgroup slots; measured alone (fwd + statistics, us): 256 channels batch 2 / 4: 31.3 -> 25.8 /
  // 40.7 -> 34.3 (with 64-channel slabs at batch 4); 128 channels batch 4: 67.2 -> 59.4 (split 2: 496 workgroups instead of 992)
  int sk = 1;
  const bool ws_fits = ws && (long long)cd.N * V * cd.Cout <= options().conv3_sk_elems;
  if (ws_fits && nch >= 4) {
    const int cap = nch / 2 < 8 ? nch / 2 : 8;
    while (sk * 2 <= cap && (long long)gx * gy * sk * 2 <= 512) sk *= 2;
  }
  { const int f = options().splitk; if (ws_fits && f >= 1 && f <= 4 && f <= nch) sk = f; }
  { const int f = options().conv3_b6_flat_sk; if (ws_fits && f >= 1 && f <= 8 && f <= nch) sk = f; }      // measurement override
  if (raw_sk) {                                       // raw mode: see b6_launch
    *raw_sk = sk;
    if (dry) return 0;
    StatsArg none{nullptr, 0, 1, cd.Cout, 1};
    hipLaunchKernelGGL(kfn, dim3(gx, gy, sk), dim3(256), lds, s, X, Wp, (const float*)nullptr, Y, cd, tps, 0, none);
    return 0;
  }
  if (bw) return 0;                                   // (deep levels: the one-launch norm backward takes the raw slabs instead)
  StatsArg st{nullptr, 0, 1, cd.Cout, G > 0 ? G : 1};
  const bool stats_ok = sk == 1 && G > 0 && cd.N % G == 0;        // tiles are sample-major and never straddle samples
  if (stats_ok) { st.rows = gx / G; st.tiles_per_group = gx / G; st.partial = stat_partial; }
  if (dry) return stats_ok ? gx / G : 0;
  if (sk == 1) {
    hipLaunchKernelGGL(kfn, dim3(gx, gy, 1), dim3(256), lds, s, X, Wp, bias, Y, cd, tps, accumulate, st);
  } else {
    const long long n = (long long)cd.N * V * cd.Cout;
    StatsArg none{nullptr, 0, 1, cd.Cout, 1};
    hipLaunchKernelGGL(kfn, dim3(gx, gy, sk), dim3(256), lds, s, X, Wp, (const float*)nullptr, ws, cd, tps, 0, none);
    hipLaunchKernelGGL(k_b6_sum_slabs, dim3((int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, s, ws, sk, n, cd.Cout, bias, Y,
                       accumulate);
  }
  return st.partial ? st.rows : 0;
}

// Forward / dgrad on the bf16 pipe where option conv3_b6 allows it.  Returns the statistics rows (as conv3_fwd_impl does);
// *handled = false leaves the shape to the fp32 kernels.
int b6_last_planes() { return g_b6_last_planes; }

int b6_fwd(const float* x, const float* wp, const float* bias, float* y, const ConvDims& cd, int KD, int accumulate, void* workspace,
           double* stat_partial, int G, bool dry, hipStream_t s, bool* handled, int* raw_sk, const BwdStatsIn* bw) {
  *handled = false;
  g_b6_last_planes = 3;
  const Options& o = options();
  if (o.conv3_b6 == 0) return 0;
  const long long vox = (long long)cd.N * cd.D * cd.H * cd.W;
  float* ws = (float*)workspace;
  int rows = 0;
  if (KD == 3) {
    const bool forced = o.conv3_b6 >= 2;
    if (cd.Cout16 == 16 && cd.Cin16 == 16) {
      if (o.conv3_b6 >= 3 || (o.conv3_b6 == 1 && (o.conv3_b6_levels & 4) && vox >= 256LL * 1024)) {
        rows = b6_launch<3, 4, 8, 8, 1, 1>(x, wp, bias, y, cd, accumulate, ws, stat_partial, G, dry, s, raw_sk, bw);
        *handled = true;
      }
    } else if (cd.Cout16 % 64 == 0) {
      if (forced || ((o.conv3_b6_levels & 2) && vox >= o.conv3_b6_minvox)) {
        // 64-channel slabs.  Mid level (>= 16 K voxels): 64-voxel 4x4x4 tiles -- they tile 28x28x20 exactly and give 490 workgroups
        // (128-voxel tiles: 280 workgroups, one per CU, nothing to overlap with: 66-71 us against 47-51 us).  Deep level
        // (14x14x10, from 2 K voxels): 2x8x4 tiles (37 % padding; 4x8x4 wastes 57 %): 40 us against 52 us for the fp32 kernel, LA step
        // 7.90 against 7.97 ms.  The 7x7x5 level stays with the fp32 kernel (35 vs 32 us).
        if (vox >= 16LL * 1024) {
          rows = b6_launch<3, 4, 4, 4, 4, 1>(x, wp, bias, y, cd, accumulate, ws, stat_partial, G, dry, s, raw_sk, bw);
        } else if (o.conv3_b6_flat && 64 + 2 * (cd.H * cd.W + cd.W + 1) <= kB6FlatAvMax) {
          // flat 64-voxel tiles; narrower slabs for the smallest volumes (7x7x5: 8 tiles) so that the grid still covers the CUs at the
          // split-K the launcher picks
          const long long gx8 = (long long)cd.N * cdiv(cd.D * cd.H * cd.W, 64) * (cd.Cout16 / 64) * 8;     // workgroups with 64-channel slabs at split 8
          const int nt = o.conv3_b6_flat >= 2 ? (o.conv3_b6_flat == 2 ? 2 : (o.conv3_b6_flat == 4 ? 4 : 1))
                                              : ((vox >= 2048 || (cd.Cin16 >= 256 && gx8 >= 512)) ? 4 : 2);
          if (nt == 4) rows = b6_launch_flat<3, 4, 1>(x, wp, bias, y, cd, accumulate, ws, stat_partial, G, dry, s, raw_sk, bw);
          else if (nt == 2) rows = b6_launch_flat<3, 2, 1>(x, wp, bias, y, cd, accumulate, ws, stat_partial, G, dry, s, raw_sk, bw);
          else rows = b6_launch_flat<3, 1, 1>(x, wp, bias, y, cd, accumulate, ws, stat_partial, G, dry, s, raw_sk, bw);
        } else {
          rows = b6_launch<3, 2, 8, 4, 4, 1>(x, wp, bias, y, cd, accumulate, ws, stat_partial, G, dry, s, raw_sk, bw);
        }
        *handled = true;
      }
    } else if (cd.Cout16 % 32 == 0) {
      if (forced || ((o.conv3_b6_levels & 1) && vox >= o.conv3_b6_minvox)) {
        if (vox >= 64LL * 1024 || cd.W % 8 == 0) rows = b6_launch<3, 4, 8, 8, 2, 1>(x, wp, bias, y, cd, accumulate, ws, stat_partial, G, dry, s, raw_sk, bw);
        else rows = b6_launch<3, 4, 4, 8, 2, 1>(x, wp, bias, y, cd, accumulate, ws, stat_partial, G, dry, s, raw_sk, bw);
        *handled = true;
      }
    }
  } else {
    // 2-D (U-Net): ACDC step 5.18 -> 4.24 ms with the 32- to 256-channel levels (and their weight gradients) on the bf16 pipe
    const bool on = o.conv3_b6 >= 2 || ((o.conv3_b6_levels & 8) && vox >= o.conv3_b6_minvox);
    if (cd.Cout16 == 16 && cd.Cin16 <= o.conv3_b6_cin16max) {
      // the U-Net's 16-channel layers at full resolution (16 -> 16 and, after the skip concatenation, 32 -> 16): 16x16 tiles on the
      // persistent direct-weight kernel, as the 3-D 16-channel level (work items = (tile, cin chunk))
      if (o.conv3_b6 >= 3 || (o.conv3_b6 == 1 && (o.conv3_b6_levels & 8) && (o.conv3_b6_levels & 4) && vox >= 256LL * 1024)) {
        rows = b6_launch<1, 1, 16, 16, 1, 1>(x, wp, bias, y, cd, accumulate, ws, stat_partial, G, dry, s, raw_sk, bw);
        *handled = true;
      }
    } else if (cd.Cout16 % 64 == 0 && on) {
      // (flat 64-pixel tiles, k_c3f<1,..>, lose here: ACDC step 4.28 / 4.44 vs 4.14 ms for the levels up to 16 K / 64 K pixels)
      rows = b6_launch<1, 1, 8, 16, 4, 1>(x, wp, bias, y, cd, accumulate, ws, stat_partial, G, dry, s, raw_sk, bw);
      *handled = true;
    } else if (cd.Cout16 % 32 == 0 && on) {
      // 32-channel slabs: from 64 K pixels on 16x16 tiles with direct weight fragments (k_c3d, as the 3-D 32-channel level) -- the
      // staged 8x16 kernel ran the 16 -> 32 dgrad at 256x256 at 54 TFLOP/s-eq; ACDC step 4.11 -> 4.05 ms (cfg2d: 0 staged, 2 always direct)
      if (o.conv3_b6_cfg2d >= 2 || (o.conv3_b6_cfg2d == 1 && vox >= 64LL * 1024))
        rows = b6_launch<1, 1, 16, 16, 2, 1>(x, wp, bias, y, cd, accumulate, ws, stat_partial, G, dry, s, raw_sk, bw);
      else rows = b6_launch<1, 1, 8, 16, 2, 2>(x, wp, bias, y, cd, accumulate, ws, stat_partial, G, dry, s, raw_sk, bw);
      *handled = true;
    }
  }
  return rows;
}

}  // namespace bcp
