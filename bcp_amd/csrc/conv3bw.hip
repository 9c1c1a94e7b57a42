// bcp_amd/csrc/conv3bw.hip -- weight gradient of the 3x3x3 / 3x3 convolution with fp32 numerics on the BF16 matrix pipe.
//
//     dW[tap][ci][co] = sum over voxels v of  X[v + off(tap)][ci] * dY[v][co]
//
// Same arithmetic as conv3b.hip: both operands enter the LDS as THREE bf16 pieces (x = p0 + p1 + p2 up to 2^-26 |x|) and every
// K = 32 block takes six v_mfma_f32_16x16x32_bf16 (a0 b0 + a0 b1 + a1 b0 + a0 b2 + a1 b1 + a2 b0, fp32 accumulation in the matrix
// core) -- fp32-equivalent, ~2.5x fewer matrix-pipe cycles than the eight v_mfma_f32_16x16x4_f32 it replaces, and bf16 MFMAs leave
// issue slots for the LDS / VALU work around them, which fp32 MFMAs on gfx950 do not (tools/probe/overlap_probe.hip).
//
// GEMM view per tap: rows = 16 input channels (one chunk), columns = 16 * NT output channels, K = voxels.  The MFMA wants, per
// lane, 8 consecutive K values of ONE channel -- but both tensors are channel-innermost, in HBM and in the LDS ([voxel][16 ci]
// halo planes as in conv3b.hip, [voxel][CT co] gradient-tile planes).  ds_read_b64_tr_b16 closes the gap: within a 16-lane group,
// lane i receives as element j the (i & 3)-th bf16 of the 8 bytes addressed by lane 4 j + (i >> 2) (measured:
// tools/probe/tr_probe.hip).  With lane a pointing at (voxel a >> 2, channel quad a & 3) every lane i ends up with channel i of
// four consecutive voxels -- a transposed fragment straight out of the row-major image, and a tap shift is just a different row
// address (no shifted copies, no alignment problem: rows are 32 bytes).  Two such reads make one 8-element MFMA operand; K slot
// r * 4 + j of lane group lg is voxel r * 16 + lg * 4 + j of the 32-voxel K block, for X and dY alike, which keeps each read's 16
// voxels contiguous (conflict-free).
//
// Workgroup = 256 threads = 4 waves; the waves split the taps (wave w owns taps w, w + 4, ...), every wave walks all voxels of the
// tile; a workgroup loops over a group of spatial tiles for one (cin chunk, cout slab) and writes ONE partial [T][16][CT] slab,
// summed by k_wgrad_reduce(_deep) of conv3.hip (deterministic, no atomics).  Two workgroups per CU.
//
// Reference op: autograd of nn.Conv3d(k=3,pad=1) / nn.Conv2d(k=3,pad=1) (networks/VNet.py:17, networks/unet.py:19-25).
#include "conv3_defs.h"
#include "../../include/bcp_hip.h"

namespace bcp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#ifndef BCP_DS_READ_TR16_B64      // (the host simulator supplies its own)
typedef short bcp_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 ds_read_tr16(const unsigned short* p) {
  return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bcp_s16x4*)p));
}
#else
__device__ __forceinline__ bf16x4 ds_read_tr16(const unsigned short* p) {
  const unsigned long long u = BCP_DS_READ_TR16_B64(p);
  return __builtin_bit_cast(bf16x4, u);
}
#endif
__device__ __forceinline__ bf16x8 cat8(bf16x4 lo, bf16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

template <class TL>
constexpr int w6_voff(int m) {                       // Tile::voff as a constant expression
  return ((m / (TL::TW_ * TL::TH_)) * TL::HH + (m / TL::TW_) % TL::TH_) * TL::HW + m % TL::TW_;
}
template <class TL>
constexpr bool w6_voff_linear() {
  for (int m = 0; m < TL::M; ++m)
    if (w6_voff<TL>(m) != w6_voff<TL>(m % 32) + (m / 32) * w6_voff<TL>(32)) return false;
  return true;
}

// PL (round 4): 3 = three bf16 planes per operand, six MFMAs per K block; 2 = two fp16 planes pre-scaled by powers of two from the
// tensors' own maxima (cd.xamax for X, cd.yamax for dY -- left by the norm passes that wrote them), three MFMAs per K block; the
// partial slabs are scaled back on the way out (conv3_defs.h).  ds_read_b64_tr_b16 transposes 16-bit elements of either type.
template <int PL> struct W6Pipe;
template <> struct W6Pipe<3> {
  using frag = bf16x8; using half4 = bf16x4;
  static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ void split(const float4& v, float, unsigned short* base, int ps) { split_store4(v, base, ps); }
};
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <> struct W6Pipe<2> {
  using frag = f16x8; using half4 = f16x4;
  static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ void split(const float4& v, float s, unsigned short* base, int ps) { split_store4_f16(v, s, base, ps); }
};

// DIR (round 6, the deep levels): ONE tile group per (cin chunk, cout slab) -- the workgroup walks every tile of the launch and writes its
// finished [CT][16][T] block STRAIGHT into the torch gradient dW[co][ci][tap] (transposed through the LDS the planes no longer need: rows of
// 16 * T contiguous floats per output channel), `partial` unused: no slab is written, re-read or reduced and no reduce kernel follows.
// Rounds 2-5 cut the 7x7x5 x 256 / 14x14x10 x 128 launches into 4 / 16 tile groups to fill 512 workgroup slots: 28 MB of partial slabs
// written and read back for a 7 MB / 1.8 MB gradient (67.6 / 57.8 MB per launch in profiles/r05_pmc_ops.json against 8.1 / 5.8 MB of
// operands + result).
template <int KD, int TD, int TH, int TW, int NT, int PL = 3, bool DIR = false>
__global__ __launch_bounds__(256) void k_w6(const float* __restrict__ X, const float* __restrict__ dY, float* __restrict__ partial,
                                            ConvDims cd, int tiles_total, int tiles_per_group, int accumulate) {
  using TL = Tile<KD, TD, TH, TW>;
  using PP = W6Pipe<PL>;
  using frag_t = typename PP::frag;
  auto rd = [](const unsigned short* lo, const unsigned short* hi) __attribute__((always_inline)) -> frag_t {
    return __builtin_bit_cast(frag_t, cat8(ds_read_tr16(lo), ds_read_tr16(hi)));
  };
  constexpr int T = TL::T, CT = NT * 16, M = TL::M, KB = M / 32;
  constexpr int TPW = (T + 3) / 4;                      // taps per wave
  constexpr int XPLANE = TL::HV * 16;                   // bf16 elements per halo piece plane ([HV][16])
  constexpr int YS = CT + 16;                           // gradient-tile row stride (bank spread of the transposed reads)
  constexpr int YPLANE = M * YS;
  constexpr int NY4 = (M * (CT / 4) + 255) / 256;       // dY-tile float4 per thread
  static_assert(TW % 4 == 0 && M % 32 == 0, "a transposed read covers 4 consecutive voxels of a W row");
  using HF = HaloFetch<TL>;

  HIP_DYNAMIC_SHARED(float4, smem4)
  unsigned short* Xb = reinterpret_cast<unsigned short*>(smem4);   // [PL][HV][16]
  unsigned short* Yb = Xb + PL * XPLANE;                           // [PL][M][YS]
  float xsc = 1.f, ysc = 1.f, osc = 1.f;
  if (PL == 2) {
    const int ex = f16_scale_exp(amax_read(cd.xamax)), ey = f16_scale_exp(amax_read(cd.yamax));
    xsc = ldexpf(1.f, ex); ysc = ldexpf(1.f, ey); osc = ldexpf(1.f, -(ex + ey));
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  // work item (tile group, cin chunk, cout slab).  XCD-aware (cd.xcd): the workgroups an XCD receives (linear id % 8) take a
  // CONTIGUOUS range of the item list ordered (group, cout slab, cin chunk) with the cin chunk fastest -- the items of one tile
  // group read the same dY tile and the two 64-byte halves of the same X lines, and neighbouring groups share halo planes: on one
  // XCD, at about the same time, they meet in its L2 (grid order put them 5 XCDs apart: X fetched ~2x, dY once per cin chunk)
  int cc = blockIdx.y, cslab = blockIdx.z, grp = blockIdx.x;
  if (cd.xcd) {
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int w = xcd_tile(lin, gridDim.x * gridDim.y * gridDim.z, 0);
    cc = w % (int)gridDim.y;
    cslab = (w / (int)gridDim.y) % (int)gridDim.z;
    grp = w / (int)(gridDim.y * gridDim.z);
  }
  const int cout0 = cslab * CT;

  f32x4 acc[TPW][NT];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane source addresses of the transposed reads: lane (li, lg) points at voxel r*16 + lg*4 + (li >> 2) of K block kb,
  // channel quad li & 3
  // A K block is 32 consecutive tile voxels and every tile shape used here has 32 | TW * TH (or, 2-D, 32 = two W rows), so the
  // halo offset of voxel kb * 32 + j is voff(j) + kb * voff(32): two base addresses per operand and a constant step -- no
  // per-block address table (indexed by the runtime kb it lived in scratch memory for the TW = 4 tiles: two scratch loads at the
  // head of every K block, on the critical path of the LDS reads).
  constexpr int XSTEP = w6_voff<TL>(32) * 16, YSTEP = 32 * YS;
  static_assert(w6_voff_linear<TL>(), "k_w6: halo offsets must be linear in the K block");
  int xo[2], yo[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int m = r * 16 + lg * 4 + (li >> 2);
    xo[r] = TL::voff(m) * 16 + (li & 3) * 4;
    yo[r] = m * YS + (li & 3) * 4;
  }
  int toff[TPW];                                     // halo row offset of this wave's taps (a slot past T recomputes tap 0: never stored)
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = wave + 4 * t;
    toff[t] = (tap < T ? TL::tapoff(tap) : 0) * 16;
  }

  int t_end = (grp + 1) * tiles_per_group;
  if (t_end > tiles_total) t_end = tiles_total;
  int tile = grp * tiles_per_group;
  if (tile >= t_end) return;

  HF hf;
  hf.init(cd, reinterpret_cast<float*>(smem4));      // (its fp32 LDS slot is not used)
  const bool slab_full = cout0 + CT <= cd.Cout;
  float4 px[HF::NP], py[NY4];
  unsigned xvm = 0, yvm = 0;                          // validity bits of the registers in flight
  // global -> registers, branch-free (see HaloFetch::fetch_nb): rows outside the volume read element 0 and are zeroed at the stash
  auto fetch = [&](int tl) __attribute__((always_inline)) {
    int n, d0, h0, w0;
    tile_origin(cd, tl, TD, TH, TW, n, d0, h0, w0);
    xvm = hf.fetch_nb(X, cd, n, d0, h0, w0, cc, px);
    yvm = 0;
#pragma unroll
    for (int u = 0; u < NY4; ++u) {
      const int q = (threadIdx.x + u * 256) % (M * (CT / 4));
      const int m = q / (CT / 4), c4 = q % (CT / 4);
      const int tw = m % TW, th = (m / TW) % TH, td = m / (TW * TH);
      const int d = d0 + td, h = h0 + th, w = w0 + tw;
      const int co = cout0 + c4 * 4;
      const unsigned ok = (d < cd.D && h < cd.H && w < cd.W && (slab_full || co < cd.Cout)) ? 1u : 0u;
      const unsigned off = ok ? (unsigned)(((((long long)n * cd.D + d) * cd.H + h) * cd.W + w) * cd.Cout + co) : 0u;
      py[u] = ld4(dY + off);
      yvm |= ok << u;
    }
  };
  auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < HF::NP; ++u)
      if (hf.act && u * HF::RPP + hf.r0 < HF::HR) {
        const float4 v = ((xvm >> u) & 1u) ? px[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        PP::split(v, xsc, Xb + ((u * HF::RPP + hf.r0) * TL::HW + hf.hw) * 16 + hf.part * 4, XPLANE);
      }
#pragma unroll
    for (int u = 0; u < NY4; ++u) {
      const int q = threadIdx.x + u * 256;
      if (q < M * (CT / 4)) {
        const float4 v = ((yvm >> u) & 1u) ? py[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        PP::split(v, ysc, Yb + (q / (CT / 4)) * YS + (q % (CT / 4)) * 4, YPLANE);
      }
    }
  };
  fetch(tile);
  stash();
  BCP_LDS_BARRIER();
  for (;;) {
    const bool has_next = tile + 1 < t_end;
    fetch(has_next ? tile + 1 : tile);       // (the last tile re-reads itself: no conditional load in the loop)
    // A fragments as a prefetched STREAM (round 3, as k_c3d): hipcc issued the six transposed reads of a tap right in front of its
    // MFMAs (read, s_waitcnt, MFMA: one LDS round trip exposed per tap); here tap t + 1's fragments -- tap 0 of the next K block after
    // the last tap -- are requested before tap t's MFMAs and sched_barrier keeps them there.
    frag_t an[PL];
#pragma unroll
    for (int s = 0; s < PL; ++s) an[s] = rd(Xb + s * XPLANE + xo[0] + toff[0], Xb + s * XPLANE + xo[1] + toff[0]);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const unsigned short* Yk = Yb + kb * YSTEP;
      const unsigned short* Xk = Xb + kb * XSTEP;
      frag_t b[NT][PL];
#pragma unroll
      for (int s = 0; s < PL; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          b[nt][s] = rd(Yk + s * YPLANE + yo[0] + nt * 16, Yk + s * YPLANE + yo[1] + nt * 16);
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        frag_t a[PL];
#pragma unroll
        for (int s = 0; s < PL; ++s) a[s] = an[s];
        if (t + 1 < TPW) {
#pragma unroll
          for (int s = 0; s < PL; ++s) an[s] = rd(Xk + s * XPLANE + xo[0] + toff[t + 1], Xk + s * XPLANE + xo[1] + toff[t + 1]);
        } else if (kb + 1 < KB) {
#pragma unroll
          for (int s = 0; s < PL; ++s) an[s] = rd(Xk + XSTEP + s * XPLANE + xo[0] + toff[0], Xk + XSTEP + s * XPLANE + xo[1] + toff[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // rows = input channels (A = X^T fragment), columns = output channels (B = dY fragment); smallest terms first
#define BCP_W6(I, J)                                                                                                   \
  _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                                      \
      acc[t][nt] = PP::mfma(a[I], b[nt][J], acc[t][nt]);
        if constexpr (PL == 3) { BCP_W6(2, 0) BCP_W6(1, 1) BCP_W6(0, 2) BCP_W6(1, 0) BCP_W6(0, 1) BCP_W6(0, 0) }
        else { BCP_W6(1, 0) BCP_W6(0, 1) BCP_W6(0, 0) }
#undef BCP_W6
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!has_next) break;
    BCP_LDS_BARRIER();                        // every wave is done reading the planes
    stash();
    BCP_LDS_BARRIER();
    ++tile;
  }
  if constexpr (DIR) {
    // dW[co][ci][tap] (+)= acc: lane (li, lg) holds ci = lg*4 + r, co = nt*16 + li of its wave's taps.  LDS image [CT][16 * T (+1: the 16
    // output channels of a store instruction land in 16 different banks)], then 256 threads sweep each output channel's 16 * T contiguous
    // floats (a 1728-byte run of dW at T = 27)
    constexpr int RS = 16 * T + 1;
    float* Ts = reinterpret_cast<float*>(smem4);
    BCP_LDS_BARRIER();                          // every wave is done reading the planes
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tap = wave + 4 * t;
      if (tap < T) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) Ts[(nt * 16 + li) * RS + (lg * 4 + r) * T + tap] = acc[t][nt][r] * osc;
      }
    }
    BCP_LDS_BARRIER();
    float* dW = partial;                        // (DIR: the argument IS the gradient tensor)
    const int ci_n = cd.Cin - cc * 16 < 16 ? cd.Cin - cc * 16 : 16;
    for (int q = threadIdx.x; q < CT * 16 * T; q += 256) {
      const int col = q / (16 * T), e = q - col * (16 * T);
      if (cout0 + col < cd.Cout && e < ci_n * T) {
        float* o = dW + ((long long)(cout0 + col) * cd.Cin + cc * 16) * T + e;
        const float v = Ts[col * RS + e];
        *o = accumulate ? (*o + v) : v;
      }
    }
    return;
  }
  // partial[grp][tap][ci][co]: lane (li, lg) holds ci = lg*4 + r (rows), co = li (columns)
  float* P = partial + (long long)grp * T * cd.Cin16 * cd.Cout16;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = wave + 4 * t;
    if (tap < T) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          P[((long long)tap * cd.Cin16 + cc * 16 + lg * 4 + r) * cd.Cout16 + cout0 + nt * 16 + li] = acc[t][nt][r] * osc;
    }
  }
}

struct W6Plan { int cfg, TD, TH, TW, NT, groups; bool direct, deep; };

static bool w6_plan(W6Plan& p, const ConvDims& cd, int KD) {
  const Options& o = options();
  if (o.wgrad_b6 == 0 || cd.Cin16 % 16 || cd.Cout16 % 16) return false;
  if (cd.Cout16 % 32 && !(o.wgrad_b6 >= 2 || (o.wgrad_b6_levels & 4))) return false;      // 16-channel slabs: see wgrad_b6_levels
  const long long vox = (long long)cd.N * cd.D * cd.H * cd.W;
  const bool forced = o.wgrad_b6 >= 2;
  if (KD == 3) {
    if (!forced && vox < o.wgrad_b6_minvox) return false;
    if (cd.W % 8 == 0 || cd.W >= 32) { p.cfg = 0; p.TD = 4; p.TH = 4; p.TW = 8; }      // 128 voxels
    else { p.cfg = 1; p.TD = 4; p.TH = 8; p.TW = 4; }
    if (vox < 16LL * 1024) { p.cfg = 2; p.TD = 2; p.TH = 8; p.TW = 4; }                 // deep level: 64 voxels
  } else {
    if (!forced && !((o.wgrad_b6_levels & 8) && vox >= o.wgrad_b6_minvox)) return false;
    p.cfg = 3; p.TD = 1; p.TH = 8; p.TW = 16;
  }
  p.NT = cd.Cout16 % 32 ? 1 : 2;
  p.direct = false;
  p.deep = false;
  int slots = o.wgrad_b6_slots > 0 ? o.wgrad_b6_slots : 512;
  if (p.cfg == 2 && o.wgrad_b6_deep) {
    // round 6, deep levels (< 16 K voxels: a few dozen tiles): the matrix work is microseconds, what the launch costs is the partial
    // slabs.  Few tile groups -- ONE where the channel blocks alone fill the slots (256 -> 256: 16 x 16 blocks of 16 x 16 channels),
    // then the kernel writes the gradient itself (k_w6<..., DIR>); narrow slabs (NT = 1) double the blocks.  Measured alone
    // (tools/probe/wgrad_deep_probe.py, gpurun_out/r06_s2): 2x7x7x5 x 256 35.5 -> 23.7 us (67.6 -> 8.1 MB), 2x14x14x10 x 128 42.6 -> 38.2
    // (four groups: 57.8 -> ~20 MB), pancreas 2x6^3 x 256 32.1 -> 23.9, 2x12^3 x 128 41.6 -> 32.2.  What did NOT help, each measured on
    // the way: two / three register sets of tiles in flight (hipcc drains vmcnt at the loop head), the operands by LDS-DMA two to four
    // tiles ahead (5-8 us SLOWER: the read-back is more LDS traffic), LDS fragment reads two / three tap steps ahead -- the tile walk is
    // bound by its LDS reads (four transposed reads per three MFMAs at NT = 1) and ~1 us of stash / barriers per tile, not by a latency
    p.deep = true;
    p.NT = (o.wgrad_b6_deep_nt == 2 && cd.Cout16 % 32 == 0) ? 2 : 1;
    slots = o.wgrad_b6_deep_slots > 0 ? o.wgrad_b6_deep_slots : 256;
    if (o.wgrad_b6_deep_tile == 1) { p.cfg = 1; p.TD = 4; p.TH = 8; p.TW = 4; }      // 128-voxel tiles: half the tiles to walk
  }
  const int tiles = cd.N * cdiv(cd.D, p.TD) * cdiv(cd.H, p.TH) * cdiv(cd.W, p.TW);
  const int chan_blocks = (cd.Cin16 / 16) * (cd.Cout16 / (p.NT * 16));
  int g = cdiv(slots, chan_blocks);
  if (g > tiles) g = tiles;
  if (g < 1) g = 1;
  const int tpg = cdiv(tiles, g);
  p.groups = cdiv(tiles, tpg);
  p.direct = p.deep && p.groups == 1;
  return true;
}

size_t b6_wgrad_workspace_bytes(const ConvDims& cd, int KD) {
  W6Plan p;
  if (!w6_plan(p, cd, KD)) return 0;
  return (size_t)p.groups * KD * 9 * cd.Cin16 * cd.Cout16 * sizeof(float);
}

template <int KD, int TD, int TH, int TW, int NT, bool DIR = false>
static int w6_launch(const float* X, const float* dY, float* partial, ConvDims cd, int groups, int accumulate, hipStream_t s) {
  using TL = Tile<KD, TD, TH, TW>;
  constexpr int CT = NT * 16;
  const bool f16 = cd.xamax != nullptr && cd.yamax != nullptr && options().conv3_f16 != 0;      // both operands' |max| known: two fp16 planes
  size_t lds = (size_t)(f16 ? 2 : 3) * (TL::HV * 16 * 2 + (size_t)TL::M * (CT + 16) * 2);
  if (DIR && lds < (size_t)CT * (16 * TL::T + 1) * sizeof(float)) lds = (size_t)CT * (16 * TL::T + 1) * sizeof(float);      // the epilogue's transposed image
  cd.tiles_d = cdiv(cd.D, TD); cd.tiles_h = cdiv(cd.H, TH); cd.tiles_w = cdiv(cd.W, TW);
  const int tiles = cd.N * cd.tiles_d * cd.tiles_h * cd.tiles_w;
  const int tpg = cdiv(tiles, groups);
  auto kfn = k_w6<KD, TD, TH, TW, NT, 3, DIR>;
  if (f16) kfn = k_w6<KD, TD, TH, TW, NT, 2, DIR>;
  if (lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const dim3 grid(cdiv(tiles, tpg), cd.Cin16 / 16, cd.Cout16 / CT);
  hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, X, dY, partial, cd, tiles, tpg, accumulate);
  return cdiv(tiles, tpg);
}

// Partial weight-gradient slabs on the bf16 pipe where option wgrad_b6 allows it: returns the number of slabs written to
// `partial` ([G][T][Cin16][Cout16], for k_wgrad_reduce(_deep)), 0 when the shape is left to the fp32 kernels, -1 when the kernel wrote
// (accumulate: added to) the gradient `dw` itself (deep levels, one tile group: nothing left to reduce).
int b6_wgrad(const float* x, const float* dy, float* partial, float* dw, int accumulate, const ConvDims& cd, int KD, hipStream_t s) {
  W6Plan p;
  if (!w6_plan(p, cd, KD)) return 0;
  if (p.deep) {
    // deep level (round 6): ONE tile group -> the kernel writes (accumulate: adds to) dW itself; more -> slabs + reduce as before
    float* out = p.direct ? dw : partial;
    const int g = p.direct ? 1 : p.groups, acc = p.direct ? accumulate : 0;
    int G = 0;
#define BCP_W6_D1(TD_, TH_, TW_, NT_)                                                                     \
  do {                                                                                                    \
    if (p.direct) G = w6_launch<3, TD_, TH_, TW_, NT_, true>(x, dy, out, cd, g, acc, s);                  \
    else G = w6_launch<3, TD_, TH_, TW_, NT_, false>(x, dy, out, cd, g, acc, s);                          \
  } while (0)
    if (p.cfg == 2) { if (p.NT == 1) BCP_W6_D1(2, 8, 4, 1); else BCP_W6_D1(2, 8, 4, 2); }
    else { if (p.NT == 1) BCP_W6_D1(4, 8, 4, 1); else BCP_W6_D1(4, 8, 4, 2); }
#undef BCP_W6_D1
    return p.direct ? -1 : G;
  }
  if (p.NT == 1) {
    switch (p.cfg) {
      case 0: return w6_launch<3, 4, 4, 8, 1>(x, dy, partial, cd, p.groups, 0, s);
      case 1: return w6_launch<3, 4, 8, 4, 1>(x, dy, partial, cd, p.groups, 0, s);
      case 2: return w6_launch<3, 2, 8, 4, 1>(x, dy, partial, cd, p.groups, 0, s);
      default: return w6_launch<1, 1, 8, 16, 1>(x, dy, partial, cd, p.groups, 0, s);
    }
  }
  switch (p.cfg) {
    case 0: return w6_launch<3, 4, 4, 8, 2>(x, dy, partial, cd, p.groups, 0, s);
    case 1: return w6_launch<3, 4, 8, 4, 2>(x, dy, partial, cd, p.groups, 0, s);
    case 2: return w6_launch<3, 2, 8, 4, 2>(x, dy, partial, cd, p.groups, 0, s);
    default: return w6_launch<1, 1, 8, 16, 2>(x, dy, partial, cd, p.groups, 0, s);
  }
}

}  // namespace bcp
