// bcp_amd/csrc/cc.hip -- largest-connected-component filter on the device.
//
// Reference: LargestCC_pancreas (LA_BCP_train.py:65-77, 26-connectivity; pancreas_utils.py:284-296,
// connectivity=2 -> 18), get_ACDC_2DLargestCC (ACDC_BCP_train.py:89-109, 8-connectivity per class 1..3).
// The reference copies every pseudo-label to the host, runs skimage.measure.label on one CPU thread and
// copies back -- the pipeline stall of the step.  Here everything stays on the device, no host sync:
//
//   1. k_cc_local   one workgroup per 4x8x16 (2-D: 1x16x32) tile: union-find of the tile in LDS
//                   (ds atomics), component sizes counted in LDS; writes L[v] = global index of the
//                   voxel's tile-local root and lsize[root] = local component size.
//   2. k_cc_border  unions across tile faces only (global atomicMin link-to-smaller-root, path halving), one thread per
//                   face voxel, between tile-local roots.
//   3. k_cc_count   one global atomicAdd per TILE-LOCAL root into its global root (a percolating blob
//                   costs ~#tiles atomics on its root word instead of one per voxel).
//   4. k_cc_select  roots only: 64-bit atomicMax on key = size<<32 | ~root.
//   5. k_cc_write   keep voxels whose root is the selected one.
//
// Links always point to the smaller index, so a component's root is its FIRST voxel in raster order and
// "largest, ties -> lowest label id" (np.argmax(np.bincount(labels.flat)[1:])+1 over raster-ordered labels)
// is exactly the atomicMax above.  Multi-class maps are labelled in one pass (neighbours join only when
// their class is equal) == the reference's per-class loop, because classes are disjoint.
#include "common.h"
#include <cstdlib>
#include "../../include/bcp_hip.h"

namespace bcp {

__device__ __forceinline__ int uf_find(const int* L, int x) {
  int p = L[x];
  while (p != x) { x = p; p = L[x]; }
  return x;
}

// find with path halving (plain stores of an ANCESTOR over a parent link: links only ever move towards the root, so a racing
// reader sees either the old parent or a nearer-to-root one -- the scheme of ECL-CC's `representative`).  Without it the giant
// percolating component of a noise-like pseudo-label map (every tile-local root chained into one tree) made each redundant
// border union walk the whole chain: k_cc_border 170 us of a 250 us chain at 2 x 112x112x80, 50 % foreground.
__device__ __forceinline__ int uf_find_c(int* L, int x) {
  int curr = L[x];
  if (curr != x) {
    int prev = x, next;
    while (curr > (next = L[curr])) {
      L[prev] = next;
      prev = curr;
      curr = next;
    }
  }
  return curr;
}

__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  for (;;) {
    a = uf_find_c(L, a);
    b = uf_find_c(L, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }   // a > b: hang a under b
    const int old = atomicMin(&L[a], b);
    if (old == a) return;
    a = old;                                          // somebody re-parented a meanwhile: retry from there
  }
}

struct CcDims { int N, D, H, W, tiles_d, tiles_h, tiles_w, conn; };

// forward half of the neighbourhood (linear offset > 0), filtered by connectivity
template <class F>
__device__ __forceinline__ void for_fwd_neighbours(int conn, F&& body) {
  for (int dd = 0; dd <= 1; ++dd)
    for (int dh = -1; dh <= 1; ++dh)
      for (int dw = -1; dw <= 1; ++dw) {
        if (dd == 0 && (dh < 0 || (dh == 0 && dw <= 0))) continue;
        if ((dd != 0) + (dh != 0) + (dw != 0) > conn) continue;
        body(dd, dh, dw);
      }
}

// (round 6) SRC: where a voxel's class comes from -- 0: the uint8 map `seg`; 1 / 2: straight from the network's logits (two channels +
// threshold as k_plabel_bin, four channels as k_plabel_argmax4: common.h plabel_*_of, the same bits), the map written to `seg` on the way
// for the launches behind this one.  The pseudo-label launch in front of the chain (21 us + a boundary on the teacher's tail, the step's
// critical path between the forward passes and the loss) is gone.
struct CcSrc { const float* logits; float thres; };
template <int TD, int TH, int TW, int SRC>
__global__ __launch_bounds__(256) void k_cc_local(uint8_t* __restrict__ seg, CcSrc src, int* __restrict__ L, int* __restrict__ lsize,
                                                  int* __restrict__ size, CcDims cd, unsigned long long* __restrict__ best, int nbest) {
  // (round 6) the selection table k_cc_select max-reduces into, three launches later: cleared here instead of by a memset launch of its own
  // on the teacher stream's tail (the step's critical path)
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < nbest; i += 256) best[i] = 0ull;
  constexpr int TV = TD * TH * TW, VPT = TV / 256;
  static_assert(TV % 256 == 0, "a whole number of voxels per thread");
  __shared__ int Ls[TV];
  __shared__ int Cnt[TV];
  __shared__ __attribute__((aligned(16))) uint8_t Ss[TV];
  const int tw = blockIdx.x % cd.tiles_w, th = (blockIdx.x / cd.tiles_w) % cd.tiles_h;
  const int td = (blockIdx.x / (cd.tiles_w * cd.tiles_h)) % cd.tiles_d, n = blockIdx.x / (cd.tiles_w * cd.tiles_h * cd.tiles_d);
  const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
  const long long nbase = (long long)n * cd.D * cd.H * cd.W;
  int gidx[VPT];
  // (round 6) from the logits with W % 4 == 0: a thread labels QUADS of four consecutive voxels of a row -- 16-byte loads of the logits
  // and ONE 4-byte store of the map per quad (the voxel-per-lane form below read 8 bytes per lane and wrote the map byte by byte:
  // k_cc_local 54.8 -> 76.3 us in the step, more than the pseudo-label launch it replaced) -- a quad is wholly inside or outside the volume
  const bool quads = SRC != 0 && (cd.W & 3) == 0 && (reinterpret_cast<uintptr_t>(seg) & 3u) == 0;
  if constexpr (SRC != 0) {
    if (quads) {
      for (int q = threadIdx.x; q < TV / 4; q += 256) {
        const int i0 = q * 4;
        const int lw = i0 % TW, lh = (i0 / TW) % TH, ld = i0 / (TW * TH);
        const int d = d0 + ld, h = h0 + lh, w = w0 + lw;
        uchar4 s4 = make_uchar4(0, 0, 0, 0);
        if (d < cd.D && h < cd.H && w < cd.W) {
          const long long gv = nbase + ((long long)d * cd.H + h) * cd.W + w;      // a multiple of 4
          if constexpr (SRC == 1) {
            const float4 a = ld4(src.logits + gv * 2), b = ld4(src.logits + gv * 2 + 4);
            s4 = make_uchar4(plabel_bin_of(a.x, a.y, src.thres), plabel_bin_of(a.z, a.w, src.thres), plabel_bin_of(b.x, b.y, src.thres),
                             plabel_bin_of(b.z, b.w, src.thres));
          } else {
            const float4 a = ld4(src.logits + gv * 4), b = ld4(src.logits + gv * 4 + 4), c = ld4(src.logits + gv * 4 + 8), e = ld4(src.logits + gv * 4 + 12);
            s4 = make_uchar4(plabel_argmax4_of(a.x, a.y, a.z, a.w), plabel_argmax4_of(b.x, b.y, b.z, b.w), plabel_argmax4_of(c.x, c.y, c.z, c.w),
                             plabel_argmax4_of(e.x, e.y, e.z, e.w));
          }
          *reinterpret_cast<uchar4*>(seg + gv) = s4;
        }
        *reinterpret_cast<uchar4*>(&Ss[i0]) = s4;
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int u = 0; u < VPT; ++u) {
    const int i = threadIdx.x + u * 256;
    const int lw = i % TW, lh = (i / TW) % TH, ld = i / (TW * TH);
    const int d = d0 + ld, h = h0 + lh, w = w0 + lw;
    const bool in = d < cd.D && h < cd.H && w < cd.W;
    gidx[u] = in ? (int)(nbase + ((long long)d * cd.H + h) * cd.W + w) : -1;
    uint8_t s = 0;
    if (quads) s = Ss[i];
    else if (in) {
      if constexpr (SRC == 0) s = seg[gidx[u]];
      else if constexpr (SRC == 1) {
        const float2 x = *reinterpret_cast<const float2*>(src.logits + (long long)gidx[u] * 2);
        s = plabel_bin_of(x.x, x.y, src.thres);
        seg[gidx[u]] = s;
      } else {
        const float4 x = ld4(src.logits + (long long)gidx[u] * 4);
        s = plabel_argmax4_of(x.x, x.y, x.z, x.w);
        seg[gidx[u]] = s;
      }
    }
    if (!quads) Ss[i] = s;
    Ls[i] = s ? i : -1;
    Cnt[i] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < VPT; ++u) {
    const int i = threadIdx.x + u * 256;
    const uint8_t cls = Ss[i];
    if (cls) {
      const int lw = i % TW, lh = (i / TW) % TH, ld = i / (TW * TH);
      for_fwd_neighbours(cd.conn, [&](int dd, int dh, int dw) {
        const int d2 = ld + dd, h2 = lh + dh, w2 = lw + dw;
        if (d2 < TD && h2 >= 0 && h2 < TH && w2 >= 0 && w2 < TW) {
          const int j = (d2 * TH + h2) * TW + w2;
          if (Ss[j] == cls) uf_union(Ls, i, j);
        }
      });
    }
  }
  __syncthreads();
  int root[VPT];
#pragma unroll
  for (int u = 0; u < VPT; ++u) {
    const int i = threadIdx.x + u * 256;
    root[u] = -1;
    if (Ss[i]) {
      root[u] = uf_find(Ls, i);
      atomicAdd(&Cnt[root[u]], 1);
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < VPT; ++u) {
    const int i = threadIdx.x + u * 256;
    if (gidx[u] < 0) continue;
    int lab = -1, ls = 0;
    if (root[u] >= 0) {
      const int r = root[u];
      const int rw = r % TW, rh = (r / TW) % TH, rd = r / (TW * TH);
      lab = (int)(nbase + ((long long)(d0 + rd) * cd.H + (h0 + rh)) * cd.W + (w0 + rw));
      if (r == i) ls = Cnt[i];
    }
    L[gidx[u]] = lab;
    lsize[gidx[u]] = ls;
    size[gidx[u]] = 0;
  }
}

// Only voxels on a tile face have out-of-tile forward neighbours: one thread per (tile, face voxel) -- the d = TD-1 plane plus
// the h / w boundary ring of the other slices (676 of 2048 voxels of an 8x16x16 tile; a 2-D tile has just its ring) -- so
// every lane of a wave has union work to do instead of one in three.
template <int TD, int TH, int TW>
struct CcFace {
  static constexpr int FACE = TD > 1 ? TH * TW : 0;
  static constexpr int RING = TH * TW - (TH - 2) * (TW - 2);
  static constexpr int NB = FACE + (TD > 1 ? TD - 1 : 1) * RING;
};

template <int TD, int TH, int TW>
__global__ __launch_bounds__(256) void k_cc_border(const uint8_t* __restrict__ seg, int* __restrict__ L, CcDims cd, int dedupe) {
  using F = CcFace<TD, TH, TW>;
  const long long V = (long long)cd.D * cd.H * cd.W;
  const long long ntiles = (long long)cd.N * cd.tiles_d * cd.tiles_h * cd.tiles_w;
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // (round 6: no early return -- every lane of a wave walks the 13 neighbour offsets in lockstep for the pair exchange below; an idle lane
  // carries cls = 0)
  bool live = id < ntiles * F::NB;
  const int tile = live ? (int)(id / F::NB) : 0, bi = live ? (int)(id % F::NB) : 0;
  int ld, lh, lw;
  if (bi < F::FACE) { ld = TD - 1; lh = bi / TW; lw = bi % TW; }
  else {
    const int j = bi - F::FACE, k = j % F::RING;
    ld = j / F::RING;
    if (k < TW) { lh = 0; lw = k; }
    else if (k < 2 * TW) { lh = TH - 1; lw = k - TW; }
    else { lh = 1 + (k - 2 * TW) / 2; lw = ((k - 2 * TW) & 1) ? TW - 1 : 0; }
  }
  const int tw = tile % cd.tiles_w, th = (tile / cd.tiles_w) % cd.tiles_h;
  const int td = (tile / (cd.tiles_w * cd.tiles_h)) % cd.tiles_d, n = tile / (cd.tiles_w * cd.tiles_h * cd.tiles_d);
  const int d = td * TD + ld, h = th * TH + lh, w = tw * TW + lw;
  live = live && d < cd.D && h < cd.H && w < cd.W;
  const long long v = (long long)n * V + ((long long)d * cd.H + h) * cd.W + w;
  const uint8_t cls = live ? seg[v] : 0;
  // unions go between TILE-LOCAL ROOTS (L[x] after k_cc_local, or a nearer-to-root ancestor later on): the up to 13
  // out-of-tile neighbours of one voxel mostly belong to one or two components of the neighbouring tiles, and joining the
  // same pair again costs two more pointer chases through global memory -- skip the pairs this thread has just done.
  const int rv = cls ? L[v] : -1;
  int seen0 = -1, seen1 = -1;
  const int lane = threadIdx.x & 63;
  for_fwd_neighbours(cd.conn, [&](int dd, int dh, int dw) {
    int ru = -1;
    if (cls) {
      const int d2 = d + dd, h2 = h + dh, w2 = w + dw;
      if (d2 < cd.D && h2 >= 0 && h2 < cd.H && w2 >= 0 && w2 < cd.W) {
        const bool same_tile = (ld + dd < TD) && (lh + dh >= 0) && (lh + dh < TH) && (lw + dw >= 0) && (lw + dw < TW);
        if (!same_tile) {
          const long long u = v + ((long long)dd * cd.H + dh) * cd.W + dw;
          if (seg[u] == cls) ru = L[u];
        }
      }
    }
    // (round 6) neighbouring lanes are neighbouring face voxels and mostly join the SAME pair of tile-local roots at the same offset
    // (a noise-like map: one percolating component through every tile): a lane whose pair is the previous lane's leaves it to that lane --
    // which joins it now, has joined it before (its own `seen`), or leaves it to ITS predecessor; the first lane of a run always acts.
    // Hundreds of redundant find chains per tile pair were what the kernel's 70-90 us were made of.
    bool mine = ru >= 0 && ru != seen0 && ru != seen1;
    if (dedupe) {
      const int pru = __shfl_up(ru, 1), prv = __shfl_up(rv, 1);
      if (lane > 0 && ru >= 0 && pru == ru && prv == rv) mine = false;
    }
    if (mine) {
      uf_union(L, rv, ru);
      seen1 = seen0;
      seen0 = ru;
    }
  });
}

__global__ __launch_bounds__(256) void k_cc_count(int* __restrict__ L, const int* __restrict__ lsize, int* __restrict__ size,
                                                  long long n) {
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long long)gridDim.x * blockDim.x) {
    const int ls = lsize[v];
    if (ls > 0) atomicAdd(&size[uf_find_c(L, (int)v)], ls);   // (leaves every tile-local root one hop from its global root)
  }
}

// (round 6) count AND select: the atomicAdd that adds a tile-local root's size to its global root returns the root's running total; the
// maximum over ALL running totals' keys (size << 32 | ~root) is the maximum over the FINAL ones -- every running total of a root is
// dominated by that root's final total, which the last adder sees -- so the selection needs no pass of its own over the finished sizes
// (k_cc_select: 17.6 us + a boundary on the critical path).  Per-thread running maxima per class, one atomicMax per class and block, as
// k_cc_select; blockIdx.y = sample.  Order-independent: same `best` bits whatever order the atomics land in.
__global__ __launch_bounds__(256) void k_cc_count_select(const uint8_t* __restrict__ seg, int* __restrict__ L, const int* __restrict__ lsize,
                                                         int* __restrict__ size, unsigned long long* __restrict__ best, long long V,
                                                         int nclass) {
  __shared__ unsigned long long red[4][8];
  const int sample = blockIdx.y;
  unsigned long long loc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long base = (long long)sample * V;
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < V; r += (long long)gridDim.x * blockDim.x) {
    const long long v = base + r;
    const int ls = lsize[v];
    if (ls <= 0) continue;
    const int root = uf_find_c(L, (int)v);                       // (leaves every tile-local root one hop from its global root)
    const int tot = atomicAdd(&size[root], ls) + ls;
    const unsigned long long key = ((unsigned long long)(unsigned)tot << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)root);
    const int c = seg[v] - 1;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k == c && key > loc[k]) loc[k] = key;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < nclass; ++k) {
    unsigned long long m = loc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o); m = t > m ? t : m; }
    if (lane == 0) red[wave][k] = m;
  }
  __syncthreads();
  if ((int)threadIdx.x < nclass) {
    unsigned long long m = red[0][threadIdx.x];
    for (int w = 1; w < 4; ++w) m = red[w][threadIdx.x] > m ? red[w][threadIdx.x] : m;
    if (m) atomicMax(&best[(long long)sample * nclass + threadIdx.x], m);
  }
}

// (round 6) the same per TILE: one workgroup takes the tile-local roots of one k_cc_local tile, finds their global roots and adds up the sizes
// of the ones that share a global root in an LDS table first (open addressing keyed on the root; a tile holds at most TV / 2 local roots, the
// table has TV slots) -- ONE global atomicAdd per (tile, global root).  On a noise-like map every tile holds 5-20 tile-local pieces of the
// one percolating component: ~10^4 atomics on that root's size word, serialised in one L2 channel, were what k_cc_count_select's 50-58 us in
// the step were made of.
template <int TD, int TH, int TW>
__global__ __launch_bounds__(256) void k_cc_count_select_tile(const uint8_t* __restrict__ seg, int* __restrict__ L, const int* __restrict__ lsize,
                                                              int* __restrict__ size, unsigned long long* __restrict__ best, CcDims cd, int nclass) {
  constexpr int TV = TD * TH * TW, VPT = TV / 256;
  __shared__ int hkey[TV];
  __shared__ int hval[TV];
  __shared__ unsigned long long red[4][8];
  const int tw = blockIdx.x % cd.tiles_w, th = (blockIdx.x / cd.tiles_w) % cd.tiles_h;
  const int td = (blockIdx.x / (cd.tiles_w * cd.tiles_h)) % cd.tiles_d, n = blockIdx.x / (cd.tiles_w * cd.tiles_h * cd.tiles_d);
  const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
  const long long nbase = (long long)n * cd.D * cd.H * cd.W;
#pragma unroll
  for (int u = 0; u < VPT; ++u) { hkey[threadIdx.x + u * 256] = -1; hval[threadIdx.x + u * 256] = 0; }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < VPT; ++u) {
    const int i = threadIdx.x + u * 256;
    const int lw = i % TW, lh = (i / TW) % TH, ld = i / (TW * TH);
    const int d = d0 + ld, h = h0 + lh, w = w0 + lw;
    if (d >= cd.D || h >= cd.H || w >= cd.W) continue;
    const int v = (int)(nbase + ((long long)d * cd.H + h) * cd.W + w);
    const int ls = lsize[v];
    if (ls <= 0) continue;
    const int root = uf_find_c(L, v);                            // (leaves every tile-local root one hop from its global root)
    unsigned hs = ((unsigned)root * 2654435761u) >> 7;
    for (;;) {
      hs &= (unsigned)(TV - 1);
      const int old = atomicCAS(&hkey[hs], -1, root);
      if (old == -1 || old == root) { atomicAdd(&hval[hs], ls); break; }
      ++hs;
    }
  }
  __syncthreads();
  unsigned long long loc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < VPT; ++u) {
    const int sidx = threadIdx.x + u * 256;
    const int root = hkey[sidx];
    if (root < 0) continue;
    const int add = hval[sidx];
    const int tot = atomicAdd(&size[root], add) + add;
    const unsigned long long key = ((unsigned long long)(unsigned)tot << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)root);
    const int c = seg[root] - 1;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k == c && key > loc[k]) loc[k] = key;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < nclass; ++k) {
    unsigned long long m = loc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o); m = t > m ? t : m; }
    if (lane == 0) red[wave][k] = m;
  }
  __syncthreads();
  if ((int)threadIdx.x < nclass) {
    unsigned long long m = red[0][threadIdx.x];
    for (int w = 1; w < 4; ++w) m = red[w][threadIdx.x] > m ? red[w][threadIdx.x] : m;
    if (m) atomicMax(&best[(long long)n * nclass + threadIdx.x], m);
  }
}

// blockIdx.y = sample: every thread keeps a running maximum per class, the block reduces them (64-bit max over shuffles +
// one LDS hop) and issues ONE atomicMax per class.  (Noise maps have ~10^5 roots; one atomic -- or even one racy read -- per
// root on N * nclass hot words cost 0.43 ms per ACDC step.)
__global__ __launch_bounds__(256) void k_cc_select(const uint8_t* __restrict__ seg, const int* __restrict__ L,
                                                   const int* __restrict__ size, unsigned long long* __restrict__ best, long long V,
                                                   long long n, int nclass) {
  __shared__ unsigned long long red[4][8];
  const int sample = blockIdx.y;
  unsigned long long loc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long base = (long long)sample * V;
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < V; r += (long long)gridDim.x * blockDim.x) {
    const long long v = base + r;
    if (L[v] != (int)v) continue;  // global roots only
    const unsigned long long key = ((unsigned long long)(unsigned)size[v] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)v);
    const int c = seg[v] - 1;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k == c && key > loc[k]) loc[k] = key;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < nclass; ++k) {
    unsigned long long m = loc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o); m = t > m ? t : m; }
    if (lane == 0) red[wave][k] = m;
  }
  __syncthreads();
  if ((int)threadIdx.x < nclass) {
    unsigned long long m = red[0][threadIdx.x];
    for (int w = 1; w < 4; ++w) m = red[w][threadIdx.x] > m ? red[w][threadIdx.x] : m;
    if (m) atomicMax(&best[(long long)sample * nclass + threadIdx.x], m);
  }
}

__global__ __launch_bounds__(256) void k_cc_write(const uint8_t* __restrict__ seg, const int* __restrict__ L,
                                                  const unsigned long long* __restrict__ best, uint8_t* __restrict__ out_u8,
                                                  float* __restrict__ out_f32, long long V, long long n, int nclass) {
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long long)gridDim.x * blockDim.x) {
    uint8_t o = 0;
    const uint8_t cls = seg[v];
    if (cls) {
      const int r = uf_find(L, (int)v);
      const unsigned long long key = best[(v / V) * nclass + (cls - 1)];
      const unsigned root = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
      o = ((unsigned)r == root) ? cls : 0;
    }
    if (out_u8) out_u8[v] = o;
    if (out_f32) out_f32[v] = (float)o;
  }
}

static inline int border_grid(const CcDims& cd, int nb) {
  return (int)(((long long)cd.N * cd.tiles_d * cd.tiles_h * cd.tiles_w * nb + 255) / 256);
}

}  // namespace bcp

using namespace bcp;

extern "C" size_t bcp_cc_workspace_bytes(int N, int D, int H, int W, int nclass) {
  const size_t n = (size_t)N * D * H * W;
  return n * 3 * sizeof(int) + (size_t)N * nclass * sizeof(unsigned long long) + 64;
}

template <int SRC>
static int cc_run(uint8_t* seg, CcSrc src, uint8_t* out_u8, float* out_f32, int N, int D, int H, int W, int nclass, int connectivity,
                  void* workspace, hipStream_t s) {
  const long long V = (long long)D * H * W, n = V * N;
  int* L = reinterpret_cast<int*>(workspace);
  int* lsize = L + n;
  int* size = lsize + n;
  unsigned long long* best =
      reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(size + n) + 15) & ~(uintptr_t)15);
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  CcDims cd;
  cd.N = N; cd.D = D; cd.H = H; cd.W = W; cd.conn = connectivity;
  // local tiles: 8x16x16 / 32x64 voxels (8 per thread) when the volume has at least ~2 such tiles per CU, else 4x8x16 / 16x32;
  // bigger tiles leave fewer voxels on tile faces for the global-atomic border pass (51 % -> 33 % in 3-D)
  const int tile_opt = options().cc_tile;   // 0: by size, 1: small tiles, 2: big tiles (tests / measurements)
  const bool big = tile_opt ? tile_opt == 2 : (n >= 512LL * 2048);
  if (D > 1 && big) {
    cd.tiles_d = cdiv(D, 8); cd.tiles_h = cdiv(H, 16); cd.tiles_w = cdiv(W, 16);
    hipLaunchKernelGGL((k_cc_local<8, 16, 16, SRC>), dim3(N * cd.tiles_d * cd.tiles_h * cd.tiles_w), dim3(256), 0, s, seg, src, L, lsize, size, cd, best, N * nclass);
    hipLaunchKernelGGL((k_cc_border<8, 16, 16>), dim3(border_grid(cd, CcFace<8, 16, 16>::NB)), dim3(256), 0, s, seg, L, cd, options().cc_border_dedupe);
  } else if (D > 1) {
    cd.tiles_d = cdiv(D, 4); cd.tiles_h = cdiv(H, 8); cd.tiles_w = cdiv(W, 16);
    hipLaunchKernelGGL((k_cc_local<4, 8, 16, SRC>), dim3(N * cd.tiles_d * cd.tiles_h * cd.tiles_w), dim3(256), 0, s, seg, src, L, lsize, size, cd, best, N * nclass);
    hipLaunchKernelGGL((k_cc_border<4, 8, 16>), dim3(border_grid(cd, CcFace<4, 8, 16>::NB)), dim3(256), 0, s, seg, L, cd, options().cc_border_dedupe);
  } else if (big) {
    cd.tiles_d = 1; cd.tiles_h = cdiv(H, 32); cd.tiles_w = cdiv(W, 64);
    hipLaunchKernelGGL((k_cc_local<1, 32, 64, SRC>), dim3(N * cd.tiles_h * cd.tiles_w), dim3(256), 0, s, seg, src, L, lsize, size, cd, best, N * nclass);
    hipLaunchKernelGGL((k_cc_border<1, 32, 64>), dim3(border_grid(cd, CcFace<1, 32, 64>::NB)), dim3(256), 0, s, seg, L, cd, options().cc_border_dedupe);
  } else {
    cd.tiles_d = 1; cd.tiles_h = cdiv(H, 16); cd.tiles_w = cdiv(W, 32);
    hipLaunchKernelGGL((k_cc_local<1, 16, 32, SRC>), dim3(N * cd.tiles_h * cd.tiles_w), dim3(256), 0, s, seg, src, L, lsize, size, cd, best, N * nclass);
    hipLaunchKernelGGL((k_cc_border<1, 16, 32>), dim3(border_grid(cd, CcFace<1, 16, 32>::NB)), dim3(256), 0, s, seg, L, cd, options().cc_border_dedupe);
  }
  const int ntile = N * cd.tiles_d * cd.tiles_h * cd.tiles_w;
  if (options().cc_fuse_select != 0 && options().cc_count_tile != 0) {
    if (D > 1 && big) hipLaunchKernelGGL((k_cc_count_select_tile<8, 16, 16>), dim3(ntile), dim3(256), 0, s, seg, L, lsize, size, best, cd, nclass);
    else if (D > 1) hipLaunchKernelGGL((k_cc_count_select_tile<4, 8, 16>), dim3(ntile), dim3(256), 0, s, seg, L, lsize, size, best, cd, nclass);
    else if (big) hipLaunchKernelGGL((k_cc_count_select_tile<1, 32, 64>), dim3(ntile), dim3(256), 0, s, seg, L, lsize, size, best, cd, nclass);
    else hipLaunchKernelGGL((k_cc_count_select_tile<1, 16, 32>), dim3(ntile), dim3(256), 0, s, seg, L, lsize, size, best, cd, nclass);
  } else if (options().cc_fuse_select != 0) {
    int gx = (int)((V + 255) / 256);          // one voxel per thread up to 2048 workgroups per launch (k_cc_count's geometry)
    const int cap = 2048 / N < 1 ? 1 : 2048 / N;
    gx = gx < 1 ? 1 : (gx > cap ? cap : gx);
    hipLaunchKernelGGL(k_cc_count_select, dim3(gx, N), dim3(256), 0, s, seg, L, lsize, size, best, V, nclass);
  } else {
    hipLaunchKernelGGL(k_cc_count, dim3(grid), dim3(256), 0, s, L, lsize, size, n);
    int gx = (int)((V + 255) / 256 / 4);      // ~4 voxels per thread
    const int cap = options().cc_select_blocks > 0 ? options().cc_select_blocks : 256;
    gx = gx < 1 ? 1 : (gx > cap ? cap : gx);  // (round 6, measured: 1024 blocks per sample make the chain 162 -> 155 us alone and move nothing in the step: gpurun_out/r06_s17)
    hipLaunchKernelGGL(k_cc_select, dim3(gx, N), dim3(256), 0, s, seg, L, size, best, V, n, nclass);
  }
  hipLaunchKernelGGL(k_cc_write, dim3(grid), dim3(256), 0, s, seg, L, best, out_u8, out_f32, V, n, nclass);
  return 0;
}

extern "C" int bcp_cc_largest(const uint8_t* seg, uint8_t* out_u8, float* out_f32, int N, int D, int H, int W, int nclass,
                              int connectivity, void* workspace, void* stream) {
  if (bcp::options().whatif & 4) return BCP_OK;      // MEASUREMENT ONLY (common.h Options::whatif)

  BCP_REQUIRE(seg && (out_u8 || out_f32) && workspace, "bcp_cc_largest: null pointer");
  BCP_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && nclass >= 1 && nclass <= 8, "bcp_cc_largest: bad extents");
  BCP_REQUIRE(connectivity >= 1 && connectivity <= 3, "bcp_cc_largest: connectivity must be 1..3 (number of axes that may differ)");
  BCP_REQUIRE((long long)D * H * W * N < (1LL << 31), "bcp_cc_largest: volume too large for 32-bit labels");
  cc_run<0>(const_cast<uint8_t*>(seg) /* SRC 0 only reads it */, CcSrc{nullptr, 0.f}, out_u8, out_f32, N, D, H, W, nclass, connectivity, workspace,
            (hipStream_t)stream);
  BCP_CHECK_LAUNCH("bcp_cc_largest");
  return BCP_OK;
}

// (round 6) pseudo-label + largest-CC in one chain: get_cut_mask(out, nms=1) (LA_BCP_train.py:57-63, pancreas_utils.py:275-281; C = 2,
// nclass = 1, thres) and get_ACDC_masks(output, nms=1) (ACDC_BCP_train.py:112-117; C = 4, nclass = 3, thres ignored) straight from the
// channel-last logits.  seg_out receives what bcp_plabel_bin / bcp_plabel_argmax4 would have written (the unfiltered map; the chain's
// later launches read it), out_* what bcp_cc_largest(seg_out, ..) would: the same bits as the two calls.
extern "C" int bcp_plabel_cc_largest(const float* logits, int C, float thres, uint8_t* seg_out, uint8_t* out_u8, float* out_f32, int N, int D,
                                     int H, int W, int nclass, int connectivity, void* workspace, void* stream) {
  BCP_REQUIRE(logits && seg_out && (out_u8 || out_f32) && workspace, "bcp_plabel_cc_largest: null pointer");
  BCP_REQUIRE((C == 2 && nclass == 1) || (C == 4 && nclass == 3), "bcp_plabel_cc_largest: C = 2 / nclass = 1 (threshold) or C = 4 / nclass = 3 (argmax)");
  BCP_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "bcp_plabel_cc_largest: bad extents");
  BCP_REQUIRE(connectivity >= 1 && connectivity <= 3, "bcp_plabel_cc_largest: connectivity must be 1..3 (number of axes that may differ)");
  BCP_REQUIRE((long long)D * H * W * N < (1LL << 31), "bcp_plabel_cc_largest: volume too large for 32-bit labels");
  BCP_REQUIRE(aligned16(logits), "bcp_plabel_cc_largest: alignment");
  if (bcp::options().whatif & 4) return BCP_OK;      // MEASUREMENT ONLY (common.h Options::whatif)
  if (C == 2) cc_run<1>(seg_out, CcSrc{logits, thres}, out_u8, out_f32, N, D, H, W, nclass, connectivity, workspace, (hipStream_t)stream);
  else cc_run<2>(seg_out, CcSrc{logits, 0.f}, out_u8, out_f32, N, D, H, W, nclass, connectivity, workspace, (hipStream_t)stream);
  BCP_CHECK_LAUNCH("bcp_plabel_cc_largest");
  return BCP_OK;
}
