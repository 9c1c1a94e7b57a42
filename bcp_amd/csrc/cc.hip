// bcp_amd/csrc/cc.hip -- largest-connected-component filter on the device.
//
// Reference: LargestCC_pancreas (LA_BCP_train.py:65-77, 26-connectivity; pancreas_utils.py:284-296,
// connectivity=2 -> 18), get_ACDC_2DLargestCC (ACDC_BCP_train.py:89-109, 8-connectivity per class 1..3).
// The reference copies every pseudo-label to the host, runs skimage.measure.label on one CPU
// thread and copies back -- the pipeline stall of the step.  Here: lock-free union-find
// (atomicMin link-to-smaller-root), so a component's root is its FIRST voxel in raster order;
// "largest, ties -> lowest label id" (np.argmax(np.bincount(...)[1:])+1) therefore becomes one
// 64-bit atomicMax on key = size<<32 | ~root.  Multi-class maps are labelled in one pass
// (neighbours join only when their class is equal), which equals the reference's per-class loop
// because classes are disjoint.  No host sync anywhere.
#include "common.h"
#include "../../include/bcp_hip.h"

namespace bcp {

__device__ __forceinline__ int uf_find(const int* L, int x) {
  int p = L[x];
  while (p != x) { x = p; p = L[x]; }
  return x;
}

__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  for (;;) {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }   // a > b: hang a under b
    const int old = atomicMin(&L[a], b);
    if (old == a) return;
    a = old;                                          // somebody re-parented a meanwhile: retry from there
  }
}

__global__ __launch_bounds__(256) void k_cc_init(const uint8_t* __restrict__ seg, int* __restrict__ L, int* __restrict__ size,
                                                 long long n) {
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long long)gridDim.x * blockDim.x) {
    L[v] = seg[v] ? (int)v : -1;
    size[v] = 0;
  }
}

// conn: maximum number of non-zero offset components (3D: 1 -> 6-conn, 2 -> 18, 3 -> 26; 2D (D==1): 1 -> 4, 2 -> 8)
__global__ __launch_bounds__(256) void k_cc_merge(const uint8_t* __restrict__ seg, int* __restrict__ L, int N, int D, int H, int W,
                                                  int conn) {
  const long long V = (long long)D * H * W, total = V * N;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (long long)gridDim.x * blockDim.x) {
    const uint8_t cls = seg[v];
    if (!cls) continue;
    const long long r = v % V;
    const int w = (int)(r % W), h = (int)((r / W) % H), d = (int)(r / ((long long)W * H));
    // forward half of the neighbourhood: linear offset > 0
    for (int dd = 0; dd <= 1; ++dd)
      for (int dh = -1; dh <= 1; ++dh)
        for (int dw = -1; dw <= 1; ++dw) {
          if (dd == 0 && (dh < 0 || (dh == 0 && dw <= 0))) continue;
          const int nz = (dd != 0) + (dh != 0) + (dw != 0);
          if (nz > conn) continue;
          const int d2 = d + dd, h2 = h + dh, w2 = w + dw;
          if (d2 >= D || h2 < 0 || h2 >= H || w2 < 0 || w2 >= W) continue;
          const long long u = v + ((long long)dd * H + dh) * W + dw;
          if (seg[u] == cls) uf_union(L, (int)v, (int)u);
        }
  }
}

__global__ __launch_bounds__(256) void k_cc_flatten(int* __restrict__ L, int* __restrict__ size, long long n) {
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long long)gridDim.x * blockDim.x) {
    if (L[v] < 0) continue;
    const int r = uf_find(L, (int)v);
    atomicAdd(&size[r], 1);
    // no path write here: other threads still walk the forest; roots are re-found in the write pass
  }
}

__global__ __launch_bounds__(256) void k_cc_select(const uint8_t* __restrict__ seg, const int* __restrict__ L,
                                                   const int* __restrict__ size, unsigned long long* __restrict__ best, long long V,
                                                   long long n, int nclass) {
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long long)gridDim.x * blockDim.x) {
    if (L[v] != (int)v) continue;  // roots only
    const int sample = (int)(v / V);
    const unsigned long long key = ((unsigned long long)(unsigned)size[v] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)v);
    atomicMax(&best[(long long)sample * nclass + (seg[v] - 1)], key);
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

}  // namespace bcp

using namespace bcp;

extern "C" size_t bcp_cc_workspace_bytes(int N, int D, int H, int W, int nclass) {
  const size_t n = (size_t)N * D * H * W;
  return n * 2 * sizeof(int) + (size_t)N * nclass * sizeof(unsigned long long) + 64;
}

extern "C" int bcp_cc_largest(const uint8_t* seg, uint8_t* out_u8, float* out_f32, int N, int D, int H, int W, int nclass,
                              int connectivity, void* workspace, void* stream) {
  BCP_REQUIRE(seg && (out_u8 || out_f32) && workspace, "bcp_cc_largest: null pointer");
  BCP_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && nclass >= 1 && nclass <= 8, "bcp_cc_largest: bad extents");
  BCP_REQUIRE(connectivity >= 1 && connectivity <= 3, "bcp_cc_largest: connectivity must be 1..3 (number of axes that may differ)");
  const long long V = (long long)D * H * W, n = V * N;
  BCP_REQUIRE(n < (1LL << 31), "bcp_cc_largest: volume too large for 32-bit labels");
  hipStream_t s = (hipStream_t)stream;
  int* L = reinterpret_cast<int*>(workspace);
  int* size = L + n;
  unsigned long long* best =
      reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(size + n) + 15) & ~(uintptr_t)15);
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipMemsetAsync(best, 0, (size_t)N * nclass * sizeof(unsigned long long), s);
  hipLaunchKernelGGL(k_cc_init, dim3(grid), dim3(256), 0, s, seg, L, size, n);
  hipLaunchKernelGGL(k_cc_merge, dim3(grid), dim3(256), 0, s, seg, L, N, D, H, W, connectivity);
  hipLaunchKernelGGL(k_cc_flatten, dim3(grid), dim3(256), 0, s, L, size, n);
  hipLaunchKernelGGL(k_cc_select, dim3(grid), dim3(256), 0, s, seg, L, size, best, V, n, nclass);
  hipLaunchKernelGGL(k_cc_write, dim3(grid), dim3(256), 0, s, seg, L, best, out_u8, out_f32, V, n, nclass);
  BCP_CHECK_LAUNCH("bcp_cc_largest");
  return BCP_OK;
}
