// bcp_amd/csrc/norm.hip -- BatchNorm (train) / InstanceNorm + activation (+dropout, +residual) forward
// and backward for channels-last activations [rows][C] on gfx950.
//
// Reference: nn.BatchNorm3d/2d in train() mode (networks/VNet.py:18-26, networks/unet.py:21-28),
// nn.InstanceNorm3d(affine=False) (pancreas/Vnet.py:93), nn.ReLU / nn.LeakyReLU(0.01),
// nn.Dropout3d(p=0.5) (VNet.py:165,211) / nn.Dropout(p) (unet.py:23), skip add after the ReLU
// (VNet.py:220-233).
//
// All four kernels are pure HBM streams over [rows][C] with C the fastest dim: a thread owns one
// float4 channel group and walks rows, so a wave reads 1 KiB contiguous per instruction.
// Statistics accumulate in fp64 per thread -> LDS -> per-block partials -> 1-block finalize (no
// atomics, bitwise reproducible).
//   groups G = 1            : BatchNorm over all rows (N*spatial)
//   groups G = N            : InstanceNorm, rows_per_group = spatial
#include "common.h"
#include "../../include/bcp_hip.h"

namespace bcp {

struct NormEpilogue {
  const float* chan_scale;     // nullable [N][C]: Dropout3d keep*1/(1-p) per (sample, channel)
  const uint8_t* elem_mask;    // nullable [rows][C]: elementwise Dropout keep mask
  float elem_scale;            // 1/(1-p) for elem_mask
  long long rows_per_sample;   // spatial size (rows per n) for chan_scale indexing
  int act;
  const unsigned long long* mask_seed;   // nullable (round 4): device seed of an elementwise Dropout whose keep bits are EVALUATED here
  float p_keep;                          //   (bern_keep, common.h: the bits bcp_bernoulli_dev would write) instead of read from elem_mask
};
// the four mask bytes of elements e .. e+3: loaded, or evaluated from the seed
#define BCP_NORM_MASK_PROLOGUE(ep)                                                                                  \
  unsigned mseed_lo = 0, mseed_hi = 0;                                                                              \
  if ((ep).mask_seed) { const unsigned long long s_ = *(ep).mask_seed; mseed_lo = (unsigned)s_; mseed_hi = (unsigned)(s_ >> 32); } \
  const bool has_mask = (ep).elem_mask != nullptr || (ep).mask_seed != nullptr;
#define BCP_NORM_MASK4(ep, e) ((ep).mask_seed ? bern_keep4((e), mseed_lo, mseed_hi, (ep).p_keep) : *reinterpret_cast<const uchar4*>((ep).elem_mask + (e)))

// ------------------------------------------------------------------ column reductions
// MODE 0: (sum x, sum x^2) of y.   MODE 1: (sum dz, sum dz*xhat) for the backward pass.
//
// All three streaming kernels below walk "segments": a segment is a run of rows that shares every per-channel
// parameter -- one normalisation group, or one sample of it when a per-(sample, channel) Dropout3d scale is present.
// blockIdx.y = segment, so group / sample indices are block-uniform, a thread's float4 column never changes, and the
// per-channel parameters are loaded ONCE into registers; the element loop is 4-way unrolled with the loads up front
// (memory-level parallelism for an HBM stream) and contains no division.
// SL (round 4): the streamed operand -- y in MODE 0, da in MODE 1 -- arrives as the nslab raw split-K slabs of the conv that produced it
// (bcp_conv3_fwd_raw) and is summed HERE on the way in, in k_b6_sum_slabs' order (bias first, then the slabs front to back: bit-identical
// to the slab-sum launch this replaces), and written once to sum_out for the apply pass: the deep levels' conv -> k_b6_sum_slabs ->
// statistics -> finalize -> apply chain loses a launch and a round trip of the tensor through the L2.
struct SlabSrc { const float* slabs; int nslab; long long stride; const float* bias; float* sum_out; };

template <int MODE, bool SL = false>
__global__ __launch_bounds__(256) void k_col_partial(const float* __restrict__ y, const float* __restrict__ da,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     NormEpilogue ep, long long seg_rows, int spg /* segments per group */, int C,
                                                     double* __restrict__ partial /* [G][spg * gridDim.x][C][2] */, SlabSrc sl,
                                                     float* __restrict__ amax_clear_out /* nullable: the |max| slots of the tensor the fused apply pass
                                                                                           behind this launch writes (k_norm_apply_fin: no finalize launch clears them) */) {
  constexpr int U = 4;
  BCP_NORM_MASK_PROLOGUE(ep)
  if (amax_clear_out && blockIdx.x == 0 && blockIdx.y == 0) amax_clear(amax_clear_out);
  const int C4 = C >> 2;
  const int col = threadIdx.x % C4;        // float4 column
  const int slot = threadIdx.x / C4;       // row slot within a pass
  const int slots = 256 / C4;
  const int seg = blockIdx.y, g = seg / spg, nbps = gridDim.x;
  const long long chunk = (seg_rows + nbps - 1) / nbps;
  const long long r0 = (long long)blockIdx.x * chunk;
  long long r1 = r0 + chunk;
  if (r1 > seg_rows) r1 = seg_rows;
  const long long sbase = (long long)seg * seg_rows;

  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  float scv[4] = {1, 1, 1, 1}, shv[4] = {0, 0, 0, 0}, muv[4] = {0, 0, 0, 0}, rsv[4] = {1, 1, 1, 1}, csv[4] = {1, 1, 1, 1};
  if (MODE == 1) {
    const float4 sc = ld4(scale + (long long)g * C + col * 4), sh = ld4(shift + (long long)g * C + col * 4);
    const float4 mu = ld4(mean + (long long)g * C + col * 4), rs = ld4(rstd + (long long)g * C + col * 4);
    scv[0] = sc.x; scv[1] = sc.y; scv[2] = sc.z; scv[3] = sc.w;
    shv[0] = sh.x; shv[1] = sh.y; shv[2] = sh.z; shv[3] = sh.w;
    muv[0] = mu.x; muv[1] = mu.y; muv[2] = mu.z; muv[3] = mu.w;
    rsv[0] = rs.x; rsv[1] = rs.y; rsv[2] = rs.z; rsv[3] = rs.w;
    if (ep.chan_scale) {
      const float4 c4 = ld4(ep.chan_scale + (sbase / ep.rows_per_sample) * C + col * 4);
      csv[0] = c4.x; csv[1] = c4.y; csv[2] = c4.z; csv[3] = c4.w;
    }
  }
  auto accum = [&](const float4& v, const float4& d4, const uchar4& m4) {
    const float vv[4] = {v.x, v.y, v.z, v.w};
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { s1[k] += (double)vv[k]; s2[k] += (double)vv[k] * (double)vv[k]; }
    } else {
      const float dz[4] = {d4.x, d4.y, d4.z, d4.w};
      float cs[4] = {csv[0], csv[1], csv[2], csv[3]};
      if (has_mask) {
        cs[0] *= m4.x ? ep.elem_scale : 0.f; cs[1] *= m4.y ? ep.elem_scale : 0.f;
        cs[2] *= m4.z ? ep.elem_scale : 0.f; cs[3] *= m4.w ? ep.elem_scale : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float z = (vv[k] - muv[k]) * scv[k] + shv[k];
        const float g1 = dz[k] * cs[k] * act_grad(z, ep.act);
        const float xh = (vv[k] - muv[k]) * rsv[k];
        s1[k] += (double)g1;
        s2[k] += (double)g1 * (double)xh;
      }
    }
  };
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (SL && sl.bias) bias4 = ld4(sl.bias + col * 4);
  // (SL) sum of the slabs at element offset e, bias first, slabs front to back; stored for the apply pass
  auto slab_sum = [&](long long e) __attribute__((always_inline)) -> float4 {
    float4 a = bias4;
    for (int k = 0; k < sl.nslab; ++k) {
      const float4 p = ld4(sl.slabs + (long long)k * sl.stride + e);
      a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    }
    st4(sl.sum_out + e, a);
    return a;
  };
  if (slot < slots) {
    long long r = r0 + slot;
    for (; r + (long long)(U - 1) * slots < r1; r += (long long)U * slots) {
      float4 v[U], d4[U];
      uchar4 m4[U];
      if (SL) {
        // slab loop outermost, the U rows' loads of a slab issued together (memory-level parallelism), same per-element add order
        float4 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = bias4;
        for (int k = 0; k < sl.nslab; ++k) {
          float4 p[U];
#pragma unroll
          for (int u = 0; u < U; ++u) p[u] = ld4(sl.slabs + (long long)k * sl.stride + ((sbase + r + (long long)u * slots) * C4 + col) * 4);
#pragma unroll
          for (int u = 0; u < U; ++u) { a[u].x += p[u].x; a[u].y += p[u].y; a[u].z += p[u].z; a[u].w += p[u].w; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long long e = ((sbase + r + (long long)u * slots) * C4 + col) * 4;
          st4(sl.sum_out + e, a[u]);
          if (MODE == 0) v[u] = a[u];
          else {
            d4[u] = a[u];
            v[u] = ld4(y + e);
            if (has_mask) m4[u] = BCP_NORM_MASK4(ep, e);
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long long e = ((sbase + r + (long long)u * slots) * C4 + col) * 4;
          v[u] = ld4(y + e);
          if (MODE == 1) {
            d4[u] = ld4(da + e);
            if (has_mask) m4[u] = BCP_NORM_MASK4(ep, e);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) accum(v[u], d4[u], m4[u]);
    }
    for (; r < r1; r += slots) {
      const long long e = ((sbase + r) * C4 + col) * 4;
      float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
      uchar4 m4 = make_uchar4(0, 0, 0, 0);
      float4 v;
      if (SL && MODE == 0) v = slab_sum(e);
      else v = ld4(y + e);
      if (MODE == 1) {
        d4 = SL ? slab_sum(e) : ld4(da + e);
        if (has_mask) m4 = BCP_NORM_MASK4(ep, e);
      }
      accum(v, d4, m4);
    }
  }
  // block reduce over the row slots: xor-shuffles inside a wave (lanes with equal column are C4 apart), then one LDS hop
  // over the <= 4 waves.  (The previous version let C4 threads walk all 256/C4 slots serially -- for C = 16 that tail
  // took as long as the streaming loop itself.)
  double acc8[8] = {s1[0], s1[1], s1[2], s1[3], s2[0], s2[1], s2[2], s2[3]};
  for (int off = C4; off < 64; off <<= 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc8[k] += __shfl_xor(acc8[k], off);
  }
  constexpr int NE = 4;                                   // LDS entries: one per wave (C4 <= 64) or per 64-column slice
  __shared__ double red[NE][64][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool leader = (C4 >= 64) || lane < C4;
  if (leader) {
#pragma unroll
    for (int k = 0; k < 8; ++k) red[wave][lane][k] = acc8[k];
  }
  __syncthreads();
  if ((int)threadIdx.x < C4) {
    // C4 <= 64: column col lives at lane col of every wave.  C4 = 128 / 256: column col lives in wave (col / 64) + j * (C4 / 64)
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int wpr = C4 >= 64 ? C4 / 64 : 1;               // waves per row slot
    const int w0 = C4 >= 64 ? col / 64 : 0;
    for (int w = w0; w < 4; w += wpr) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += red[w][col & 63][k];
    }
    const long long prow = (long long)g * spg * nbps + (long long)(seg - g * spg) * nbps + blockIdx.x;
    double* out = partial + (prow * C + col * 4) * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) { out[k * 2] = a[k]; out[k * 2 + 1] = a[4 + k]; }
  }
}

// Sum the per-block partials of 16 channels of one group: 1024 threads = 32 doubles (16 channels x {s1, s2}, one 256-byte
// row of the partial table) x 32 row-slots, four independent loads in flight per thread, then an LDS tree.  Threads 0..15
// return true with the two sums of channel chunk*16 + tid.  (A single thread walking ~1000 partials serially cost more than
// the streaming pass itself; 16 slots with one load in flight left this kernel at ~8 us on the step's critical path.)
constexpr int kFinalizeThreads = 1024;
__device__ __forceinline__ bool reduce_partials(const double* __restrict__ partial, int nb, int C, int g, int chunk,
                                                double& s1, double& s2) {
  __shared__ double red[32][33];
  __shared__ double fin[32];
  const int e = threadIdx.x & 31, slot = threadIdx.x >> 5;
  const double* p = partial + ((long long)g * nb * C + chunk * 16) * 2 + e;
  const long long rs = (long long)C * 2;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int b = slot;
  for (; b + 96 < nb; b += 128) {
    const double v0 = p[b * rs], v1 = p[(b + 32) * rs], v2 = p[(b + 64) * rs], v3 = p[(b + 96) * rs];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; b < nb; b += 32) a0 += p[b * rs];
  red[slot][e] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x < 32) {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int k = 0; k < 32; k += 2) { t0 += red[k][e]; t1 += red[k + 1][e]; }
    fin[e] = t0 + t1;
  }
  __syncthreads();
  if (threadIdx.x >= 16) return false;
  s1 = fin[threadIdx.x * 2];
  s2 = fin[threadIdx.x * 2 + 1];
  return true;
}

// forward finalize: block = (group g, 16-channel chunk)
__global__ __launch_bounds__(kFinalizeThreads) void k_norm_finalize(const double* __restrict__ partial, int nb, int G, int C,
                                                       long long rows_per_group, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, float momentum, float eps,
                                                       float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ scale,
                                                       float* __restrict__ shift, float* __restrict__ var_unb, float* __restrict__ amax_out) {
  if (amax_out && blockIdx.x == 0) amax_clear(amax_out);      // the apply pass that follows max-reduces |a| into the slots
  const int chunks = C >> 4;
  const int g = blockIdx.x / chunks, chunk = blockIdx.x % chunks, c = chunk * 16 + (threadIdx.x & 15);
  double s1, s2;
  if (!reduce_partials(partial, nb, C, g, chunk, s1, s2)) return;
  const int idx = g * C + c;
  const double n = (double)rows_per_group;
  const double m = s1 / n;
  double var = s2 / n - m * m;
  if (var < 0.0) var = 0.0;
  const double r = 1.0 / sqrt(var + (double)eps);
  mean[idx] = (float)m;
  rstd[idx] = (float)r;
  const double ga = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
  scale[idx] = (float)(ga * r);
  shift[idx] = (float)be;   // z = (y - mean) * scale + beta: subtract the mean FIRST (no cancellation against mean*scale)
  var_unb[idx] = (float)(n > 1.0 ? var * n / (n - 1.0) : var);   // torch's running_var uses the UNBIASED variance
}

// running = (1-momentum)*running + momentum*stat, applied group after group IN ORDER: a call with G groups is
// bit-for-bit G consecutive BatchNorm calls (the BCP step normalises its two student / teacher batches separately).
__device__ __forceinline__ void update_running(const float* __restrict__ mean, const float* __restrict__ var_unb, int G, int C,
                                               float* __restrict__ running_mean, float* __restrict__ running_var, float momentum) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double rm = (double)running_mean[c], rv = (double)running_var[c];
    for (int g = 0; g < G; ++g) {
      rm = (1.0 - (double)momentum) * rm + (double)momentum * (double)mean[g * C + c];
      rv = (1.0 - (double)momentum) * rv + (double)momentum * (double)var_unb[g * C + c];
      rm = (double)(float)rm;   // the reference rounds to fp32 after every call
      rv = (double)(float)rv;
    }
    running_mean[c] = (float)rm;
    running_var[c] = (float)rv;
  }
}

// backward finalize: dgamma/dbeta (+= or =) and the two per-(g,c) means used by the apply pass
__global__ __launch_bounds__(kFinalizeThreads) void k_norm_bwd_finalize(const double* __restrict__ partial, int nb, int G, int C,
                                                           long long rows_per_group, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int accumulate, float* __restrict__ c1,
                                                           float* __restrict__ c2, float* __restrict__ raw /* [2][G][C] */,
                                                           float* __restrict__ amax_out) {
  if (amax_out && blockIdx.x == 0) amax_clear(amax_out);      // the apply pass that follows max-reduces |dy| into the slots
  const int chunks = C >> 4;
  const int g = blockIdx.x / chunks, chunk = blockIdx.x % chunks, c = chunk * 16 + (threadIdx.x & 15);
  double s1, s2;
  if (!reduce_partials(partial, nb, C, g, chunk, s1, s2)) return;
  const int idx = g * C + c;
  c1[idx] = (float)(s1 / (double)rows_per_group);
  c2[idx] = (float)(s2 / (double)rows_per_group);
  raw[idx] = (float)s1;
  raw[(long long)G * C + idx] = (float)s2;
}

// ------------------------------------------------------------------ apply passes
// row-reversed position of float4 index p inside a segment (same column): (rows-1-r)*C4 + col
#define REV(p) (nv - C4 - (p) + 2 * col)
// a = act(y*scale + shift) [* chan_scale] [* elem_mask*elem_scale] [+ residual]      (segments: see k_col_partial)
__global__ __launch_bounds__(256) void k_norm_running_only(const float* __restrict__ mean, const float* __restrict__ var_unb, int G, int C,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var, float momentum) {
  update_running(mean, var_unb, G, C, running_mean, running_var, momentum);
}

__global__ __launch_bounds__(256) void k_norm_apply(const float* __restrict__ y, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, const float* __restrict__ mean,
                                                    const float* __restrict__ residual, NormEpilogue ep, long long seg_rows,
                                                    int spg, int G, int C, float* __restrict__ out,
                                                    const float* __restrict__ var_unb, float* __restrict__ running_mean,
                                                    float* __restrict__ running_var, float momentum, float* __restrict__ amax_out,
                                                    long long ldo4 /* row stride of out in float4: C / 4, or wider when out is the first C
                                                                      channels of a concat buffer (bcp_norm_fwd out_ld) */) {
  constexpr int U = 4;
  BCP_NORM_MASK_PROLOGUE(ep)
  float amax = 0.f;            // max |a| of what this thread writes (round 4: the fp16 pre-scale of the conv that reads a, conv3_defs.h)
  if (blockIdx.x == 0 && blockIdx.y == 0 && running_mean) update_running(mean, var_unb, G, C, running_mean, running_var, momentum);
  const int C4 = C >> 2;
  const int c4sh = 31 - __clz(C4);
  const int col = threadIdx.x & (C4 - 1);          // C4 is a power of two dividing 256: fixed for the whole loop
  const int seg = blockIdx.y, g = seg / spg;
  const long long nv = seg_rows * C4, base = (long long)seg * nv;
  const long long stride = (long long)gridDim.x * 256;
  const float4 sc = ld4(scale + (long long)g * C + col * 4), sh = ld4(shift + (long long)g * C + col * 4);
  const float4 mu = ld4(mean + (long long)g * C + col * 4);
  float4 cs = make_float4(1.f, 1.f, 1.f, 1.f);
  if (ep.chan_scale) cs = ld4(ep.chan_scale + (((long long)seg * seg_rows) / ep.rows_per_sample) * C + col * 4);
  auto one = [&](long long i, const float4& v, const float4& r4, const uchar4& m4) {
    float o[4] = {act_fwd((v.x - mu.x) * sc.x + sh.x, ep.act), act_fwd((v.y - mu.y) * sc.y + sh.y, ep.act),
                  act_fwd((v.z - mu.z) * sc.z + sh.z, ep.act), act_fwd((v.w - mu.w) * sc.w + sh.w, ep.act)};
    if (ep.chan_scale) { o[0] *= cs.x; o[1] *= cs.y; o[2] *= cs.z; o[3] *= cs.w; }
    if (has_mask) {
      o[0] *= m4.x ? ep.elem_scale : 0.f; o[1] *= m4.y ? ep.elem_scale : 0.f;
      o[2] *= m4.z ? ep.elem_scale : 0.f; o[3] *= m4.w ? ep.elem_scale : 0.f;
    }
    if (residual) { o[0] += r4.x; o[1] += r4.y; o[2] += r4.z; o[3] += r4.w; }
    const long long io = ldo4 == C4 ? i : ((i - col) >> c4sh) * ldo4 + col;      // (i - col) / C4 = row: C4 is a power of two
    st4(out + io * 4, make_float4(o[0], o[1], o[2], o[3]));
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float t = fabsf(o[k]); amax = (t > amax || t != t) ? t : amax; }
  };
  // Back to front: the statistics pass (or the conv epilogue) that ran just before this kernel swept the tensor front to
  // back, so its tail is what the 256 MB memory-side cache still holds.  (col stays fixed: nv and stride are multiples of C4.)
  long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; j + (U - 1) * stride < nv; j += U * stride) {
    float4 v[U], r4[U];
    uchar4 m4[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = base + REV(j + u * stride);
      v[u] = ld4(y + i * 4);
      if (residual) r4[u] = ld4(residual + i * 4);
      if (has_mask) m4[u] = BCP_NORM_MASK4(ep, i * 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one(base + REV(j + u * stride), v[u], r4[u], m4[u]);
  }
  for (; j < nv; j += stride) {
    const long long i = base + REV(j);
    float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uchar4 m4 = make_uchar4(0, 0, 0, 0);
    const float4 v = ld4(y + i * 4);
    if (residual) r4 = ld4(residual + i * 4);
    if (has_mask) m4 = BCP_NORM_MASK4(ep, i * 4);
    one(i, v, r4, m4);
  }
  if (amax_out) block_amax_publish(amax, amax_out);
}

// dy = scale * (dz - c1 - xhat*c2),  dz = da * epilogue' * act'(z)
__global__ __launch_bounds__(256) void k_norm_bwd_apply(const float* __restrict__ y, const float* __restrict__ da,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ c1, const float* __restrict__ c2,
                                                        NormEpilogue ep, long long seg_rows, int spg, int G, int C,
                                                        float* __restrict__ dy, const float* __restrict__ raw,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                        float* __restrict__ amax_out) {
  constexpr int U = 4;
  BCP_NORM_MASK_PROLOGUE(ep)
  float amax = 0.f;            // max |dy| of what this thread writes: the fp16 pre-scale of the dgrad / weight-gradient kernels that read dy
  if (blockIdx.x == 0 && blockIdx.y == 0 && dgamma) {   // parameter gradients: sum the groups in order (deterministic)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float gb = accumulate ? dbeta[c] : 0.f, gg = accumulate ? dgamma[c] : 0.f;
      for (int g = 0; g < G; ++g) { gb += raw[g * C + c]; gg += raw[(long long)G * C + g * C + c]; }
      dbeta[c] = gb;
      dgamma[c] = gg;
    }
  }
  const int C4 = C >> 2;
  const int col = threadIdx.x & (C4 - 1);
  const int seg = blockIdx.y, g = seg / spg;
  const long long nv = seg_rows * C4, base = (long long)seg * nv;
  const long long stride = (long long)gridDim.x * 256;
  const long long gc = (long long)g * C + col * 4;
  const float4 sc = ld4(scale + gc), sh = ld4(shift + gc), mu = ld4(mean + gc), rs = ld4(rstd + gc);
  const float4 k1 = ld4(c1 + gc), k2 = ld4(c2 + gc);
  float4 csl = make_float4(1.f, 1.f, 1.f, 1.f);
  if (ep.chan_scale) csl = ld4(ep.chan_scale + (((long long)seg * seg_rows) / ep.rows_per_sample) * C + col * 4);
  const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
  const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
  const float k1v[4] = {k1.x, k1.y, k1.z, k1.w}, k2v[4] = {k2.x, k2.y, k2.z, k2.w};
  auto one = [&](long long i, const float4& v, const float4& d4, const uchar4& m4) {
    float cs[4] = {csl.x, csl.y, csl.z, csl.w};
    if (has_mask) {
      cs[0] *= m4.x ? ep.elem_scale : 0.f; cs[1] *= m4.y ? ep.elem_scale : 0.f;
      cs[2] *= m4.z ? ep.elem_scale : 0.f; cs[3] *= m4.w ? ep.elem_scale : 0.f;
    }
    const float vv[4] = {v.x, v.y, v.z, v.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float z = (vv[k] - muv[k]) * scv[k] + shv[k];
      const float dz = dd[k] * cs[k] * act_grad(z, ep.act);
      const float xh = (vv[k] - muv[k]) * rsv[k];
      o[k] = scv[k] * (dz - k1v[k] - xh * k2v[k]);
    }
    st4(dy + i * 4, make_float4(o[0], o[1], o[2], o[3]));
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float t = fabsf(o[q]); amax = (t > amax || t != t) ? t : amax; }
  };
  long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; j + (U - 1) * stride < nv; j += U * stride) {
    float4 v[U], d4[U];
    uchar4 m4[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = base + REV(j + u * stride);
      v[u] = ld4(y + i * 4);
      d4[u] = ld4(da + i * 4);
      if (has_mask) m4[u] = BCP_NORM_MASK4(ep, i * 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one(base + REV(j + u * stride), v[u], d4[u], m4[u]);
  }
  for (; j < nv; j += stride) {
    const long long i = base + REV(j);
    uchar4 m4 = make_uchar4(0, 0, 0, 0);
    const float4 v = ld4(y + i * 4), d4 = ld4(da + i * 4);
    if (has_mask) m4 = BCP_NORM_MASK4(ep, i * 4);
    one(i, v, d4, m4);
  }
  if (amax_out) block_amax_publish(amax, amax_out);
}


// ------------------------------------------------------------------ apply passes that finalise the statistics themselves (round 6)
// A dependent launch costs ~4 us on the step's critical path whatever it does (measured: the LA step with every finalize launch left out is
// 7.7 % faster), and the finalize kernel does ~nothing.  Where the statistics pass left FEW partial rows (nb <= kFinMaxRows: the deep
// levels, tensors of a few MB) the apply pass reduces them itself: a workgroup owns a 32-channel chunk (blockIdx.z) of a row block of one
// segment, sums the nb x 32 x 2 doubles of its chunk (<= 64 KB, L2-resident) in a fixed order -- every workgroup gets the same bits -- and
// forms mean / rstd / scale / shift (forward) or c1 / c2 (backward) in the LDS.  The first workgroup of a group writes the statistics table
// (the backward pass reads it), workgroup (0, 0, chunk) the running statistics / parameter gradients of its chunk, group after group in
// order.  Same arithmetic per element as k_norm_finalize + k_norm_apply / k_norm_bwd_finalize + k_norm_bwd_apply; the partial rows are
// summed in ascending order per slot (the finalize kernels sum them in 32 slots and a tree), so the statistics can differ in the last
// bit of their fp64 sums.
constexpr int kFinMaxRows = 128;       // partial rows per group the fused apply passes accept
constexpr int kFinChunk = 32;          // channels per workgroup

// sums over the nb partial rows of group g for the cw channels of chunk c0 (both statistics: one double2 per row and channel): -> fin[2 * cw]
// (LDS), all threads return after the barrier behind it.  red: 512 doubles of LDS.  Thread = (channel j, row slot sl); its <= kFinMaxRows / S
// rows are ALL requested before the first add (one round trip: with four loads per loop trip the prologue was eight dependent round
// trips and cost what the launch it replaces had), and added in ascending order; the S slots are added in order by thread j.
__device__ __forceinline__ void fin_reduce_chunk(const double* __restrict__ partial, int nb, int C, int g, int c0, int cw, double* red,
                                                 double* fin) {
  const int S = 256 / cw;                                     // 8 (cw = 32) or 16 row slots
  const int j = threadIdx.x % cw, sl = threadIdx.x / cw;
  const long long rs = (long long)C * 2;
  const double* p = partial + ((long long)g * nb * C + c0 + j) * 2;
  constexpr int R = kFinMaxRows / 8;                          // rows per thread at most
  double2 v[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int b = sl + k * S;
    v[k] = b < nb ? *reinterpret_cast<const double2*>(p + b * rs) : make_double2(0.0, 0.0);
  }
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int k = 0; k < R; ++k) { a0 += v[k].x; a1 += v[k].y; }
  __syncthreads();                                            // (a previous call's readers of red / fin are done)
  red[threadIdx.x * 2] = a0;
  red[threadIdx.x * 2 + 1] = a1;
  __syncthreads();
  if ((int)threadIdx.x < 2 * cw) {
    const int c = threadIdx.x >> 1, st = threadIdx.x & 1;
    double t = 0.0;
    for (int k = 0; k < S; ++k) t += red[(k * cw + c) * 2 + st];
    fin[threadIdx.x] = t;
  }
  __syncthreads();
}

struct FinFwd {
  const double* partial; int nb;
  const float *gamma, *beta; float *running_mean, *running_var; float momentum, eps;
  float* stats;                // [5][G][C]
};

__global__ __launch_bounds__(256) void k_norm_apply_fin(const float* __restrict__ y, const float* __restrict__ residual, NormEpilogue ep,
                                                        long long seg_rows, int spg, int G, int C, long long rows_per_group, FinFwd f,
                                                        float* __restrict__ out, float* __restrict__ amax_out, long long ldo) {
  constexpr int U = 4;
  __shared__ double red[512];
  __shared__ double fin[2 * kFinChunk];
  __shared__ float pmu[kFinChunk], psc[kFinChunk], psh[kFinChunk];
  BCP_NORM_MASK_PROLOGUE(ep)
  const int cw = C < kFinChunk ? C : kFinChunk, c0 = blockIdx.z * cw;
  const int seg = blockIdx.y, g = seg / spg;
  float *mean = f.stats, *rstd = f.stats + (long long)G * C, *scale = f.stats + 2LL * G * C, *shift = f.stats + 3LL * G * C;
  float* var_unb = f.stats + 4LL * G * C;
  const double n = (double)rows_per_group;
  // one group's statistics of this chunk: thread c < cw holds channel c0 + c
  auto group_stats = [&](int gg, float& m_f, float& r_f, float& sc_f, float& sh_f, float& vu_f) {
    fin_reduce_chunk(f.partial, f.nb, C, gg, c0, cw, red, fin);
    if ((int)threadIdx.x < cw) {
      const int c = c0 + threadIdx.x;
      const double s1 = fin[threadIdx.x * 2], s2 = fin[threadIdx.x * 2 + 1];
      const double m = s1 / n;
      double var = s2 / n - m * m;
      if (var < 0.0) var = 0.0;
      const double r = 1.0 / sqrt(var + (double)f.eps);
      const double ga = f.gamma ? (double)f.gamma[c] : 1.0, be = f.beta ? (double)f.beta[c] : 0.0;
      m_f = (float)m; r_f = (float)r; sc_f = (float)(ga * r); sh_f = (float)be;
      vu_f = (float)(n > 1.0 ? var * n / (n - 1.0) : var);
    }
  };
  // running statistics of this chunk, the groups in order: an EXTRA workgroup per chunk (blockIdx.x == gridDim.x - 1, launched only when there
  // are running statistics) that does nothing else -- inside a working block the G - 1 extra reductions were the launch's tail
  if (f.running_mean && blockIdx.x == gridDim.x - 1) {
    if (blockIdx.y != 0) return;
    double rm = 0.0, rv = 0.0;
    if ((int)threadIdx.x < cw) { rm = (double)f.running_mean[c0 + threadIdx.x]; rv = (double)f.running_var[c0 + threadIdx.x]; }
    for (int gg = 0; gg < G; ++gg) {
      float m_f = 0.f, r_f = 0.f, sc_f = 0.f, sh_f = 0.f, vu_f = 0.f;
      group_stats(gg, m_f, r_f, sc_f, sh_f, vu_f);
      if ((int)threadIdx.x < cw) {      // update_running's arithmetic: fp32 rounding after every group
        rm = (double)(float)((1.0 - (double)f.momentum) * rm + (double)f.momentum * (double)m_f);
        rv = (double)(float)((1.0 - (double)f.momentum) * rv + (double)f.momentum * (double)vu_f);
      }
    }
    if ((int)threadIdx.x < cw) { f.running_mean[c0 + threadIdx.x] = (float)rm; f.running_var[c0 + threadIdx.x] = (float)rv; }
    return;
  }
  const int nbx = f.running_mean ? (int)gridDim.x - 1 : (int)gridDim.x;      // working blocks along x
  const int cw4 = cw >> 2, col = threadIdx.x % cw4, slot = threadIdx.x / cw4, slots = 256 / cw4;
  const long long sbase = (long long)seg * seg_rows;
  const long long stride = (long long)nbx * slots;
  long long r = (long long)blockIdx.x * slots + slot;
  // the first trip's loads go out BEFORE the statistics are reduced: their latency runs under the prologue (at the deep levels a thread has one trip)
  float4 v0[U], q0[U];
  uchar4 k0[U];
  if (out) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long rr = r + u * stride;
      v0[u] = q0[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      k0[u] = make_uchar4(0, 0, 0, 0);
      if (rr < seg_rows) {
        const long long e = (sbase + rr) * C + c0 + col * 4;
        v0[u] = ld4(y + e);
        if (residual) q0[u] = ld4(residual + e);
        if (has_mask) k0[u] = BCP_NORM_MASK4(ep, e);
      }
    }
  }
  {
    float m_f = 0.f, r_f = 0.f, sc_f = 0.f, sh_f = 0.f, vu_f = 0.f;
    group_stats(g, m_f, r_f, sc_f, sh_f, vu_f);
    if ((int)threadIdx.x < cw) {
      pmu[threadIdx.x] = m_f; psc[threadIdx.x] = sc_f; psh[threadIdx.x] = sh_f;
      if (blockIdx.x == 0 && seg == g * spg) {      // the group's first workgroup leaves the table the backward pass reads
        const int idx = g * C + c0 + threadIdx.x;
        mean[idx] = m_f; rstd[idx] = r_f; scale[idx] = sc_f; shift[idx] = sh_f; var_unb[idx] = vu_f;
      }
    }
    __syncthreads();
  }
  if (!out) return;                                            // statistics only
  const float4 mu = *reinterpret_cast<const float4*>(pmu + col * 4), sc = *reinterpret_cast<const float4*>(psc + col * 4);
  const float4 sh = *reinterpret_cast<const float4*>(psh + col * 4);
  float4 cs = make_float4(1.f, 1.f, 1.f, 1.f);
  if (ep.chan_scale) cs = ld4(ep.chan_scale + (sbase / ep.rows_per_sample) * C + c0 + col * 4);
  float amax = 0.f;
  auto one = [&](long long row, const float4& v, const float4& r4, const uchar4& m4) {
    float o[4] = {act_fwd((v.x - mu.x) * sc.x + sh.x, ep.act), act_fwd((v.y - mu.y) * sc.y + sh.y, ep.act),
                  act_fwd((v.z - mu.z) * sc.z + sh.z, ep.act), act_fwd((v.w - mu.w) * sc.w + sh.w, ep.act)};
    if (ep.chan_scale) { o[0] *= cs.x; o[1] *= cs.y; o[2] *= cs.z; o[3] *= cs.w; }
    if (has_mask) {
      o[0] *= m4.x ? ep.elem_scale : 0.f; o[1] *= m4.y ? ep.elem_scale : 0.f;
      o[2] *= m4.z ? ep.elem_scale : 0.f; o[3] *= m4.w ? ep.elem_scale : 0.f;
    }
    if (residual) { o[0] += r4.x; o[1] += r4.y; o[2] += r4.z; o[3] += r4.w; }
    st4(out + row * ldo + c0 + col * 4, make_float4(o[0], o[1], o[2], o[3]));
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float t = fabsf(o[k]); amax = (t > amax || t != t) ? t : amax; }
  };
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (r + u * stride < seg_rows) one(sbase + r + u * stride, v0[u], q0[u], k0[u]);
  r += U * stride;
  for (; r + (U - 1) * stride < seg_rows; r += U * stride) {
    float4 v[U], r4[U];
    uchar4 m4[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long e = (sbase + r + u * stride) * C + c0 + col * 4;
      v[u] = ld4(y + e);
      if (residual) r4[u] = ld4(residual + e);
      if (has_mask) m4[u] = BCP_NORM_MASK4(ep, e);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one(sbase + r + u * stride, v[u], r4[u], m4[u]);
  }
  for (; r < seg_rows; r += stride) {
    const long long e = (sbase + r) * C + c0 + col * 4;
    float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uchar4 m4 = make_uchar4(0, 0, 0, 0);
    const float4 v = ld4(y + e);
    if (residual) r4 = ld4(residual + e);
    if (has_mask) m4 = BCP_NORM_MASK4(ep, e);
    one(sbase + r, v, r4, m4);
  }
  if (amax_out) block_amax_publish(amax, amax_out);
}

struct FinBwd {
  const double* partial; int nb;
  float *dgamma, *dbeta; int accumulate;
};

__global__ __launch_bounds__(256) void k_norm_bwd_apply_fin(const float* __restrict__ y, const float* __restrict__ da,
                                                            const float* __restrict__ stats, NormEpilogue ep, long long seg_rows, int spg,
                                                            int G, int C, long long rows_per_group, FinBwd f, float* __restrict__ dy,
                                                            float* __restrict__ amax_out) {
  constexpr int U = 4;
  __shared__ double red[512];
  __shared__ double fin[2 * kFinChunk];
  __shared__ float pc1[kFinChunk], pc2[kFinChunk];
  BCP_NORM_MASK_PROLOGUE(ep)
  const int cw = C < kFinChunk ? C : kFinChunk, c0 = blockIdx.z * cw;
  const int seg = blockIdx.y, g = seg / spg;
  const float *mean = stats, *rstd = stats + (long long)G * C, *scale = stats + 2LL * G * C, *shift = stats + 3LL * G * C;
  // parameter gradients of this chunk, the groups in order (k_norm_bwd_apply's block (0, 0)): an EXTRA workgroup per chunk, as in the forward kernel
  if (f.dgamma && blockIdx.x == gridDim.x - 1) {
    if (blockIdx.y != 0) return;
    float gb = 0.f, gg = 0.f;
    if ((int)threadIdx.x < cw && f.accumulate) { gb = f.dbeta[c0 + threadIdx.x]; gg = f.dgamma[c0 + threadIdx.x]; }
    for (int q = 0; q < G; ++q) {
      fin_reduce_chunk(f.partial, f.nb, C, q, c0, cw, red, fin);
      if ((int)threadIdx.x < cw) { gb += (float)fin[threadIdx.x * 2]; gg += (float)fin[threadIdx.x * 2 + 1]; }
    }
    if ((int)threadIdx.x < cw) { f.dbeta[c0 + threadIdx.x] = gb; f.dgamma[c0 + threadIdx.x] = gg; }
    return;
  }
  const int nbx = f.dgamma ? (int)gridDim.x - 1 : (int)gridDim.x;
  const int cw4 = cw >> 2, col = threadIdx.x % cw4, slot = threadIdx.x / cw4, slots = 256 / cw4;
  const long long sbase = (long long)seg * seg_rows;
  const long long stride = (long long)nbx * slots;
  long long r = (long long)blockIdx.x * slots + slot;
  float4 v0[U], d0[U];
  uchar4 k0[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {      // the first trip's loads in front of the prologue
    const long long rr = r + u * stride;
    v0[u] = d0[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    k0[u] = make_uchar4(0, 0, 0, 0);
    if (rr < seg_rows) {
      const long long e = (sbase + rr) * C + c0 + col * 4;
      v0[u] = ld4(y + e);
      d0[u] = ld4(da + e);
      if (has_mask) k0[u] = BCP_NORM_MASK4(ep, e);
    }
  }
  const long long gc = (long long)g * C + c0 + col * 4;
  const float4 sc = ld4(scale + gc), sh = ld4(shift + gc), mu = ld4(mean + gc), rs = ld4(rstd + gc);
  float4 csl = make_float4(1.f, 1.f, 1.f, 1.f);
  if (ep.chan_scale) csl = ld4(ep.chan_scale + (sbase / ep.rows_per_sample) * C + c0 + col * 4);
  fin_reduce_chunk(f.partial, f.nb, C, g, c0, cw, red, fin);
  if ((int)threadIdx.x < cw) {
    pc1[threadIdx.x] = (float)(fin[threadIdx.x * 2] / (double)rows_per_group);
    pc2[threadIdx.x] = (float)(fin[threadIdx.x * 2 + 1] / (double)rows_per_group);
  }
  __syncthreads();
  const float4 k1 = *reinterpret_cast<const float4*>(pc1 + col * 4), k2 = *reinterpret_cast<const float4*>(pc2 + col * 4);
  const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
  const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
  const float k1v[4] = {k1.x, k1.y, k1.z, k1.w}, k2v[4] = {k2.x, k2.y, k2.z, k2.w};
  float amax = 0.f;
  auto one = [&](long long e, const float4& v, const float4& d4, const uchar4& m4) {
    float cs[4] = {csl.x, csl.y, csl.z, csl.w};
    if (has_mask) {
      cs[0] *= m4.x ? ep.elem_scale : 0.f; cs[1] *= m4.y ? ep.elem_scale : 0.f;
      cs[2] *= m4.z ? ep.elem_scale : 0.f; cs[3] *= m4.w ? ep.elem_scale : 0.f;
    }
    const float vv[4] = {v.x, v.y, v.z, v.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float z = (vv[k] - muv[k]) * scv[k] + shv[k];
      const float dz = dd[k] * cs[k] * act_grad(z, ep.act);
      const float xh = (vv[k] - muv[k]) * rsv[k];
      o[k] = scv[k] * (dz - k1v[k] - xh * k2v[k]);
    }
    st4(dy + e, make_float4(o[0], o[1], o[2], o[3]));
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float t = fabsf(o[q]); amax = (t > amax || t != t) ? t : amax; }
  };
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (r + u * stride < seg_rows) one((sbase + r + u * stride) * C + c0 + col * 4, v0[u], d0[u], k0[u]);
  r += U * stride;
  for (; r + (U - 1) * stride < seg_rows; r += U * stride) {
    float4 v[U], d4[U];
    uchar4 m4[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long e = (sbase + r + u * stride) * C + c0 + col * 4;
      v[u] = ld4(y + e);
      d4[u] = ld4(da + e);
      if (has_mask) m4[u] = BCP_NORM_MASK4(ep, e);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one((sbase + r + u * stride) * C + c0 + col * 4, v[u], d4[u], m4[u]);
  }
  for (; r < seg_rows; r += stride) {
    const long long e = (sbase + r) * C + c0 + col * 4;
    uchar4 m4 = make_uchar4(0, 0, 0, 0);
    const float4 v = ld4(y + e), d4 = ld4(da + e);
    if (has_mask) m4 = BCP_NORM_MASK4(ep, e);
    one(e, v, d4, m4);
  }
  if (amax_out) block_amax_publish(amax, amax_out);
}

// ------------------------------------------------------------------ the smallest levels: the whole norm in ONE launch (round 6)
// Where a group has a few hundred rows (LA 7x7x5, pancreas 6^3: <= 512 / GS rows) a 1024-thread workgroup OWNS a 32-channel chunk of GS
// groups for all their rows: 128-byte row segments (whole cache lines -- round 3's one-launch form owned FOUR channels and was bound by one
// line per lane, tools/attic/norm_small.hip), at most four rows per thread held in registers, the slab sum, the statistics, their
// finalisation and the apply pass without a second read or a launch boundary in between.  GS = G when the groups are order-dependent
// (running statistics / parameter gradients: G <= 2, both groups in one workgroup, finished in order), else 1 (grid.y = G).  At most 32
// workgroups per launch: each writes its |max| into a slot of its own (no atomics, nobody has to clear them).  Same arithmetic per element
// as the streaming kernels; the fp64 statistics are summed in a different fixed order (rows of a thread, lanes, waves).
constexpr int kOwnThreads = 1024, kOwnMaxP = 4;

struct OwnCommon {
  const float* y;              // fwd: pre-norm input when nslab == 0; bwd: the saved pre-norm tensor
  const float* slabs; int nslab; long long slab_stride; const float* bias;   // fwd: conv slabs (+ bias); bwd: da slabs
  float* sum_out;              // fwd: y written here when nslab > 0; bwd: da sum (nullable)
  int G, C, R, GS, cw;         // groups, channels, rows per group, groups per workgroup, channels per workgroup (8 / 16 / 32)
  NormEpilogue ep;
  float* amax_out;
};

// block reduction of 8 doubles per thread over the row slots of every group slot: -> fin[gs][32 channels][2] in LDS
__device__ __forceinline__ void own_reduce(double (&acc8)[8], int GS, int cw, double* red /* [16][8][8] */, double* fin /* [GS][64] */) {
  const int cw4 = cw >> 2;
  // lanes with equal column are cw4 apart
  for (int off = cw4; off < 64; off <<= 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc8[k] += __shfl_xor(acc8[k], off);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < cw4) {
#pragma unroll
    for (int k = 0; k < 8; ++k) red[(wave * 8 + lane) * 8 + k] = acc8[k];
  }
  __syncthreads();
  const int wpg = 16 / GS;                                     // waves per group slot
  if ((int)threadIdx.x < GS * 2 * cw) {
    const int gs = threadIdx.x / (2 * cw), j = threadIdx.x % (2 * cw), c = j >> 1, st = j & 1;      // channel c of the chunk, statistic st
    const int col = c >> 2, k = (c & 3) + st * 4;
    double t = 0.0;
    for (int w = 0; w < wpg; ++w) t += red[((gs * wpg + w) * 8 + col) * 8 + k];
    fin[gs * 64 + j] = t;
  }
  __syncthreads();
}

struct OwnFwd {
  const float *gamma, *beta; float *running_mean, *running_var; float momentum, eps;
  const float* residual; float* stats; float* out; long long ldo;
};

template <int P>
__global__ __launch_bounds__(kOwnThreads) void k_norm_own_fwd(OwnCommon a, OwnFwd f) {
  __shared__ double red[16 * 8 * 8];
  __shared__ double fin[2 * 64];
  __shared__ float pmu[2 * 32], psc[2 * 32], psh[2 * 32];
  __shared__ float amax_red[16];
  const NormEpilogue& ep = a.ep;
  BCP_NORM_MASK_PROLOGUE(ep)
  const int C = a.C, R = a.R, GS = a.GS, cw = a.cw, cw4 = cw >> 2, c0 = blockIdx.x * cw;
  const int col = threadIdx.x & (cw4 - 1), slot = threadIdx.x / cw4;     // 1024 / cw4 row slots
  const int spg = (kOwnThreads / cw4) / GS;                      // row slots per group slot
  const int gs = slot / spg, rs = slot % spg;
  const int g = blockIdx.y * GS + gs;
  float4 v[P], q[P];
  uchar4 m4[P];
  bool ok[P];
  long long eo[P];
  const bool SL = a.nslab > 0;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (SL && a.bias) bias4 = ld4(a.bias + c0 + col * 4);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int r = rs + p * spg;
    ok[p] = r < R;
    eo[p] = ((long long)g * R + r) * C + c0 + col * 4;
    v[p] = SL ? bias4 : make_float4(0.f, 0.f, 0.f, 0.f);
    q[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    m4[p] = make_uchar4(0, 0, 0, 0);
    if (ok[p]) {
      if (!SL) v[p] = ld4(a.y + eo[p]);
      if (f.residual) q[p] = ld4(f.residual + eo[p]);
      if (has_mask) m4[p] = BCP_NORM_MASK4(ep, eo[p]);
    }
  }
  if (SL) {      // bias first, then the slabs front to back (k_b6_sum_slabs' order); up to four slabs' loads go out before the first add
    constexpr int KU = P >= 4 ? 2 : 4;      // slabs requested together (P = 4: two, or the loads spill)
    for (int k0 = 0; k0 < a.nslab; k0 += KU) {
      float4 t[KU][P];
#pragma unroll
      for (int kk = 0; kk < KU; ++kk)
#pragma unroll
        for (int p = 0; p < P; ++p)
          t[kk][p] = (k0 + kk < a.nslab && ok[p]) ? ld4(a.slabs + (long long)(k0 + kk) * a.slab_stride + eo[p]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int kk = 0; kk < KU; ++kk)
        if (k0 + kk < a.nslab) {
#pragma unroll
          for (int p = 0; p < P; ++p) { v[p].x += t[kk][p].x; v[p].y += t[kk][p].y; v[p].z += t[kk][p].z; v[p].w += t[kk][p].w; }
        }
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (ok[p]) st4(a.sum_out + eo[p], v[p]);
  }
  double acc8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int p = 0; p < P; ++p)
    if (ok[p]) {
      const float vv[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { acc8[k] += (double)vv[k]; acc8[4 + k] += (double)vv[k] * (double)vv[k]; }
    }
  own_reduce(acc8, GS, cw, red, fin);
  float *mean = f.stats, *rstd = f.stats + (long long)a.G * C, *scale = f.stats + 2LL * a.G * C, *shift = f.stats + 3LL * a.G * C;
  float* var_unb = f.stats + 4LL * a.G * C;
  float vu_mine = 0.f, m_mine = 0.f;
  if ((int)threadIdx.x < GS * cw) {      // thread = (group slot, channel): k_norm_finalize's arithmetic
    const int gq = threadIdx.x / cw, c = threadIdx.x % cw, gg = blockIdx.y * GS + gq;
    const double n = (double)R;
    const double s1 = fin[gq * 64 + c * 2], s2 = fin[gq * 64 + c * 2 + 1];
    const double m = s1 / n;
    double var = s2 / n - m * m;
    if (var < 0.0) var = 0.0;
    const double r = 1.0 / sqrt(var + (double)f.eps);
    const double ga = f.gamma ? (double)f.gamma[c0 + c] : 1.0, be = f.beta ? (double)f.beta[c0 + c] : 0.0;
    const int idx = gg * C + c0 + c;
    m_mine = (float)m;
    vu_mine = (float)(n > 1.0 ? var * n / (n - 1.0) : var);
    mean[idx] = m_mine; rstd[idx] = (float)r; scale[idx] = (float)(ga * r); shift[idx] = (float)be; var_unb[idx] = vu_mine;
    pmu[gq * 32 + c] = m_mine; psc[gq * 32 + c] = (float)(ga * r); psh[gq * 32 + c] = (float)be;
  }
  __syncthreads();
  if (f.running_mean) {      // (GS = G here) the groups in order, fp32 rounding after each: update_running's arithmetic
    __shared__ float gm[2 * 32], gv[2 * 32];
    if ((int)threadIdx.x < GS * cw) { gm[(threadIdx.x / cw) * 32 + threadIdx.x % cw] = m_mine; gv[(threadIdx.x / cw) * 32 + threadIdx.x % cw] = vu_mine; }
    __syncthreads();
    if ((int)threadIdx.x < cw) {
      double rm = (double)f.running_mean[c0 + threadIdx.x], rv = (double)f.running_var[c0 + threadIdx.x];
      for (int gq = 0; gq < GS; ++gq) {
        rm = (double)(float)((1.0 - (double)f.momentum) * rm + (double)f.momentum * (double)gm[gq * 32 + threadIdx.x]);
        rv = (double)(float)((1.0 - (double)f.momentum) * rv + (double)f.momentum * (double)gv[gq * 32 + threadIdx.x]);
      }
      f.running_mean[c0 + threadIdx.x] = (float)rm;
      f.running_var[c0 + threadIdx.x] = (float)rv;
    }
  }
  const float4 mu = *reinterpret_cast<const float4*>(pmu + gs * 32 + col * 4), sc = *reinterpret_cast<const float4*>(psc + gs * 32 + col * 4);
  const float4 sh = *reinterpret_cast<const float4*>(psh + gs * 32 + col * 4);
  float amax = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (!ok[p]) continue;
    const long long row = (long long)g * R + rs + p * spg;
    float o[4] = {act_fwd((v[p].x - mu.x) * sc.x + sh.x, ep.act), act_fwd((v[p].y - mu.y) * sc.y + sh.y, ep.act),
                  act_fwd((v[p].z - mu.z) * sc.z + sh.z, ep.act), act_fwd((v[p].w - mu.w) * sc.w + sh.w, ep.act)};
    if (ep.chan_scale) {
      const float4 cs = ld4(ep.chan_scale + (row / ep.rows_per_sample) * C + c0 + col * 4);
      o[0] *= cs.x; o[1] *= cs.y; o[2] *= cs.z; o[3] *= cs.w;
    }
    if (has_mask) {
      o[0] *= m4[p].x ? ep.elem_scale : 0.f; o[1] *= m4[p].y ? ep.elem_scale : 0.f;
      o[2] *= m4[p].z ? ep.elem_scale : 0.f; o[3] *= m4[p].w ? ep.elem_scale : 0.f;
    }
    if (f.residual) { o[0] += q[p].x; o[1] += q[p].y; o[2] += q[p].z; o[3] += q[p].w; }
    st4(f.out + row * f.ldo + c0 + col * 4, make_float4(o[0], o[1], o[2], o[3]));
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float t = fabsf(o[k]); amax = (t > amax || t != t) ? t : amax; }
  }
  if (a.amax_out) {      // this workgroup's slot is its own; workgroup 0 also defines the slots nobody owns
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(amax, o); amax = (t > amax || t != t) ? t : amax; }
    if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = amax;
    __syncthreads();
    const int nwg = gridDim.x * gridDim.y, me = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) {
      float m = amax_red[0];
      for (int k = 1; k < 16; ++k) { const float t = amax_red[k]; m = (t > m || t != t) ? t : m; }
      if (m != m) m = __uint_as_float(0x7fc00000u);
      a.amax_out[me * kAmaxStride] = m;
    }
    if (me == 0 && (int)threadIdx.x >= nwg && (int)threadIdx.x < kAmaxSlots) a.amax_out[threadIdx.x * kAmaxStride] = 0.f;
  }
}

struct OwnBwd {
  const float* stats; const float* da; float *dgamma, *dbeta; int accumulate; float* dy;
};

template <int P>
__global__ __launch_bounds__(kOwnThreads) void k_norm_own_bwd(OwnCommon a, OwnBwd f) {
  __shared__ double red[16 * 8 * 8];
  __shared__ double fin[2 * 64];
  __shared__ float pc1[2 * 32], pc2[2 * 32];
  __shared__ float amax_red[16];
  const NormEpilogue& ep = a.ep;
  BCP_NORM_MASK_PROLOGUE(ep)
  const int C = a.C, R = a.R, GS = a.GS, cw = a.cw, cw4 = cw >> 2, c0 = blockIdx.x * cw;
  const int col = threadIdx.x & (cw4 - 1), slot = threadIdx.x / cw4;
  const int spg = (kOwnThreads / cw4) / GS;
  const int gs = slot / spg, rs = slot % spg;
  const int g = blockIdx.y * GS + gs;
  const float *mean = f.stats, *rstd = f.stats + (long long)a.G * C, *scale = f.stats + 2LL * a.G * C, *shift = f.stats + 3LL * a.G * C;
  const long long gc = (long long)g * C + c0 + col * 4;
  const float4 sc = ld4(scale + gc), sh = ld4(shift + gc), mu = ld4(mean + gc), rsd = ld4(rstd + gc);
  const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
  const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rsd.x, rsd.y, rsd.z, rsd.w};
  float4 v[P], d[P];
  uchar4 m4[P];
  bool ok[P];
  long long eo[P];
  const bool SL = a.nslab > 0;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int r = rs + p * spg;
    ok[p] = r < R;
    eo[p] = ((long long)g * R + r) * C + c0 + col * 4;
    v[p] = d[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    m4[p] = make_uchar4(0, 0, 0, 0);
    if (ok[p]) {
      v[p] = ld4(a.y + eo[p]);
      if (!SL) d[p] = ld4(f.da + eo[p]);
      if (has_mask) m4[p] = BCP_NORM_MASK4(ep, eo[p]);
    }
  }
  if (SL) {
    constexpr int KU = P >= 4 ? 2 : 4;      // slabs requested together (P = 4: two, or the loads spill)
    for (int k0 = 0; k0 < a.nslab; k0 += KU) {
      float4 t[KU][P];
#pragma unroll
      for (int kk = 0; kk < KU; ++kk)
#pragma unroll
        for (int p = 0; p < P; ++p)
          t[kk][p] = (k0 + kk < a.nslab && ok[p]) ? ld4(a.slabs + (long long)(k0 + kk) * a.slab_stride + eo[p]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int kk = 0; kk < KU; ++kk)
        if (k0 + kk < a.nslab) {
#pragma unroll
          for (int p = 0; p < P; ++p) { d[p].x += t[kk][p].x; d[p].y += t[kk][p].y; d[p].z += t[kk][p].z; d[p].w += t[kk][p].w; }
        }
    }
    if (a.sum_out) {
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (ok[p]) st4(a.sum_out + eo[p], d[p]);
    }
  }
  // dz and xhat of every element stay in registers (k_col_partial<1>'s arithmetic)
  float dzv[P][4], xhv[P][4];
  double acc8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int p = 0; p < P; ++p) {
    float cs[4] = {1.f, 1.f, 1.f, 1.f};
    if (ok[p] && ep.chan_scale) {
      const long long row = (long long)g * R + rs + p * spg;
      const float4 c4 = ld4(ep.chan_scale + (row / ep.rows_per_sample) * C + c0 + col * 4);
      cs[0] = c4.x; cs[1] = c4.y; cs[2] = c4.z; cs[3] = c4.w;
    }
    if (has_mask) {
      cs[0] *= m4[p].x ? ep.elem_scale : 0.f; cs[1] *= m4[p].y ? ep.elem_scale : 0.f;
      cs[2] *= m4[p].z ? ep.elem_scale : 0.f; cs[3] *= m4[p].w ? ep.elem_scale : 0.f;
    }
    const float vv[4] = {v[p].x, v[p].y, v[p].z, v[p].w}, dd[4] = {d[p].x, d[p].y, d[p].z, d[p].w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float z = (vv[k] - muv[k]) * scv[k] + shv[k];
      dzv[p][k] = dd[k] * cs[k] * act_grad(z, ep.act);
      xhv[p][k] = (vv[k] - muv[k]) * rsv[k];
      if (ok[p]) { acc8[k] += (double)dzv[p][k]; acc8[4 + k] += (double)dzv[p][k] * (double)xhv[p][k]; }
    }
  }
  own_reduce(acc8, GS, cw, red, fin);
  if ((int)threadIdx.x < GS * cw) {
    const int gq = threadIdx.x / cw, c = threadIdx.x % cw;
    pc1[gq * 32 + c] = (float)(fin[gq * 64 + c * 2] / (double)R);
    pc2[gq * 32 + c] = (float)(fin[gq * 64 + c * 2 + 1] / (double)R);
  }
  if (f.dgamma && (int)threadIdx.x < cw) {      // (GS = G) the groups in order: k_norm_bwd_apply's block (0, 0)
    float gb = f.accumulate ? f.dbeta[c0 + threadIdx.x] : 0.f, gg = f.accumulate ? f.dgamma[c0 + threadIdx.x] : 0.f;
    for (int gq = 0; gq < GS; ++gq) { gb += (float)fin[gq * 64 + threadIdx.x * 2]; gg += (float)fin[gq * 64 + threadIdx.x * 2 + 1]; }
    f.dbeta[c0 + threadIdx.x] = gb;
    f.dgamma[c0 + threadIdx.x] = gg;
  }
  __syncthreads();
  const float4 k1 = *reinterpret_cast<const float4*>(pc1 + gs * 32 + col * 4), k2 = *reinterpret_cast<const float4*>(pc2 + gs * 32 + col * 4);
  const float k1v[4] = {k1.x, k1.y, k1.z, k1.w}, k2v[4] = {k2.x, k2.y, k2.z, k2.w};
  float amax = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (!ok[p]) continue;
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = scv[k] * (dzv[p][k] - k1v[k] - xhv[p][k] * k2v[k]);
    st4(f.dy + eo[p], make_float4(o[0], o[1], o[2], o[3]));
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float t = fabsf(o[k]); amax = (t > amax || t != t) ? t : amax; }
  }
  if (a.amax_out) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(amax, o); amax = (t > amax || t != t) ? t : amax; }
    if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = amax;
    __syncthreads();
    const int nwg = gridDim.x * gridDim.y, me = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) {
      float m = amax_red[0];
      for (int k = 1; k < 16; ++k) { const float t = amax_red[k]; m = (t > m || t != t) ? t : m; }
      if (m != m) m = __uint_as_float(0x7fc00000u);
      a.amax_out[me * kAmaxStride] = m;
    }
    if (me == 0 && (int)threadIdx.x >= nwg && (int)threadIdx.x < kAmaxSlots) a.amax_out[threadIdx.x * kAmaxStride] = 0.f;
  }
}

// -> GS (groups per workgroup) when the one-launch form serves the shape, else 0; cw = channels per workgroup: the narrowest of 8 / 16 / 32 that
// keeps the launch at <= 32 workgroups (a workgroup moves its bytes at one CU's rate: eight 32-channel workgroups were slower than the
// two-launch chain at LA's 7x7x5 x 256 level).  ordered: running statistics / parameter gradients tie the groups together
static inline int own_gs(int G, long long R, int C, bool ordered, int& cw) {
  if (options().norm_own == 0 || (C & 31) != 0 || C > 1024) return 0;
  const int GS = ordered ? G : 1;
  if (GS > 2 || (GS == 2 && (G & 1))) return 0;
  cw = 8;
  while (cw < 32 && (long long)(C / cw) * (G / GS) > kAmaxSlots) cw <<= 1;
  if ((long long)(C / cw) * (G / GS) > kAmaxSlots) return 0;
  if (R > 512 || R > (long long)kOwnMaxP * ((kOwnThreads / (cw / 4)) / GS)) return 0;      // (a few hundred rows: beyond that the narrow chunks are bound by one cache line per lane -- pancreas 12^3 x 128 through here: -2.3 % on the step)
  return GS;
}
#define BCP_OWN_LAUNCH(KERN, R_, GS_, G_, C_, s_, a_, b_)                                                              \
  do {                                                                                                                \
    const int rpp_ = (kOwnThreads / ((a_).cw / 4)) / (GS_);                                                             \
    const int P_ = (int)(((R_) + rpp_ - 1) / rpp_);                                                                   \
    const dim3 grid_((C_) / (a_).cw, (G_) / (GS_));                                                                        \
    if (P_ <= 1) hipLaunchKernelGGL((KERN<1>), grid_, dim3(kOwnThreads), 0, s_, a_, b_);                              \
    else if (P_ == 2) hipLaunchKernelGGL((KERN<2>), grid_, dim3(kOwnThreads), 0, s_, a_, b_);                         \
    else hipLaunchKernelGGL((KERN<4>), grid_, dim3(kOwnThreads), 0, s_, a_, b_);                                      \
  } while (0)

// the fused apply passes serve: option norm_fuse_fin on, few partial rows, 16 <= C <= 256 (C a power of two: whole chunks)
static inline bool fin_fused_ok(int nb, int C) { return options().norm_fuse_fin != 0 && nb >= 1 && nb <= kFinMaxRows && C >= 16 && C <= 256; }
static inline dim3 fin_grid(long long seg_rows, int nseg, int C, bool runner) {
  const int cw = C < kFinChunk ? C : kFinChunk, slots = 256 / (cw / 4);
  long long gx = (seg_rows + (long long)slots * 4 - 1) / ((long long)slots * 4);      // four rows per thread and trip
  const long long cap = 1024 / ((long long)nseg * (C / cw)) < 1 ? 1 : 1024 / ((long long)nseg * (C / cw));
  if (gx > cap) gx = cap;
  return dim3((unsigned)(gx < 1 ? 1 : gx) + (runner ? 1u : 0u), (unsigned)nseg, (unsigned)(C / cw));
}

static inline bool skip_fin(long long rpg) { const int w = options().whatif; return (w & 1) || ((w & 2) && rpg <= 4096); }
static inline bool skip_app(long long rpg) { return ((options().whatif & 2) && rpg <= 4096) || ((options().whatif & 16) && rpg >= 100000); }
static inline bool skip_bstat(long long rpg) { return (options().whatif & 32) && rpg >= 100000; }
static inline int norm_blocks(long long rows_per_group, int C) {
  // enough blocks to fill the chip, but at least ~64 rows per thread-slot to amortise the LDS reduce
  const int slots = 256 / (C / 4);
  long long nb = rows_per_group / ((long long)slots * 16);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return (int)nb;
}

// statistics pass that also sums split-K slabs (k_col_partial<MODE, true>): it moves nslab + 1 times the tensor and serves the deep levels
// only (<= 4096 rows per group), where norm_blocks' ~64 rows per thread-slot would leave 3-15 workgroups to do it (round 4, measured:
// LA step 6.10 vs 6.02 ms with the slab-sum launch) -- one row per thread-slot and pass, at most ~256 workgroups per launch
static inline int norm_blocks_slabs(long long rows_per_group, int C, int G, bool sizing = false /* workspace size: whatever the options say later */) {
  const int slots = 256 / (C / 4);
  long long nb = (rows_per_group + slots - 1) / slots;
  long long cap = G >= 256 ? 1 : 256 / G;
  // (round 6) the fused apply pass behind this launch re-reads the partial rows in EVERY workgroup (k_norm_apply_fin): fewer, longer blocks
  if (!sizing && options().norm_fuse_fin != 0 && C <= 256 && cap > options().norm_fin_rows && options().norm_fin_rows > 0) cap = options().norm_fin_rows;
  if (nb > cap) nb = cap;
  const int plain = norm_blocks(rows_per_group, C);
  return (int)(nb < plain ? plain : nb);
}

static inline int apply_grid(long long nvec, int nseg) {
  const int vec = options().norm_apply_vec > 0 ? options().norm_apply_vec : 4, capw = options().norm_apply_cap > 0 ? options().norm_apply_cap : 2048;
  long long g = (nvec + 256 * vec - 1) / (256 * vec);      // 4 float4 per thread per trip
  const long long cap = capw / nseg < 1 ? 1 : capw / nseg;
  if (g > cap) g = cap;
  return (int)(g < 1 ? 1 : g);
}

// segments (see k_col_partial): the finer of (group, sample) when a per-sample channel scale is present
struct Segs { long long seg_rows; int spg; int nbps; };
static inline Segs make_segs(int G, long long rows_per_group, int C, const NormEpilogue& ep, int nblocks = 0) {
  Segs sg{rows_per_group, 1, nblocks > 0 ? nblocks : norm_blocks(rows_per_group, C)};
  if (ep.chan_scale && ep.rows_per_sample < rows_per_group) {
    sg.seg_rows = ep.rows_per_sample;
    sg.spg = (int)(rows_per_group / ep.rows_per_sample);
    sg.nbps = sg.nbps / sg.spg < 1 ? 1 : sg.nbps / sg.spg;
  }
  return sg;
}
static constexpr int kMaxSamplesPerGroup = 64;

// ---- finalize steps for producers that keep the statistics pass AND the apply pass in their own kernels (the fused first layer,
// csrc/conv3.hip bcp_conv3_c1_norm_fwd / _bwd): same kernels as bcp_norm_fwd / bcp_norm_bwd run between their two passes
void norm_fwd_finalize_launch(const double* partial, int nb, int G, int C, long long rows_per_group, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, float* stats, hipStream_t s, float* amax_clear_or_null) {
  float *mean = stats, *rstd = stats + (long long)G * C, *scale = stats + 2LL * G * C, *shift = stats + 3LL * G * C, *var_unb = stats + 4LL * G * C;
  if (!skip_fin(rows_per_group)) hipLaunchKernelGGL(k_norm_finalize, dim3(G * (C / 16)), dim3(kFinalizeThreads), 0, s, partial, nb, G, C, rows_per_group, gamma, beta, running_mean,
                     running_var, momentum, eps, mean, rstd, scale, shift, var_unb, amax_clear_or_null);
  if (running_mean) hipLaunchKernelGGL(k_norm_running_only, dim3(1), dim3(256), 0, s, mean, var_unb, G, C, running_mean, running_var, momentum);
}

__global__ __launch_bounds__(256) void k_norm_bwd_params(const float* __restrict__ raw, int G, int C, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, int accumulate) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {      // k_norm_bwd_apply's block (0, 0): the groups in order
    float gb = accumulate ? dbeta[c] : 0.f, gg = accumulate ? dgamma[c] : 0.f;
    for (int g = 0; g < G; ++g) { gb += raw[g * C + c]; gg += raw[(long long)G * C + g * C + c]; }
    dbeta[c] = gb;
    dgamma[c] = gg;
  }
}

// c1c2raw: float[4][G][C] -- the two means the apply pass subtracts, then the raw sums
void norm_bwd_finalize_launch(const double* partial, int nb, int G, int C, long long rows_per_group, float* dgamma, float* dbeta, int accumulate,
                              float* c1c2raw, hipStream_t s, float* amax_clear_or_null) {
  float *c1 = c1c2raw, *c2 = c1 + (long long)G * C, *raw = c2 + (long long)G * C;
  if (!skip_fin(rows_per_group)) hipLaunchKernelGGL(k_norm_bwd_finalize, dim3(G * (C / 16)), dim3(kFinalizeThreads), 0, s, partial, nb, G, C, rows_per_group, dgamma, dbeta,
                     accumulate, c1, c2, raw, amax_clear_or_null);
  if (dgamma) hipLaunchKernelGGL(k_norm_bwd_params, dim3(1), dim3(256), 0, s, raw, G, C, dgamma, dbeta, accumulate);
}

}  // namespace bcp

using namespace bcp;

extern "C" size_t bcp_norm_workspace_bytes(int G, long long rows_per_group, int C) {
  if (G < 1 || C < 16 || rows_per_group < 1) return 0;
  // per-block fp64 partials, then the two per-(g,c) backward means (c1, c2)
  // (the slab-summing statistics pass of the deep levels uses more, smaller blocks: size the partial rows for whichever is larger)
  const int nbmax = rows_per_group <= 4096 ? norm_blocks_slabs(rows_per_group, C, G, true) : norm_blocks(rows_per_group, C);
  return (size_t)G * (nbmax + kMaxSamplesPerGroup) * C * 2 * sizeof(double) + (size_t)4 * G * C * sizeof(float);
}

static int check_norm_args(const char* fn, int G, long long rows_per_group, int C) {
  BCP_REQUIRE(G >= 1 && rows_per_group >= 1, "%s: bad extents", fn);
  BCP_REQUIRE(C >= 16 && C <= 1024 && (C & (C - 1)) == 0, "%s: C=%d unsupported (need a power of two in 16..1024)", fn, C);
  return BCP_OK;
}

extern "C" int bcp_norm_fwd(const float* y, int G, long long rows_per_group, int C, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, int act,
                            const float* chan_scale, long long rows_per_sample, const uint8_t* elem_mask, float elem_scale,
                            const unsigned long long* mask_seed /* nullable: evaluate the Dropout keep bits from this device seed */, float mask_p_keep,
                            const float* residual, float* stats /* [5][G][C]: mean, rstd, scale, beta, unbiased var */, void* workspace,
                            const double* partial_in, int nb_in, float* out, long long out_ld /* row stride of out in floats; 0: C */,
                            float* amax_out /* nullable: max |out| (fp16 pre-scale of the conv that reads it) */, void* stream) {
  if (int rc = check_norm_args("bcp_norm_fwd", G, rows_per_group, C)) return rc;
  if (out_ld == 0) out_ld = C;
  BCP_REQUIRE(out_ld >= C && (out_ld & 3) == 0, "bcp_norm_fwd: out_ld=%lld (need a multiple of 4 >= C)", out_ld);
  BCP_REQUIRE(y && stats && workspace, "bcp_norm_fwd: null pointer");
  BCP_REQUIRE(aligned16(y) && (!out || aligned16(out)) && aligned16(stats), "bcp_norm_fwd: alignment");
  BCP_REQUIRE(out || !residual, "bcp_norm_fwd: statistics-only mode (out = NULL) takes no residual");
  hipStream_t s = (hipStream_t)stream;
  NormEpilogue ep{chan_scale, elem_mask, elem_scale, rows_per_sample > 0 ? rows_per_sample : rows_per_group, act, mask_seed, mask_p_keep};
  BCP_REQUIRE(!chan_scale || (rows_per_group % ep.rows_per_sample == 0 && rows_per_group / ep.rows_per_sample <= kMaxSamplesPerGroup),
              "bcp_norm_fwd: a group must hold 1..%d whole samples", kMaxSamplesPerGroup);
  const Segs sg = make_segs(G, rows_per_group, C, ep);
  const int nseg = G * sg.spg, nb = sg.nbps * sg.spg;
  double* partial = reinterpret_cast<double*>(workspace);
  float *mean = stats, *rstd = stats + (long long)G * C, *scale = stats + 2LL * G * C, *shift = stats + 3LL * G * C;
  float* var_unb = stats + 4LL * G * C;
  int own_cw = 0;
  if (const int GS = (out && !partial_in) ? own_gs(G, rows_per_group, C, running_mean != nullptr, own_cw) : 0) {      // (round 6) the smallest levels: ONE launch
    const OwnCommon oc{y, nullptr, 0, 0, nullptr, nullptr, G, C, (int)rows_per_group, GS, own_cw, ep, amax_out};
    const OwnFwd of{gamma, beta, running_mean, running_var, momentum, eps, residual, stats, out, out_ld};
    BCP_OWN_LAUNCH(k_norm_own_fwd, rows_per_group, GS, G, C, s, oc, of);
    BCP_CHECK_LAUNCH("bcp_norm_fwd");
    return BCP_OK;
  }
  const bool fused = out && !partial_in && fin_fused_ok(nb, C);      // (round 6) the apply pass finalises the statistics itself: no finalize launch
  if (partial_in) {   // statistics partials were produced by the conv epilogue (bcp_conv3_fwd_stats)
    if (!skip_fin(rows_per_group)) hipLaunchKernelGGL(k_norm_finalize, dim3(G * (C / 16)), dim3(kFinalizeThreads), 0, s, partial_in, nb_in, G, C, rows_per_group, gamma, beta,
                       running_mean, running_var, momentum, eps, mean, rstd, scale, shift, var_unb, out ? amax_out : (float*)nullptr);
  } else {
    hipLaunchKernelGGL((k_col_partial<0>), dim3(sg.nbps, nseg), dim3(256), 0, s, y, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, ep, sg.seg_rows, sg.spg, C, partial, SlabSrc{},
                       fused ? amax_out : (float*)nullptr);
    if (!fused && !skip_fin(rows_per_group)) hipLaunchKernelGGL(k_norm_finalize, dim3(G * (C / 16)), dim3(kFinalizeThreads), 0, s, partial, nb, G, C, rows_per_group, gamma, beta,
                       running_mean, running_var, momentum, eps, mean, rstd, scale, shift, var_unb, out ? amax_out : (float*)nullptr);
  }
  if (fused)
    hipLaunchKernelGGL(k_norm_apply_fin, fin_grid(sg.seg_rows, nseg, C, running_mean != nullptr), dim3(256), 0, s, y, residual, ep, sg.seg_rows, sg.spg, G, C, rows_per_group,
                       FinFwd{partial, nb, gamma, beta, running_mean, running_var, momentum, eps, stats}, out, amax_out, out_ld);
  else if (out && skip_app(rows_per_group)) {}
  else if (out)
    hipLaunchKernelGGL(k_norm_apply, dim3(apply_grid(sg.seg_rows * (C / 4), nseg), nseg), dim3(256), 0, s, y, scale, shift, mean, residual,
                       ep, sg.seg_rows, sg.spg, G, C, out, var_unb, running_mean, running_var, momentum, amax_out, (long long)(out_ld / 4));
  else if (running_mean)   // statistics only: the consumer applies the normalisation itself (bcp_pw16_fwd_norm); the apply pass also carries the running-statistics update
    hipLaunchKernelGGL(k_norm_running_only, dim3(1), dim3(256), 0, s, mean, var_unb, G, C, running_mean, running_var, momentum);
  BCP_CHECK_LAUNCH("bcp_norm_fwd");
  return BCP_OK;
}

extern "C" int bcp_norm_bwd(const float* y, const float* da, int G, long long rows_per_group, int C, const float* stats,
                            int act, const float* chan_scale, long long rows_per_sample, const uint8_t* elem_mask,
                            float elem_scale, const unsigned long long* mask_seed, float mask_p_keep, float* dgamma, float* dbeta, int accumulate, void* workspace,
                            const double* partial_in, int nb_in, float* dy, float* amax_out /* nullable: max |dy| */, void* stream) {
  if (int rc = check_norm_args("bcp_norm_bwd", G, rows_per_group, C)) return rc;
  BCP_REQUIRE(y && da && stats && workspace && dy, "bcp_norm_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  NormEpilogue ep{chan_scale, elem_mask, elem_scale, rows_per_sample > 0 ? rows_per_sample : rows_per_group, act, mask_seed, mask_p_keep};
  BCP_REQUIRE(!chan_scale || (rows_per_group % ep.rows_per_sample == 0 && rows_per_group / ep.rows_per_sample <= kMaxSamplesPerGroup),
              "bcp_norm_bwd: a group must hold 1..%d whole samples", kMaxSamplesPerGroup);
  const Segs sg = make_segs(G, rows_per_group, C, ep);
  const int nseg = G * sg.spg, nb = sg.nbps * sg.spg;
  double* partial = reinterpret_cast<double*>(workspace);
  const float *mean = stats, *rstd = stats + (long long)G * C, *scale = stats + 2LL * G * C, *shift = stats + 3LL * G * C;
  // c1/c2 live behind the partials in the workspace
  float* c1 = reinterpret_cast<float*>(partial + (size_t)G * (norm_blocks(rows_per_group, C) + kMaxSamplesPerGroup) * C * 2);
  float* c2 = c1 + (long long)G * C;
  float* raw = c2 + (long long)G * C;
  int own_cw = 0;
  if (const int GS = !partial_in ? own_gs(G, rows_per_group, C, dgamma != nullptr, own_cw) : 0) {
    const OwnCommon oc{y, nullptr, 0, 0, nullptr, nullptr, G, C, (int)rows_per_group, GS, own_cw, ep, amax_out};
    const OwnBwd ob{stats, da, dgamma, dbeta, accumulate, dy};
    BCP_OWN_LAUNCH(k_norm_own_bwd, rows_per_group, GS, G, C, s, oc, ob);
    BCP_CHECK_LAUNCH("bcp_norm_bwd");
    return BCP_OK;
  }
  const bool fused = !partial_in && fin_fused_ok(nb, C);
  if (partial_in) {   // (sum dz, sum dz * xhat) partials handed in by the caller (no kernel of the library produces them any more)
    BCP_REQUIRE(!chan_scale && !elem_mask && !mask_seed && nb_in > 0, "bcp_norm_bwd: fused statistics do not cover dropout epilogues");
    if (!skip_fin(rows_per_group)) hipLaunchKernelGGL(k_norm_bwd_finalize, dim3(G * (C / 16)), dim3(kFinalizeThreads), 0, s, partial_in, nb_in, G, C, rows_per_group, dgamma,
                       dbeta, accumulate, c1, c2, raw, amax_out);
  } else {
    if (!skip_bstat(rows_per_group)) hipLaunchKernelGGL((k_col_partial<1>), dim3(sg.nbps, nseg), dim3(256), 0, s, y, da, scale, shift, mean, rstd, ep, sg.seg_rows, sg.spg,
                       C, partial, SlabSrc{}, fused ? amax_out : (float*)nullptr);
    if (!fused && !skip_fin(rows_per_group)) hipLaunchKernelGGL(k_norm_bwd_finalize, dim3(G * (C / 16)), dim3(kFinalizeThreads), 0, s, partial, nb, G, C, rows_per_group, dgamma,
                       dbeta, accumulate, c1, c2, raw, amax_out);
  }
  if (fused)
    hipLaunchKernelGGL(k_norm_bwd_apply_fin, fin_grid(sg.seg_rows, nseg, C, dgamma != nullptr), dim3(256), 0, s, y, da, stats, ep, sg.seg_rows, sg.spg, G, C, rows_per_group,
                       FinBwd{partial, nb, dgamma, dbeta, accumulate}, dy, amax_out);
  else if (!skip_app(rows_per_group)) hipLaunchKernelGGL(k_norm_bwd_apply, dim3(apply_grid(sg.seg_rows * (C / 4), nseg), nseg), dim3(256), 0, s, y, da, scale, shift, mean,
                     rstd, c1, c2, ep, sg.seg_rows, sg.spg, G, C, dy, raw, dgamma, dbeta, accumulate, amax_out);
  BCP_CHECK_LAUNCH("bcp_norm_bwd");
  return BCP_OK;
}

// ---- deep levels: the producing conv left its raw split-K slabs (bcp_conv3_fwd_raw); the statistics pass sums them on its way in
// (k_col_partial<MODE, SL = true>) and writes the sum once for the apply pass.  Bit-identical to k_b6_sum_slabs + bcp_norm_fwd / _bwd.
extern "C" int bcp_norm_slabs_ok(int G, long long rows_per_group, int C) {
  return (options().norm_slabs != 0 && G >= 1 && rows_per_group >= 1 && rows_per_group <= 4096 && C >= 16 && C <= 1024 && (C & (C - 1)) == 0) ? 1 : 0;
}

extern "C" int bcp_norm_fwd_slabs(const float* slabs, int nslab, long long slab_stride, const float* bias, float* ysum, int G,
                                  long long rows_per_group, int C, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float momentum, float eps, int act, const float* chan_scale, long long rows_per_sample,
                                  const uint8_t* elem_mask, float elem_scale, const unsigned long long* mask_seed, float mask_p_keep,
                                  const float* residual, float* stats, void* workspace, float* out,
                                  float* amax_out, void* stream) {
  if (int rc = check_norm_args("bcp_norm_fwd_slabs", G, rows_per_group, C)) return rc;
  BCP_REQUIRE(rows_per_group <= 4096, "bcp_norm_fwd_slabs: rows_per_group=%lld > 4096 (check bcp_norm_slabs_ok)", rows_per_group);
  BCP_REQUIRE(slabs && ysum && stats && workspace && nslab >= 1 && nslab <= 64, "bcp_norm_fwd_slabs: null pointer / bad slab count");
  BCP_REQUIRE(aligned16(slabs) && aligned16(ysum) && (!out || aligned16(out)) && aligned16(stats) && (slab_stride & 3) == 0 && (!bias || aligned16(bias)),
              "bcp_norm_fwd_slabs: alignment");
  BCP_REQUIRE(out || !residual, "bcp_norm_fwd_slabs: statistics-only mode (out = NULL) takes no residual");
  hipStream_t s = (hipStream_t)stream;
  NormEpilogue ep{chan_scale, elem_mask, elem_scale, rows_per_sample > 0 ? rows_per_sample : rows_per_group, act, mask_seed, mask_p_keep};
  BCP_REQUIRE(!chan_scale || (rows_per_group % ep.rows_per_sample == 0 && rows_per_group / ep.rows_per_sample <= kMaxSamplesPerGroup),
              "bcp_norm_fwd_slabs: a group must hold 1..%d whole samples", kMaxSamplesPerGroup);
  const Segs sg = make_segs(G, rows_per_group, C, ep, norm_blocks_slabs(rows_per_group, C, G));
  const int nseg = G * sg.spg, nb = sg.nbps * sg.spg;
  double* partial = reinterpret_cast<double*>(workspace);
  float *mean = stats, *rstd = stats + (long long)G * C, *scale = stats + 2LL * G * C, *shift = stats + 3LL * G * C;
  float* var_unb = stats + 4LL * G * C;
  const SlabSrc sl{slabs, nslab, slab_stride, bias, ysum};
  int own_cw = 0;
  if (const int GS = out ? own_gs(G, rows_per_group, C, running_mean != nullptr, own_cw) : 0) {
    const OwnCommon oc{nullptr, slabs, nslab, slab_stride, bias, ysum, G, C, (int)rows_per_group, GS, own_cw, ep, amax_out};
    const OwnFwd of{gamma, beta, running_mean, running_var, momentum, eps, residual, stats, out, (long long)C};
    BCP_OWN_LAUNCH(k_norm_own_fwd, rows_per_group, GS, G, C, s, oc, of);
    BCP_CHECK_LAUNCH("bcp_norm_fwd_slabs");
    return BCP_OK;
  }
  const bool fused = out && fin_fused_ok(nb, C);
  hipLaunchKernelGGL((k_col_partial<0, true>), dim3(sg.nbps, nseg), dim3(256), 0, s, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, ep, sg.seg_rows, sg.spg, C, partial, sl,
                     fused ? amax_out : (float*)nullptr);
  if (!fused && !skip_fin(rows_per_group)) hipLaunchKernelGGL(k_norm_finalize, dim3(G * (C / 16)), dim3(kFinalizeThreads), 0, s, partial, nb, G, C, rows_per_group, gamma, beta,
                     running_mean, running_var, momentum, eps, mean, rstd, scale, shift, var_unb, out ? amax_out : (float*)nullptr);
  if (fused)
    hipLaunchKernelGGL(k_norm_apply_fin, fin_grid(sg.seg_rows, nseg, C, running_mean != nullptr), dim3(256), 0, s, ysum, residual, ep, sg.seg_rows, sg.spg, G, C, rows_per_group,
                       FinFwd{partial, nb, gamma, beta, running_mean, running_var, momentum, eps, stats}, out, amax_out, (long long)C);
  else if (out && skip_app(rows_per_group)) {}
  else if (out)
    hipLaunchKernelGGL(k_norm_apply, dim3(apply_grid(sg.seg_rows * (C / 4), nseg), nseg), dim3(256), 0, s, ysum, scale, shift, mean, residual,
                       ep, sg.seg_rows, sg.spg, G, C, out, var_unb, running_mean, running_var, momentum, amax_out, (long long)(C / 4));
  else if (running_mean)
    hipLaunchKernelGGL(k_norm_running_only, dim3(1), dim3(256), 0, s, mean, var_unb, G, C, running_mean, running_var, momentum);
  BCP_CHECK_LAUNCH("bcp_norm_fwd_slabs");
  return BCP_OK;
}

extern "C" int bcp_norm_bwd_slabs(const float* y, const float* da_slabs, int nslab, long long slab_stride, float* da_sum, int G,
                                  long long rows_per_group, int C, const float* stats, int act, const float* chan_scale,
                                  long long rows_per_sample, const uint8_t* elem_mask, float elem_scale, const unsigned long long* mask_seed,
                                  float mask_p_keep, float* dgamma, float* dbeta,
                                  int accumulate, void* workspace, float* dy, float* amax_out, void* stream) {
  if (int rc = check_norm_args("bcp_norm_bwd_slabs", G, rows_per_group, C)) return rc;
  BCP_REQUIRE(rows_per_group <= 4096, "bcp_norm_bwd_slabs: rows_per_group=%lld > 4096 (check bcp_norm_slabs_ok)", rows_per_group);
  BCP_REQUIRE(y && da_slabs && da_sum && stats && workspace && dy && nslab >= 1 && nslab <= 64, "bcp_norm_bwd_slabs: null pointer / bad slab count");
  BCP_REQUIRE(aligned16(y) && aligned16(da_slabs) && aligned16(dy) && aligned16(da_sum) && (slab_stride & 3) == 0, "bcp_norm_bwd_slabs: alignment");
  BCP_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bcp_norm_bwd_slabs: dgamma and dbeta come together");
  hipStream_t s = (hipStream_t)stream;
  NormEpilogue ep{chan_scale, elem_mask, elem_scale, rows_per_sample > 0 ? rows_per_sample : rows_per_group, act, mask_seed, mask_p_keep};
  BCP_REQUIRE(!chan_scale || (rows_per_group % ep.rows_per_sample == 0 && rows_per_group / ep.rows_per_sample <= kMaxSamplesPerGroup),
              "bcp_norm_bwd_slabs: a group must hold 1..%d whole samples", kMaxSamplesPerGroup);
  const Segs sg = make_segs(G, rows_per_group, C, ep, norm_blocks_slabs(rows_per_group, C, G));
  const int nseg = G * sg.spg, nb = sg.nbps * sg.spg;
  double* partial = reinterpret_cast<double*>(workspace);
  const float *mean = stats, *rstd = stats + (long long)G * C, *scale = stats + 2LL * G * C, *shift = stats + 3LL * G * C;
  float* c1 = reinterpret_cast<float*>(partial + (size_t)G * (norm_blocks_slabs(rows_per_group, C, G) + kMaxSamplesPerGroup) * C * 2);
  float* c2 = c1 + (long long)G * C;
  float* raw = c2 + (long long)G * C;
  const SlabSrc sl{da_slabs, nslab, slab_stride, nullptr, da_sum};
  int own_cw = 0;
  if (const int GS = own_gs(G, rows_per_group, C, dgamma != nullptr, own_cw)) {
    const OwnCommon oc{y, da_slabs, nslab, slab_stride, nullptr, da_sum, G, C, (int)rows_per_group, GS, own_cw, ep, amax_out};
    const OwnBwd ob{stats, nullptr, dgamma, dbeta, accumulate, dy};
    BCP_OWN_LAUNCH(k_norm_own_bwd, rows_per_group, GS, G, C, s, oc, ob);
    BCP_CHECK_LAUNCH("bcp_norm_bwd_slabs");
    return BCP_OK;
  }
  const bool fused = fin_fused_ok(nb, C);
  hipLaunchKernelGGL((k_col_partial<1, true>), dim3(sg.nbps, nseg), dim3(256), 0, s, y, (const float*)nullptr, scale, shift, mean, rstd, ep,
                     sg.seg_rows, sg.spg, C, partial, sl, fused ? amax_out : (float*)nullptr);
  if (!fused && !skip_fin(rows_per_group)) hipLaunchKernelGGL(k_norm_bwd_finalize, dim3(G * (C / 16)), dim3(kFinalizeThreads), 0, s, partial, nb, G, C, rows_per_group, dgamma,
                     dbeta, accumulate, c1, c2, raw, amax_out);
  if (fused)
    hipLaunchKernelGGL(k_norm_bwd_apply_fin, fin_grid(sg.seg_rows, nseg, C, dgamma != nullptr), dim3(256), 0, s, y, da_sum, stats, ep, sg.seg_rows, sg.spg, G, C, rows_per_group,
                       FinBwd{partial, nb, dgamma, dbeta, accumulate}, dy, amax_out);
  else if (!skip_app(rows_per_group)) hipLaunchKernelGGL(k_norm_bwd_apply, dim3(apply_grid(sg.seg_rows * (C / 4), nseg), nseg), dim3(256), 0, s, y, da_sum, scale, shift, mean,
                     rstd, c1, c2, ep, sg.seg_rows, sg.spg, G, C, dy, raw, dgamma, dbeta, accumulate, amax_out);
  BCP_CHECK_LAUNCH("bcp_norm_bwd_slabs");
  return BCP_OK;
}
