// tools/attic/norm_small.hip -- NOT part of libbcp_hip.so since round 4 (kept for the record).
// The one-launch norm for the deep levels (rows_per_group <= 4096): statistics + finalize + apply (+ the split-K slab sum) in ONE
// launch per layer instead of 3-5.  Measured slower in round 3: a workgroup that owns four channels of every row touches one cache
// line per lane (TA-bound: 31 us for 8 MB) while the back-to-back chain it replaces costs 12 us; LA step 6.67 vs 6.54 ms.
// Superseded by bcp_norm_fwd_slabs / bcp_norm_bwd_slabs (csrc/norm.hip): the slab sum folded into the ROW-MAJOR statistics pass.
// The text below is the kernels and entry points as they stood in csrc/norm.hip at the end of round 3 (needs that file's helpers).

// ------------------------------------------------------------------ small groups: the whole norm in ONE launch
// Deep levels (LA 14x14x10 / 7x7x5, pancreas 12^3 / 6^3, the U-Net's 16x16 level): a group has a few thousand rows, the tensor a
// few MB, and the three-kernel chain above (statistics -> finalize -> apply, + the split-K slab sum in front of it) is ~5 us of
// launch / dependency latency per kernel with nothing to stream.  Here a workgroup OWNS four channels of a group for all its rows:
// every thread keeps its <= RPT rows (one float4 each) in registers, so the slab sum, the statistics, their finalisation and the
// apply pass need no second read and no cross-workgroup step -- one launch instead of four (forward) / three (backward).
// Same arithmetic per element as the streaming kernels; the statistics are summed in fp64 in a different (fixed) order.
//   grid.x = C / 4 (float4 column), grid.y = group chunks (one chunk holding ALL groups when running statistics or parameter
//   gradients make the groups order-dependent); GU groups are in flight per trip (independent loads up front).
static constexpr int kSmallMaxRows = 4096;       // rows per group: 256 threads x RPT <= 16

template <int NV>
__device__ __forceinline__ void small_block_reduce(double (&v)[NV], double* red /* [4][NV] */, double* fin /* [NV] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();                                       // (the previous trip's readers of red / fin are done)
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
  }
  __syncthreads();
  if ((int)threadIdx.x < NV) fin[threadIdx.x] = (red[threadIdx.x] + red[NV + threadIdx.x]) + (red[2 * NV + threadIdx.x] + red[3 * NV + threadIdx.x]);
  __syncthreads();
}

template <int RPT, int GU>
__global__ __launch_bounds__(256) void k_norm_small_fwd(const float* __restrict__ slabs, int nslab, long long slab_stride,
                                                        const float* __restrict__ bias, float* __restrict__ ysum, int G, int R, int C,
                                                        int groups_per_block, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                                                        float eps, NormEpilogue ep, const float* __restrict__ residual,
                                                        float* __restrict__ stats, float* __restrict__ out) {
  __shared__ double red[4 * GU * 8];
  __shared__ double fin[GU * 8];
  // channel quad of this workgroup: the gridDim.x / 8 workgroups an XCD receives (linear id % 8) take NEIGHBOURING quads, so the 128-byte
  // lines of a row (8 quads) are fetched into one or two L2s instead of all eight (8 MB of slabs crossed the fabric as 44 MB)
  const int nq = gridDim.x;
  const int c0 = ((nq & 7) == 0 ? (int)(blockIdx.x & 7) * (nq >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x) * 4;
  const int g_begin = blockIdx.y * groups_per_block, g_end = g_begin + groups_per_block < G ? g_begin + groups_per_block : G;
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) b4 = ld4(bias + c0);
  float *mean = stats, *rstd = stats + (long long)G * C, *scale = stats + 2LL * G * C, *shift = stats + 3LL * G * C;
  float* var_unb = stats + 4LL * G * C;
  double rm = 0.0, rv = 0.0;                             // running statistics of channel c0 + tid (threads 0..3)
  if (running_mean && threadIdx.x < 4) { rm = (double)running_mean[c0 + threadIdx.x]; rv = (double)running_var[c0 + threadIdx.x]; }
  for (int g0 = g_begin; g0 < g_end; g0 += GU) {
    float4 v[GU][RPT];
    // ---- slab sum (the order of k_b6_sum_slabs: bias first, then the slabs front to back).  The SLAB loop is the outer one and all
    // GU * RPT loads of a slab are issued before the first add: a runtime-count loop around each element's loads compiled to one
    // dependent round trip per (element, slab) -- 64 in a row, 40 us for 8 MB (measured, round 3); now nslab round trips.
    long long eoff[GU][RPT];
#pragma unroll
    for (int u = 0; u < GU; ++u)
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int r = threadIdx.x + k * 256;
        const bool ok = g0 + u < g_end && r < R;
        eoff[u][k] = ok ? ((long long)(g0 + u) * R + r) * C + c0 : -1;
        v[u][k] = b4;
      }
    for (int s = 0; s < nslab; ++s) {
      const float* sl = slabs + (long long)s * slab_stride;
      float4 p[GU][RPT];
#pragma unroll
      for (int u = 0; u < GU; ++u)
#pragma unroll
        for (int k = 0; k < RPT; ++k) p[u][k] = ld4(sl + (eoff[u][k] >= 0 ? eoff[u][k] : (long long)c0));      // (unconditional loads: row 0 stands in)
#pragma unroll
      for (int u = 0; u < GU; ++u)
#pragma unroll
        for (int k = 0; k < RPT; ++k) { v[u][k].x += p[u][k].x; v[u][k].y += p[u][k].y; v[u][k].z += p[u][k].z; v[u][k].w += p[u][k].w; }
    }
    if (ysum) {
#pragma unroll
      for (int u = 0; u < GU; ++u)
#pragma unroll
        for (int k = 0; k < RPT; ++k)
          if (eoff[u][k] >= 0) st4(ysum + eoff[u][k], v[u][k]);
    }
    double acc[GU * 8];
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < RPT; ++k)
        if (g0 + u < g_end && (int)threadIdx.x + k * 256 < R) {
          const float vv[4] = {v[u][k].x, v[u][k].y, v[u][k].z, v[u][k].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) { s1[j] += (double)vv[j]; s2[j] += (double)vv[j] * (double)vv[j]; }
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[u * 8 + j] = s1[j]; acc[u * 8 + 4 + j] = s2[j]; }
    }
    small_block_reduce<GU * 8>(acc, red, fin);
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      if (g0 + u >= g_end) break;                        // uniform
      const int g = g0 + u;
      float mu[4], sc[4], sh[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {                      // k_norm_finalize's arithmetic, redundantly in every thread
        const double n = (double)R, m = fin[u * 8 + j] / n;
        double var = fin[u * 8 + 4 + j] / n - m * m;
        if (var < 0.0) var = 0.0;
        const double rr = 1.0 / sqrt(var + (double)eps);
        const double ga = gamma ? (double)gamma[c0 + j] : 1.0, be = beta ? (double)beta[c0 + j] : 0.0;
        mu[j] = (float)m; sc[j] = (float)(ga * rr); sh[j] = (float)be;
        if ((int)threadIdx.x == j) {
          const int idx = g * C + c0 + j;
          const float vu = (float)(n > 1.0 ? var * n / (n - 1.0) : var);
          mean[idx] = mu[j]; rstd[idx] = (float)rr; scale[idx] = sc[j]; shift[idx] = sh[j]; var_unb[idx] = vu;
          if (running_mean) {                            // update_running's arithmetic, group after group
            rm = (1.0 - (double)momentum) * rm + (double)momentum * (double)mu[j];
            rv = (1.0 - (double)momentum) * rv + (double)momentum * (double)vu;
            rm = (double)(float)rm; rv = (double)(float)rv;
          }
        }
      }
      if (!out) continue;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int r = threadIdx.x + k * 256;
        if (r < R) {
          const long long row = (long long)g * R + r, e = row * C + c0;
          const float4 y4 = v[u][k];
          float o[4] = {act_fwd((y4.x - mu[0]) * sc[0] + sh[0], ep.act), act_fwd((y4.y - mu[1]) * sc[1] + sh[1], ep.act),
                        act_fwd((y4.z - mu[2]) * sc[2] + sh[2], ep.act), act_fwd((y4.w - mu[3]) * sc[3] + sh[3], ep.act)};
          if (ep.chan_scale) {
            const float4 cs = ld4(ep.chan_scale + (row / ep.rows_per_sample) * C + c0);
            o[0] *= cs.x; o[1] *= cs.y; o[2] *= cs.z; o[3] *= cs.w;
          }
          if (ep.elem_mask) {
            const uchar4 m4 = *reinterpret_cast<const uchar4*>(ep.elem_mask + e);
            o[0] *= m4.x ? ep.elem_scale : 0.f; o[1] *= m4.y ? ep.elem_scale : 0.f;
            o[2] *= m4.z ? ep.elem_scale : 0.f; o[3] *= m4.w ? ep.elem_scale : 0.f;
          }
          if (residual) { const float4 r4 = ld4(residual + e); o[0] += r4.x; o[1] += r4.y; o[2] += r4.z; o[3] += r4.w; }
          st4(out + e, make_float4(o[0], o[1], o[2], o[3]));
        }
      }
    }
  }
  if (running_mean && threadIdx.x < 4) { running_mean[c0 + threadIdx.x] = (float)rm; running_var[c0 + threadIdx.x] = (float)rv; }
}

// backward twin: da = sum of the dgrad's split-K slabs (no bias), dz = da * epilogue' * act'(z), the two sums, dy
template <int RPT, int GU>
__global__ __launch_bounds__(256) void k_norm_small_bwd(const float* __restrict__ y, const float* __restrict__ da_slabs, int nslab,
                                                        long long slab_stride, float* __restrict__ da_sum, int G, int R, int C,
                                                        int groups_per_block, const float* __restrict__ stats, NormEpilogue ep,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                        float* __restrict__ dy) {
  __shared__ double red[4 * GU * 8];
  __shared__ double fin[GU * 8];
  // channel quad of this workgroup: the gridDim.x / 8 workgroups an XCD receives (linear id % 8) take NEIGHBOURING quads, so the 128-byte
  // lines of a row (8 quads) are fetched into one or two L2s instead of all eight (8 MB of slabs crossed the fabric as 44 MB)
  const int nq = gridDim.x;
  const int c0 = ((nq & 7) == 0 ? (int)(blockIdx.x & 7) * (nq >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x) * 4;
  const int g_begin = blockIdx.y * groups_per_block, g_end = g_begin + groups_per_block < G ? g_begin + groups_per_block : G;
  const float *mean = stats, *rstd = stats + (long long)G * C, *scale = stats + 2LL * G * C, *shift = stats + 3LL * G * C;
  float gb = 0.f, gg = 0.f;                               // parameter gradients of channel c0 + tid (threads 0..3), groups in order
  if (dgamma && threadIdx.x < 4 && accumulate) { gb = dbeta[c0 + threadIdx.x]; gg = dgamma[c0 + threadIdx.x]; }
  for (int g0 = g_begin; g0 < g_end; g0 += GU) {
    float4 v[GU][RPT], d[GU][RPT];
    long long eoff[GU][RPT];
#pragma unroll
    for (int u = 0; u < GU; ++u)
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int r = threadIdx.x + k * 256;
        const bool ok = g0 + u < g_end && r < R;
        eoff[u][k] = ok ? ((long long)(g0 + u) * R + r) * C + c0 : -1;
        v[u][k] = ld4(y + (ok ? eoff[u][k] : (long long)c0));
        d[u][k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    for (int s = 0; s < nslab; ++s) {                     // slab loop outermost, loads of a slab issued together: see k_norm_small_fwd
      const float* sl = da_slabs + (long long)s * slab_stride;
      float4 p[GU][RPT];
#pragma unroll
      for (int u = 0; u < GU; ++u)
#pragma unroll
        for (int k = 0; k < RPT; ++k) p[u][k] = ld4(sl + (eoff[u][k] >= 0 ? eoff[u][k] : (long long)c0));
#pragma unroll
      for (int u = 0; u < GU; ++u)
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          if (nslab == 1) d[u][k] = p[u][k];            // (bit-identical to the plain tensor: no 0 + x)
          else { d[u][k].x += p[u][k].x; d[u][k].y += p[u][k].y; d[u][k].z += p[u][k].z; d[u][k].w += p[u][k].w; }
        }
    }
    if (da_sum) {
#pragma unroll
      for (int u = 0; u < GU; ++u)
#pragma unroll
        for (int k = 0; k < RPT; ++k)
          if (eoff[u][k] >= 0) st4(da_sum + eoff[u][k], d[u][k]);
    }
    // dz replaces da in the registers; xhat is recomputed in the apply loop
    float mu[GU][4], sc[GU][4], sh[GU][4], rs[GU][4];
    double acc[GU * 8];
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int g = g0 + u < g_end ? g0 + u : g_end - 1;
      const float4 m4 = ld4(mean + (long long)g * C + c0), r4 = ld4(rstd + (long long)g * C + c0);
      const float4 s4 = ld4(scale + (long long)g * C + c0), h4 = ld4(shift + (long long)g * C + c0);
      mu[u][0] = m4.x; mu[u][1] = m4.y; mu[u][2] = m4.z; mu[u][3] = m4.w;
      rs[u][0] = r4.x; rs[u][1] = r4.y; rs[u][2] = r4.z; rs[u][3] = r4.w;
      sc[u][0] = s4.x; sc[u][1] = s4.y; sc[u][2] = s4.z; sc[u][3] = s4.w;
      sh[u][0] = h4.x; sh[u][1] = h4.y; sh[u][2] = h4.z; sh[u][3] = h4.w;
      double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int r = threadIdx.x + k * 256;
        if (g0 + u < g_end && r < R) {
          const long long row = (long long)g * R + r;
          float cs[4] = {1.f, 1.f, 1.f, 1.f};
          if (ep.chan_scale) { const float4 c4 = ld4(ep.chan_scale + (row / ep.rows_per_sample) * C + c0); cs[0] = c4.x; cs[1] = c4.y; cs[2] = c4.z; cs[3] = c4.w; }
          if (ep.elem_mask) {
            const uchar4 k4 = *reinterpret_cast<const uchar4*>(ep.elem_mask + row * C + c0);
            cs[0] *= k4.x ? ep.elem_scale : 0.f; cs[1] *= k4.y ? ep.elem_scale : 0.f;
            cs[2] *= k4.z ? ep.elem_scale : 0.f; cs[3] *= k4.w ? ep.elem_scale : 0.f;
          }
          const float vv[4] = {v[u][k].x, v[u][k].y, v[u][k].z, v[u][k].w}, dd[4] = {d[u][k].x, d[u][k].y, d[u][k].z, d[u][k].w};
          float dz[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float z = (vv[j] - mu[u][j]) * sc[u][j] + sh[u][j];
            dz[j] = dd[j] * cs[j] * act_grad(z, ep.act);
            const float xh = (vv[j] - mu[u][j]) * rs[u][j];
            s1[j] += (double)dz[j];
            s2[j] += (double)dz[j] * (double)xh;
          }
          d[u][k] = make_float4(dz[0], dz[1], dz[2], dz[3]);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[u * 8 + j] = s1[j]; acc[u * 8 + 4 + j] = s2[j]; }
    }
    small_block_reduce<GU * 8>(acc, red, fin);
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      if (g0 + u >= g_end) break;                        // uniform
      const int g = g0 + u;
      float k1[4], k2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {                      // k_norm_bwd_finalize's arithmetic
        k1[j] = (float)(fin[u * 8 + j] / (double)R);
        k2[j] = (float)(fin[u * 8 + 4 + j] / (double)R);
        if (dgamma && (int)threadIdx.x == j) { gb += (float)fin[u * 8 + j]; gg += (float)fin[u * 8 + 4 + j]; }
      }
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int r = threadIdx.x + k * 256;
        if (r < R) {
          const long long e = ((long long)g * R + r) * C + c0;
          const float vv[4] = {v[u][k].x, v[u][k].y, v[u][k].z, v[u][k].w}, dz[4] = {d[u][k].x, d[u][k].y, d[u][k].z, d[u][k].w};
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xh = (vv[j] - mu[u][j]) * rs[u][j];
            o[j] = sc[u][j] * (dz[j] - k1[j] - xh * k2[j]);
          }
          st4(dy + e, make_float4(o[0], o[1], o[2], o[3]));
        }
      }
    }
  }
  if (dgamma && threadIdx.x < 4) { dbeta[c0 + threadIdx.x] = gb; dgamma[c0 + threadIdx.x] = gg; }
}

static inline int small_rpt(long long rows_per_group) {
  int rpt = 1;
  while ((long long)rpt * 256 < rows_per_group) rpt <<= 1;
  return rpt;
}


// ---- entry points
// ---- one-launch variants for small groups (k_norm_small_*): rows_per_group <= 4096, any C % 4 == 0
extern "C" int bcp_norm_small_ok(int G, long long rows_per_group, int C) {
  return (options().norm_small != 0 && G >= 1 && rows_per_group >= 1 && rows_per_group <= kSmallMaxRows && C >= 4 && (C & 3) == 0) ? 1 : 0;
}

#define BCP_SMALL_DISPATCH(KERNEL, ...)                                                                          \
  do {                                                                                                           \
    const int rpt = small_rpt(rows_per_group);                                                                   \
    const bool two = rpt <= 8 && gpb >= 2;                                                                       \
    const dim3 grid(C / 4, (G + gpb - 1) / gpb), block(256);                                                     \
    switch (rpt) {                                                                                               \
      case 1: if (two) hipLaunchKernelGGL((KERNEL<1, 2>), grid, block, 0, s, __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<1, 1>), grid, block, 0, s, __VA_ARGS__); break; \
      case 2: if (two) hipLaunchKernelGGL((KERNEL<2, 2>), grid, block, 0, s, __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<2, 1>), grid, block, 0, s, __VA_ARGS__); break; \
      case 4: if (two) hipLaunchKernelGGL((KERNEL<4, 2>), grid, block, 0, s, __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<4, 1>), grid, block, 0, s, __VA_ARGS__); break; \
      case 8: if (two) hipLaunchKernelGGL((KERNEL<8, 2>), grid, block, 0, s, __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<8, 1>), grid, block, 0, s, __VA_ARGS__); break; \
      default: hipLaunchKernelGGL((KERNEL<16, 1>), grid, block, 0, s, __VA_ARGS__); break;                       \
    }                                                                                                            \
  } while (0)

extern "C" int bcp_norm_fwd_small(const float* slabs, int nslab, long long slab_stride, const float* bias, float* ysum, int G,
                                  long long rows_per_group, int C, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float momentum, float eps, int act, const float* chan_scale, long long rows_per_sample,
                                  const uint8_t* elem_mask, float elem_scale, const float* residual, float* stats, float* out, void* stream) {
  BCP_REQUIRE(slabs && stats && nslab >= 1 && nslab <= 64, "bcp_norm_fwd_small: null pointer / bad slab count");
  BCP_REQUIRE(G >= 1 && rows_per_group >= 1 && rows_per_group <= kSmallMaxRows && C >= 4 && (C & 3) == 0,
              "bcp_norm_fwd_small: rows_per_group=%lld C=%d unsupported (need rows <= %d, C %% 4 == 0)", rows_per_group, C, kSmallMaxRows);
  BCP_REQUIRE(aligned16(slabs) && (!out || aligned16(out)) && (!ysum || aligned16(ysum)) && (slab_stride & 3) == 0, "bcp_norm_fwd_small: alignment");
  BCP_REQUIRE(ysum || (nslab == 1 && !bias), "bcp_norm_fwd_small: summing slabs / adding a bias needs the ysum output");
  BCP_REQUIRE(out || !residual, "bcp_norm_fwd_small: statistics-only mode (out = NULL) takes no residual");
  hipStream_t s = (hipStream_t)stream;
  NormEpilogue ep{chan_scale, elem_mask, elem_scale, rows_per_sample > 0 ? rows_per_sample : rows_per_group, act};
  const int R = (int)rows_per_group;
  const int gpb = running_mean ? G : (G >= 2 ? 2 : 1);        // running statistics are updated group after group: one workgroup walks them all
  BCP_SMALL_DISPATCH(k_norm_small_fwd, slabs, nslab, slab_stride, bias, ysum, G, R, C, gpb, gamma, beta, running_mean, running_var, momentum, eps,
                     ep, residual, stats, out);
  BCP_CHECK_LAUNCH("bcp_norm_fwd_small");
  return BCP_OK;
}

extern "C" int bcp_norm_bwd_small(const float* y, const float* da_slabs, int nslab, long long slab_stride, float* da_sum, int G,
                                  long long rows_per_group, int C, const float* stats, int act, const float* chan_scale,
                                  long long rows_per_sample, const uint8_t* elem_mask, float elem_scale, float* dgamma, float* dbeta,
                                  int accumulate, float* dy, void* stream) {
  BCP_REQUIRE(y && da_slabs && stats && dy && nslab >= 1 && nslab <= 64, "bcp_norm_bwd_small: null pointer / bad slab count");
  BCP_REQUIRE(G >= 1 && rows_per_group >= 1 && rows_per_group <= kSmallMaxRows && C >= 4 && (C & 3) == 0,
              "bcp_norm_bwd_small: rows_per_group=%lld C=%d unsupported (need rows <= %d, C %% 4 == 0)", rows_per_group, C, kSmallMaxRows);
  BCP_REQUIRE(aligned16(y) && aligned16(da_slabs) && aligned16(dy) && (!da_sum || aligned16(da_sum)) && (slab_stride & 3) == 0, "bcp_norm_bwd_small: alignment");
  BCP_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bcp_norm_bwd_small: dgamma and dbeta come together");
  hipStream_t s = (hipStream_t)stream;
  NormEpilogue ep{chan_scale, elem_mask, elem_scale, rows_per_sample > 0 ? rows_per_sample : rows_per_group, act};
  const int R = (int)rows_per_group;
  const int gpb = dgamma ? G : (G >= 2 ? 2 : 1);              // parameter gradients sum the groups in order
  BCP_SMALL_DISPATCH(k_norm_small_bwd, y, da_slabs, nslab, slab_stride, da_sum, G, R, C, gpb, stats, ep, dgamma, dbeta, accumulate, dy);
  BCP_CHECK_LAUNCH("bcp_norm_bwd_small");
  return BCP_OK;
}
