// bcp_amd/csrc/loss.hip -- fused masked Dice + CE ("mix_loss") forward / backward for gfx950.
//
// Reference semantics:
//   LA / pancreas  utils/BCP_utils.py:58-69 + utils/losses.py:47-77  (per-(n,c) soft Dice, smooth 1e-5)
//   ACDC           ACDC_BCP_train.py:167-179 + utils/losses.py:102-134 (per-class Dice over the batch,
//                  squared denominators, smooth 1e-10; returns dice and ce separately)
// Every voxel lies in exactly one of the two complementary masks (M = outside the box -> "image"
// term, 1-M = inside -> "patch" term), so ONE pass over the logits accumulates both terms:
// 8 B logits + 2 x 1 B labels per voxel (C=2).  Partial sums are fp64 (wavefront shuffles ->
// LDS -> one fp64 atomic per block per quantity); a 1-block finalize kernel turns them into the
// loss scalar(s) and the per-(n,term,class) coefficient table the backward pass needs, so the
// backward is a second single pass (8 B logits re-read + 8 B dlogits written per voxel).
#include "common.h"
#include "../../include/bcp_hip.h"

namespace bcp {

// accumulator layout (doubles):
//   acc[((n*2 + t)*C + c)*3 + {0: inter, 1: union|z, 2: ysum}]   n < N
//   tail: acc[N*2*C*3 + t*2 + {0: ce_sum, 1: count}]
// coefficient layout (floats) written by the finalize kernel:
//   coef[((n*2 + t)*C + c)*2 + {0: A, 1: B}],  tail coef[N*2*C*2 + t] = CE coefficient
// LA flavour:  dL/dP_c = A*1[y==c] - B          (A,B already hold -1/2 * w_t/(N*C) and signs)
// ACDC flavour: dL/dP_c = A*1[y==c] - B*P_c

template <int C>
struct Softmax {
  float p[C];
  __device__ __forceinline__ void compute(const float* x) {
    float m = x[0];
#pragma unroll
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float s = 0.f;
    if constexpr (C == 2) {
      // (round 6) the larger logit's term is expf(0) = 1 exactly: one expf per voxel instead of two, the same bits (the forward kernel is
      // VALU-bound: ~250 instructions per voxel, 4 cycles each per wave)
      const bool hi1 = x[1] > x[0];                    // m == x[1] (ties: m == x[0] == x[1], both terms expf(0))
      const float e = expf((x[0] - m) + (x[1] - m));   // one difference is +0 exactly, so the sum IS the other one; a NaN / inf - inf in either still reaches s
      p[0] = hi1 ? e : 1.f;
      p[1] = hi1 ? 1.f : e;
      s = p[0] + p[1];
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) { p[c] = expf(x[c] - m); s += p[c]; }
    }
    const float inv = 1.0f / s;
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] *= inv;
    lse = m + logf(s);
  }
  float lse;
};

__device__ __forceinline__ bool in_box(int d, int h, int w, const int* bx) {
  return (d >= bx[0]) & (d < bx[1]) & (h >= bx[2]) & (h < bx[3]) & (w >= bx[4]) & (w < bx[5]);
}

struct BoxArg { int v[6]; };  // d0,d1,h0,h1,w0,w1 (half-open)

// One block handles a contiguous range of voxels of ONE sample n = blockIdx.y.
template <int C, bool ACDC>
__global__ __launch_bounds__(256) void k_mixloss_fwd(const float* __restrict__ logits, const uint8_t* __restrict__ img_l,
                                                     const uint8_t* __restrict__ patch_l,
                                                     const uint8_t* __restrict__ mask /* nullable: 1 = image term */,
                                                     BoxArg box, int D, int H, int W, double* __restrict__ acc, int N,
                                                     unsigned* __restrict__ ticket,
                                                     const uint8_t* __restrict__ img_l2, const uint8_t* __restrict__ patch_l2) {
  // PAIR mode (round 5, img_l2 != NULL): both mix_loss calls of a step in one launch -- logits = [2N] samples, the second N with label maps
  // of their own (img_l2 / patch_l2), the mask shared; blockIdx.y = sample of the pair, rows of partials per sample as before
  const int ny = blockIdx.y, half = (img_l2 && ny >= N) ? 1 : 0, n = ny - half * N;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 3) ticket[threadIdx.x] = 0u;      // for k_mixloss_reduce's last-arriver hand-overs
  const long long V = (long long)D * H * W;
  const float* lg = logits + (long long)ny * V * C;
  const uint8_t* la = (half ? img_l2 : img_l) + (long long)n * V;
  const uint8_t* lb = (half ? patch_l2 : patch_l) + (long long)n * V;
  const uint8_t* mk = mask ? mask + (long long)n * V : nullptr;
  double s[2][C][3];
  double ce[2] = {0.0, 0.0}, cnt[2] = {0.0, 0.0};
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < C; ++c) s[t][c][0] = s[t][c][1] = s[t][c][2] = 0.0;

  // one voxel: softmax + branch-free accumulation into the term the voxel belongs to.  (round 6) Into fp32 partials of ONE trip -- an
  // aligned group of <= 4 consecutive voxels (flush): <= 3 fp32 roundings per partial, and the grouping is a property of the voxel index,
  // not of the launch geometry (the pair launch and the single launches cut the volume differently and still add the same numbers) -- and
  // from there into the fp64 accumulators: a quarter of the fp64 adds and conversions (14 per voxel before: the kernel ran at 0.9 TB/s)
  float fs[2][C][3], fce[2], fcnt[2];
  auto trip_clear = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int c = 0; c < C; ++c) fs[t][c][0] = fs[t][c][1] = fs[t][c][2] = 0.f;
      fce[t] = 0.f; fcnt[t] = 0.f;
    }
  };
  auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        s[t][c][0] += (double)fs[t][c][0];
        s[t][c][1] += (double)fs[t][c][1];
        if (ACDC) s[t][c][2] += (double)fs[t][c][2];
      }
      ce[t] += (double)fce[t];
      cnt[t] += (double)fcnt[t];
    }
    trip_clear();
  };
  trip_clear();
  auto voxel = [&](const float (&x)[C], int t, int y) __attribute__((always_inline)) {
    Softmax<C> sm;
    sm.compute(x);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const float m = (tt == t) ? 1.f : 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float oh = (y == c) ? 1.f : 0.f;
        fs[tt][c][0] += sm.p[c] * oh * m;
        if (ACDC) {
          fs[tt][c][1] += sm.p[c] * sm.p[c] * m;
          fs[tt][c][2] += oh * m;
        } else {
          fs[tt][c][1] += (sm.p[c] + oh) * m;
        }
      }
      float xy = x[0];
#pragma unroll
      for (int c = 1; c < C; ++c) xy = (y == c) ? x[c] : xy;
      fce[tt] += (sm.lse - xy) * m;
      fcnt[tt] += m;
    }
  };
  auto term_of = [&](long long v) __attribute__((always_inline)) -> int {
    if (mk) return mk[v] ? 0 : 1;
    const unsigned vu = (unsigned)v, q1 = vu / (unsigned)W;          // V < 2^31 (checked by the entry point): 32-bit divisions
    const int w = (int)(vu - q1 * (unsigned)W);
    const int d = (int)(q1 / (unsigned)H);
    const int h = (int)(q1 - (unsigned)d * (unsigned)H);
    return in_box(d, h, w, box.v) ? 1 : 0;
  };
  // 16 bytes of logits per lane and iteration (VP voxels), two iterations in flight; V % VP == 0 is the launcher's condition for
  // this path (sample bases stay 16-byte aligned), the scalar loop below serves the rest
  constexpr int VP = 4 / C;
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (VP >= 1 && C * VP == 4 && (V % VP) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15u) == 0) {
    // a trip = G = 2 consecutive float4 of one lane (32 bytes: four voxels at C = 2, two at C = 4), both requested before the first is used
    constexpr int G = 2;
    const long long VV = V / VP, VG = VV / G;
    const bool row_trip = !mk && (W % (G * VP)) == 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < VG; i += stride) {
      float4 q[G];
#pragma unroll
      for (int g = 0; g < G; ++g) q[g] = ld4(lg + (i * G + g) * 4);
      // the trip's G * VP consecutive voxels lie in one row when W is a multiple of that (80, 96, 256 in the reference's configurations):
      // one (d, h, w) decomposition -- two 32-bit divisions -- per trip instead of per voxel
      const long long v0 = i * G * VP;
      int td = 0, th = 0, tw0 = 0;
      if (row_trip) {
        const unsigned vu = (unsigned)v0, q1 = vu / (unsigned)W;
        tw0 = (int)(vu - q1 * (unsigned)W);
        td = (int)(q1 / (unsigned)H);
        th = (int)(q1 - (unsigned)td * (unsigned)H);
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float xv[4] = {q[g].x, q[g].y, q[g].z, q[g].w};
#pragma unroll
        for (int u = 0; u < VP; ++u) {
          const long long v = v0 + g * VP + u;
          float x[C];
#pragma unroll
          for (int c = 0; c < C; ++c) x[c] = xv[u * C + c];
          const int t = row_trip ? (in_box(td, th, tw0 + g * VP + u, box.v) ? 1 : 0) : term_of(v);
          voxel(x, t, t ? lb[v] : la[v]);
        }
      }
      flush();
    }
    // the float4 behind the last whole trip (VV % G of them), one trip of their own -- by the block and lane that would own the next trip
    if ((long long)blockIdx.x * blockDim.x + threadIdx.x == VG % stride && VG * G < VV) {
      for (long long j = VG * G; j < VV; ++j) {
        const float4 q = ld4(lg + j * 4);
        const float xv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int u = 0; u < VP; ++u) {
          const long long v = j * VP + u;
          float x[C];
#pragma unroll
          for (int c = 0; c < C; ++c) x[c] = xv[u * C + c];
          const int t = term_of(v);
          voxel(x, t, t ? lb[v] : la[v]);
        }
      }
      flush();
    }
  } else {
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += stride) {
      float x[C];
#pragma unroll
      for (int c = 0; c < C; ++c) x[c] = lg[v * C + c];
      const int t = term_of(v);
      voxel(x, t, t ? lb[v] : la[v]);
      flush();
    }
  }
  // block reduction: wave shuffles then LDS, then one fp64 atomic per quantity
  __shared__ double red[4][2 * C * 3 + 4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double r = wave_sum(s[t][c][k]);
        if (lane == 0) red[wid][(t * C + c) * 3 + k] = r;
      }
    const double r1 = wave_sum(ce[t]), r2 = wave_sum(cnt[t]);
    if (lane == 0) { red[wid][2 * C * 3 + t * 2] = r1; red[wid][2 * C * 3 + t * 2 + 1] = r2; }
  }
  __syncthreads();
  // one row of per-block partials [n][block][2*C*3 quantities of sample n | 4 CE / count values]: no atomics -- 2048 blocks
  // adding into the same 28 fp64 addresses serialised in L2 for ~40 us at the LA size -- and a fixed summation order
  const int nq = 2 * C * 3 + 4;
  if ((int)threadIdx.x < nq)
    acc[((long long)ny * gridDim.x + blockIdx.x) * nq + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[0] = LA: loss ; ACDC: dice.   out[1] = ACDC: ce (LA: ce part, informational). out[2] = LA dice part.
template <int C, bool ACDC>
__device__ void mixloss_finalize(const double* __restrict__ acc, float* __restrict__ coef, float* __restrict__ out, int N,
                                 float w_img, float w_patch, const float* __restrict__ prev, float* __restrict__ total) {
  double tail[4] = {0.0, 0.0, 0.0, 0.0};
  for (int n = 0; n < N; ++n)
    for (int q = 0; q < 4; ++q) tail[q] += acc[(long long)N * 2 * C * 3 + (long long)n * 4 + q];
  const double wt[2] = {(double)w_img, (double)w_patch};
  double dice_total = 0.0, ce_total = 0.0;
  for (int t = 0; t < 2; ++t) {
    const double cecoef = wt[t] / (tail[t * 2 + 1] + 1e-16);
    ce_total += cecoef * tail[t * 2];
    coef[(long long)N * 2 * C * 2 + t] = (float)cecoef;
  }
  if (!ACDC) {
    const double smooth = 1e-5;
    for (int t = 0; t < 2; ++t) {
      double dsum = 0.0;
      for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
          const double* q = acc + (((long long)n * 2 + t) * C + c) * 3;
          const double I = q[0], U = q[1];
          dsum += (2.0 * I + smooth) / (U + smooth);
          // d/dP_c of -(w/(N*C)) * (2I+s)/(U+s)  at a voxel with label y:
          //   -(w/(N*C)) * [ 2*1[y==c]/(U+s) - (2I+s)/(U+s)^2 ]
          const double k = -wt[t] / ((double)N * C);
          float* cf = coef + (((long long)n * 2 + t) * C + c) * 2;
          cf[0] = (float)(k * 2.0 / (U + smooth));
          cf[1] = (float)(k * (2.0 * I + smooth) / ((U + smooth) * (U + smooth)));
        }
      dice_total += wt[t] * (1.0 - dsum / ((double)N * C));
    }
    out[0] = (float)((dice_total + ce_total) / 2.0);
    out[1] = (float)ce_total;
    out[2] = (float)dice_total;
  } else {
    const double smooth = 1e-10;
    for (int t = 0; t < 2; ++t) {
      double lsum = 0.0;
      for (int c = 0; c < C; ++c) {
        double I = 0.0, Z = 0.0, Y = 0.0;
        for (int n = 0; n < N; ++n) {
          const double* q = acc + (((long long)n * 2 + t) * C + c) * 3;
          I += q[0]; Z += q[1]; Y += q[2];
        }
        const double den = Z + Y + smooth;
        lsum += 1.0 - (2.0 * I + smooth) / den;
        // d/dP_c of (w/C) * [1 - (2I+s)/den]:  -(w/C) * [ 2*1[y==c]/den - (2I+s)*2*P_c/den^2 ]
        const double k = -wt[t] / (double)C;
        for (int n = 0; n < N; ++n) {
          float* cf = coef + (((long long)n * 2 + t) * C + c) * 2;
          cf[0] = (float)(k * 2.0 / den);
          cf[1] = (float)(k * (2.0 * I + smooth) * 2.0 / (den * den));
        }
      }
      dice_total += wt[t] * lsum / (double)C;
    }
    out[0] = (float)dice_total;
    out[1] = (float)ce_total;
    out[2] = (float)((dice_total + ce_total) / 2.0);
  }
  // the step's total over both mix_loss calls, in the reference's fp32 order (round 4: the torch adds / division that followed are gone):
  //   LA / pancreas  loss = loss_l + loss_u                                        LA_BCP_train.py:255, train_pancreas.py:166
  //   ACDC           loss = ((unl_dice + l_dice) + (unl_ce + l_ce)) / 2            ACDC_BCP_train.py:381-384
  if (total) {
    if (!ACDC) total[0] = prev[0] + out[0];
    else total[0] = ((prev[0] + out[0]) + (prev[1] + out[1])) / 2.f;
  }
}

template <int C, bool ACDC>
__global__ __launch_bounds__(256) void k_mixloss_reduce(const double* __restrict__ partial, int nb, double* __restrict__ acc, int N,
                                                        unsigned* __restrict__ ticket, float* __restrict__ coef, float* __restrict__ out,
                                                        float w_img, float w_patch, const float* __restrict__ prev, float* __restrict__ total,
                                                        int pair, float w_img2, float w_patch2) {
  // PAIR mode (pair != 0; grid = 2N blocks): blocks 0 .. N-1 are the first call, N .. 2N-1 the second, each half with reduced rows, ticket,
  // coefficient table and out3 of its own (the second half's right behind the first's); the last arriver of a half finalises that half as
  // the single call would, then the LATER of the two finalisers adds the step's total from both out3 in the first-call-then-second order
  // block n: the nb per-block rows of sample n -> acc[n][2*C*3] and tailp[n][4], in a fixed order (thread t sums rows t, t + 256, ...;
  // then lanes, waves).  The LAST block to arrive (ticket zeroed by the forward kernel, a kernel boundary earlier) turns the N reduced
  // rows into the loss scalar(s) and the coefficient table: no finalize launch.  The hand-over is a few hundred bytes per block:
  // release fence -> ticket -> acquire fence in the last arriver only.
  constexpr int nq = 2 * C * 3 + 4;
  static_assert(nq <= 32, "one 32-lane slot per row");
  __shared__ double wred[8][32];
  __shared__ unsigned s_last;
  const int ny = blockIdx.x, half = (pair && ny >= N) ? 1 : 0, n = ny - half * N;
  acc += (size_t)half * N * nq;                    // this half's reduced rows [N][2*C*3] | [N][4]
  coef += (size_t)half * (N * 2 * C * 2 + 2);
  out += half * 3;
  // eight row slots of 32 lanes: a row's nq doubles are read by consecutive lanes (coalesced; round 4 -- with thread = row the 224-byte
  // rows were read at a 224-byte stride: 19 us for the ACDC launch); slot s sums rows s, s + 8, ... in order, then the slots in order
  const int q = threadIdx.x & 31, rs = threadIdx.x >> 5;
  double v = 0.0;
  if (q < nq) {
    // eight loads in flight per thread (the plain loop issued them one by one: 64 dependent L2 round trips = 19 us for the LA launch);
    // rows are added in ascending order either way
    const double* base = partial + (long long)ny * nb * nq + q;
    int r = rs;
    for (; r + 56 < nb; r += 64) {
      double t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = base[(long long)(r + 8 * k) * nq];
#pragma unroll
      for (int k = 0; k < 8; ++k) v += t[k];
    }
    for (; r < nb; r += 8) v += base[(long long)r * nq];
  }
  wred[rs][q] = v;
  __syncthreads();
  if ((int)threadIdx.x < nq) {
    double t = wred[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += wred[k][threadIdx.x];
    if ((int)threadIdx.x < 2 * C * 3) acc[(long long)n * 2 * C * 3 + threadIdx.x] = t;
    else acc[(long long)N * 2 * C * 3 + (long long)n * 4 + (threadIdx.x - 2 * C * 3)] = t;
    __threadfence();
  }
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(ticket + half, 1u) == (unsigned)(N - 1)) ? 1u : 0u;
  __syncthreads();
  if (s_last) {
    // the N reduced rows come into the LDS with ALL threads loading (one dependent global round trip instead of ~N * nq of them in the
    // single finalising thread: that serial walk was 19 us of the launch), then thread 0 does the scalar arithmetic out of the LDS
    constexpr int kStageRows = 32;
    __shared__ double stage[kStageRows * nq];
    if (threadIdx.x == 0) __threadfence();
    __syncthreads();
    const int tot = N * nq;
    const bool staged = N <= kStageRows;
    if (staged)
      for (int i = threadIdx.x; i < tot; i += 256) stage[i] = acc[i];          // acc = [N][2*C*3] | [N][4]: the layout mixloss_finalize reads
    __syncthreads();
    if (threadIdx.x == 0) {
      if (!pair) mixloss_finalize<C, ACDC>(staged ? stage : acc, coef, out, N, w_img, w_patch, prev, total);
      else {
        mixloss_finalize<C, ACDC>(staged ? stage : acc, coef, out, N, half ? w_img2 : w_img, half ? w_patch2 : w_patch, nullptr, nullptr);
        __threadfence();
        if (atomicAdd(ticket + 2, 1u) == 1u) {       // the other half's out3 is final and visible: the step's total, as the second call's finalize forms it
          __threadfence();
          const float* o1 = out - half * 3;          // first call's out3
          const float a0 = __hip_atomic_load(o1 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a1 = __hip_atomic_load(o1 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const float b0 = __hip_atomic_load(o1 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b1 = __hip_atomic_load(o1 + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!ACDC) total[0] = a0 + b0;
          else total[0] = ((a0 + b0) + (a1 + b1)) / 2.f;
        }
      }
    }
  }
}

template <int C, bool ACDC>
__global__ __launch_bounds__(256) void k_mixloss_bwd(const float* __restrict__ logits, const uint8_t* __restrict__ img_l,
                                                     const uint8_t* __restrict__ patch_l, const uint8_t* __restrict__ mask,
                                                     BoxArg box, int D, int H, int W, const float* __restrict__ coef,
                                                     float* __restrict__ dlogits, int N, float g_dice, float g_ce,
                                                     const float* __restrict__ g_dev /* nullable [g_dev_n] */, int g_dev_n,
                                                     const uint8_t* __restrict__ img_l2, const uint8_t* __restrict__ patch_l2) {
  const int ny = blockIdx.y, half = (img_l2 && ny >= N) ? 1 : 0, n = ny - half * N;      // PAIR mode: see k_mixloss_fwd
  coef += (size_t)half * (N * 2 * C * 2 + 2);
  const long long V = (long long)D * H * W;
  const float* lg = logits + (long long)ny * V * C;
  float* dl = dlogits + (long long)ny * V * C;
  const uint8_t* la = (half ? img_l2 : img_l) + (long long)n * V;
  const uint8_t* lb = (half ? patch_l2 : patch_l) + (long long)n * V;
  const uint8_t* mk = mask ? mask + (long long)n * V : nullptr;
  __shared__ float cf[2][C][2];
  __shared__ float cce[2];
  if ((int)threadIdx.x < 2 * C * 2) (&cf[0][0][0])[threadIdx.x] = coef[(long long)n * 2 * C * 2 + threadIdx.x];
  if ((int)threadIdx.x < 2) cce[threadIdx.x] = coef[(long long)N * 2 * C * 2 + threadIdx.x];
  __syncthreads();
  if (g_dev) { g_dice *= g_dev[0]; g_ce *= g_dev[g_dev_n > 1 ? 1 : 0]; }  // upstream gradients stay on the device (no host sync)
  auto term_of = [&](long long v) __attribute__((always_inline)) -> int {
    if (mk) return mk[v] ? 0 : 1;
    const unsigned vu = (unsigned)v, q1 = vu / (unsigned)W;          // V < 2^31 (checked by the entry point): 32-bit divisions
    const int w = (int)(vu - q1 * (unsigned)W);
    const int d = (int)(q1 / (unsigned)H);
    const int h = (int)(q1 - (unsigned)d * (unsigned)H);
    return in_box(d, h, w, box.v) ? 1 : 0;
  };
  auto voxel = [&](const float (&x)[C], int t, int y, float (&o)[C]) __attribute__((always_inline)) {
    Softmax<C> sm;
    sm.compute(x);
    float gp[C];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float oh = (y == c) ? 1.f : 0.f;
      gp[c] = ACDC ? (cf[t][c][0] * oh - cf[t][c][1] * sm.p[c]) : (cf[t][c][0] * oh - cf[t][c][1]);
      dot += gp[c] * sm.p[c];
    }
    const float kce = g_ce * cce[t];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float oh = (y == c) ? 1.f : 0.f;
      o[c] = g_dice * (sm.p[c] * (gp[c] - dot)) + kce * (sm.p[c] - oh);
    }
  };
  constexpr int VP = 4 / C;
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (VP >= 1 && C * VP == 4 && (V % VP) == 0 && ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15u) == 0) {
    const long long VV = V / VP;
    const bool row_vec = !mk && VP > 1 && (W % VP) == 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < VV; i += stride) {
      const float4 q = ld4(lg + i * 4);
      const float xv[4] = {q.x, q.y, q.z, q.w};
      float ov[4];
      // (round 6) the float4's VP voxels lie in one row when W % VP == 0: one (d, h, w) decomposition for them (see k_mixloss_fwd)
      int td = 0, th = 0, tw0 = 0;
      if (row_vec) {
        const unsigned vu = (unsigned)(i * VP), q1 = vu / (unsigned)W;
        tw0 = (int)(vu - q1 * (unsigned)W);
        td = (int)(q1 / (unsigned)H);
        th = (int)(q1 - (unsigned)td * (unsigned)H);
      }
#pragma unroll
      for (int u = 0; u < VP; ++u) {
        const long long v = i * VP + u;
        float x[C], o[C];
#pragma unroll
        for (int c = 0; c < C; ++c) x[c] = xv[u * C + c];
        const int t = row_vec ? (in_box(td, th, tw0 + u, box.v) ? 1 : 0) : term_of(v);
        voxel(x, t, t ? lb[v] : la[v], o);
#pragma unroll
        for (int c = 0; c < C; ++c) ov[u * C + c] = o[c];
      }
      st4(dl + i * 4, make_float4(ov[0], ov[1], ov[2], ov[3]));
    }
  } else {
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += stride) {
      float x[C], o[C];
#pragma unroll
      for (int c = 0; c < C; ++c) x[c] = lg[v * C + c];
      const int t = term_of(v);
      voxel(x, t, t ? lb[v] : la[v], o);
#pragma unroll
      for (int c = 0; c < C; ++c) dl[v * C + c] = o[c];
    }
  }
}

constexpr int kLossPartialRows = 512;  // most forward blocks per sample = rows of per-block partials the reduce kernel sums

// pair: the second call's label maps and weights (img_l2 == NULL: one call); N = samples PER CALL
struct PairArg { const uint8_t* img_l2; const uint8_t* patch_l2; float w_img2, w_patch2; };
template <int C, bool ACDC>
static int launch_fwd(const float* logits, const uint8_t* img_l, const uint8_t* patch_l, const uint8_t* mask, const int* box6,
                      int N, int D, int H, int W, float w_img, float w_patch, double* acc,
                      float* coef, float* out, const float* prev, float* total, hipStream_t s, const PairArg pa = PairArg{nullptr, nullptr, 0.f, 0.f}) {
  const int NT = pa.img_l2 ? 2 * N : N;            // samples of the launch
  BoxArg bx;
  bx.v[0] = box6[0]; bx.v[1] = box6[0] + box6[3];
  bx.v[2] = box6[1]; bx.v[3] = box6[1] + box6[4];
  bx.v[4] = box6[2]; bx.v[5] = box6[2] + box6[5];
  const long long V = (long long)D * H * W;
  // 16 bytes of logits per lane and iteration: ~2 iterations per thread at the LA size, >= 2048 workgroups per launch at most
  long long nbl = (V * C / 4 + 511) / 512;
  if (nbl * NT > 1024) nbl = (1024 + NT - 1) / NT;
  const int nb = (int)(nbl < 1 ? 1 : (nbl > kLossPartialRows ? kLossPartialRows : nbl));
  double* red = acc + (size_t)NT * kLossPartialRows * (2 * C * 3 + 4);          // per call: [N][2*C*3] | [N][4]
  unsigned* ticket = reinterpret_cast<unsigned*>(red + (size_t)NT * (2 * C * 3 + 4));      // three words (one per call, one for the pair)
  hipLaunchKernelGGL((k_mixloss_fwd<C, ACDC>), dim3(nb, NT), dim3(256), 0, s, logits, img_l, patch_l, mask, bx, D, H, W, acc, N, ticket,
                     pa.img_l2, pa.patch_l2);
  hipLaunchKernelGGL((k_mixloss_reduce<C, ACDC>), dim3(NT), dim3(256), 0, s, acc, nb, red, N, ticket, coef, out, w_img, w_patch, prev, total,
                     pa.img_l2 ? 1 : 0, pa.w_img2, pa.w_patch2);
  return 0;
}

template <int C, bool ACDC>
static int launch_bwd(const float* logits, const uint8_t* img_l, const uint8_t* patch_l, const uint8_t* mask, const int* box6,
                      int N, int D, int H, int W, const float* coef, float* dlogits, float g_dice, float g_ce, const float* g_dev, int g_dev_n, hipStream_t s,
                      const PairArg pa = PairArg{nullptr, nullptr, 0.f, 0.f}) {
  const int NT = pa.img_l2 ? 2 * N : N;
  BoxArg bx;
  bx.v[0] = box6[0]; bx.v[1] = box6[0] + box6[3];
  bx.v[2] = box6[1]; bx.v[3] = box6[1] + box6[4];
  bx.v[4] = box6[2]; bx.v[5] = box6[2] + box6[5];
  const long long V = (long long)D * H * W;
  long long gb = (V * C / 4 + 255) / 256;          // one 16-byte vector per thread, at most ~4096 workgroups per launch
  if (gb * NT > 4096) gb = (4096 + NT - 1) / NT;
  hipLaunchKernelGGL((k_mixloss_bwd<C, ACDC>), dim3((int)(gb < 1 ? 1 : gb), NT), dim3(256), 0, s, logits, img_l, patch_l, mask, bx, D, H,
                     W, coef, dlogits, N, g_dice, g_ce, g_dev, g_dev_n, pa.img_l2, pa.patch_l2);
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------------------
// utils/losses.py:79-134 `DiceLoss.forward(inputs, target, mask, weight, softmax)` AS A CLASS: `inputs` are PROBABILITIES (the
// ACDC script hands it F.softmax(output), ACDC_BCP_train.py:170-176), any dense layout (channel / voxel / sample strides in
// elements: torch's softmax returns NCHW-contiguous, the networks here produce NHWC), per-class sums over the whole batch,
// squared denominators, smooth 1e-10 (utils/losses.py:94 and :105: masked and unmasked alike), per-class weights.  The fused step uses bcp_mixloss_* above;
// this pair exists so the reference's own loss body runs unchanged on the seam.
// mask_mode: 0 none, 1 dense uint8 (non-zero = counted), 2 box (1 OUTSIDE the box), 3 box complement (1 inside)
constexpr int kDiceProbRows = 512;
constexpr int kDiceProbMaxC = 8;
struct DiceW { float w[kDiceProbMaxC]; };

__device__ __forceinline__ float dice_prob_mask(const uint8_t* mk, long long nv, int mode, BoxArg box, unsigned v, int H, int W) {
  if (mode == 0) return 1.f;
  if (mode == 1) return mk[nv] ? 1.f : 0.f;
  const unsigned q1 = v / (unsigned)W;
  const int w = (int)(v - q1 * (unsigned)W);
  const int d = (int)(q1 / (unsigned)H);
  const int h = (int)(q1 - (unsigned)d * (unsigned)H);
  const bool in = in_box(d, h, w, box.v);
  return (mode == 2) ? (in ? 0.f : 1.f) : (in ? 1.f : 0.f);
}

template <int C>
__global__ __launch_bounds__(256) void k_dice_prob_fwd(const float* __restrict__ p, long long cs, long long vs, long long ns,
                                                       const uint8_t* __restrict__ target, const uint8_t* __restrict__ mask, int mode,
                                                       BoxArg box, int N, int D, int H, int W, double* __restrict__ partial) {
  const long long V = (long long)D * H * W, T = (long long)N * V;
  double s[C * 3];
#pragma unroll
  for (int i = 0; i < C * 3; ++i) s[i] = 0.0;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < T; i += stride) {
    const long long n = i / V, v = i - n * V;
    const float m = dice_prob_mask(mask, i, mode, box, (unsigned)v, H, W);
    const int y = target[i];
    const float* q = p + n * ns + v * vs;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float pc = q[c * cs], oh = (y == c) ? 1.f : 0.f;
      s[c * 3 + 0] += (double)(pc * oh * m);
      s[c * 3 + 1] += (double)(pc * pc * m);
      s[c * 3 + 2] += (double)(oh * m);
    }
  }
  __shared__ double red[4 * C * 3];
  block_sum_256<C * 3>(s, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < C * 3; ++i) partial[(long long)blockIdx.x * (C * 3) + i] = s[i];
  }
}

// one block: rows summed in a fixed order -> loss, per-class (1 - dice) and the coefficient table {A_c, B_c}:
// dL/dp_c(v) = m(v) * (A_c * 1[y == c] + B_c * p_c(v))
template <int C>
__global__ __launch_bounds__(256) void k_dice_prob_finalize(const double* __restrict__ partial, int nb, DiceW wt, int masked,
                                                            float* __restrict__ coef, float* __restrict__ out) {
  double s[C * 3];
#pragma unroll
  for (int i = 0; i < C * 3; ++i) s[i] = 0.0;
  for (int r = threadIdx.x; r < nb; r += 256) {
#pragma unroll
    for (int i = 0; i < C * 3; ++i) s[i] += partial[(long long)r * (C * 3) + i];
  }
  __shared__ double red[4 * C * 3];
  block_sum_256<C * 3>(s, red);
  if (threadIdx.x != 0) return;
  const double smooth = 1e-10;      // _dice_loss AND _dice_mask_loss (utils/losses.py:94, :105); (void)masked
  double loss = 0.0;
  for (int c = 0; c < C; ++c) {
    const double I = s[c * 3], Z = s[c * 3 + 1], Y = s[c * 3 + 2];
    const double den = Z + Y + smooth, num = 2.0 * I + smooth;
    const double dice = 1.0 - num / den;
    loss += dice * (double)wt.w[c];
    const double k = (double)wt.w[c] / (double)C;
    coef[c * 2 + 0] = (float)(-k * 2.0 / den);
    coef[c * 2 + 1] = (float)(k * num * 2.0 / (den * den));
    out[1 + c] = (float)(1.0 - dice);          // class_wise_dice of the reference (computed there, never returned)
  }
  out[0] = (float)(loss / (double)C);
}

template <int C>
__global__ __launch_bounds__(256) void k_dice_prob_bwd(const float* __restrict__ p, long long cs, long long vs, long long ns,
                                                       const uint8_t* __restrict__ target, const uint8_t* __restrict__ mask, int mode,
                                                       BoxArg box, int N, int D, int H, int W, const float* __restrict__ coef,
                                                       const float* __restrict__ g_dev, float g, float* __restrict__ dp) {
  const long long V = (long long)D * H * W, T = (long long)N * V;
  __shared__ float cf[C * 2];
  if ((int)threadIdx.x < C * 2) cf[threadIdx.x] = coef[threadIdx.x];
  __syncthreads();
  if (g_dev) g *= g_dev[0];
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < T; i += stride) {
    const long long n = i / V, v = i - n * V;
    const float m = dice_prob_mask(mask, i, mode, box, (unsigned)v, H, W) * g;
    const int y = target[i];
    const long long o = n * ns + v * vs;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float pc = p[o + c * cs], oh = (y == c) ? 1.f : 0.f;
      dp[o + c * cs] = m * (cf[c * 2] * oh + cf[c * 2 + 1] * pc);
    }
  }
}

static inline int dice_prob_grid(long long T) {
  long long gx = (T + 255) / 256;
  if (gx > kDiceProbRows) gx = kDiceProbRows;
  return (int)(gx < 1 ? 1 : gx);
}

static inline BoxArg box_arg(const int* box6) {
  BoxArg bx;
  bx.v[0] = box6[0]; bx.v[1] = box6[0] + box6[3];
  bx.v[2] = box6[1]; bx.v[3] = box6[1] + box6[4];
  bx.v[4] = box6[2]; bx.v[5] = box6[2] + box6[5];
  return bx;
}

}  // namespace bcp

using namespace bcp;

extern "C" size_t bcp_dice_prob_workspace_bytes(int C) {
  // [per-block partial rows (doubles) | coef floats C*2]
  return (size_t)kDiceProbRows * (size_t)C * 3 * sizeof(double) + (size_t)C * 2 * sizeof(float) + 16;
}

static inline float* dice_prob_coef(void* ws, int C) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (size_t)kDiceProbRows * (size_t)C * 3 * sizeof(double));
}

template <int CC>
static void dice_prob_launch_fwd(const float* probs, long long cs, long long vs, long long ns, const uint8_t* target, const uint8_t* mask,
                                 int mode, BoxArg bx, int N, int D, int H, int W, DiceW wt, double* partial, float* coef, float* out,
                                 int nb, hipStream_t s) {
  hipLaunchKernelGGL((k_dice_prob_fwd<CC>), dim3(nb), dim3(256), 0, s, probs, cs, vs, ns, target, mask, mode, bx, N, D, H, W, partial);
  hipLaunchKernelGGL((k_dice_prob_finalize<CC>), dim3(1), dim3(256), 0, s, partial, nb, wt, mode != 0 ? 1 : 0, coef, out);
}

template <int CC>
static void dice_prob_launch_bwd(const float* probs, long long cs, long long vs, long long ns, const uint8_t* target, const uint8_t* mask,
                                 int mode, BoxArg bx, int N, int D, int H, int W, const float* coef, const float* g_dev, float g,
                                 float* dprobs, int gx, hipStream_t s) {
  hipLaunchKernelGGL((k_dice_prob_bwd<CC>), dim3(gx), dim3(256), 0, s, probs, cs, vs, ns, target, mask, mode, bx, N, D, H, W, coef, g_dev,
                     g, dprobs);
}

extern "C" int bcp_dice_prob_fwd(const float* probs, long long cstride, long long vstride, long long nstride, const uint8_t* target,
                                 const uint8_t* mask_or_null, int mask_mode, const int* box6, int N, int D, int H, int W, int C,
                                 const float* weight_host_or_null, void* workspace, float* out /* [1 + C] */, void* stream) {
  BCP_REQUIRE(probs && target && workspace && out, "bcp_dice_prob_fwd: null pointer");
  BCP_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && (long long)D * H * W < (1LL << 31), "bcp_dice_prob_fwd: bad extents");
  BCP_REQUIRE(C >= 2 && C <= 4, "bcp_dice_prob_fwd: C = %d unsupported (2..4)", C);
  BCP_REQUIRE(mask_mode >= 0 && mask_mode <= 3 && (mask_mode != 1 || mask_or_null) && (mask_mode < 2 || box6),
              "bcp_dice_prob_fwd: mask_mode %d without its mask / box", mask_mode);
  BCP_REQUIRE(cstride > 0 && vstride > 0 && nstride > 0, "bcp_dice_prob_fwd: strides must be positive");
  DiceW wt;
  for (int c = 0; c < kDiceProbMaxC; ++c) wt.w[c] = (weight_host_or_null && c < C) ? weight_host_or_null[c] : 1.f;
  BoxArg bx = {};
  if (mask_mode >= 2) bx = box_arg(box6);
  const long long T = (long long)N * D * H * W;
  const int nb = dice_prob_grid(T);
  double* partial = reinterpret_cast<double*>(workspace);
  float* coef = dice_prob_coef(workspace, C);
  hipStream_t s = (hipStream_t)stream;
  if (C == 2) dice_prob_launch_fwd<2>(probs, cstride, vstride, nstride, target, mask_or_null, mask_mode, bx, N, D, H, W, wt, partial, coef, out, nb, s);
  else if (C == 3) dice_prob_launch_fwd<3>(probs, cstride, vstride, nstride, target, mask_or_null, mask_mode, bx, N, D, H, W, wt, partial, coef, out, nb, s);
  else dice_prob_launch_fwd<4>(probs, cstride, vstride, nstride, target, mask_or_null, mask_mode, bx, N, D, H, W, wt, partial, coef, out, nb, s);
  BCP_CHECK_LAUNCH("bcp_dice_prob_fwd");
  return BCP_OK;
}

extern "C" int bcp_dice_prob_bwd(const float* probs, long long cstride, long long vstride, long long nstride, const uint8_t* target,
                                 const uint8_t* mask_or_null, int mask_mode, const int* box6, int N, int D, int H, int W, int C,
                                 const void* workspace, const float* g_dev_or_null, float g, float* dprobs, void* stream) {
  BCP_REQUIRE(probs && target && workspace && dprobs, "bcp_dice_prob_bwd: null pointer");
  BCP_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && (long long)D * H * W < (1LL << 31), "bcp_dice_prob_bwd: bad extents");
  BCP_REQUIRE(C >= 2 && C <= 4, "bcp_dice_prob_bwd: C = %d unsupported (2..4)", C);
  BCP_REQUIRE(mask_mode >= 0 && mask_mode <= 3 && (mask_mode != 1 || mask_or_null) && (mask_mode < 2 || box6),
              "bcp_dice_prob_bwd: mask_mode %d without its mask / box", mask_mode);
  BoxArg bx = {};
  if (mask_mode >= 2) bx = box_arg(box6);
  const long long T = (long long)N * D * H * W;
  long long gx = (T + 255) / 256;
  if (gx > 2048) gx = 2048;
  const float* coef = dice_prob_coef(const_cast<void*>(workspace), C);
  hipStream_t s = (hipStream_t)stream;
  if (C == 2) dice_prob_launch_bwd<2>(probs, cstride, vstride, nstride, target, mask_or_null, mask_mode, bx, N, D, H, W, coef, g_dev_or_null, g, dprobs, (int)gx, s);
  else if (C == 3) dice_prob_launch_bwd<3>(probs, cstride, vstride, nstride, target, mask_or_null, mask_mode, bx, N, D, H, W, coef, g_dev_or_null, g, dprobs, (int)gx, s);
  else dice_prob_launch_bwd<4>(probs, cstride, vstride, nstride, target, mask_or_null, mask_mode, bx, N, D, H, W, coef, g_dev_or_null, g, dprobs, (int)gx, s);
  BCP_CHECK_LAUNCH("bcp_dice_prob_bwd");
  return BCP_OK;
}

extern "C" size_t bcp_mixloss_workspace_bytes(int N, int C) {
  // [acc doubles | coef floats], both 16-B aligned
  const size_t acc = (size_t)N * (kLossPartialRows + 1) * (2 * C * 3 + 4) * sizeof(double) + 16;   // per-block rows + the reduced row per sample + the ticket
  const size_t coef = ((size_t)N * 2 * C * 2 + 2) * sizeof(float);
  return ((acc + 15) / 16) * 16 + ((coef + 15) / 16) * 16;
}

static inline float* coef_ptr(void* ws, int N, int C) {
  const size_t acc = (size_t)N * (kLossPartialRows + 1) * (2 * C * 3 + 4) * sizeof(double) + 16;   // per-block rows + the reduced row per sample + the ticket
  return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ((acc + 15) / 16) * 16);
}

extern "C" int bcp_mixloss_fwd(const float* logits, const uint8_t* img_l, const uint8_t* patch_l, const uint8_t* mask_or_null,
                               const int* box6, int N, int D, int H, int W, int C, int flavour, float w_img, float w_patch,
                               void* workspace, float* out3, const float* prev_out3_or_null, float* total_or_null, void* stream) {
  BCP_REQUIRE(logits && img_l && patch_l && box6 && workspace && out3, "bcp_mixloss_fwd: null pointer");
  BCP_REQUIRE((prev_out3_or_null == nullptr) == (total_or_null == nullptr), "bcp_mixloss_fwd: prev_out3 and total come together");
  BCP_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && (long long)D * H * W < (1LL << 31), "bcp_mixloss_fwd: bad extents");
  BCP_REQUIRE((flavour == BCP_LOSS_LA && C == 2) || (flavour == BCP_LOSS_ACDC && C == 4),
              "bcp_mixloss_fwd: flavour/C combination unsupported (LA: C=2, ACDC: C=4), got flavour=%d C=%d", flavour, C);
  double* acc = reinterpret_cast<double*>(workspace);
  float* coef = coef_ptr(workspace, N, C);
  if (flavour == BCP_LOSS_LA)
    launch_fwd<2, false>(logits, img_l, patch_l, mask_or_null, box6, N, D, H, W, w_img, w_patch, acc, coef, out3, prev_out3_or_null,
                         total_or_null, (hipStream_t)stream);
  else
    launch_fwd<4, true>(logits, img_l, patch_l, mask_or_null, box6, N, D, H, W, w_img, w_patch, acc, coef, out3, prev_out3_or_null,
                        total_or_null, (hipStream_t)stream);
  BCP_CHECK_LAUNCH("bcp_mixloss_fwd");
  return BCP_OK;
}

// ---- round 5: both mix_loss calls of a self-training step (LA_BCP_train.py:252-254, ACDC_BCP_train.py:370-377, train_pancreas.py:160-165) in ONE
// launch pair: logits = [2N] samples (the grouped student forward's output), call 1 = samples 0 .. N-1 with (img_l, patch_l, w_img, w_patch),
// call 2 = samples N .. 2N-1 with (img_l2, patch_l2, w_img2, w_patch2), the mask / box shared.  out6 = the two calls' out3 back to back,
// total = the step's loss (what bcp_mixloss_fwd's prev / total form).  Bit-identical to the two calls.  workspace: bcp_mixloss_pair_workspace_bytes.
static inline size_t pair_acc_bytes(int N, int C) { return (size_t)2 * N * (kLossPartialRows + 1) * (2 * C * 3 + 4) * sizeof(double) + 16; }
extern "C" size_t bcp_mixloss_pair_workspace_bytes(int N, int C) {
  const size_t coef = (size_t)2 * ((size_t)N * 2 * C * 2 + 2) * sizeof(float);
  return ((pair_acc_bytes(N, C) + 15) / 16) * 16 + ((coef + 15) / 16) * 16;
}
static inline float* pair_coef_ptr(void* ws, int N, int C) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ((pair_acc_bytes(N, C) + 15) / 16) * 16);
}

extern "C" int bcp_mixloss_pair_fwd(const float* logits, const uint8_t* img_l, const uint8_t* patch_l, const uint8_t* img_l2, const uint8_t* patch_l2,
                                    const uint8_t* mask_or_null, const int* box6, int N, int D, int H, int W, int C, int flavour,
                                    float w_img, float w_patch, float w_img2, float w_patch2, void* workspace, float* out6, float* total,
                                    void* stream) {
  BCP_REQUIRE(logits && img_l && patch_l && img_l2 && patch_l2 && box6 && workspace && out6 && total, "bcp_mixloss_pair_fwd: null pointer");
  BCP_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && (long long)D * H * W < (1LL << 31), "bcp_mixloss_pair_fwd: bad extents");
  BCP_REQUIRE((flavour == BCP_LOSS_LA && C == 2) || (flavour == BCP_LOSS_ACDC && C == 4),
              "bcp_mixloss_pair_fwd: flavour/C combination unsupported (LA: C=2, ACDC: C=4), got flavour=%d C=%d", flavour, C);
  double* acc = reinterpret_cast<double*>(workspace);
  float* coef = pair_coef_ptr(workspace, N, C);
  const PairArg pa{img_l2, patch_l2, w_img2, w_patch2};
  if (flavour == BCP_LOSS_LA)
    launch_fwd<2, false>(logits, img_l, patch_l, mask_or_null, box6, N, D, H, W, w_img, w_patch, acc, coef, out6, nullptr, total, (hipStream_t)stream, pa);
  else
    launch_fwd<4, true>(logits, img_l, patch_l, mask_or_null, box6, N, D, H, W, w_img, w_patch, acc, coef, out6, nullptr, total, (hipStream_t)stream, pa);
  BCP_CHECK_LAUNCH("bcp_mixloss_pair_fwd");
  return BCP_OK;
}

extern "C" int bcp_mixloss_pair_bwd(const float* logits, const uint8_t* img_l, const uint8_t* patch_l, const uint8_t* img_l2, const uint8_t* patch_l2,
                                    const uint8_t* mask_or_null, const int* box6, int N, int D, int H, int W, int C, int flavour,
                                    const void* workspace, float g_dice, float g_ce, const float* g_dev_or_null, int g_dev_n, float* dlogits,
                                    void* stream) {
  BCP_REQUIRE(logits && img_l && patch_l && img_l2 && patch_l2 && box6 && workspace && dlogits, "bcp_mixloss_pair_bwd: null pointer");
  BCP_REQUIRE(!g_dev_or_null || g_dev_n == 1 || g_dev_n == 2, "bcp_mixloss_pair_bwd: g_dev_n=%d (1: one upstream gradient for both terms, 2: {dice, ce})", g_dev_n);
  BCP_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && (long long)D * H * W < (1LL << 31), "bcp_mixloss_pair_bwd: bad extents");
  BCP_REQUIRE((flavour == BCP_LOSS_LA && C == 2) || (flavour == BCP_LOSS_ACDC && C == 4), "bcp_mixloss_pair_bwd: flavour/C");
  const float* coef = pair_coef_ptr(const_cast<void*>(workspace), N, C);
  const PairArg pa{img_l2, patch_l2, 0.f, 0.f};
  if (flavour == BCP_LOSS_LA)
    launch_bwd<2, false>(logits, img_l, patch_l, mask_or_null, box6, N, D, H, W, coef, dlogits, g_dice, g_ce, g_dev_or_null, g_dev_n, (hipStream_t)stream, pa);
  else
    launch_bwd<4, true>(logits, img_l, patch_l, mask_or_null, box6, N, D, H, W, coef, dlogits, g_dice, g_ce, g_dev_or_null, g_dev_n, (hipStream_t)stream, pa);
  BCP_CHECK_LAUNCH("bcp_mixloss_pair_bwd");
  return BCP_OK;
}

extern "C" int bcp_mixloss_bwd(const float* logits, const uint8_t* img_l, const uint8_t* patch_l, const uint8_t* mask_or_null,
                               const int* box6, int N, int D, int H, int W, int C, int flavour, const void* workspace,
                               float g_dice, float g_ce, const float* g_dev_or_null, int g_dev_n, float* dlogits, void* stream) {
  BCP_REQUIRE(logits && img_l && patch_l && box6 && workspace && dlogits, "bcp_mixloss_bwd: null pointer");
  BCP_REQUIRE(!g_dev_or_null || g_dev_n == 1 || g_dev_n == 2, "bcp_mixloss_bwd: g_dev_n=%d (1: one upstream gradient for both terms, 2: {dice, ce})", g_dev_n);
  BCP_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && (long long)D * H * W < (1LL << 31), "bcp_mixloss_bwd: bad extents");
  BCP_REQUIRE((flavour == BCP_LOSS_LA && C == 2) || (flavour == BCP_LOSS_ACDC && C == 4), "bcp_mixloss_bwd: flavour/C");
  const float* coef = coef_ptr(const_cast<void*>(workspace), N, C);
  if (flavour == BCP_LOSS_LA)
    launch_bwd<2, false>(logits, img_l, patch_l, mask_or_null, box6, N, D, H, W, coef, dlogits, g_dice, g_ce, g_dev_or_null, g_dev_n, (hipStream_t)stream);
  else
    launch_bwd<4, true>(logits, img_l, patch_l, mask_or_null, box6, N, D, H, W, coef, dlogits, g_dice, g_ce, g_dev_or_null, g_dev_n, (hipStream_t)stream);
  BCP_CHECK_LAUNCH("bcp_mixloss_bwd");
  return BCP_OK;
}
