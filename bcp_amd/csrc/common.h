// bcp_amd/csrc/common.h -- shared device/host helpers for libbcp_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#define BCP_OK 0
#define BCP_EINVAL (-1)   // bad argument (shape / alignment / divisibility)
#define BCP_ELAUNCH (-2)  // HIP reported a launch error
#define BCP_EUNSUP (-3)   // configuration not supported by this build

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace bcp {

void set_error(const char* fmt, ...);

#define BCP_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      ::bcp::set_error(__VA_ARGS__);    \
      return BCP_EINVAL;                \
    }                                   \
  } while (0)

#define BCP_CHECK_LAUNCH(name)                                          \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) {                                            \
      ::bcp::set_error("%s: %s", name, hipGetErrorString(e__));         \
      return BCP_ELAUNCH;                                               \
    }                                                                   \
  } while (0)

// Process-wide tuning / test switches.  Set ONLY through bcp_set_option() (include/bcp_hip.h): the launch paths never read
// the environment (round 1 called getenv() up to 15 times per launch).  0 / empty = automatic choice.
struct Options {
  int conv3_p = 0;          // cap on persistent workgroups of the resident / pipeline convs (tests: force multi-tile loops)
  int splitk = 0;           // streaming conv: split-K factor 1..4
  long long conv3_sk_elems = 1LL << 20;   // largest conv output (elements) that may run split-K over its cin chunks: the slab workspace is 8 x that (bcp_conv3_fwd_workspace_bytes)
  int conv3_b6_cin16max = 32;   // 2-D layers with 16 output channels on the bf16 pipe: widest input (measurement switch)
  int conv3_b6_pipe = 1;        // ... and of those the 3-D ones as the LDS-DMA software pipeline k_c3p (0: k_c3h, register-staged weights)
  int conv3_b6_w22 = 1;         // 64-voxel x 64-channel staged tiles: waves arranged 2 x 2 (k_c3h) instead of 1 x 4 (k_c3b)
  int conv3_b6_cfg2d = 1;       // 2-D 32-channel slabs: 1 = direct-weight 16x16 tiles from 64 K pixels, 2 = always, 0 = staged 8x16 tiles
  int conv3_b6_flat_sk = 0; // flat bf16-pipe tiles: split-K factor 1..8 (0: the launcher's rule)
  int res_pcu = 0;          // resident conv: persistent workgroups per CU 1..4
  int res_nt = 0;           // resident conv: widest channel slab (1, 2, 4)
  long long res_tile2d_vox = 10000;   // 2-D resident conv: 16x16 tiles from this many pixels per launch on
  int conv3_cfg[4] = {0, 0, 0, 0};    // streaming conv: TD, TH, TW, NT
  int wgrad_nt = 0;
  long long wgrad_tile[5] = {0, 0, 0, 0, 1LL << 60};   // TD, TH, TW, min voxels, max voxels
  int tn_groups = 0;        // k2s2 / 1x1 weight-gradient GEMM: cap on voxel groups (tests: force multi-chunk groups)
  int pack_sections = 7;    // MEASUREMENT ONLY (bcp_conv3_pack_many): which sections of the weight packs are written -- 1 fp32, 2 three bf16 planes, 4 two fp16 planes; anything but 7 is only valid when no launch reads the others
  int cc_tile = 0;          // largest-CC tile flavour
  int cc_select_blocks = 0; // largest-CC: workgroups per sample of the root-selection pass (0 = 256; measurement switch)
  int conv3_b6 = 1;         // fp32 conv on the bf16 matrix pipe (three-piece operands, conv3b.hip): 0 off, 1 where measured faster, 2 wherever valid
  int conv3_b6_levels = 15;  // automatic choice (conv3_b6 = 1): bit 0 = 32-channel slabs (256-voxel tiles), bit 1 = 64-channel slabs, bit 2 = the 16 -> 16 layers (persistent k_c3d with cross-tile halo prefetch: 176 vs 243-258 us alone, step 7.00 vs 7.18 ms; one tile per workgroup it was 209-228 us and no step gain), bit 3 = the 2-D instances (ACDC step 5.18 -> 4.24 ms together with the weight gradients).  LA step, interleaved A/B (ms per step): off 8.87, 32-channel level 8.32, + 64-channel level 8.05 -- the latter although ALONE that kernel is slower than the exclusive pipeline kernel it replaces (66-71 vs 61 us): two workgroups per CU leave room for the other stream
  int conv3_b6_minvox = 256;     // automatic choice: smallest launch (voxels, batch included) that goes to the bf16-pipe kernels
  int conv3_b6_flat = 1;    // deep levels (64-channel slabs, < 16 K voxels): flat 64-voxel tiles with per-lane validity masks (k_c3f) instead of bricks: 128 channels @14x14x10 35 vs 41 us (fp32 kernel 52), 256 @7x7x5 31 vs 33 (fp32); LA step 6.87 vs 6.96 ms.  2 / 3: force 32- / 16-channel slabs (measurements)
  int conv3_b6_direct = 1;  // bf16-pipe forward: weight fragments straight from global memory (k_c3d, no stage barriers) instead of an LDS stage (k_c3b): 1 = for the 256-voxel x 32-channel tiles (78-82 vs 84-89 us alone, 7.32 vs 7.34 ms per step), 2 = everywhere (measurements)
  int wgrad_b6 = 1;         // weight gradient on the bf16 matrix pipe (conv3bw.hip): 0 off, 1 where measured faster, 2 wherever valid.  LA step (interleaved A/B): off 7.82 ms, 32/64-channel levels 7.52, + 128-channel level 7.38
  int wgrad_b6_minvox = 256;     // (7x7x5 level included: 39 vs 57 us alone, 7.30 vs 7.36 ms per step)
  int norm_slabs = 1;       // deep levels (<= 4096 rows per group): the conv leaves its raw split-K slabs and the norm's row-major statistics pass sums them on its way in (bcp_norm_fwd_slabs / _bwd_slabs): no k_b6_sum_slabs launch (27 per LA step).  0: slab-sum launches (round 3)
  int fuse_bwd_stats = 1;   // dgrad epilogue of the bf16-pipe kernels accumulates the consumer norm layer's backward statistics (bcp_conv3_dgrad_bwdstats): no k_col_partial<1> pass over (y, da) for conv -> conv edges
  int conv3_xcd = 11;       // (round 4: 1 -> 11.  With two fp16 planes the kernels issue half the MFMAs and the fabric matters more: the persistent 16-channel kernel walking contiguous eighths (bit 2) is now FASTER alone (110 vs 117 us, fetch halved), one weight stream per XCD at the 128-channel level (bit 8) still slower alone (23.6 vs 22.0 us) but the LA step is 5.31 vs 5.34 ms with both, three interleaved pairs, pancreas 4.89 vs 4.92: tools/sessions/r04_s15.sh) // bf16-pipe kernels: XCD-aware workgroup -> tile order (each XCD walks a contiguous eighth of the tile list: halo overlap hits its own L2; the flat deep-level kernel deals WEIGHT STREAMS to XCDs).  Bits for measurements: 2 = also the persistent 16-channel kernel (slower), 8 = k_c3q deals weight streams to XCDs also when there are only 8 of them (128-channel level: minimal fabric traffic, slower alone, step unchanged), 16 = k_c3q in tile order
  int conv3_f16 = 1;        // round 4: bf16-pipe kernels that have a two-plane fp16 instance (k_c3d) use it when the launch carries the input tensor's |max| (per-tensor power-of-two pre-scales, conv3_defs.h): three MFMAs per K block instead of six.  0: three bf16 planes everywhere
  int wgrad_b6_slots = 0;   // weight gradient on the matrix pipe: workgroups per launch the tile groups are cut for (0 = 512: two per CU); measurement switch
  int mix_c1 = 1;           // round 5: single-channel copy-paste mix as k_mix_box_c1 (one multiply-high per float4 instead of six divisions; b read inside the box only).  0: the general kernel (measurement switch)
  int wgrad_b6_deep = 1;    // round 6: deep-level weight gradients (< 16 K voxels) with few tile groups; ONE group -> the kernel writes dW itself (k_w6 DIR), no partial slabs, no reduce launch.  0: rounds 2-5 (512 slots of [T][16][32] slabs)
  int wgrad_b6_deep_nt = 0; // ... n-tiles per workgroup there: 0 = 1 (16-channel slabs: twice the channel blocks), 1, 2 (measurement switch)
  int wgrad_b6_deep_slots = 0;  // ... workgroups the tile groups are cut for (0 = 256)
  int wgrad_b6_deep_tile = 1;   // ... 1 = 128-voxel tiles (4x8x4: half the tiles to walk; 7x7x5 x 256 direct 23.7 vs 32.0 us), 0 = 64-voxel tiles (2x8x4, rounds 2-5)
  int k2_stats = 2;         // round 6: k2s2 / transposed conv forwards leave the norm statistics of their output (k_gemm_nn<.., STATS>): 2 = where the output is >= 2^24 elements (the top level: the pass saved is 19-23 us, the epilogue costs ~11 us at every level; LA 786.9 -> 789.6, pancreas 850.0 -> 853.8 volumes/s, interleaved A/B), 1 = wherever the shape allows (788.2 / 848.2), 0 = the norm's own statistics pass
  int up_recompute = 0;     // (the host takes it only in forwards without a backward pass unless VNet.UP_RECOMPUTE_GRAD: teacher-only also measured no faster, gpurun_out/r06_s50) round 6, measured and NOT adopted (kept with its kernel / network checks): the transposed conv + its norm with the conv output recomputed instead of stored (bcp_up_fwd_norm / bcp_up_norm_bwd): 2 = where that output is >= 2^24 elements (the top V-Net level), 1 = wherever the shape allows, 0 = off.  640 MB less per LA step and SLOWER: 788.6 (off) -> 768.1 volumes/s (2), pancreas 849.5 -> 837.1 (gpurun_out/r06_s21) -- the GEMM passes move their bytes at ~3.3 TB/s where the streaming norm passes they replace run at 5+
  int k2_bwd_stats = 0;     // round 6, measured and NOT adopted (kept for the record and its kernel check): the k2s2 / transposed-conv DGRADS leave the backward statistics of the norm layer in front of them (k_gemm_nn<.., 2>, bcp_down_dgrad_bwdstats / bcp_up_dgrad_bwdstats).  2 = where the output is >= 2^22 elements, 1 = wherever the shape allows, 0 = the norm's own pass (k_col_partial<1>).  LA 788.2 (off) / 786.9 (1) / 784.2 (2), pancreas 853.0 / 852.3 / 849.4 volumes/s (gpurun_out/r06_s16): the epilogue re-reads y, so all it saves is the da read, and it costs the GEMM its occupancy
  int norm_fuse_fin = 1;    // round 6: norm layers whose statistics pass leaves <= 128 partial rows per group (the deep levels): the apply pass finalises the statistics itself (k_norm_apply_fin / k_norm_bwd_apply_fin), no finalize launch.  0: finalize launches everywhere
  int norm_fin_rows = 128;   // ... partial rows per group the slab-summing statistics pass leaves in front of a fused apply pass (128: as without it)
  int gemm_pipe = 1;        // round 6: k_gemm_nn instances that walk several row blocks per workgroup (the statistics / recompute epilogues) request the next block's operands under the current block's MFMAs and epilogue.  0: block after block
  int gemm_stat_r = 0;      // ... measurement switch: row blocks per workgroup of those instances (0 = stat_plan's choice: <= ~1024 partial rows per group, >= 512 workgroups)
  int gemm_walk = 0;        // ... measurement switch: plain k_gemm_nn launches of >= 2 W workgroups as ~W walking workgroups (k_gemm_nn<.., 7>); 0 = one row block per workgroup
  int cc_border_dedupe = 1; // round 6: k_cc_border lanes leave a (root, root) pair that the previous lane also holds at the same neighbour offset to that lane.  0: every lane joins every pair it sees
  int cc_count_tile = 0;    // round 6, measured and NOT adopted as the default (p = 0.5 noise: chain 166.8 -> 172.1 us alone, p = 0.1: 137.6 -> 112.1; the step inside the noise, gpurun_out/r06_s59 -- the chain is bound by the find walks, not by the atomics): ... one workgroup per tile, sizes of tile-local roots that share a global root added up in an LDS table first (k_cc_count_select_tile): one global atomic per (tile, global root).  0: one per tile-local root
  int cc_fuse_select = 1;   // round 6: the largest-CC chain's size count also max-reduces the (size, root) keys (k_cc_count_select): no k_cc_select launch on the teacher's tail.  0: two launches
  int norm_apply_cap = 2048;   // round 6 (measurement switches): workgroups per apply-pass launch at most ...
  int norm_apply_vec = 4;      // ... and float4 per thread the grid is sized for (4 = one unrolled trip; 1 = every thread one float4, no loop)
  int wgrad_reduce_flat = 1;   // round 6: many-group weight-gradient slab sums read slab-contiguous float4 (k_wgrad_reduce_flat).  0: k_wgrad_reduce_deep
  int norm_own = 0;         // round 6, measured and NOT adopted (kept with its kernel checks): norm layers with a few hundred rows per group (LA 7x7x5, pancreas 6^3) in ONE launch (k_norm_own_fwd / _bwd: a 1024-thread workgroup owns an 8..32-channel chunk of all rows).  LA 800.2 (on) vs 800.7 (off), 32-channel chunks 804.4 vs 810.4 volumes/s (gpurun_out/r06_s35, s36): the statistics pass + fused apply pass it replaces spread over 4x the workgroups, and 1024-thread workgroups wait for a whole CU beside the other stream's convs
  int whatif = 0;           // MEASUREMENT ONLY (wrong results): bit 0 = no finalize launches of the norm layers, bit 1 = no finalize and no apply launches where rows per group <= 4096, bit 2 = no largest-CC launches, bit 3 = no conv / k2 weight-gradient launches, bit 4 = no forward / backward apply pass at >= 100000 rows per group, bit 5 = no backward statistics pass there (prices a change before it is built)
  int wgrad_b6_levels = 15; // bit 3: 2-D; bit 2: also the 16-channel slabs (one n-tile per wave): 187 vs 270 us alone, 7.28 vs 7.36 ms per step
};
Options& options();

// ---- pseudo-label arithmetic shared by k_plabel_bin / k_plabel_argmax4 (elementwise.hip) and the largest-CC kernels that label straight
// from the logits (cc.hip, bcp_plabel_cc_largest): ONE definition so both paths produce the same bits
// LA / pancreas: softmax over 2 channels, (p1 >= thres)            (LA_BCP_train.py:57-60)
__device__ __forceinline__ unsigned char plabel_bin_of(float x0, float x1, float thres) {
  const float m = fmaxf(x0, x1);
  const float e0 = expf(x0 - m), e1 = expf(x1 - m);
  const float p1 = e1 / (e0 + e1);
  return (p1 >= thres) ? 1 : 0;
}
// ACDC: softmax over 4 channels then argmax, first maximum wins       (ACDC_BCP_train.py:112-114)
__device__ __forceinline__ unsigned char plabel_argmax4_of(float x0, float x1, float x2, float x3) {
  const float m = fmaxf(fmaxf(x0, x1), fmaxf(x2, x3));
  const float e0 = expf(x0 - m), e1 = expf(x1 - m), e2 = expf(x2 - m), e3 = expf(x3 - m);
  const float s = e0 + e1 + e2 + e3;
  const float p0 = e0 / s, p1 = e1 / s, p2 = e2 / s, p3 = e3 / s;
  int best = 0;
  float pb = p0;
  if (p1 > pb) { pb = p1; best = 1; }
  if (p2 > pb) { pb = p2; best = 2; }
  if (p3 > pb) { pb = p3; best = 3; }
  return (unsigned char)best;
}


static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: a wave with global loads in
// flight (a kernel prefetching its next halo or weight stage under the current one) would park at the barrier for a full HBM round trip and
// hold every other wave of the workgroup with it.  Here the wave waits for its own LDS operations (lgkmcnt) and joins the
// barrier; registers being filled by outstanding global loads are private and need no fence.
#ifndef BCP_LDS_BARRIER
#define BCP_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// ---- LDS-DMA (gfx950 global_load_lds_dwordx4): 16 bytes per lane from a PER-LANE global address to LDS at a WAVE-UNIFORM base +
// 16 * lane.  The data is ordered for a ds_read by the issuing wave's counted vmcnt followed by a barrier the reader has passed:
// BCP_VM_LDS_BARRIER(N) = "at most N younger vector-memory operations may still be in flight" + the LDS barrier above.  N counts
// LDS-DMAs ONLY: the counter retires LDS-DMAs in order among themselves, but an ordinary load issued later can retire earlier (k_c3p).
#ifndef BCP_GLDS16
typedef __attribute__((address_space(3))) void* bcp_lds_ptr_t;
#define BCP_GLDS16(gsrc, lds_wave_base) __builtin_amdgcn_global_load_lds((gsrc), (bcp_lds_ptr_t)(lds_wave_base), 16, 0, 0)
#define BCP_VM_LDS_BARRIER(N) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory")
#endif

// ---- timing-only pause (s_sleep n = ~64 n clocks); the host simulator defines it away
#ifndef BCP_S_SLEEP
#define BCP_S_SLEEP(n) __builtin_amdgcn_s_sleep(n)
#endif

// ---- wavefront (64-lane) reductions; every lane of the wave must call.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// block-wide sum of NV doubles per thread (blockDim.x == 256, 4 waves); result valid in thread 0.
template <int NV>
__device__ __forceinline__ void block_sum_256(double (&v)[NV], double* lds /* [4*NV] */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) lds[wid * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = lds[i] + lds[NV + i] + lds[2 * NV + i] + lds[3 * NV + i];
  }
}

// ---- per-tensor |max| slots (round 4: the power-of-two pre-scale of the two-plane fp16 conv instances, conv3_defs.h).  A tensor's |max| lives
// in kAmaxSlots floats, one per 128-byte line (kAmaxStride apart): the pass that writes the tensor max-reduces every workgroup's maximum
// into slot (workgroup id % kAmaxSlots), the kernel that reads the tensor takes the maximum over the slots.  ONE slot serialised ~2000
// atomics per launch on one L2 address (k_norm_apply 12.9 -> 24.3 us at the 32-channel level); spread over 32 lines they are noise.
constexpr int kAmaxSlots = 32, kAmaxStride = 32, kAmaxFloats = kAmaxSlots * kAmaxStride;
// the kernel in front of the producing pass clears the slots (threads 0 .. 31 of ONE workgroup)
__device__ __forceinline__ void amax_clear(float* __restrict__ slots) {
  if (threadIdx.x < kAmaxSlots) slots[threadIdx.x * kAmaxStride] = 0.f;
}
// block maximum of a non-negative per-thread value -> at most ONE atomic per block, and only when it would raise the block's slot (read
// past the L1: a stale read only costs a redundant atomic).  fmaxf drops NaNs, so a NaN in the tensor is forwarded explicitly: the
// consumer then sees NaN, takes scale 1, and the NaN reaches its output as it would on the bf16 / fp32 paths.  Non-negative floats
// order like their bit patterns.
__device__ __forceinline__ void block_amax_publish(float m, float* __restrict__ slots) {
  __shared__ float amax_red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(m, o); m = (t > m || t != t) ? t : m; }
  if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) { const float t = amax_red[k]; m = (t > m || t != t) ? t : m; }
    if (m != m) m = __uint_as_float(0x7fc00000u);                              // canonical positive NaN: above every number as an unsigned
    float* slot = slots + ((blockIdx.x + blockIdx.y * gridDim.x) % kAmaxSlots) * kAmaxStride;
    const float cur = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!(m <= cur)) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(m));
  }
}
// the tensor's |max| as its consumer sees it: maximum over the slots (every wave reads them itself: one load per lane, five shuffles)
__device__ __forceinline__ float amax_read(const float* __restrict__ slots) {
  float m = slots[(threadIdx.x & (kAmaxSlots - 1)) * kAmaxStride];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) { const float t = __shfl_xor(m, o); m = (t > m || t != t) ? t : m; }
  return m;
}

// ---- counter-hash Bernoulli keep bit of element i under a 64-bit seed (bcp_bernoulli / bcp_bernoulli_dev write exactly these bits as a
// mask; round 4: the norm kernels can also evaluate them in place -- NormEpilogue::mask_seed -- so that an elementwise Dropout needs no
// mask tensor and no launch of its own)
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool bern_keep(long long i, unsigned seed_lo, unsigned seed_hi, float p_keep) {
  const unsigned h = mix32((unsigned)i ^ mix32(seed_lo + 0x9e3779b9u * (unsigned)(i >> 32)) ^ seed_hi);
  return (float)(h >> 8) * (1.0f / 16777216.0f) < p_keep;
}
// keep bits of elements e .. e+3 as the four bytes a mask load would return
__device__ __forceinline__ uchar4 bern_keep4(long long e, unsigned seed_lo, unsigned seed_hi, float p_keep) {
  return make_uchar4(bern_keep(e, seed_lo, seed_hi, p_keep) ? 1 : 0, bern_keep(e + 1, seed_lo, seed_hi, p_keep) ? 1 : 0,
                     bern_keep(e + 2, seed_lo, seed_hi, p_keep) ? 1 : 0, bern_keep(e + 3, seed_lo, seed_hi, p_keep) ? 1 : 0);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// activation codes shared with the host side
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2 };
__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == ACT_RELU) return z > 0.f ? z : 0.f;
  if (act == ACT_LRELU) return z > 0.f ? z : 0.01f * z;
  return z;
}
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == ACT_LRELU) return z > 0.f ? 1.f : 0.01f;
  return 1.f;
}

}  // namespace bcp
