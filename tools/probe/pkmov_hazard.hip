// tools/probe/pkmov_hazard.hip -- micro-reproducer for the round-4 deviation.
// Every lane issues four 16-byte loads, waits with a PARTIAL s_waitcnt vmcnt(1) -- "the first three have landed" -- and reads the third
// load's second dword with ONE instruction directly behind the wait; the instruction is the variant under test.  Launched beside kernels
// that keep the CUs' vector-memory return path busy (tools/probe/pkmov_hazard.py: the library's bf16-pipe convs with LDS-DMA weight stages
// + a copy / GEMM load), a wrong value means the register still held what it held BEFORE the load.
//   variant 0: v_pk_mov_b32 ... op_sel:[1,0]   (what the round-4 build of k_bilinear2x_fwd did: hipcc's choice for a float2 shuffle)
//   variant 1: v_mov_b32                        (what the rebuilt kernel does)
//   variant 2: v_pk_mov_b32 behind s_waitcnt vmcnt(0)
//   variant 3: v_pk_mov_b32 ... op_sel:[0,0]   (low halves)
//   variant 4: v_pk_mov_b32 behind vmcnt(1) + s_nop 7
//   variant 5: v_pk_add_f32 (another packed consumer: reads both halves of the pair)
// out[i] = {value seen by the variant, the same register read by a plain v_mov_b32 a few instructions later, expected}
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probe/libpkmov.so tools/probe/pkmov_hazard.hip
#include <hip/hip_runtime.h>


#define PK_CLOBBERS "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "memory"
// WAIT: the s_waitcnt in front of the consumer; USE: the consumer (must leave the value in v56)
#define PK_BODY(WAIT, USE)                                                                                                  \
  asm volatile("v_mov_b32 v49, 0xc0ffee00\n\t" /* sentinel: what a too-early read of the third load's second dword sees */  \
               "global_load_dwordx4 v[40:43], %2, off\n\t"                                                                  \
               "global_load_dwordx4 v[44:47], %3, off\n\t"                                                                  \
               "global_load_dwordx4 v[48:51], %4, off\n\t"                                                                  \
               "global_load_dwordx4 v[52:55], %5, off\n\t"                                                                  \
               "v_mov_b32 v58, 0\n\t"                                                                                       \
               "v_mov_b32 v59, 0\n\t" WAIT "\n\t" USE "\n\t"                                                                \
               "v_mov_b32 %0, v56\n\t"                                                                                      \
               "s_nop 7\n\t"                                                                                                \
               "v_mov_b32 %1, v49\n\t"                                                                                      \
               "s_waitcnt vmcnt(0)"                                                                                         \
               : "=v"(seen), "=v"(later)                                                                                    \
               : "v"(pa), "v"(pb), "v"(pc), "v"(pd)                                                                         \
               : PK_CLOBBERS)

template <int V>
__global__ __launch_bounds__(256) void k_pk(const float* __restrict__ x, float* __restrict__ out, int rows, int C4) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int c4 = i % C4, r = (i / C4) % (rows - 9);
  const float* pa = x + ((long long)r * C4 + c4) * 4;
  const float* pb = pa + (long long)C4 * 4;            // next row
  const float* pc = pa + (long long)8 * C4 * 4;        // eight rows on (another line, like the upsample's second image row)
  const float* pd = pc + (long long)C4 * 4;
  float seen, later;
  if (V == 0) PK_BODY("s_waitcnt vmcnt(1)", "v_pk_mov_b32 v[56:57], v[48:49], v[40:41] op_sel:[1,0]");
  if (V == 1) PK_BODY("s_waitcnt vmcnt(1)", "v_mov_b32 v56, v49");
  if (V == 2) PK_BODY("s_waitcnt vmcnt(0)", "v_pk_mov_b32 v[56:57], v[48:49], v[40:41] op_sel:[1,0]");
  if (V == 3) PK_BODY("s_waitcnt vmcnt(1)", "v_pk_mov_b32 v[56:57], v[48:49], v[40:41] op_sel:[0,0]\n\tv_mov_b32 v56, v49");   // the pk_mov reads v48 / v40; v56 then from a plain move
  if (V == 4) PK_BODY("s_waitcnt vmcnt(1)\n\ts_nop 7", "v_pk_mov_b32 v[56:57], v[48:49], v[40:41] op_sel:[1,0]");
  if (V == 5) PK_BODY("s_waitcnt vmcnt(1)", "v_pk_add_f32 v[56:57], v[48:49], v[58:59] op_sel:[1,0]");                          // v56 = v49 + 0
  // 6 / 7: the round-4 kernel's exact shape -- destination pair == second source pair, two such moves back to back, behind a FULL wait (6) or
  // the partial one (7)
  if (V == 6) PK_BODY("s_waitcnt vmcnt(0)", "v_pk_mov_b32 v[40:41], v[48:49], v[40:41] op_sel:[1,0]\n\tv_pk_mov_b32 v[42:43], v[50:51], v[42:43] op_sel:[1,0]\n\tv_mov_b32 v56, v40");
  if (V == 7) PK_BODY("s_waitcnt vmcnt(1)", "v_pk_mov_b32 v[40:41], v[48:49], v[40:41] op_sel:[1,0]\n\tv_pk_mov_b32 v[42:43], v[50:51], v[42:43] op_sel:[1,0]\n\tv_mov_b32 v56, v40");
  // 8 / 9 / 10: the instruction tools/probe/isa_bisect.py found guilty in the round-4 upsample kernel -- v_pk_mul_f32 with the halves of its
  // SECOND source crossed (low result = S0.lo * S1.HI, high = S0.hi * S1.LO).  S1 = (2.0, 1.0) written by two v_mov_b32, so the low result
  // must be the loaded dword v48 itself; 9: S0 written by the VALU too (v48 copied to v60 first); 10: the plain form on pre-swapped constants
  if (V == 8) PK_BODY("v_mov_b32 v58, 2.0\n\tv_mov_b32 v59, 1.0\n\ts_waitcnt vmcnt(0)", "v_pk_mul_f32 v[56:57], v[48:49], v[58:59] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32 v49, v48");
  if (V == 9) PK_BODY("v_mov_b32 v58, 2.0\n\tv_mov_b32 v59, 1.0\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32 v52, v48\n\tv_mov_b32 v53, v49", "v_pk_mul_f32 v[56:57], v[52:53], v[58:59] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32 v49, v48");
  if (V == 10) PK_BODY("v_mov_b32 v58, 1.0\n\tv_mov_b32 v59, 2.0\n\ts_waitcnt vmcnt(0)", "v_pk_mul_f32 v[56:57], v[48:49], v[58:59]\n\tv_mov_b32 v49, v48");
  // 11: the same crossing on a packed ADD (S1 = (2.0, 0.0): low = v48 + 0.0);  12: on the FIRST source instead (low = S0.HI * S1.lo, the form the
  // bisection found innocent): S0 = (1.0, loaded) -> low = v48 * 1.0 with S0 = v[58:59] = (2.0, v48)
  if (V == 11) PK_BODY("v_mov_b32 v58, 2.0\n\tv_mov_b32 v59, 0\n\ts_waitcnt vmcnt(0)", "v_pk_add_f32 v[56:57], v[48:49], v[58:59] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32 v49, v48");
  if (V == 12) PK_BODY("v_mov_b32 v58, 2.0\n\tv_mov_b32 v60, 1.0\n\tv_mov_b32 v61, 3.0\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32 v59, v48", "v_pk_mul_f32 v[56:57], v[58:59], v[60:61] op_sel:[1,0] op_sel_hi:[0,1]\n\tv_mov_b32 v49, v48");
  // 13 .. 17: which operand selections of a packed multiply / fma are affected (S0 = (x, x), constants chosen so that the low result must be x)
  if (V == 13) PK_BODY("v_mov_b32 v58, 2.0\n\tv_mov_b32 v59, 1.0\n\tv_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\ts_waitcnt vmcnt(0)", "v_pk_fma_f32 v[56:57], v[48:49], v[58:59], v[60:61] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\tv_mov_b32 v49, v48");   // fma, second source crossed
  if (V == 14) PK_BODY("v_mov_b32 v58, 2.0\n\tv_mov_b32 v59, 1.0\n\ts_waitcnt vmcnt(0)", "v_pk_mul_f32 v[56:57], v[48:49], v[58:59] op_sel:[0,1]\n\tv_mov_b32 v49, v48");                  // S1.hi broadcast (both halves take S1.hi)
  if (V == 15) PK_BODY("v_mov_b32 v58, 1.0\n\tv_mov_b32 v59, 2.0\n\ts_waitcnt vmcnt(0)", "v_pk_mul_f32 v[56:57], v[48:49], v[58:59] op_sel_hi:[1,0]\n\tv_mov_b32 v49, v48");               // S1.lo broadcast
  if (V == 16) PK_BODY("v_mov_b32 v58, 2.0\n\tv_mov_b32 v59, 1.0\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32 v52, 3.0\n\tv_mov_b32 v53, v48", "v_pk_mul_f32 v[56:57], v[52:53], v[58:59] op_sel:[1,1] op_sel_hi:[0,0]\n\tv_mov_b32 v49, v48");   // both sources crossed
  if (V == 17) PK_BODY("v_mov_b32 v58, 1.0\n\tv_mov_b32 v59, 1.0\n\ts_waitcnt vmcnt(0)", "v_pk_mul_f32 v[56:57], v[48:49], v[58:59] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32 v49, v48");  // the guilty form with S1 = (1.0, 1.0): is it S1.hi read as 0, or the product?
  out[(long long)i * 3 + 0] = seen;
  out[(long long)i * 3 + 1] = later;
  out[(long long)i * 3 + 2] = (V >= 8) ? pc[0] : pc[1];
}

// ---- stand-alone NEIGHBOURS (no library): which instruction class on the other stream makes the guilty multiply fail?
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float nb_f32x4;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 nb_bf16x8;
typedef __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16 nb_f16x8;
template <int KIND>
__global__ __launch_bounds__(256) void k_nb(float* __restrict__ sink, const float* __restrict__ src, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int t = threadIdx.x;
  nb_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float v = (float)t * 1e-3f, w = 1.0001f;
  if (KIND == 0) {                                   // dense bf16 MFMA, as the bf16-pipe convs issue
    nb_bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(v + k); b[k] = (__bf16)(w + k); }
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  } else if (KIND == 1) {                            // fp16 MFMA (the two-plane instances)
    nb_f16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(v + k); b[k] = (_Float16)(w + k); }
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  } else if (KIND == 2) {                            // fp32 MFMA
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v, w, acc, 0, 0, 0);
  } else if (KIND == 3) {                            // LDS-DMA: global -> LDS, 16 bytes per lane
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    for (int i = 0; i < iters; ++i) {
      __builtin_amdgcn_global_load_lds(src + ((size_t)(blockIdx.x * 256 + t) * 4 + (size_t)(i & 63) * 262144) % (1 << 22), (lds_ptr_t)(lds + (t >> 6) * 256), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    acc[0] = lds[t];
  } else if (KIND == 4) {                            // plain VALU fma chain
    for (int i = 0; i < iters * 8; ++i) v = __builtin_fmaf(v, w, 0.5f);
    acc[0] = v;
  } else {                                           // LDS traffic (ds_write / ds_read)
    for (int i = 0; i < iters; ++i) { lds[(t * 4 + i) & 4095] = v; __syncthreads(); v += lds[(t * 7 + i) & 4095]; }
    acc[0] = v;
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[t] = acc[0];
}
extern "C" int nb_run(int kind, float* sink, const float* src, int iters, int blocks, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (kind) {
    case 0: hipLaunchKernelGGL(k_nb<0>, dim3(blocks), dim3(256), 0, s, sink, src, iters); break;
    case 1: hipLaunchKernelGGL(k_nb<1>, dim3(blocks), dim3(256), 0, s, sink, src, iters); break;
    case 2: hipLaunchKernelGGL(k_nb<2>, dim3(blocks), dim3(256), 0, s, sink, src, iters); break;
    case 3: hipLaunchKernelGGL(k_nb<3>, dim3(blocks), dim3(256), 0, s, sink, src, iters); break;
    case 4: hipLaunchKernelGGL(k_nb<4>, dim3(blocks), dim3(256), 0, s, sink, src, iters); break;
    default: hipLaunchKernelGGL(k_nb<5>, dim3(blocks), dim3(256), 0, s, sink, src, iters); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int pk_run(int variant, const float* x, float* out, int rows, int C4, void* stream) {
  const int total = (rows - 9) * C4;
  dim3 grid(total / 256), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (variant) {
    case 0: hipLaunchKernelGGL(k_pk<0>, grid, block, 0, s, x, out, rows, C4); break;
    case 1: hipLaunchKernelGGL(k_pk<1>, grid, block, 0, s, x, out, rows, C4); break;
    case 2: hipLaunchKernelGGL(k_pk<2>, grid, block, 0, s, x, out, rows, C4); break;
    case 3: hipLaunchKernelGGL(k_pk<3>, grid, block, 0, s, x, out, rows, C4); break;
    case 4: hipLaunchKernelGGL(k_pk<4>, grid, block, 0, s, x, out, rows, C4); break;
    case 5: hipLaunchKernelGGL(k_pk<5>, grid, block, 0, s, x, out, rows, C4); break;
    case 6: hipLaunchKernelGGL(k_pk<6>, grid, block, 0, s, x, out, rows, C4); break;
    case 7: hipLaunchKernelGGL(k_pk<7>, grid, block, 0, s, x, out, rows, C4); break;
    case 8: hipLaunchKernelGGL(k_pk<8>, grid, block, 0, s, x, out, rows, C4); break;
    case 9: hipLaunchKernelGGL(k_pk<9>, grid, block, 0, s, x, out, rows, C4); break;
    case 10: hipLaunchKernelGGL(k_pk<10>, grid, block, 0, s, x, out, rows, C4); break;
    case 11: hipLaunchKernelGGL(k_pk<11>, grid, block, 0, s, x, out, rows, C4); break;
    case 12: hipLaunchKernelGGL(k_pk<12>, grid, block, 0, s, x, out, rows, C4); break;
    case 13: hipLaunchKernelGGL(k_pk<13>, grid, block, 0, s, x, out, rows, C4); break;
    case 14: hipLaunchKernelGGL(k_pk<14>, grid, block, 0, s, x, out, rows, C4); break;
    case 15: hipLaunchKernelGGL(k_pk<15>, grid, block, 0, s, x, out, rows, C4); break;
    case 16: hipLaunchKernelGGL(k_pk<16>, grid, block, 0, s, x, out, rows, C4); break;
    case 17: hipLaunchKernelGGL(k_pk<17>, grid, block, 0, s, x, out, rows, C4); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
