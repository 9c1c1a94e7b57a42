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


#define PK_CLOBBERS "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "memory"
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
  out[(long long)i * 3 + 0] = seen;
  out[(long long)i * 3 + 1] = later;
  out[(long long)i * 3 + 2] = pc[1];
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
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
