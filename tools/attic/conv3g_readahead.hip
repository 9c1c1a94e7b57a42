// NOT part of the product: the fragment read-ahead variant of k_c3b (bcp_amd/csrc/conv3b.hip), measured and dropped in round 2.
// 64-channel level alone: 54.8 vs 55.8 us at batch 2, 107 vs 102 us at batch 4 -- the LDS pipe of these kernels is THROUGHPUT-bound
// (15 ds_read_b128 per wave and stage x 8 waves per CU = 960 LDS cycles against 768 cycles of MFMA), so hiding the read latency
// buys nothing; what would help is fewer reads per MFMA (2 x 2 wave split of the 64 x 64 tile: 12 instead of 15).
// Kept for the record (drops into conv3b.hip after k_c3b; launcher hook: swap kfn, 4 weight buffers in the LDS size).

// ------------------------------------------------------------------------------------------------
// k_c3b with the NEXT stage's MFMA fragments read ahead (one tap pair per stage).  In k_c3b a wave's stage is serial -- LDS fragment
// reads, wait, 6*MT*NT MFMAs, weight stash, barrier -- and at the mid / deep levels a workgroup's life is a chain of such stages
// (64 channels: 4 chunks x 14 stages), so whatever a stage does not overlap is paid 56 times; ablation of the 64-channel kernel (59 us
// alone): LDS fragment reads 12.8 us, weight fetch + stash 12.5 us.  Here the weight stages live in a RING OF FOUR LDS buffers: while
// stage g's MFMAs run, the fragments of stage g + 1 (buffer (g + 1) & 3, stashed during stage g - 1, published by the barrier that
// ended it) are already on their way into a second register set, and stage g + 2's weights go from registers into buffer (g + 2) & 3,
// which no wave can still be reading (its last reader was stage g - 2's read-ahead, two barriers ago).  The halo planes change at a
// chunk boundary, so a chunk's first stage reads its voxel fragments after the re-stash; every other stage has them early.
// No conditional loads anywhere (see k_c3b): the read-ahead past a chunk's last stage re-reads that stage's voxel rows.
// ------------------------------------------------------------------------------------------------
template <int KD, int TD, int TH, int TW, int NT>
__global__ __launch_bounds__(256) void k_c3g(const float* __restrict__ X, const float* __restrict__ Wp, const float* __restrict__ bias,
                                             float* __restrict__ Y, ConvDims cd, int accumulate, StatsArg st) {
  using TL = Tile<KD, TD, TH, TW>;
  constexpr int MT = TL::MT, T = TL::T, TP = (T + 1) / 2, CT = NT * 16;
  constexpr int S = TP;                                        // one tap pair per stage
  static_assert(S % 2 == 0 && S >= 6, "k_c3g: even number of tap pairs (3x3x3)");
  constexpr int XPLANE = TL::HV * XSB;
  constexpr int WPLANE = CT * 32;
  constexpr int WSTAGE = 3 * WPLANE;
  constexpr int NW4 = (12 * CT + 255) / 256;
  using HF = HaloFetch<TL>;

  HIP_DYNAMIC_SHARED(float4, smem4)
  unsigned short* Xb = reinterpret_cast<unsigned short*>(smem4);   // [3][HV][XSB]
  unsigned short* Wb = Xb + 3 * XPLANE;                            // [4][3][CT][32]
  double* Ss = reinterpret_cast<double*>(Wb + 4 * WSTAGE);         // [4][CT][2] statistics scratch

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  int n, d0, h0, w0;
  tile_origin(cd, blockIdx.x, TD, TH, TW, n, d0, h0, w0);
  const int cout0 = blockIdx.y * CT;

  int voff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) voff[mt] = TL::voff((wave * MT + mt) * 16 + b6_row<TW>(li)) * XSB + (lg & 1) * 8;
  const int woff = li * 32 + ((lg ^ ((li & 8) ? 2 : 0)) * 8);
  HF hf;
  hf.init(cd, reinterpret_cast<float*>(smem4));

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = cd.Cin16 >> 4;
  const int c_begin = (int)((long long)nchunks * blockIdx.z / gridDim.z), c_end = (int)((long long)nchunks * (blockIdx.z + 1) / gridDim.z);
  Y += (long long)blockIdx.z * cd.N * cd.D * cd.H * cd.W * cd.Cout;

  const unsigned short* Wb16 = reinterpret_cast<const unsigned short*>(Wp + (long long)T * cd.Cin16 * cd.Cout16);
  auto wfetch_at = [&](int cc, int sg, float4 (&wpre)[NW4]) __attribute__((always_inline)) {     // stage sg (may run past S) of chunk cc
    if (sg >= S) { sg -= S; ++cc; }
    if (cc >= c_end) { cc = c_end - 1; sg = S - 1; }
#pragma unroll
    for (int u = 0; u < NW4; ++u) {
      const int q = (threadIdx.x + u * 256) % (12 * CT);
      const int kq = q & 3, co = (q >> 2) % CT, sp = q / (4 * CT);
      wpre[u] = *reinterpret_cast<const float4*>(Wb16 + ((((long long)cc * TP + sg) * 3 + sp) * cd.Cout16 + cout0 + co) * 32 + kq * 8);
    }
  };
  auto wstash = [&](unsigned short* Wbuf, const float4 (&wpre)[NW4]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NW4; ++u) {
      const int q = threadIdx.x + u * 256;
      const int kq = q & 3, co = (q >> 2) % CT, sp = q / (4 * CT);
      // (the value goes through a select, as in k_c3b: a plain conditional store of wpre[u] leaves the register sets in scratch
      //  memory with this compiler -- 112 bytes of private segment, scratch loads in front of every stash)
      const float4 v = (q >= 0) ? wpre[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < 12 * CT) *reinterpret_cast<float4*>(Wbuf + sp * WPLANE + co * 32 + ((kq ^ ((co & 8) ? 2 : 0)) * 8)) = v;
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

  struct Frag { bf16x8 a[MT][3], b[NT][3]; };
  auto read_b = [&](Frag& F, int g) __attribute__((always_inline)) {          // weight fragments of global stage g
    const unsigned short* Wc = Wb + (g & 3) * WSTAGE + woff;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) F.b[nt][s] = *reinterpret_cast<const bf16x8*>(Wc + s * WPLANE + nt * 16 * 32);
  };
  auto read_a = [&](Frag& F, int sg) __attribute__((always_inline)) {         // voxel fragments of tap pair sg (clamped: see above)
    const int tp = sg < S ? sg : S - 1;
    const int t0 = 2 * tp, t1 = 2 * tp + 1 < T ? 2 * tp + 1 : T - 1;
    const int tA = ((t0 / 9) * TL::HH + (t0 / 3) % 3) * TL::HW + t0 % 3, tB = ((t1 / 9) * TL::HH + (t1 / 3) % 3) * TL::HW + t1 % 3;
    const int toff = ((lg >> 1) ? tB : tA) * XSB;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) F.a[mt][s] = *reinterpret_cast<const bf16x8*>(Xb + s * XPLANE + voff[mt] + toff);
  };
  auto mma = [&](const Frag& F) __attribute__((always_inline)) {
#define BCP_B6(I, J)                                                                                            \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)            \
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.b[nt][J], F.a[mt][I], acc[mt][nt], 0, 0, 0);
    BCP_B6(2, 0) BCP_B6(1, 1) BCP_B6(0, 2) BCP_B6(1, 0) BCP_B6(0, 1) BCP_B6(0, 0)
#undef BCP_B6
  };

  // prologue: stages 0 and 1 into ring buffers 0 and 1, stages 2 and 3 in flight in the two register sets
  float4 hpre[HF::NP], W0[NW4], W1[NW4];
  hfetch(c_begin, hpre);
  wfetch_at(c_begin, 0, W0);
  wfetch_at(c_begin, 1, W1);
  hstash(hpre);
  wstash(Wb, W0);
  wstash(Wb + WSTAGE, W1);
  wfetch_at(c_begin, 2, W0);
  wfetch_at(c_begin, 3, W1);
  BCP_LDS_BARRIER();
  Frag F0, F1;
  read_b(F0, 0);

  int g = 0;                                         // global stage counter of this workgroup (ring position)
  // stage g of chunk cc, tap pair sg: Fc holds its fragments (the voxel part is read here at a chunk's first stage), Fn receives
  // stage g + 1's; Wn holds stage g + 2's weights and is refilled with stage g + 4's
  auto stage = [&](int cc, int sg, Frag& Fc, Frag& Fn, float4 (&Wn)[NW4]) __attribute__((always_inline)) {
    read_b(Fn, g + 1);
    read_a(Fn, sg + 1);
    mma(Fc);
    wstash(Wb + ((g + 2) & 3) * WSTAGE, Wn);
    wfetch_at(cc, sg + 4, Wn);
    BCP_LDS_BARRIER();
    ++g;
  };
  constexpr int HPF = S - 4;
#pragma unroll 1
  for (int cc = c_begin; cc < c_end; ++cc) {
    if (cc > c_begin) {
      // (the barrier that ended the previous chunk's last stage: every wave is done with its halo planes)
      hstash(hpre);
      BCP_LDS_BARRIER();
    }
    read_a(F0, 0);
#pragma unroll 1
    for (int sg = 0; sg < HPF; sg += 2) { stage(cc, sg, F0, F1, W0); stage(cc, sg + 1, F1, F0, W1); }
    hfetch(cc + 1 < c_end ? cc + 1 : cc, hpre);
#pragma unroll 1
    for (int sg = HPF; sg < S; sg += 2) { stage(cc, sg, F0, F1, W0); stage(cc, sg + 1, F1, F0, W1); }
  }

  b6_epilogue<TL, TD, TH, TW, NT>(acc, Y, bias, cd, n, d0, h0, w0, cout0, accumulate, st, Ss);
}

