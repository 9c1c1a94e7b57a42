// tools/attic/conv3b_flatd.hip -- NOT part of libbcp_hip.so since round 4 (kept for the record).
// k_c3g: flat deep-level tiles with DIRECT weight fragments and MT m-tiles per wave (option conv3_b6_flatd): measured no faster than the
// staged / pipelined flat kernels (128 channels 36.6 vs 34.8 us, 256 channels 30.4 vs 24.7 us alone).  Text as it stood in
// csrc/conv3b.hip at the end of round 3 (needs that file's helpers).

// ------------------------------------------------------------------------------------------------
// FLAT tiles + DIRECT weight fragments (round 3): the deep-level kernel with the structure of the best mid-level one.  k_c3f gives
// a wave ONE 16-voxel m-tile against a 64-channel slab (15 LDS fragment reads per 24 MFMAs: its tap loop is LDS-bound, and every
// stage ends in a barrier that hands the weight buffers over); here a wave owns MT m-tiles (BM = 64 * MT consecutive voxels per
// workgroup, waves along M) against a 32-channel slab, takes its weight fragments straight from the pre-split pack one tap pair
// ahead (k_c3d: 6 KB per 48 MFMAs at MT = 4, no LDS stage, NO barrier in the tap loop) and reads 3 * (MT + 2) fragments per
// 12 * MT MFMAs.  A 256-voxel tile also halves the flat halo over-read (BM + 2R rows staged per BM outputs: 2.2x instead of 5.8x
// at 14x14x10) and the number of times the layer's weights cross the L2.  Split-K over the cin chunks fills the chip; the slabs go to
// the one-launch norm kernels raw (bcp_conv3_fwd_raw) or are summed by k_b6_sum_slabs.
// ------------------------------------------------------------------------------------------------
template <int KD, int MT, int NT, int AVMAX>
__global__ __launch_bounds__(256) void k_c3g(const float* __restrict__ X, const float* __restrict__ Wp, const float* __restrict__ bias,
                                             float* __restrict__ Y, ConvDims cd, int tiles_per_sample, int accumulate, StatsArg st) {
  constexpr int BM = 64 * MT, T = KD * 9, TP = (T + 1) / 2, TPE = (TP + 1) & ~1, CT = NT * 16, PD = KD == 3 ? 1 : 0;
  constexpr int XPLANE = AVMAX * XSB;
  constexpr int NP = (AVMAX * 4 + 255) / 256;                  // halo float4 per thread (row, 4-channel part)

  HIP_DYNAMIC_SHARED(float4, smem4)
  unsigned short* Xb = reinterpret_cast<unsigned short*>(smem4);   // [3][AVMAX][XSB]
  double* Ss = reinterpret_cast<double*>(Xb + 3 * XPLANE);         // [4][CT][2] statistics scratch

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int V = cd.D * cd.H * cd.W, HW = cd.H * cd.W;
  const int R = PD * HW + cd.W + 1, AV = BM + 2 * R;               // AV <= AVMAX (checked by the launcher)
  const int bx = cd.xcd ? xcd_tile(blockIdx.x, gridDim.x, gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) : (int)blockIdx.x;
  const int n = bx / tiles_per_sample, m0 = (bx % tiles_per_sample) * BM;
  const int cout0 = blockIdx.y * CT;

  // this lane's MT voxels (one per m-tile), their halo rows and the validity bits of their 27 neighbours
  int vrow[MT];
  unsigned vbits[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int ml = (wave * MT + mt) * 16 + li, mv = m0 + ml;
    vrow[mt] = (ml + R) * XSB + (lg & 1) * 8;
    const int w = mv % cd.W, h = (mv / cd.W) % cd.H, d = mv / HW;
    unsigned vb = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int kw = t % 3, kh = (t / 3) % 3, kd = t / 9;
      const bool ok = (unsigned)(w + kw - 1) < (unsigned)cd.W && (unsigned)(h + kh - 1) < (unsigned)cd.H && (unsigned)(d + kd - PD) < (unsigned)cd.D;
      vb |= (ok ? 1u : 0u) << t;
    }
    vbits[mt] = mv < V ? vb : 0u;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = cd.Cin16 >> 4;
  const int c_begin = (int)((long long)nchunks * blockIdx.z / gridDim.z), c_end = (int)((long long)nchunks * (blockIdx.z + 1) / gridDim.z);
  Y += (long long)blockIdx.z * cd.N * V * cd.Cout;

  // lane (li, lg): output channel li of n-tile nt, k quarter lg of pre-split pack row [chunk][pair][piece][cout][32 k]  (k_c3d)
  const unsigned short* Wl = reinterpret_cast<const unsigned short*>(Wp + (long long)T * cd.Cin16 * cd.Cout16) + (long long)(cout0 + li) * 32 + lg * 8;
  const long long piece_stride = (long long)cd.Cout16 * 32;
  auto bload = [&](int cc, int tp, bf16x8 (&b)[NT][3]) __attribute__((always_inline)) {
    if (B6_ABLATE & 64) { cc = c_begin; tp = 0; }       // (measurement: cache-resident weight fetches)
    const unsigned short* p = Wl + ((long long)cc * TP + (tp < TP ? tp : TP - 1)) * 3 * piece_stride;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt][s] = *reinterpret_cast<const bf16x8*>(p + s * piece_stride + nt * 16 * 32);
  };
  // halo: row r of the flat range = voxel m0 - R + r of sample n (zero outside [0, V) and beyond Cin); branch-free loads (k_c3f)
  unsigned hvm = 0;
  float4 hpre[NP];
  const long long xbase = (long long)n * V * cd.Cin;
  auto hfetch = [&](int cc) __attribute__((always_inline)) {
    hvm = 0;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int q = threadIdx.x + u * 256, r = q >> 2, part = q & 3;
      const int gm = m0 - R + r;
      const unsigned ok = (r < AV && (unsigned)gm < (unsigned)V && cc * 16 + part * 4 < cd.Cin) ? 1u : 0u;
      const unsigned off = ok ? (unsigned)(xbase + (long long)gm * cd.Cin + cc * 16 + part * 4) : 0u;
      hpre[u] = ld4(X + off);
      hvm |= ok << u;
    }
  };
  auto hstash = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int q = threadIdx.x + u * 256, r = q >> 2, part = q & 3;
      if (r < AV) {
        const float4 v = ((hvm >> u) & 1u) ? hpre[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        split_store4(v, Xb + r * XSB + part * 4, XPLANE);
      }
    }
  };

  bf16x8 B0[NT][3], B1[NT][3];
  hfetch(c_begin);
  bload(c_begin, 0, B0);
  hstash();
  BCP_LDS_BARRIER();
  constexpr int HPF = TPE >= 6 ? TPE - 4 : 0;        // pair in front of which the next chunk's halo is fetched
  const bf16x8 zero = __builtin_bit_cast(bf16x8, make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll 1
  for (int cc = c_begin; cc < c_end; ++cc) {
    if (cc > c_begin) {
      BCP_LDS_BARRIER();                             // every wave is done with the previous chunk's halo planes
      hstash();
      BCP_LDS_BARRIER();
    }
    const int ccn = cc + 1 < c_end ? cc + 1 : cc;
#pragma unroll
    for (int tp = 0; tp < TPE; ++tp) {
      // the next pair's weight fragments (the next chunk's pair 0 after the last one) into the other register set
      if (tp & 1) { if (tp + 1 < TPE) bload(cc, tp + 1, B0); else bload(ccn, 0, B0); }
      else bload(cc, tp + 1, B1);
      if (tp == HPF) hfetch(ccn);                    // (the last chunk re-reads its own halo: no conditional load)
      if (tp < TP) {
        const int t0 = 2 * tp, t1 = 2 * tp + 1 < T ? 2 * tp + 1 : T - 1;
        const int oA = ((t0 / 9) - PD) * HW + ((t0 / 3) % 3 - 1) * cd.W + (t0 % 3 - 1);      // wave-uniform row offsets of the two taps
        const int oB = ((t1 / 9) - PD) * HW + ((t1 / 3) % 3 - 1) * cd.W + (t1 % 3 - 1);
        const int toff = ((lg >> 1) ? oB : oA) * XSB;
        const int tsel = (lg >> 1) ? t1 : t0;
        bf16x8 a[MT][3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const bool ok = ((vbits[mt] >> tsel) & 1u) != 0;
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(Xb + s * XPLANE + vrow[mt] + toff);
            a[mt][s] = ok ? v : zero;
          }
        }
#define BCP_B6(BS, I, J)                                                                                        \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)            \
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BS[nt][J], a[mt][I], acc[mt][nt], 0, 0, 0);
        if (tp & 1) { BCP_B6(B1, 2, 0) BCP_B6(B1, 1, 1) BCP_B6(B1, 0, 2) BCP_B6(B1, 1, 0) BCP_B6(B1, 0, 1) BCP_B6(B1, 0, 0) }
        else { BCP_B6(B0, 2, 0) BCP_B6(B0, 1, 1) BCP_B6(B0, 0, 2) BCP_B6(B0, 1, 0) BCP_B6(B0, 0, 1) BCP_B6(B0, 0, 0) }
#undef BCP_B6
      }
    }
  }

  // epilogue: lane (li, lg) holds voxel m0 + (wave*MT + mt)*16 + li, channels lg*4 .. lg*4+3 of each n-tile: 16-byte stores into flat rows
  double s1[NT][4], s2[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.0; s2[nt][r] = 0.0; }
  const bool vec = cout0 + CT <= cd.Cout && (cd.Cout & 3) == 0;     // uniform
  const bool want_stats = st.partial != nullptr;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int mv = m0 + (wave * MT + mt) * 16 + li;
    if (mv < V) {
      float* yrow = Y + ((long long)n * V + mv) * cd.Cout + cout0 + lg * 4;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = cout0 + nt * 16 + lg * 4 + r;
          float v = acc[mt][nt][r] + ((bias && co < cd.Cout) ? bias[co] : 0.f);
          if (accumulate && co < cd.Cout) v += yrow[nt * 16 + r];
          o[r] = v;
          if (want_stats && co < cd.Cout) { s1[nt][r] += (double)v; s2[nt][r] += (double)v * (double)v; }
        }
        if (vec) st4(yrow + nt * 16, make_float4(o[0], o[1], o[2], o[3]));
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (cout0 + nt * 16 + lg * 4 + r < cd.Cout) yrow[nt * 16 + r] = o[r];
        }
      }
    }
  }
  if (want_stats) {
    const int gg = bx / st.tiles_per_group, row = bx % st.tiles_per_group;
    BCP_LDS_BARRIER();
    stats_flush_t<NT>(s1, s2, Ss, st.partial + ((long long)gg * st.rows + row) * st.C * 2, cout0, cd.Cout);
  }
}


static constexpr int kB6FlatDAvMax = 576;   // flat halo rows (BM + 2 R) of the direct-weight flat instances: 3 x 576 x 32 B = 54 KB (two workgroups per CU)

template <int KD, int MT, int NT>
static int b6_launch_flatd(const float* X, const float* Wp, const float* bias, float* Y, ConvDims cd, int accumulate, float* ws,
                           double* stat_partial, int G, bool dry, hipStream_t s, int* raw_sk, const BwdStatsIn* bw) {
  constexpr int CT = NT * 16, BM = 64 * MT;
  const int V = cd.D * cd.H * cd.W, tps = cdiv(V, BM);
  const size_t lds = (size_t)3 * kB6FlatDAvMax * XSB * 2 + (size_t)4 * CT * 2 * sizeof(double);
  auto kfn = k_c3g<KD, MT, NT, kB6FlatDAvMax>;
  if (lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int gx = cd.N * tps, gy = cd.Cout16 / CT;
  const int nch = cd.Cin16 / 16;
  // split-K over the cin chunks: as many workgroups as the 512 co-resident slots take (a power of two <= 8, >= 1 chunk each): the chain
  // of a workgroup is 14 barrier-free tap pairs per chunk
  int sk = 1;
  const bool ws_fits = ws && (long long)cd.N * V * cd.Cout <= (1LL << 20);
  if (ws_fits) {
    const int cap = nch < 8 ? nch : 8;
    while (sk * 2 <= cap && (long long)gx * gy * sk * 2 <= 512) sk *= 2;
  }
  { const int f = options().splitk; if (ws_fits && f >= 1 && f <= 4 && f <= nch) sk = f; }
  { const int f = options().conv3_b6_flat_sk; if (ws_fits && f >= 1 && f <= 8 && f <= nch) sk = f; }      // measurement override
  if (raw_sk) {                                       // raw mode: see b6_launch
    *raw_sk = sk;
    if (dry) return 0;
    StatsArg none{nullptr, 0, 1, cd.Cout, 1};
    hipLaunchKernelGGL(kfn, dim3(gx, gy, sk), dim3(256), lds, s, X, Wp, (const float*)nullptr, Y, cd, tps, 0, none);
    return 0;
  }
  if (bw) return 0;
  StatsArg st{nullptr, 0, 1, cd.Cout, G > 0 ? G : 1};
  const bool stats_ok = sk == 1 && G > 0 && cd.N % G == 0;        // tiles are sample-major and never straddle samples
  if (stats_ok) { st.rows = gx / G; st.tiles_per_group = gx / G; st.partial = stat_partial; }
  if (dry) return stats_ok ? gx / G : 0;
  if (sk == 1) {
    hipLaunchKernelGGL(kfn, dim3(gx, gy, 1), dim3(256), lds, s, X, Wp, bias, Y, cd, tps, accumulate, st);
  } else {
    const long long n = (long long)cd.N * V * cd.Cout;
    StatsArg none{nullptr, 0, 1, cd.Cout, 1};
    hipLaunchKernelGGL(kfn, dim3(gx, gy, sk), dim3(256), lds, s, X, Wp, (const float*)nullptr, ws, cd, tps, 0, none);
    hipLaunchKernelGGL(k_b6_sum_slabs, dim3((int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, s, ws, sk, n, cd.Cout, bias, Y,
                       accumulate);
  }
  return st.partial ? st.rows : 0;
}

