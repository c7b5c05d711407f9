// Every grouped weight-gradient launch of a backward pass as ONE persistent launch (gfx950, bf16).
//
// gemm_wgg.h runs one launch per group (a DeepSVG block's four products, the heads, the encoder tail): six launches per
// c2 step, each a single round of (tile, k-slice) units on the chip.  Measured per launch (profiles/r04_step_dump.txt):
// 50-58 us of which ~8 us are pipeline fill, accumulator dump and launch ramp, and the heads / encoder groups have only
// 176 / 112 units for 256 CUs.  Here the units of ALL groups are dealt round-robin to one workgroup per CU
// (unit u of a group goes to workgroup (base + u) mod nwg; bases are multiples of 8, so a group's k-slice kz still
// runs on XCD kz & 7 and the tiles that share its operand panels read them from one L2), and a workgroup walks its units
// back to back:
//   * the memory waves' software pipeline (XD k-tiles in registers, two LDS stages) runs ACROSS unit boundaries: during
//     the last XD + 1 steps of a unit they already load / stage the first k-tiles of the next one, so there is no
//     pipeline fill per unit (a unit whose job masks rows -- row codes staged in LDS -- starts with its own prologue);
//   * the math waves dump their accumulators straight to the unit's partial slab in ACCUMULATOR ORDER (16 fully
//     coalesced 1 KB stores per wave, no LDS round trip, no barrier with the memory waves) and go on;
//   * k-slices are padded to a multiple of XD k-tiles with dead tiles (out-of-range loads: zeros), so every unit starts
//     on register set 0 / stage 0 and the step loop stays unrolled over compile-time register sets.
// The split-K reduction is mfp_wgrad_reduce's (gemm_wgg.h: wgg_reduce_kernel, slab layout 1 = accumulator order): same
// fixed summation order, bit-identical gradients.
// References: the backward of every Dense and of the embedding sums -- architecture/transformer.py:85-98,163-169,
// decoder.py:39-43, encoder.py:74-92,156-160.
#pragma once

constexpr int WGM_MAX_JOBS = 40, WGM_MAX_GROUPS = 12;
struct WgmJob {
  const unsigned short* A; const unsigned short* B; const unsigned char* rowcode;
  int M, N, lda, ldb, tiles_n, tile0, colsum, pad_;
};
struct WgmGroup {
  float* ws; float* ws_col; long long zstride;
  int job0, njobs, ntiles, splitk, base, units, tpg, fresh;
};
struct WgmParams { WgmJob job[WGM_MAX_JOBS]; WgmGroup grp[WGM_MAX_GROUPS]; int ngroups, K, nwg, pad_; };

struct WgmUnit {
  int valid, g, u;
  int ji, tile, kz, m0, n0, nk, nkp, splitk, colsum, fresh, has_rc;
};

// The next unit of workgroup w after (U.g, U.u); U.g = -1: the first.  Uniform (scalar) arithmetic only.
template <int XD>
__device__ __forceinline__ void wgm_next(const WgmParams& p, int w, int ktiles, WgmUnit& U) {
  int g = U.g, u = U.u;
  bool started = g >= 0;
  for (;;) {
    if (started) u += p.nwg;
    else { g = 0; u = ((w - p.grp[0].base) % p.nwg + p.nwg) % p.nwg; started = true; }
    while (g < p.ngroups && u >= p.grp[g].units) {
      ++g;
      if (g < p.ngroups) u = ((w - p.grp[g].base) % p.nwg + p.nwg) % p.nwg;
    }
    if (g >= p.ngroups) { U.valid = 0; U.g = g; U.u = u; return; }
    const WgmGroup& G = p.grp[g];
    const int x = u & 7, j = u >> 3;
    int kz, tile;
    if (G.splitk >= 8) { kz = (j / G.ntiles) * 8 + x; tile = j % G.ntiles; }
    else {
      kz = x % G.splitk; tile = (x / G.splitk) * G.tpg + j;
      if (tile >= G.ntiles) continue;
    }
    int ji = G.job0;
    for (int q = 1; q < G.njobs; ++q) ji = tile >= p.job[G.job0 + q].tile0 ? G.job0 + q : ji;
    const WgmJob& jb = p.job[ji];
    const int bid = tile - jb.tile0, tm = bid / jb.tiles_n, tn = bid % jb.tiles_n;
    U.valid = 1; U.g = g; U.u = u; U.ji = ji; U.tile = tile; U.kz = kz;
    U.m0 = tm * 128; U.n0 = tn * 128; U.splitk = G.splitk;
    U.nk = kz < ktiles ? (ktiles - kz + G.splitk - 1) / G.splitk : 0;
    U.nkp = (U.nk + XD - 1) / XD * XD;
    U.colsum = (jb.colsum != 0 && tn == 0) ? 1 : 0;
    U.fresh = G.fresh; U.has_rc = jb.rowcode != nullptr ? 1 : 0;
    return;
  }
}

template <int XD>
__global__ __launch_bounds__(512, 1) void gemm_wgm_kernel(WgmParams p) {
  constexpr int BM = 128, BN = 128, BK = 64, PAD = 8, LDS_S = BM + PAD;
  constexpr int TILE_E = BK * LDS_S;
  constexpr int STAGE_B = 2 * TILE_E * 2;
  constexpr int CH = BK * (BM / 8) / 256;
  static_assert(CH == 4, "64 x 128 tile = 1024 chunks over 256 memory threads");
  // ONE shared array (a second __shared__ object makes hipcc drain vmcnt in front of LDS reads): stages | column sums | row codes
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* const colsum_s = reinterpret_cast<float*>(smem_raw + 2 * STAGE_B);      // [16][BM]
  unsigned char* const rc_s = smem_raw + 2 * STAGE_B + 16 * BM * 4;              // [WG_MAX_KCHUNK]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int w = blockIdx.x;
  const int ktiles = (p.K + BK - 1) / BK, kend = p.K;
  WgmUnit cur;
  cur.g = -1; cur.u = 0;
  wgm_next<XD>(p, w, ktiles, cur);
  if (!cur.valid) return;
  WgmUnit nxt = cur;
  wgm_next<XD>(p, w, ktiles, nxt);
  bool first = true;

  if (wave < 4) {
    // ======================================================================== MATH waves
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    while (cur.valid) {
      if (first || cur.fresh) { __syncthreads(); __syncthreads(); }      // row codes staged; stage 0 filled
      first = false;
      for (int t = 0; t < cur.nkp; ++t) {
        const unsigned short* As = reinterpret_cast<const unsigned short*>(smem_raw + (t & 1) * STAGE_B);
        const unsigned short* Bs = As + TILE_E;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
          bf16x8 xf[4], wf[4];
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const unsigned short* ptr = &As[(ks * 32 + lg * 8 + (li >> 2)) * LDS_S + wm * 64 + a * 16 + (li & 3) * 4];
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 4 * LDS_S));
            xf[a] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const unsigned short* ptr = &Bs[(ks * 32 + lg * 8 + (li >> 2)) * LDS_S + wn * 64 + (li & 3) * 16 + b * 4];
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 4 * LDS_S));
            wf[b] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
      }
      // the unit's partial tile, in accumulator order: float4 ((wave * 16 + a * 4 + b) * 64 + lane) =
      // C[wm * 64 + a * 16 + li][wn * 64 + lg * 16 + b * 4 .. + 3] (wgg_reduce_kernel, layout 1)
      {
        const WgmGroup& G = p.grp[cur.g];
        float* slab = G.ws + cur.kz * G.zstride + (long long)cur.tile * (BM * BN);
        const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc(slab, 0, BM * BN * 4, 0x00020000);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[a][b]), rss,
                                                   (unsigned int)(((wave * 16 + a * 4 + b) * 64 + lane) * 16), 0, 0);
            acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
      }
      cur = nxt;
      wgm_next<XD>(p, w, ktiles, nxt);
    }
    __syncthreads();      // (the memory waves' last column-sum hand-over)
    return;
  }

  // ========================================================================== MEMORY waves
  const int mt = tid - 256;
  const int krow0 = mt >> 4, ccol = (mt & 15) * 8;
  const int ls0 = (krow0 * LDS_S + ccol) * 2;
  u32x4 ra[XD][CH], rb[XD][CH];
  float csum[8], csumN[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { csum[e] = 0.f; csumN[e] = 0.f; }

  struct Addr {
    __amdgpu_buffer_rsrc_t rsa, rsb;
    unsigned int voa0, vob0, abad, bbad;
    int lda, ldb, nk, kz, splitk, has_rc, colsum;
  };
  auto make_addr = [&](const WgmUnit& U, bool usable, Addr& a) {
    const WgmJob& jb = p.job[U.ji];
    a.rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(jb.A), 0, 0x7FFFFFFF, 0x00020000);
    a.rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(jb.B), 0, 0x7FFFFFFF, 0x00020000);
    a.lda = jb.lda; a.ldb = jb.ldb;
    a.abad = U.m0 + ccol < jb.M ? 0u : 0xFFFFFFFFu;
    a.bbad = U.n0 + ccol < jb.N ? 0u : 0xFFFFFFFFu;
    a.voa0 = (unsigned int)((krow0 * jb.lda + U.m0 + ccol) * 2);
    a.vob0 = (unsigned int)((krow0 * jb.ldb + U.n0 + ccol) * 2);
    a.nk = usable ? U.nk : 0;       // 0: every load of this unit is dead (zeros)
    a.kz = U.kz; a.splitk = U.splitk; a.has_rc = U.has_rc; a.colsum = usable ? U.colsum : 0;
  };
  Addr ac, an;
  make_addr(cur, true, ac);
  make_addr(nxt.valid ? nxt : cur, nxt.valid && !nxt.fresh, an);
  int nk_max = 0;      // k-tiles whose row codes sit in rc_s (units with has_rc only)

  // tile `tt` of the stream that starts at the current unit: tt < cur.nkp -> the current unit, else the next one
  auto gload = [&](int set, int tt) {
    const bool useN = tt >= cur.nkp;
    const int t = useN ? tt - cur.nkp : tt;
    const int a_lda = useN ? an.lda : ac.lda, a_ldb = useN ? an.ldb : ac.ldb;
    const int a_nk = useN ? an.nk : ac.nk, a_kz = useN ? an.kz : ac.kz, a_sk = useN ? an.splitk : ac.splitk;
    const unsigned int voa0 = useN ? an.voa0 : ac.voa0, vob0 = useN ? an.vob0 : ac.vob0;
    const unsigned int abad = useN ? an.abad : ac.abad, bbad = useN ? an.bbad : ac.bbad;
    const bool rc = !useN && ac.has_rc;
    const int k0 = (t * a_sk + a_kz) * BK;
    const int live = (t - a_nk) >> 31;                      // -1 while t < nk
    const int soa = (k0 * a_lda * 2) & live, sob = (k0 * a_ldb * 2) & live;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k = k0 + krow0 + 16 * c;
      const unsigned int kbad = ~(unsigned int)(live & ((k - kend) >> 31));
      const unsigned int skip = (rc && rc_s[min(t, nk_max - 1) * BK + krow0 + 16 * c]) ? 0xFFFFFFFFu : 0u;
      const unsigned int oa = (voa0 + (unsigned int)(16 * c * a_lda * 2)) | abad | kbad | skip;
      const unsigned int ob = (vob0 + (unsigned int)(16 * c * a_ldb * 2)) | bbad | kbad;
      if (useN) {
        ra[set][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(an.rsa, oa, soa, 0));
        rb[set][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(an.rsb, ob, sob, 0));
      } else {
        ra[set][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ac.rsa, oa, soa, 0));
        rb[set][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ac.rsb, ob, sob, 0));
      }
    }
  };
  auto lstore = [&](int set, int stage, int tt) {
    unsigned char* st = smem_raw + stage * STAGE_B + ls0;
    const bool toN = tt >= cur.nkp;
    const bool cs = toN ? an.colsum != 0 : ac.colsum != 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      *reinterpret_cast<u32x4*>(st + 16 * c * LDS_S * 2) = ra[set][c];
      *reinterpret_cast<u32x4*>(st + TILE_E * 2 + 16 * c * LDS_S * 2) = rb[set][c];
      if (cs) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned int wv = ra[set][c][e];
          const float lo = bf16_to_f32((unsigned short)(wv & 0xffffu)), hi = bf16_to_f32((unsigned short)(wv >> 16));
          if (toN) { csumN[2 * e] += lo; csumN[2 * e + 1] += hi; }
          else { csum[2 * e] += lo; csum[2 * e + 1] += hi; }
        }
      }
    }
  };
  float* pend_col = nullptr;      // column sums of the previous unit wait in colsum_s for their hand-over to ws_col
  auto flush_col = [&]() {
    if (pend_col != nullptr) {
      if (mt < BM) {
        float s = 0.f;
#pragma unroll
        for (int gI = 0; gI < 16; ++gI) s += colsum_s[gI * BM + mt];
        pend_col[mt] = s;
      }
      pend_col = nullptr;
    }
  };

  while (cur.valid) {
    if (first || cur.fresh) {
      if (cur.has_rc) {
        const WgmJob& jb = p.job[cur.ji];
        nk_max = cur.nk;
        for (int i = mt; i < cur.nk * BK; i += 256) {      // row codes of this unit's tokens, k-tile by k-tile
          const int k = ((i >> 6) * cur.splitk + cur.kz) * BK + (i & 63);
          rc_s[i] = k < kend ? jb.rowcode[k] : 0;
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < XD; ++i) gload(i, i);
      lstore(0, 0, 0);
      gload(0, XD);
      __syncthreads();
    }
    first = false;
    auto step = [&](auto tc, int t) {
      constexpr int xi = (decltype(tc)::value + 1) % XD;
      if (t == 1) flush_col();      // (a barrier after the previous unit's colsum_s writes has passed)
      lstore(xi, (t + 1) & 1, t + 1);
      gload(xi, t + 1 + XD);
      __syncthreads();
    };
    for (int t = 0; t < cur.nkp; t += XD) wgg_static_for<0, XD>([&](auto ic) { step(ic, t + decltype(ic)::value); });
    // ---- unit done: its column sums go to LDS (handed over at step 1 of the next unit / after the last barrier)
    if (ac.colsum) {
#pragma unroll
      for (int e = 0; e < 8; ++e) colsum_s[krow0 * BM + ccol + e] = csum[e];
      const WgmGroup& G = p.grp[cur.g];
      pend_col = G.ws_col + ((long long)cur.kz * G.ntiles + cur.tile) * BM;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { csum[e] = csumN[e]; csumN[e] = 0.f; }
    cur = nxt;
    ac = an;
    ac.nk = cur.valid ? cur.nk : 0;
    ac.colsum = cur.valid ? cur.colsum : 0;
    wgm_next<XD>(p, w, ktiles, nxt);
    make_addr(nxt.valid ? nxt : cur, nxt.valid && !nxt.fresh, an);
  }
  __syncthreads();
  flush_col();
}

inline int launch_wgm(const WgmParams& p, hipStream_t st) {
  constexpr int lds = 2 * (2 * 64 * 136 * 2) + 16 * 128 * 4 + WG_MAX_KCHUNK;
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgm_kernel<WGG_XD>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_wgrad_merged: cannot raise dynamic LDS to %d: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_wgm_kernel<WGG_XD>), dim3(p.nwg), dim3(512), lds, st, p);
  return MFP_OK;
}
