// Streaming weight-gradient GEMM (gfx950, bf16 operands): partial C_z[M][N] = A[kz-chunk][M]^T B[kz-chunk][N]
// for the split-K chunk kz of the token dimension (reference: the backward of every Dense,
// architecture/transformer.py:85-98,163-169, encoder.py:88-92, decoder.py:39-43, and -- through the
// one-hot count matrix -- of the embedding sums, encoder.py:156-160).
//
// The tile kernel in gemm.hip runs this product with ONE k-tile of prefetch and 4 waves per CU: a
// workgroup waits out a full memory latency per 64 tokens (measured 24-40 us for 33-67 MB of
// operands).  Same output tiling here (128 x 128 per workgroup, 2 x 2 math waves of 64 x 64,
// ds_read_b64_tr_b16 fragments from untransposed [k][m] / [k][n] LDS images), but with the
// warp-specialised pipeline of gemm_ws.h:
//   * 4 MEMORY waves stream the A and B k-tiles (64 tokens x 128 columns each) global -> registers
//     (XD tiles in flight) -> LDS stage, accumulate the bias-gradient column sums of A on the way,
//     and at the end store the partial tile as whole rows;
//   * 4 MATH waves only read LDS and issue MFMAs; one barrier per k-tile.
// Grid: 1-D over (k-chunk, tile) with all tiles of a chunk on one XCD (see gemm.hip).
#pragma once

constexpr int WG_MAX_KCHUNK = 4096;   // tokens per split-K chunk a ROWSKIP_A launch may have (row codes in LDS)

// ROWSKIP: rows of A whose row code is non-zero count as zero rows (MFP_GEMM_ROWSKIP_A).  The codes of
// the workgroup's whole k-chunk are staged in LDS up front: a per-tile global byte load would sit in
// front of the operand loads it gates, and since vmcnt retires in order every step would then wait
// for ALL tiles in flight (measured: 1.6 us per k-tile instead of 0.4).
template <int XD, bool ROWSKIP>
__global__ __launch_bounds__(512) void gemm_wg_kernel(GemmParams p) {
  constexpr int BM = 128, BN = 128, BK = 64, PAD = 8, LDS_S = BM + PAD;   // bf16 elements per LDS row
  constexpr int TILE_E = BK * LDS_S;                                       // elements per operand tile
  constexpr int STAGE_B = 2 * TILE_E * 2;                                  // bytes per stage (A + B)
  constexpr int CH = BK * (BM / 8) / 256;                                  // 16-byte chunks per memory thread and operand
  constexpr int CS_LD = BN + 4;                                            // f32 row stride of the output stage
  static_assert(CH == 4, "64 x 128 tile = 1024 chunks over 256 memory threads");
  static_assert(BM * CS_LD * 4 <= 2 * STAGE_B, "output stage aliases the two operand stages");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ float colsum_s[16][BM];   // memory waves: per row-group column sums of A
  __shared__ unsigned char rc_s[ROWSKIP ? WG_MAX_KCHUNK : 16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
#ifdef MFP_GEMM_TRACE
  int trace_i = 0;
#define WG_STAMP() do { if (tid == 0 && trace_i < 24) p.trace[(long long)blockIdx.x * 24 + trace_i++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WG_STAMP() do {} while (0)
#endif
#ifdef MFP_GEMM_TRACE
  int mtrace_i = 0;   // memory wave 0: phases of k-tiles 4..7 (s_memtime, core clocks)
#define WG_MSTAMP(t) do { if (tid == 256 && (t) >= 4 && (t) < 8 && mtrace_i < 24) p.trace[(long long)(gridDim.x + blockIdx.x) * 24 + mtrace_i++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WG_MSTAMP(t) do {} while (0)
#endif
  WG_STAMP();
  const int tiles = p.tiles_m * p.tiles_n, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int kz = (j / tiles) * 8 + xcd, bid = j % tiles;          // splitk % 8 == 0 (host)
  const int tm = bid / p.tiles_n, tn = bid % p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = kz * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  const bool do_colsum = (p.flags & MFP_GEMM_COLSUM_A) && tn == 0;
  f32x4 acc[4][4];
  if (ROWSKIP) {
    for (int i = tid; i < p.kchunk; i += 512) rc_s[i] = kbeg + i < p.K ? p.rowcode[kbeg + i] : 0;
    __syncthreads();
  }

  if (wave < 4) {
    // ======================================================================== MATH waves
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();   // prologue barrier (stage 0 filled)
    WG_STAMP();
    for (int t = 0; t < nk; ++t) {
      const unsigned short* As = reinterpret_cast<const unsigned short*>(smem_raw + (t & 1) * STAGE_B);
      const unsigned short* Bs = As + TILE_E;
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 xf[4], wf[4];
        // operand k-order {8 lg + j, 8 lg + 4 + j}: two tr-reads of 4 token rows each
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const unsigned short* ptr = &As[(ks * 32 + lg * 8 + (li >> 2)) * LDS_S + wm * 64 + a * 16 + (li & 3) * 4];
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 4 * LDS_S));
          xf[a] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {   // the 4 lanes of a token row supply column bases 16 q + 4 b
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
      if ((t & 3) == 3) WG_STAMP();
    }
    // ---- partial tile -> LDS (rows m, 16 contiguous columns per lane), stored by all waves below
    float* Cs = reinterpret_cast<float*>(smem_raw);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        *reinterpret_cast<f32x4*>(&Cs[(wm * 64 + a * 16 + li) * CS_LD + wn * 64 + lg * 16 + b * 4]) = acc[a][b];
  } else {
    // ====================================================================== MEMORY waves
    const int mt = tid - 256;
    const unsigned short* Ag = reinterpret_cast<const unsigned short*>(p.A);
    const unsigned short* Bg = reinterpret_cast<const unsigned short*>(p.B);
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Ag), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Bg), 0, 0x7FFFFFFF, 0x00020000);
    // chunk c of a thread: token row krow0 + 16 c, columns ccol .. ccol + 7 (16 chunks per 128-column row)
    const int krow0 = mt >> 4, ccol = (mt & 15) * 8;
    const unsigned int abad = m0 + ccol < p.M ? 0u : 0xFFFFFFFFu, bbad = n0 + ccol < p.N ? 0u : 0xFFFFFFFFu;
    const unsigned int voa0 = (unsigned int)((krow0 * p.lda + m0 + ccol) * 2);
    const unsigned int vob0 = (unsigned int)((krow0 * p.ldb + n0 + ccol) * 2);
    const int ls0 = (krow0 * LDS_S + ccol) * 2;
    u32x4 ra[XD][CH], rb[XD][CH];
    float csum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[e] = 0.f;
    auto gload = [&](int set, int t) {
      const int k0 = kbeg + t * BK;
      const int live = (t - nk) >> 31;                      // -1 while t < nk
      const int soa = (k0 * p.lda * 2) & live, sob = (k0 * p.ldb * 2) & live;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int k = k0 + krow0 + 16 * c;
        const unsigned int kbad = ~(unsigned int)(live & ((k - kend) >> 31));
        const unsigned int skip = (ROWSKIP && rc_s[min(k - kbeg, p.kchunk - 1)]) ? 0xFFFFFFFFu : 0u;
        ra[set][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rsa, (voa0 + (unsigned int)(16 * c * p.lda * 2)) | abad | kbad | skip, soa, 0));
        rb[set][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rsb, (vob0 + (unsigned int)(16 * c * p.ldb * 2)) | bbad | kbad, sob, 0));
      }
    };
    auto lstore = [&](int set, int stage) {
      unsigned char* st = smem_raw + stage * STAGE_B + ls0;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        *reinterpret_cast<u32x4*>(st + 16 * c * LDS_S * 2) = ra[set][c];
        *reinterpret_cast<u32x4*>(st + TILE_E * 2 + 16 * c * LDS_S * 2) = rb[set][c];
        if (do_colsum) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned int w = ra[set][c][e];
            csum[2 * e] += bf16_to_f32((unsigned short)(w & 0xffffu));
            csum[2 * e + 1] += bf16_to_f32((unsigned short)(w >> 16));
          }
        }
      }
    };
#pragma unroll
    for (int i = 0; i < XD; ++i) gload(i, i);
    lstore(0, 0);
    gload(0, XD);
    __syncthreads();   // prologue barrier
    // step t: tile t+1 registers -> LDS stage (t+1)&1, tile t+1+XD global -> registers
    auto step = [&](auto tc, int t) {
      constexpr int xi = (decltype(tc)::value + 1) % XD;
      WG_MSTAMP(t);
      lstore(xi, (t + 1) & 1);
      WG_MSTAMP(t);
      gload(xi, t + 1 + XD);
      WG_MSTAMP(t);
      __syncthreads();
      WG_MSTAMP(t);
    };
    static_assert(XD == 4, "step loop unrolled by XD");
    int t = 0;
    for (; t + 3 < nk; t += 4) {
      step(std::integral_constant<int, 0>{}, t);
      step(std::integral_constant<int, 1>{}, t + 1);
      step(std::integral_constant<int, 2>{}, t + 2);
      step(std::integral_constant<int, 3>{}, t + 3);
    }
    if (t < nk) step(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nk) step(std::integral_constant<int, 1>{}, t + 1);
    if (t + 2 < nk) step(std::integral_constant<int, 2>{}, t + 2);
    if (do_colsum) {
#pragma unroll
      for (int e = 0; e < 8; ++e) colsum_s[krow0][ccol + e] = csum[e];
    }
  }
  WG_STAMP();
  __syncthreads();   // partial tile (and column sums) are in LDS
  WG_STAMP();
  // ---- all 8 waves: partial tile -> ws[kz][M][N] as whole rows (128 f32 = 32 lanes x 16 B)
  {
    const float* Cs = reinterpret_cast<const float*>(smem_raw);
    float* ws = p.ws + (long long)kz * p.M * p.N;
    const int r0 = tid >> 5, c4 = (tid & 31) * 4;
#pragma unroll
    for (int i = 0; i < BM / 16; ++i) {
      const int row = r0 + 16 * i;
      if (m0 + row < p.M && n0 + c4 < p.N)
        *reinterpret_cast<f32x4*>(ws + (long long)(m0 + row) * p.N + n0 + c4) =
            *reinterpret_cast<const f32x4*>(&Cs[row * CS_LD + c4]);
    }
    if (do_colsum && tid < BM && m0 + tid < p.M) {
      float s = 0.f;
#pragma unroll
      for (int gI = 0; gI < 16; ++gI) s += colsum_s[gI][tid];
      p.ws_col[(long long)kz * p.M + m0 + tid] = s;
    }
  }
#ifdef MFP_GEMM_TRACE
  __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
  WG_STAMP();
}

inline bool wg_eligible(const mfp_gemm_args* a, int splitk) {
  if ((a->flags & MFP_GEMM_ROWSKIP_A) && (a->K + splitk - 1) / splitk > WG_MAX_KCHUNK) return false;
  return !a->a_kmajor && !a->b_kmajor && a->in_dtype == MFP_BF16 && splitk >= 8 && splitk % 8 == 0 &&
         a->M % 8 == 0 && a->N % 8 == 0 && a->lda % 8 == 0 && a->ldb % 8 == 0;
}

template <bool ROWSKIP>
inline int launch_wg_t(const GemmParams& p0, int M, int N, int splitk, hipStream_t st) {
  constexpr int lds = 2 * (2 * 64 * 136 * 2);
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wg_kernel<4, ROWSKIP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_gemm: cannot raise dynamic LDS to %d: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  GemmParams p = p0;
  p.tiles_m = (M + 127) / 128;
  p.tiles_n = (N + 127) / 128;
  p.kz_xcd = 1;
  hipLaunchKernelGGL((gemm_wg_kernel<4, ROWSKIP>), dim3(p.tiles_m * p.tiles_n * splitk), dim3(512), lds, st, p);
  return MFP_OK;
}

inline int launch_wg(const GemmParams& p0, int M, int N, int splitk, hipStream_t st) {
  return (p0.flags & MFP_GEMM_ROWSKIP_A) ? launch_wg_t<true>(p0, M, N, splitk, st) : launch_wg_t<false>(p0, M, N, splitk, st);
}
