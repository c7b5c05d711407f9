// Weight-stationary streaming GEMM for the Dense layers of the MFP path (gfx950, bf16 operands).
//
//   C[M][N] = epilogue( X[M][K] * Wt[N][K]^T ),  K = 256 or 512,  M = batch*seq_len elements (huge),
//   N = 256..1384: every Dense forward of the model, and every dgrad whose weight has a transposed
//   bf16 shadow (reference: architecture/transformer.py:85-98,163-169, encoder.py:88-92,
//   decoder.py:39-43).
//
// Why a second kernel: on these shapes the weights are tiny (<= 700 KB) and re-used by all 32768
// rows, while X and C stream through HBM exactly once.  The tile kernel in gemm.hip re-stages the
// weight tile for every 64x128 output tile (2/3 of its load-path traffic) and its thousands of
// short-lived workgroups run in lock-step load / multiply / store rounds (HBM ~38 % busy,
// profiles/r01_gemm_qkv_timeline.txt).  Here instead:
//   * ONE persistent workgroup per CU (4 waves, 1 per SIMD, up to 512 VGPRs each);
//   * the workgroup's weight slice (BN = 256 columns x K=256, or 128 x 512: 128 KB) lives in
//     REGISTERS for the whole kernel -- each wave holds its 16 NQ columns x K as MFMA fragments
//     (128 VGPRs), loaded once;
//   * X streams: 16 MT-row tiles, global -> registers (two tiles in flight) -> XOR-swizzled LDS
//     (double-buffered, one barrier per tile) -> ds_read_b128 fragments shared by the 4 waves;
//   * epilogue operands (residual / accumulate / ReLU mask) are prefetched one tile ahead so the
//     in-order VMEM return queue never makes the epilogue wait behind the X prefetch;
//   * blocks that share X rows (the N/BN column slices of one row group) are placed on the same
//     XCD, so X is fetched from HBM once and re-read from that XCD's L2.
// Rows are split into `groups` contiguous ranges of 16-row units (ragged last tile per group).
#pragma once

enum { WS_EPI_PLAIN = 0, WS_EPI_F32X = 1, WS_EPI_RELUBWD = 2 };

// EPI: which extra epilogue operand streams in (none / f32 residual-or-accumulate / bf16 ReLU
// mask); DROPOUT and OUT_BF16 are compile-time too: with no run-time flag branches and
// out-of-range-dropping buffer stores the whole tile is ONE basic block, so the compiler's vmcnt
// bookkeeping is exact (with control flow in the loop it falls back to vmcnt(0) per tile).
template <int KS, int MT, int EPI, bool DROPOUT, bool OUT_BF16>
__global__ __launch_bounds__(256) void gemm_ws_kernel(GemmParams p, int groups, int slices) {
  constexpr int NQ = 32 / KS, BM = 16 * MT, WBN = 16 * NQ, BN = 4 * WBN;
  constexpr int ROWB = 64 * KS, CPR = ROWB / 16, STAGE = BM * ROWB;
  constexpr int A_CH = BM * CPR / 256;
  static_assert((BM * CPR) % 256 == 0 && NQ >= 2, "tile/thread mismatch");
  constexpr unsigned int OOB = 0xFFFFFFF0u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];  // 2 stages of X

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
#ifdef MFP_GEMM_TRACE
  int trace_i = 0;
#define WS_STAMP() do { if (tid == 0 && trace_i < 24) p.trace[(long long)blockIdx.x * 24 + trace_i++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define WS_TSTAMP(t_, i_) do { if (tid == 0 && ((t_) == 4 || (t_) == 5)) p.trace[(long long)(256 + blockIdx.x) * 24 + ((t_) - 4) * 8 + (i_)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WS_STAMP() do {} while (0)
#define WS_TSTAMP(t_, i_) do {} while (0)
#endif
  WS_STAMP();

  // ---- block -> (row group g, column slice s); the slices of a group share an XCD (b % 8)
  int g, s;
  if (slices == 1) { g = blockIdx.x; s = 0; }
  else {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    s = j % slices;
    g = (j / slices) * 8 + xcd;   // groups % 8 == 0 (host)
  }
  const int NU = (p.M + 15) >> 4;
  const int u0 = (int)((long long)NU * g / groups), u1 = (int)((long long)NU * (g + 1) / groups);
  const int row_beg = u0 * 16, row_end = min(p.M, u1 * 16);
  const int ntiles = (u1 - u0 + MT - 1) / MT;
  if (ntiles <= 0) return;
  const int n_wave = s * BN + wave * WBN;
  // Output column of quad b for the lane group q (= lg of the output lane, = i >> 2 of the weight
  // row feeding MFMA row i).  Chosen so that the four lane groups of one row write 64 CONTIGUOUS
  // bytes per store instruction (f32: quad b at 16b + 4q; bf16: quad pair at 32(b/2) + 8q): the
  // vector-memory path costs ~5 clk per 64-byte segment an instruction touches, so 16-byte pieces
  // at a 32/64-byte stride made each store 2-4x as expensive (tools/ubench/hbm_write.hip).
  auto colq = [&](int b, int q) {
    return n_wave + (OUT_BF16 ? 32 * (b >> 1) + 8 * q + 4 * (b & 1) : 16 * b + 4 * q);
  };

  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0x7FFFFFFF, 0x00020000);
  const void* xptr = (p.flags & MFP_GEMM_RESIDUAL) ? (const void*)p.residual
                     : (EPI == WS_EPI_RELUBWD ? p.aux : (const void*)p.C);
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(xptr), 0, 0x7FFFFFFF, 0x00020000);

  // ---- X staging plan: chunk ch = tid + 256 c -> row ch / CPR, LDS slot ch % CPR holds the
  // row's 16-byte chunk (slot ^ (row & 15)): the swizzle is applied on the SOURCE address so the
  // ds_write stays linear and the fragment ds_read_b128 (16 rows x one chunk per lane group) is
  // conflict-free.
  unsigned int voa[A_CH];
  int lsa[A_CH], rowc[A_CH];
#pragma unroll
  for (int c = 0; c < A_CH; ++c) {
    const int ch = tid + c * 256, row = ch / CPR, pc = ch % CPR, sc = pc ^ (row & 15);
    rowc[c] = row;
    voa[c] = (unsigned int)((row * p.lda + sc * 8) * 2);
    lsa[c] = row * ROWB + pc * 16;
  }
  constexpr int XD = 4;   // X tiles in flight in registers (HBM latency under load ~2.7 us >> one 1.4 us step)
  u32x4 xa[XD][A_CH];
  auto gload = [&](u32x4 (&dst)[A_CH], int t) {
    // branch-free: a dead tile / row past M turns the offset into 0xFFFFFFFF (out of range ->
    // zeros, no access); pure integer arithmetic so the loop body stays one basic block.
    const int row0 = row_beg + t * BM;
    const int live = (t - ntiles) >> 31;                 // -1 while t < ntiles
    const int so = (row0 * p.lda * 2) & live;
#pragma unroll
    for (int c = 0; c < A_CH; ++c) {
      const int ok = live & ((row0 + rowc[c] - p.M) >> 31);
      const unsigned int vo = voa[c] | ~(unsigned int)ok;
      dst[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, vo, so, 0));
    }
  };
  auto lstore = [&](const u32x4 (&src)[A_CH], int stage) {
#pragma unroll
    for (int c = 0; c < A_CH; ++c) *reinterpret_cast<u32x4*>(smem_raw + stage * STAGE + lsa[c]) = src[c];
  };

  // ---- epilogue operand prefetch (one tile ahead)
  f32x4 ex[2][EPI == WS_EPI_F32X ? MT : 1][EPI == WS_EPI_F32X ? NQ : 1];
  u32x4 exa[2][EPI == WS_EPI_RELUBWD ? MT : 1][EPI == WS_EPI_RELUBWD ? NQ / 2 : 1];   // bf16, per quad pair
  unsigned int rc[2][MT];
  // ROWSKIP off: read (and ignore) bytes of X instead of a null rowcode pointer
  const unsigned int rsmask = (p.flags & MFP_GEMM_ROWSKIP) ? 0xFFu : 0u;
  const unsigned char* rcp = (p.flags & MFP_GEMM_ROWSKIP) ? p.rowcode : reinterpret_cast<const unsigned char*>(p.A);
  auto xload = [&](int set, int t) {
    const int row0 = row_beg + t * BM;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int row = row0 + a * 16 + li;
      const int rok = ((t - ntiles) >> 31) & ((row - row_end) >> 31);
      if (EPI == WS_EPI_F32X) rc[set][a] = rcp[min(row, p.M - 1)];
#pragma unroll
      for (int b = 0; b < NQ; ++b) {
        const int col = colq(b, lg);
        const unsigned int bad = ~(unsigned int)(rok & ((col - p.N) >> 31));
        if (EPI == WS_EPI_F32X) {
          const unsigned int vo = (unsigned int)((row * p.ldc + col) * 4) | bad;
          ex[set][a][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, vo, 0, 0));
        }
        if (EPI == WS_EPI_RELUBWD && (b & 1) == 0) {   // OUT_BF16 layout: quads b, b+1 are adjacent
          const unsigned int vo = (unsigned int)((row * p.ldc + col) * 2) | bad;
          exa[set][a][b >> 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, vo, 0, 0));
        }
      }
    }
  };

  // ---- prologue: first two X tiles in flight, then the stationary weight fragments
#pragma unroll
  for (int i = 0; i < XD; ++i) gload(xa[i], i);
  // weight row feeding MFMA row i = 4q + e of quad b is column colq(b, q) + e (transposed MFMA:
  // the output lane (li, lg) then holds C[row li][colq(b, lg) + 0..3]).
  bf16x8 wf[NQ][KS];
#pragma unroll
  for (int b = 0; b < NQ; ++b) {
    const int n = colq(b, li >> 2) + (li & 3);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const unsigned int vo = n < p.N ? (unsigned int)((n * p.ldb + ks * 32 + lg * 8) * 2) : OOB;
      wf[b][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsb, vo, 0, 0));
    }
  }
  f32x4 bias4[NQ];
#pragma unroll
  for (int b = 0; b < NQ; ++b) {
    const int col = colq(b, lg);
    bias4[b] = ((p.flags & MFP_GEMM_BIAS) && col < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + col)
                                                        : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  WS_STAMP();
  lstore(xa[0], 0);
  WS_STAMP();
  gload(xa[0], XD);

  f32x4 acc[2][MT][NQ];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NQ; ++b) acc[1][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};   // read (masked) by step 0

  const float inv_keep = DROPOUT ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  const unsigned long long rng_off =
      p.offset + ((DROPOUT && p.step_ptr) ? (unsigned long long)(*p.step_ptr) * MFP_RNG_STEP_STRIDE : 0ull);
  const float relu_floor = (p.flags & MFP_GEMM_RELU) ? 0.f : -3.0e38f;   // branch-free ReLU switch
  const unsigned char* frag_base = smem_raw + li * ROWB;
  // Every prologue load (weights, bias, first tiles) retires HERE: a first use inside the loop
  // would make the compiler place its preheader-derived vmcnt(N<=9) waits in the loop body, which
  // in steady state drain the X prefetch and the stores every tile.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  WS_STAMP();
  __syncthreads();
  WS_STAMP();

  // One pipeline step t (0 .. ntiles):  MFMAs of tile t (stage `cur`, accumulators acc[cur])
  // INTERLEAVED with the epilogue of tile t-1 (acc[cur ^ 1], operands ex[cur ^ 1]) and with the
  // staging traffic (X(t+1) registers -> LDS, X(t+3) global -> registers).  With one wave per SIMD
  // nothing else overlaps the ~600 clk of epilogue VALU and the ~150 clk each VMEM instruction
  // needs to issue with the 1024 clk of MFMA work; back to back they made a tile 2900 clk.
  // Step 0 has no epilogue (masked rows), step ntiles multiplies a stale stage (never stored).
  auto tile = [&](auto tc, int t) {
    constexpr int cur = decltype(tc)::value & 1, xi = (decltype(tc)::value + 1) % XD;
    if (EPI != WS_EPI_PLAIN) xload(cur, t);      // consumed by the next step
    const unsigned char* st = frag_base + cur * STAGE;
    bf16x8 xf[KS][MT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int a = 0; a < MT; ++a)
        xf[ks][a] = *reinterpret_cast<const bf16x8*>(st + a * 16 * ROWB + (((ks * 4 + lg) ^ li) * 16));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NQ; ++b)
#ifdef WS_NO_MFMA
          acc[cur][a][b] = (ks == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[cur][a][b]) + __builtin_bit_cast(f32x4, xf[ks][a]) * __builtin_bit_cast(f32x4, wf[b][ks])[0];
#else
          acc[cur][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              wf[b][ks], xf[ks][a], ks == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[cur][a][b], 0, 0, 0);
#endif
    lstore(xa[xi], cur ^ 1);          // X(t+1): loaded XD steps ago
#ifndef WS_NO_LOAD
    gload(xa[xi], t + 1 + XD);
#endif
    // ---- epilogue of tile t-1 (stores past row_end / N get an out-of-range offset and are dropped)
    const int row0 = row_beg + (t - 1) * BM;
    const int tok = ~((t - 1) >> 31);            // 0 at step 0
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int row = row0 + a * 16 + li;
      const int rok = tok & ((row - row_end) >> 31);
      const bool skip = EPI == WS_EPI_F32X && (rc[cur ^ 1][a] & rsmask) != 0;
      f32x4 v[NQ];
#pragma unroll
      for (int b = 0; b < NQ; ++b) {
        const int col = colq(b, lg);
        f32x4 x = acc[cur ^ 1][a][b] + bias4[b];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], relu_floor);
        if (EPI == WS_EPI_RELUBWD) {
          const unsigned int h0 = exa[cur ^ 1][a][b >> 1][2 * (b & 1)], h1 = exa[cur ^ 1][a][b >> 1][2 * (b & 1) + 1];
          x[0] = bf16_to_f32((unsigned short)(h0 & 0xffff)) > 0.f ? x[0] : 0.f;
          x[1] = bf16_to_f32((unsigned short)(h0 >> 16)) > 0.f ? x[1] : 0.f;
          x[2] = bf16_to_f32((unsigned short)(h1 & 0xffff)) > 0.f ? x[2] : 0.f;
          x[3] = bf16_to_f32((unsigned short)(h1 >> 16)) > 0.f ? x[3] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = skip ? 0.f : x[r];
        if (DROPOUT) {
          unsigned int rnd[4];
          philox4x32(p.seed, (unsigned int)row, (unsigned int)(col >> 2), rng_off, rnd);
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = philox_keep(rnd[r], p.dropout_p) ? x[r] * inv_keep : 0.f;
        }
        if (EPI == WS_EPI_F32X) x += ex[cur ^ 1][a][b];
        v[b] = x;
      }
      if (OUT_BF16) {   // N % 8 == 0 (host): 16-byte units of 8 columns
#pragma unroll
        for (int b = 0; b < NQ; b += 2) {
          const int col = colq(b, lg);
          const unsigned int vo = (unsigned int)((row * p.ldc + col) * 2) | ~(unsigned int)(rok & ((col - p.N) >> 31));
          const u32x4 pk = {pack_bf16x2(v[b][0], v[b][1]), pack_bf16x2(v[b][2], v[b][3]),
                            pack_bf16x2(v[b + 1][0], v[b + 1][1]), pack_bf16x2(v[b + 1][2], v[b + 1][3])};
#ifndef WS_NO_STORE
          __builtin_amdgcn_raw_buffer_store_b128(pk, rsc, vo, 0, 0);
#else
          if (vo == 0x12345u) __builtin_amdgcn_raw_buffer_store_b128(pk, rsc, vo, 0, 0);
#endif
        }
      } else {
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
          const int col = colq(b, lg);
          const unsigned int vo = (unsigned int)((row * p.ldc + col) * 4) | ~(unsigned int)(rok & ((col - p.N) >> 31));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[b]), rsc, vo, 0, 0);
        }
      }
    }
    // Issue order: every fragment ds_read, then per 8 MFMAs (128 clk of matrix pipe) one VMEM
    // instruction and a slice of the epilogue VALU / ds_write work in the MFMA shadow.
    constexpr int NMFMA = KS * MT * NQ, NVMEM = A_CH + MT * (OUT_BF16 ? NQ / 2 : NQ) + (EPI == WS_EPI_PLAIN ? 0 : MT * NQ);
    constexpr int GROUPS = NMFMA / 8;
    __builtin_amdgcn_sched_group_barrier(0x100, KS * MT, 0);
#pragma unroll
    for (int gI = 0; gI < GROUPS; ++gI) {
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                   // 8 MFMA
      __builtin_amdgcn_sched_group_barrier(0x030, (NVMEM + GROUPS - 1) / GROUPS, 0);       // VMEM r/w
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                    // ds_write
      __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);                                  // VALU
    }
    __syncthreads();
    WS_STAMP();
  };

  // Pairs inside the loop, odd tail after it: a conditional second half inside the loop gives
  // the compiler a (never taken) path with fewer VMEM ops between a load and its use, and it
  // sizes every vmcnt for that path.  ntiles + 1 steps in total (the last one only drains).
  static_assert(XD == 2 || XD == 4, "step loop is unrolled by XD");
  int t = 0;
  if (XD == 4) {
    for (; t + 3 <= ntiles; t += 4) {
      tile(std::integral_constant<int, 0>{}, t);
      tile(std::integral_constant<int, 1>{}, t + 1);
      tile(std::integral_constant<int, 2>{}, t + 2);
      tile(std::integral_constant<int, 3>{}, t + 3);
    }
    if (t <= ntiles) tile(std::integral_constant<int, 0>{}, t);
    if (t + 1 <= ntiles) tile(std::integral_constant<int, 1>{}, t + 1);
    if (t + 2 <= ntiles) tile(std::integral_constant<int, 2>{}, t + 2);
  } else {
    for (; t + 1 <= ntiles; t += 2) {
      tile(std::integral_constant<int, 0>{}, t);
      tile(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t <= ntiles) tile(std::integral_constant<int, 0>{}, t);
  }
#ifdef MFP_GEMM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WS_STAMP();
#endif
}

// Host side: can this call take the weight-stationary kernel?
inline bool ws_eligible(const mfp_gemm_args* a, int splitk) {
  if (!(a->a_kmajor && a->b_kmajor) || a->in_dtype != MFP_BF16 || splitk != 1) return false;
  if (a->K != 256 && a->K != 512) return false;
  const int f = a->flags;
  if (f & (MFP_GEMM_COLSUM_A | MFP_GEMM_ROWSKIP_A)) return false;
  const bool f32x = (f & (MFP_GEMM_RESIDUAL | MFP_GEMM_ACCUM)) != 0;
  if ((f & MFP_GEMM_RESIDUAL) && (f & MFP_GEMM_ACCUM)) return false;
  if (f32x && a->out_dtype != MFP_F32) return false;
  if ((f & MFP_GEMM_RELU_BWD) && (f32x || a->out_dtype != MFP_BF16)) return false;
  if ((f & MFP_GEMM_ROWSKIP) && !f32x) return false;
  if ((f & MFP_GEMM_DROPOUT) && !f32x) return false;
  if (a->out_dtype == MFP_BF16 && a->N % 8 != 0) return false;
  if ((long long)a->M * a->ldc * 4 >= 0x7FFFFFF0ll) return false;   // 32-bit epilogue offsets
  if (a->M < 16) return false;
  return true;
}

template <int KS, int MT, int EPI, bool DROPOUT, bool OUT_BF16>
int launch_ws(const GemmParams& p, int ncu, hipStream_t st) {
  constexpr int NQ = 32 / KS, BN = 64 * NQ, STAGE = 16 * MT * 64 * KS;
  constexpr int lds = 2 * STAGE;
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel<KS, MT, EPI, DROPOUT, OUT_BF16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_gemm: cannot raise dynamic LDS to %d: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  const int slices = (p.N + BN - 1) / BN;
  const int units = (p.M + 15) / 16;
  int groups;
  if (slices == 1) {
    groups = ncu < units ? ncu : units;
  } else {
    groups = (ncu / slices) / 8 * 8;
    if (groups < 8) groups = 8;
  }
  hipLaunchKernelGGL((gemm_ws_kernel<KS, MT, EPI, DROPOUT, OUT_BF16>), dim3(groups * slices), dim3(256), lds, st, p,
                     groups, slices);
  return MFP_OK;
}

template <int KS, int MT>
int launch_ws_epi(const mfp_gemm_args* a, const GemmParams& p, int ncu, hipStream_t st) {
  const int f = a->flags;
  if (f & (MFP_GEMM_RESIDUAL | MFP_GEMM_ACCUM))
    return (f & MFP_GEMM_DROPOUT) ? launch_ws<KS, MT, WS_EPI_F32X, true, false>(p, ncu, st)
                                  : launch_ws<KS, MT, WS_EPI_F32X, false, false>(p, ncu, st);
  if (f & MFP_GEMM_RELU_BWD) return launch_ws<KS, MT, WS_EPI_RELUBWD, false, true>(p, ncu, st);
  return a->out_dtype == MFP_BF16 ? launch_ws<KS, MT, WS_EPI_PLAIN, false, true>(p, ncu, st)
                                  : launch_ws<KS, MT, WS_EPI_PLAIN, false, false>(p, ncu, st);
}
