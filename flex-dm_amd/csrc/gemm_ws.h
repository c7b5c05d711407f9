// Weight-stationary streaming GEMM for the Dense layers of the MFP path (gfx950, bf16 operands).
//
//   C[M][N] = epilogue( X[M][K] * Wt[N][K]^T ),  K = 256, 512 or 768,  M = batch*seq_len elements (huge),
//   N = 256..1384: every Dense forward of the model, and every dgrad whose weight has a transposed
//   bf16 shadow (reference: architecture/transformer.py:85-98,163-169, encoder.py:88-92,
//   decoder.py:39-43).
//
// Why a second kernel: on these shapes the weights are tiny (<= 700 KB) and re-used by all 32768
// rows, while X and C stream through HBM exactly once -- the products are bound by the WRITE
// stream (tools/ubench/hbm_write.hip: ~6.3 TB/s chip-wide = 11.7 B/clk per CU, and one wave gets
// a 1 KB store accepted only every ~350 clk).  The tile kernel in gemm.hip re-stages the weight
// tile for every 64x128 output tile and runs load / multiply / store rounds in lock-step.  Here:
//   * ONE persistent workgroup per CU, 8 waves = 4 MATH waves + 4 MEMORY waves (2 per SIMD);
//   * math waves keep the workgroup's weight slice (256 columns x K=256, 128 x 512, 64 x 768; half
//     that for the N <= 256 layers: NARROW) in REGISTERS for the whole kernel (<= 128 VGPRs of MFMA
//     fragments per lane, filled once through LDS with full-line loads), read X fragments from
//     LDS, multiply, add bias / ReLU and drop the tile into an LDS output stage -- they never
//     touch vector memory, so a store that waits for the write path never stalls an MFMA;
//   * memory waves do ALL vector memory: X tiles global -> registers (4 tiles in flight) ->
//     XOR-swizzled LDS stage; epilogue operands (residual / accumulate / ReLU mask, prefetched
//     two tiles ahead); and the output tile LDS -> (mask, dropout, residual) -> global as whole
//     contiguous rows (1 KB per store instruction);
//   * one barrier per tile orders both hand-offs (X stage t+1 filled / output stage t-1 drained
//     while tile t is multiplied);
//   * blocks that share X rows (the N/BN column slices of one row group) sit on the same XCD, so
//     X comes from HBM once and is re-read from that XCD's L2.
// Rows are split into `groups` contiguous ranges of 16-row units (ragged last tile per group).
#pragma once

enum { WS_EPI_PLAIN = 0, WS_EPI_F32X = 1, WS_EPI_RELUBWD = 2 };

// EPI: which extra epilogue operand streams in (none / f32 residual-or-accumulate / bf16 ReLU
// mask); DROPOUT and OUT_BF16 are compile-time too: with no run-time flag branches and
// out-of-range-dropping buffer accesses every pipeline step is ONE basic block, so the compiler's
// vmcnt bookkeeping is exact (with control flow in the loop it falls back to vmcnt(0) per tile).
// NARROW halves the workgroup's column slice (K = 256: 128 instead of 256 columns): half the weight
// prologue per CU and twice the row tiles per workgroup -- for the N = 256 layers, whose 128 rows per
// workgroup at full width are only 4 pipeline steps behind a 4 us prologue.
template <int KS, int MT, int EPI, bool DROPOUT, bool OUT_BF16, bool NARROW = false>
__global__ __launch_bounds__(512) void gemm_ws_kernel(GemmParams p, int groups, int slices) {
  constexpr int NQ = (KS == 8 ? 4 : (KS == 16 ? 2 : 1)) >> (NARROW ? 1 : 0), BM = 16 * MT, WBN = 16 * NQ, BN = 4 * WBN;
  constexpr int ROWB = 64 * KS, CPR = ROWB / 16, XSTAGE = BM * ROWB;      // X stage: [BM][K] bf16
  constexpr int A_CH = BM * CPR / 256;
  constexpr int OS = OUT_BF16 ? 2 : 4, OROWB = BN * OS, OCPR = OROWB / 16, OSTAGE = BM * OROWB;
  constexpr int O_CH = BM * OCPR / 256;                                   // output chunks per memory thread
  constexpr bool PAIRED = OUT_BF16 && NQ >= 2;   // bf16 quads b, b+1 adjacent: one 16-byte stage unit
  constexpr int XD = KS == 8 ? 4 : (KS <= 24 ? 2 : 1);   // X tiles in flight in registers (16 / 32-48 / 64 KB each)
  constexpr int ED = 4;                 // rotating epilogue-operand sets (3 live: in use + two in flight)
  static_assert((BM * CPR) % 256 == 0 && (BM * OCPR) % 256 == 0 && NQ >= 1, "tile/thread mismatch");
  constexpr unsigned int OOB = 0xFFFFFFF0u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* const xst = smem_raw;                 // 2 X stages
  unsigned char* const ost = smem_raw + 2 * XSTAGE;    // 2 output stages

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
#ifdef MFP_GEMM_TRACE
  int trace_i = 0;
#define WS_STAMP() do { if (tid == 0 && trace_i < 24) p.trace[(long long)blockIdx.x * 24 + trace_i++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WS_STAMP() do {} while (0)
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
  const int Mrows = p.M;
  const int NU = (Mrows + 15) >> 4;
  const int u0 = (int)((long long)NU * g / groups), u1 = (int)((long long)NU * (g + 1) / groups);
  const int row_beg = u0 * 16, row_end = min(Mrows, u1 * 16);
  const int ntiles = (u1 - u0 + MT - 1) / MT;
  if (ntiles <= 0) return;
  const int n_slice = s * BN;

  // ---- weight slice -> registers of the math waves, through LDS: all 512 threads copy half of the
  // slice (the rows of two math waves, <= 64 KB) with FULL-LINE coalesced loads into a row-major,
  // XOR-swizzled image; those two waves then pick their MFMA fragments with ds_read_b128.
  // (Fragment-shaped global loads -- 16 rows x 64 B per instruction -- took 4.4 us per launch for
  // the 128 KB slice: tools/trace_gemm_ws.py.)  The image aliases the X / output stages, which are
  // idle until the pipeline starts.
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, 0x7FFFFFFF, 0x00020000);
  constexpr int WROWS = BN / 2, WCH = WROWS * CPR / 512;
  static_assert((WROWS * CPR) % 512 == 0 && WROWS * ROWB <= 2 * XSTAGE + 2 * OSTAGE, "weight staging image");
  // row -> li of the lane that reads it (its swizzle key): row = colq(b, q) + e with li = 4q + e
  // Both staging rounds' global loads are issued up front (second round trip hidden behind the
  // first round's LDS write / fragment reads); registers are free here: fragments are not live yet.
  auto w_load = [&](int r, u32x4 (&v)[WCH]) {
#pragma unroll
    for (int c = 0; c < WCH; ++c) {
      const int ch = tid + c * 512, row = ch / CPR, pc = ch % CPR, rl = row % WBN;
      const int key = PAIRED ? 4 * ((rl >> 3) & 3) + (rl & 3) : (rl & 15);
      const int n = n_slice + r * WROWS + row;
      const unsigned int vo = n < p.N ? (unsigned int)((n * p.ldb + (pc ^ key) * 8) * 2) : OOB;
      v[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, vo, 0, 0));
    }
  };
  auto w_store = [&](const u32x4 (&v)[WCH]) {
#pragma unroll
    for (int c = 0; c < WCH; ++c) {
      const int ch = tid + c * 512;
      *reinterpret_cast<u32x4*>(smem_raw + (ch / CPR) * ROWB + (ch % CPR) * 16) = v[c];
    }
  };

  if (wave < 4) {
    // ======================================================================== MATH waves
    // Output column (local to the wave's 16 NQ block) of quad b for lane group q: f32 16b + 4q,
    // bf16 32(b/2) + 8q + 4(b%2) -- the lane's 16-byte unit (4 f32 / 8 bf16) is one stage chunk.
    auto colq = [&](int b, int q) { return PAIRED ? 32 * (b >> 1) + 8 * q + 4 * (b & 1) : 16 * b + 4 * q; };
    const int n_wave = n_slice + wave * WBN;
    // weight row feeding MFMA row i = 4q + e of quad b is column colq(b, q) + e (transposed MFMA:
    // the output lane (li, lg) then holds C[row li][colq(b, lg) + 0..3]).
    bf16x8 wf[NQ][KS];
    u32x4 wv[2][WCH];
    w_load(0, wv[0]);
    w_load(1, wv[1]);
    WS_STAMP();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      w_store(wv[r]);
      WS_STAMP();
      __syncthreads();
      if ((wave >> 1) == r) {
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
          const unsigned char* wrow = smem_raw + ((wave & 1) * WBN + colq(b, li >> 2) + (li & 3)) * ROWB;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            wf[b][ks] = *reinterpret_cast<const bf16x8*>(wrow + (((ks * 4 + lg) ^ li) * 16));
        }
      }
      __syncthreads();
      WS_STAMP();
    }
    f32x4 bias4[NQ];
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
      const int col = n_wave + colq(b, lg);
      bias4[b] = ((p.flags & MFP_GEMM_BIAS) && col < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + col)
                                                          : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float relu_floor = (p.flags & MFP_GEMM_RELU) ? 0.f : -3.0e38f;   // branch-free ReLU switch
    // fragment reads: row a*16 + li, logical chunk 4 ks + lg stored at slot chunk ^ li
    const unsigned char* frag_base = xst + li * ROWB;
    // output stage writes: row a*16 + li, byte offset `o` of the row at slot (o/16) ^ (li & 7)
    unsigned char* ostw = ost + li * OROWB;
    auto oaddr = [&](int b) {
      const int o = (wave * WBN + colq(b, lg)) * OS;
      return (((o >> 4) ^ (li & 7)) << 4) + (o & 15);
    };
    int oslot[NQ];
#pragma unroll
    for (int b = 0; b < NQ; ++b) oslot[b] = oaddr(b);
    // Every prologue load retires HERE: a first use inside the loop would make the compiler place
    // its preheader-derived vmcnt waits in the loop body.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    WS_STAMP();
    __syncthreads();
    WS_STAMP();

    auto step = [&](auto tc, int t) {
      constexpr int cur = decltype(tc)::value & 1;
      const unsigned char* st = frag_base + cur * XSTAGE;
      f32x4 acc[MT][NQ];
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
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                wf[b][ks], xf[ks][a], ks == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[a][b], 0, 0, 0);
      // fragment reads run two k-steps ahead of their MFMAs (the memory waves sharing the SIMD
      // issue no MFMAs, so nothing else hides the ds_read latency)
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT, 0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NQ, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
      }
      unsigned char* ow = ostw + cur * OSTAGE;
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        f32x4 v[NQ];
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
          v[b] = acc[a][b] + bias4[b];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[b][r] = fmaxf(v[b][r], relu_floor);
        }
        if (PAIRED) {
#pragma unroll
          for (int b = 0; b < NQ; b += 2) {
            const u32x4 pk = {pack_bf16x2(v[b][0], v[b][1]), pack_bf16x2(v[b][2], v[b][3]),
                              pack_bf16x2(v[(b + 1) % NQ][0], v[(b + 1) % NQ][1]),
                              pack_bf16x2(v[(b + 1) % NQ][2], v[(b + 1) % NQ][3])};
            *reinterpret_cast<u32x4*>(ow + a * 16 * OROWB + oslot[b]) = pk;
          }
        } else if (OUT_BF16) {   // NQ == 1: one quad = 8 bytes
          const u32x2 pk = {pack_bf16x2(v[0][0], v[0][1]), pack_bf16x2(v[0][2], v[0][3])};
          *reinterpret_cast<u32x2*>(ow + a * 16 * OROWB + oslot[0]) = pk;
        } else {
#pragma unroll
          for (int b = 0; b < NQ; ++b)
            *reinterpret_cast<f32x4*>(ow + a * 16 * OROWB + oslot[b]) = v[b];
        }
      }
      __syncthreads();
      WS_STAMP();
    };
    int t = 0;
    for (; t + 3 <= ntiles; t += 4) {
      step(std::integral_constant<int, 0>{}, t);
      step(std::integral_constant<int, 1>{}, t + 1);
      step(std::integral_constant<int, 2>{}, t + 2);
      step(std::integral_constant<int, 3>{}, t + 3);
    }
    if (t <= ntiles) step(std::integral_constant<int, 0>{}, t);
    if (t + 1 <= ntiles) step(std::integral_constant<int, 1>{}, t + 1);
    if (t + 2 <= ntiles) step(std::integral_constant<int, 2>{}, t + 2);
  } else {
    // ====================================================================== MEMORY waves
    const int mt = tid - 256;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0x7FFFFFFF, 0x00020000);
    const void* xptr = (p.flags & MFP_GEMM_RESIDUAL) ? (const void*)p.residual
                       : (EPI == WS_EPI_RELUBWD ? p.aux : (const void*)p.C);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(xptr), 0, 0x7FFFFFFF, 0x00020000);

    // ---- X staging plan: chunk ch = mt + 256 c -> row ch / CPR, LDS slot ch % CPR holds the
    // row's 16-byte chunk (slot ^ (row & 15)): the swizzle is applied on the SOURCE address so
    // the ds_write stays linear and the fragment ds_read_b128 (16 rows x one chunk per lane
    // group) is conflict-free.
    unsigned int voa[A_CH];
    int lsa[A_CH], xrow[A_CH];
#pragma unroll
    for (int c = 0; c < A_CH; ++c) {
      const int ch = mt + c * 256, row = ch / CPR, pc = ch % CPR, sc = pc ^ (row & 15);
      xrow[c] = row;
      voa[c] = (unsigned int)((row * p.lda + sc * 8) * 2);
      lsa[c] = row * ROWB + pc * 16;
    }
    u32x4 xa[XD][A_CH];
    auto gload = [&](u32x4 (&dst)[A_CH], int t) {
      // branch-free: a dead tile / row past M turns the offset into 0xFFFFFFFF (out of range ->
      // zeros, no access); pure integer arithmetic so the step stays one basic block.
      const int row0 = row_beg + t * BM;
      const int live = (t - ntiles) >> 31;                 // -1 while t < ntiles
      const int so = (row0 * p.lda * 2) & live;
#pragma unroll
      for (int c = 0; c < A_CH; ++c) {
        const int ok = live & ((row0 + xrow[c] - p.M) >> 31);
        const unsigned int vo = voa[c] | ~(unsigned int)ok;
        dst[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, vo, so, 0));
      }
    };
    auto lstore = [&](const u32x4 (&src)[A_CH], int stage) {
#pragma unroll
      for (int c = 0; c < A_CH; ++c) *reinterpret_cast<u32x4*>(xst + stage * XSTAGE + lsa[c]) = src[c];
    };

    // ---- output plan: chunk ch = mt + 256 i -> stage row ch / OCPR, logical chunk ch % OCPR: a
    // wave instruction moves 64 consecutive chunks = 1 KB of contiguous output.  256 % OCPR == 0,
    // so a thread keeps ONE chunk column and walks rows orow0 + i * ORS (nothing per-chunk is kept
    // in registers: the memory waves need theirs for data in flight).
    static_assert(256 % OCPR == 0, "output chunk column must not depend on i");
    constexpr int ORS = 256 / OCPR;
    const int orow0 = mt / OCPR, oc = mt % OCPR;
    const int ocol = n_slice + oc * (16 / OS);
    const unsigned int ocolb = (unsigned int)(ocol * OS);           // byte offset inside an output row
    const unsigned int ocbad = ocol < p.N ? 0u : 0xFFFFFFFFu;
    auto ols = [&](int i) { const int row = orow0 + i * ORS; return row * OROWB + ((oc ^ (row & 7)) * 16); };
    // epilogue operands, requested two steps before their tile is drained
    u32x4 ex[ED][EPI == WS_EPI_PLAIN ? 1 : O_CH];
    unsigned int rcbits[ED];     // bit i: row of chunk i is skipped (ROWSKIP)
    // ROWSKIP off: read (and ignore) bytes of X instead of a null rowcode pointer
    const unsigned int rsmask = (p.flags & MFP_GEMM_ROWSKIP) ? 0xFFu : 0u;
    const unsigned char* rcp = (p.flags & MFP_GEMM_ROWSKIP) ? p.rowcode : reinterpret_cast<const unsigned char*>(p.A);
    auto xload = [&](int set, int t) {
      const int row0 = row_beg + t * BM + orow0;
      const int tok = ((t - ntiles) >> 31) & ~(t >> 31);
      unsigned int bits = 0;
#pragma unroll
      for (int i = 0; i < O_CH; ++i) {
        const int row = row0 + i * ORS;
        const unsigned int bad = ocbad | ~(unsigned int)(tok & ((row - row_end) >> 31));
        if (EPI == WS_EPI_F32X) bits |= ((rcp[max(0, min(row, p.M - 1))] & rsmask) ? 1u : 0u) << i;
        if (EPI != WS_EPI_PLAIN) {   // same element size as the output (f32 residual / bf16 mask)
          const unsigned int vo = ((unsigned int)(row * p.ldc * OS) + ocolb) | bad;
          ex[set][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, vo, 0, 0));
        }
      }
      rcbits[set] = bits;
    };

    // ---- prologue: the weight image is what the first MFMA waits for, so it goes ahead of the X tiles
    {
      u32x4 wv[2][WCH];
      w_load(0, wv[0]);
      w_load(1, wv[1]);
#pragma unroll
      for (int i = 0; i < XD; ++i) gload(xa[i], i);
#pragma unroll
      for (int r = 0; r < 2; ++r) {       // weight staging rounds (barriers mirror the math waves)
        w_store(wv[r]);
        __syncthreads();
        __syncthreads();
      }
    }
    if (EPI != WS_EPI_PLAIN) xload(0, 0);
    lstore(xa[0], 0);
    gload(xa[0], XD);
    const float inv_keep = DROPOUT ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
    const unsigned long long rng_off =
        p.offset + ((DROPOUT && p.step_ptr) ? (unsigned long long)(*p.step_ptr) * MFP_RNG_STEP_STRIDE : 0ull);
    const unsigned int dkey = drop_key(p.seed, rng_off), dthr = drop_thr16(p.dropout_p);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), see the math waves
    __syncthreads();

    // step t: X(t+1) registers -> LDS stage; X(t+1+XD) global -> registers; operands of tile t+1
    // requested; output tile t-1: LDS stage -> epilogue -> global
    auto step = [&](auto tc, int t) {
      constexpr int cur = decltype(tc)::value & 1, xi = (decltype(tc)::value + 1) % XD;
      constexpr int eset = (decltype(tc)::value + ED - 1) % ED, pset = (decltype(tc)::value + 1) % ED;
      lstore(xa[xi], cur ^ 1);   // X(t+1): loaded XD steps ago
      gload(xa[xi], t + 1 + XD);
      const int row0 = row_beg + (t - 1) * BM + orow0;
      const int tok = ~((t - 1) >> 31);            // 0 at step 0 (no tile -1)
      const unsigned char* orr = ost + (cur ^ 1) * OSTAGE;
#pragma unroll
      for (int i = 0; i < O_CH; ++i) {
        const int row = row0 + i * ORS;
        const unsigned int bad = ocbad | ~(unsigned int)(tok & ((row - row_end) >> 31));
        u32x4 raw = *reinterpret_cast<const u32x4*>(orr + ols(i));
        if (EPI == WS_EPI_RELUBWD) {   // bf16 values, bf16 mask: keep where the saved activation > 0
          const u32x4 h = ex[eset][i];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned int lo = bf16_to_f32((unsigned short)(h[r] & 0xffffu)) > 0.f ? 0xFFFFu : 0u;
            const unsigned int hi = bf16_to_f32((unsigned short)(h[r] >> 16)) > 0.f ? 0xFFFF0000u : 0u;
            raw[r] &= lo | hi;
          }
        }
        if (EPI == WS_EPI_F32X) {
          f32x4 x = __builtin_bit_cast(f32x4, raw);
          const bool skip = ((rcbits[eset] >> i) & 1u) != 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = skip ? 0.f : x[r];
          if (DROPOUT) {
            bool keep[4];
            drop_keep4(drop_row(dkey, (unsigned int)row), ocolb >> 2, dthr, keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = keep[r] ? x[r] * inv_keep : 0.f;
          }
          x += __builtin_bit_cast(f32x4, ex[eset][i]);
          raw = __builtin_bit_cast(u32x4, x);
        }
        const unsigned int vo = ((unsigned int)(row * p.ldc * OS) + ocolb) | bad;
        __builtin_amdgcn_raw_buffer_store_b128(raw, rsc, vo, 0, 0);
      }
      if (EPI != WS_EPI_PLAIN) xload(pset, t + 1);   // after the drain: set `eset` is dead, 3 sets live
      __syncthreads();
    };
    int t = 0;
    for (; t + 3 <= ntiles; t += 4) {
      step(std::integral_constant<int, 0>{}, t);
      step(std::integral_constant<int, 1>{}, t + 1);
      step(std::integral_constant<int, 2>{}, t + 2);
      step(std::integral_constant<int, 3>{}, t + 3);
    }
    if (t <= ntiles) step(std::integral_constant<int, 0>{}, t);
    if (t + 1 <= ntiles) step(std::integral_constant<int, 1>{}, t + 1);
    if (t + 2 <= ntiles) step(std::integral_constant<int, 2>{}, t + 2);
  }
#ifdef MFP_GEMM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WS_STAMP();
#endif
}

// Host side: can this call take the weight-stationary kernel?
inline bool ws_eligible(const mfp_gemm_args* a, int splitk) {
  if (!(a->a_kmajor && a->b_kmajor) || a->in_dtype != MFP_BF16 || splitk != 1) return false;
  // K = 1024 (d_model 512: FFN2 forward, FFN1 input gradient) instantiates and is correct (KS = 32, one X tile
  // in flight: two spill), but measured 46 us against 41 us for the tile kernel at T = 16384 -- not routed here
  if (a->K != 256 && a->K != 512 && a->K != 768) return false;
  if (a->K == 768 && (a->flags & ~(MFP_GEMM_BIAS | MFP_GEMM_RELU))) return false;   // plain epilogue only
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

template <int KS, int MT, int EPI, bool DROPOUT, bool OUT_BF16, bool NARROW = false>
int launch_ws(const GemmParams& p, int ncu, hipStream_t st) {
  constexpr int NQ = (KS == 8 ? 4 : (KS == 16 ? 2 : 1)) >> (NARROW ? 1 : 0), BN = 64 * NQ, XSTAGE = 16 * MT * 64 * KS, OSTAGE = 16 * MT * BN * (OUT_BF16 ? 2 : 4);
  constexpr int lds = 2 * XSTAGE + 2 * OSTAGE;
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (lds > 64 * 1024 && !attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel<KS, MT, EPI, DROPOUT, OUT_BF16, NARROW>),
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
  hipLaunchKernelGGL((gemm_ws_kernel<KS, MT, EPI, DROPOUT, OUT_BF16, NARROW>), dim3(groups * slices), dim3(512), lds, st, p,
                     groups, slices);
  return MFP_OK;
}

template <int KS, int MT, bool NARROW = false>
int launch_ws_epi(const mfp_gemm_args* a, const GemmParams& p, int ncu, hipStream_t st) {
  const int f = a->flags;
  if (f & (MFP_GEMM_RESIDUAL | MFP_GEMM_ACCUM))
    return (f & MFP_GEMM_DROPOUT) ? launch_ws<KS, MT, WS_EPI_F32X, true, false, NARROW>(p, ncu, st)
                                  : launch_ws<KS, MT, WS_EPI_F32X, false, false, NARROW>(p, ncu, st);
  if (f & MFP_GEMM_RELU_BWD) return launch_ws<KS, MT, WS_EPI_RELUBWD, false, true, NARROW>(p, ncu, st);
  return a->out_dtype == MFP_BF16 ? launch_ws<KS, MT, WS_EPI_PLAIN, false, true, NARROW>(p, ncu, st)
                                  : launch_ws<KS, MT, WS_EPI_PLAIN, false, false, NARROW>(p, ncu, st);
}
