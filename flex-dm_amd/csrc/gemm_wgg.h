// Grouped streaming weight-gradient GEMM with the split-K reduction INSIDE the launch (gfx950, bf16).
//
// One launch computes up to WGG_MAX_JOBS independent products C_j[M_j][N_j] = A_j[K][M_j]^T B_j[K][N_j]
// over the same token dimension K -- the four weight gradients of a DeepSVG block (reference
// architecture/transformer.py:85-98,163-169: dWq|k|v, dWo, dW1, dW2), or the heads / encoder /
// embedding-table gradients (decoder.py:39-43, encoder.py:74-92,156-160) -- plus their bias
// gradients (column sums of A).
//
// Why grouped: one product has 4-12 output tiles of 128 x 128, so filling 256 CUs needed a 21- to
// 64-way split of the tokens, i.e. 8-16 k-tiles per workgroup wrapped in fixed costs (pipeline fill,
// partial-tile store) and 265 MB of f32 partials per step that a second kernel re-read (r01:
// gemm_wg_kernel 563 us + splitk_reduce_kernel 200 us per step).  The 32 tiles of a block's four
// products fill the chip with an 8-way split: 64 k-tiles per workgroup, 8x fewer partial bytes.
//
// Reduction: every workgroup stores its partial tile as a 64 KB slab ws[kz][tile] with write-through
// (sc1) stores, drains them and draws a ticket for the tile; the workgroup that draws the last ticket
// acquires, sums the slabs in FIXED order kz = 0 .. splitk-1 (data-parallel replicas stay bit-identical)
// and writes the gradient.  Placement-independent (any distribution of a tile's chunks over XCDs /
// CUs); the ticket word is reset by the last arriver, so the ticket array stays all-zero between
// launches.  Main loop = gemm_wg.h (4 memory waves streaming k-tiles, 4 math waves on MFMA).
//
// Grid: 1-D, block b -> XCD b & 7; k-slice kz = (j / ntiles) * 8 + xcd, tile = j % ntiles (j = b >> 3):
// all tiles of a k-slice run on one XCD, so each operand panel is fetched from HBM once and re-read
// from that XCD's L2 by the tiles that share it.  A k-slice is CYCLIC -- k-tiles kz, kz + splitk,
// kz + 2 splitk, ... of 64 tokens -- not a contiguous chunk: with contiguous chunks the 8 XCDs walk
// addresses a power-of-two distance (chunk x row pitch = 2-4 MB) apart in lockstep and pile onto the
// same HBM channels (measured: 1.7 us per k-tile, 1 TB/s for the whole chip, against 0.8 us with the
// short chunks of the ungrouped launches); cyclic slices make the chip sweep the operands front to
// back, neighbouring XCDs on neighbouring 32-64 KB blocks.
#pragma once

constexpr int WGG_MAX_JOBS = 16;      // (round 6: 16 -- the four products of up to four blocks in one launch)
#ifndef MFP_WGG_XD
#define MFP_WGG_XD 4
#endif
constexpr int WGG_XD = MFP_WGG_XD;   // k-tiles in flight (registers) per memory wave
constexpr int WGG_ZPAD = 2112;   // floats: the slabs of one tile sit (ntiles * 64 KB + 8.25 KB) apart

template <int I, int N, typename F>
__device__ __forceinline__ void wgg_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    wgg_static_for<I + 1, N>(f);
  }
}

struct WggJob {
  const unsigned short* A;      // bf16 [K][lda], M columns
  const unsigned short* B;      // bf16 [K][ldb], N columns
  float* C;                     // f32 [M][ldc]
  float* colsum;                // f32 [M] or nullptr
  const unsigned char* rowcode; // u8 [K] or nullptr: rows of A with a non-zero code count as zero
  int M, N, lda, ldb, ldc, tiles_n, tile0, pad_;
};

struct WggParams {
  WggJob job[WGG_MAX_JOBS];
  float* ws;                    // [splitk][zstride]: partial tiles [ntiles][128 * 128] (+ pad: slabs of one tile off a power-of-two stride)
  float* ws_col;                // [splitk][ntiles][128] partial column sums
  unsigned int* tickets;        // [ntiles], zero on entry, zero on exit
  long long zstride;            // floats between the slabs of consecutive k-slices
  int njobs, ntiles, K, nk_max, splitk;   // nk_max = k-tiles of the longest k-slice
  int tpg;                                // splitk < 8: tiles per XCD group = ceil(ntiles * splitk / 8)
#ifdef MFP_GEMM_TRACE
  unsigned long long* trace;   // [workgroup][24] s_memrealtime stamps (100 MHz) of thread 0
#endif
};

#ifndef MFP_WGG_OCC
#define MFP_WGG_OCC 2
#endif
// DEFER: the workgroup only publishes its partial slab (plain stores: the kernel boundary makes them visible) and exits;
// wgg_reduce_kernel sums the slabs of every deferred group of the step in ONE launch at the end of the backward pass.
// Per launch that removes the write-through drain, the ticket and the last arriver's serial read of splitk slabs
// (~10 us of the ~18 us a grouped launch costs beside its k-loop).
template <int XD, bool ROWSKIP, bool DEFER>
__global__ __launch_bounds__(512, MFP_WGG_OCC) void gemm_wgg_kernel(WggParams p) {
  constexpr int BM = 128, BN = 128, BK = 64, PAD = 8, LDS_S = BM + PAD;
  constexpr int TILE_E = BK * LDS_S;
  constexpr int STAGE_B = 2 * TILE_E * 2;
  constexpr int CH = BK * (BM / 8) / 256;
  constexpr int CS_LD = BN + 4;
  static_assert(CH == 4, "64 x 128 tile = 1024 chunks over 256 memory threads");
  static_assert(BM * CS_LD * 4 + 16 <= 2 * STAGE_B, "output stage + the last-arriver flag alias the two operand stages");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ float colsum_s[16][BM];
  __shared__ unsigned char rc_s[ROWSKIP ? WG_MAX_KCHUNK : 16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
#ifdef MFP_GEMM_TRACE
  int trace_i = 0;
#define WGG_STAMP() do { if (tid == 0 && trace_i < 24) p.trace[(long long)blockIdx.x * 24 + trace_i++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WGG_STAMP() do {} while (0)
#endif
  WGG_STAMP();   // 0: start
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  int kz, tile;
  if (p.splitk >= 8) {          // splitk % 8 == 0 (host): XCD x runs the k-slices x, x + 8, ... of every tile
    kz = (j / p.ntiles) * 8 + xcd; tile = j % p.ntiles;
  } else {                      // splitk 1 | 2 | 4 (many tiles: c5): an XCD runs ONE k-slice of a contiguous group of tiles
    kz = xcd % p.splitk; tile = (xcd / p.splitk) * p.tpg + j;       // (neighbouring tiles share operand panels)
    if (tile >= p.ntiles || tile >= (xcd / p.splitk + 1) * p.tpg) return;
  }
  int ji = 0;
  for (int q = 1; q < p.njobs; ++q) ji = tile >= p.job[q].tile0 ? q : ji;
  const WggJob& jb = p.job[ji];
  const int M = jb.M, N = jb.N, lda = jb.lda, ldb = jb.ldb;
  const int bid = tile - jb.tile0;
  const int tm = bid / jb.tiles_n, tn = bid % jb.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int Kj = p.K;
  const int ktiles = (Kj + BK - 1) / BK;
  const int nk = kz < ktiles ? (ktiles - kz + p.splitk - 1) / p.splitk : 0;      // k-tiles kz, kz + splitk, ...
  const int kend = Kj;
  const bool do_colsum = jb.colsum != nullptr && tn == 0;
  const bool skip_rows = ROWSKIP && jb.rowcode != nullptr;
  f32x4 acc[4][4];
  if (ROWSKIP) {
    for (int i = tid; i < p.nk_max * BK; i += 512) {      // row codes of this slice's tokens, k-tile by k-tile
      const int k = ((i >> 6) * p.splitk + kz) * BK + (i & 63);
      rc_s[i] = (skip_rows && k < Kj) ? jb.rowcode[k] : 0;
    }
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
    WGG_STAMP();   // 1: first tile staged
    for (int t = 0; t < nk; ++t) {
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
      if ((t & 15) == 15) WGG_STAMP();   // 2..: every 16th k-tile
    }
    float* Cs = reinterpret_cast<float*>(smem_raw);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        *reinterpret_cast<f32x4*>(&Cs[(wm * 64 + a * 16 + li) * CS_LD + wn * 64 + lg * 16 + b * 4]) = acc[a][b];
  } else {
    // ====================================================================== MEMORY waves
    const int mt = tid - 256;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(jb.A), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(jb.B), 0, 0x7FFFFFFF, 0x00020000);
    const int krow0 = mt >> 4, ccol = (mt & 15) * 8;
    const unsigned int abad = m0 + ccol < M ? 0u : 0xFFFFFFFFu, bbad = n0 + ccol < N ? 0u : 0xFFFFFFFFu;
    const unsigned int voa0 = (unsigned int)((krow0 * lda + m0 + ccol) * 2);
    const unsigned int vob0 = (unsigned int)((krow0 * ldb + n0 + ccol) * 2);
    const int ls0 = (krow0 * LDS_S + ccol) * 2;
    u32x4 ra[XD][CH], rb[XD][CH];
    float csum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[e] = 0.f;
    auto gload = [&](int set, int t) {
      const int k0 = (t * p.splitk + kz) * BK;
      const int live = (t - nk) >> 31;                      // -1 while t < nk
      const int soa = (k0 * lda * 2) & live, sob = (k0 * ldb * 2) & live;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int k = k0 + krow0 + 16 * c;
        const unsigned int kbad = ~(unsigned int)(live & ((k - kend) >> 31));
        const unsigned int skip = (ROWSKIP && rc_s[min(t, p.nk_max - 1) * BK + krow0 + 16 * c]) ? 0xFFFFFFFFu : 0u;
        ra[set][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rsa, (voa0 + (unsigned int)(16 * c * lda * 2)) | abad | kbad | skip, soa, 0));
        rb[set][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rsb, (vob0 + (unsigned int)(16 * c * ldb * 2)) | bbad | kbad, sob, 0));
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
    auto step = [&](auto tc, int t) {
      constexpr int xi = (decltype(tc)::value + 1) % XD;
      lstore(xi, (t + 1) & 1);
      gload(xi, t + 1 + XD);
      __syncthreads();
    };
    // the step loop is unrolled by XD: register set (t + 1) % XD must be a compile-time index
    int t = 0;
    for (; t + XD - 1 < nk; t += XD) wgg_static_for<0, XD>([&](auto ic) { step(ic, t + decltype(ic)::value); });
    wgg_static_for<0, XD - 1>([&](auto ic) { if (t + decltype(ic)::value < nk) step(ic, t + decltype(ic)::value); });
    if (do_colsum) {
#pragma unroll
      for (int e = 0; e < 8; ++e) colsum_s[krow0][ccol + e] = csum[e];
    }
  }
  __syncthreads();   // partial tile (and column sums) are in LDS
  WGG_STAMP();   // tile in LDS

  // (no further static __shared__ object: statics of a size that is not a multiple of 16 would shift the
  // dynamic region off its 16-byte alignment and ds_read_b64_tr_b16 would silently read the wrong bytes)
  volatile int* last_s = reinterpret_cast<volatile int*>(smem_raw + BM * CS_LD * 4);
  // ---- publish the partial tile: slab ws[kz][tile] (whole 128 x 128, 512 B rows), then a ticket.
  // Write-through (sc1) 16-byte stores, drained by every wave, need no release fence before the ticket
  // (a plain-store slab + agent-scope release cost ~5 us more per workgroup: the fence writes the whole
  // L2's dirty lines back).
  const int r0 = tid >> 5, c4 = (tid & 31) * 4;
  {
    const float* Cs = reinterpret_cast<const float*>(smem_raw);
    float* slab = p.ws + kz * p.zstride + (long long)tile * (BM * BN);
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc(slab, 0, BM * BN * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < BM / 16; ++i) {
      const int row = r0 + 16 * i;
      __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&Cs[row * CS_LD + c4]), rss,
                                             (unsigned int)((row * BN + c4) * 4), 0, DEFER ? 0 : 16 /* sc1 */);
    }
    if (do_colsum && tid < BM) {
      float s = 0.f;
#pragma unroll
      for (int gI = 0; gI < 16; ++gI) s += colsum_s[gI][tid];
      __hip_atomic_store(&p.ws_col[((long long)kz * p.ntiles + tile) * BM + tid], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (DEFER) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its write-through stores
  __syncthreads();
  WGG_STAMP();   // slab drained
  if (tid == 0) {
    const unsigned int ticket = __hip_atomic_fetch_add(&p.tickets[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == (unsigned int)(p.splitk - 1);
    if (last) {
      // Hand-off form (MI355X_MICROARCH.md, "Valid forms"): producers store the slabs write-through (sc1) and drain
      // them (vmcnt(0)) before a relaxed agent-scope ticket; the consumer does ONE agent acquire (buffer_inv sc1:
      // invalidates this CU's whole L1, for every wave of the workgroup) followed by the workgroup barrier below,
      // then plain loads.  gfx950-specific: the library is built for gfx950 only (Makefile ARCH).
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(&p.tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next launch
    }
    *last_s = last;
  }
  __syncthreads();
  WGG_STAMP();   // ticket drawn
  if (!*last_s) return;

  // ---- last arriver: C[m][n] = sum over kz (ascending) of the slabs; colsum likewise.  A thread owns 8
  // float4 of the tile and keeps 4 slabs of them (32 loads) in flight: the slabs come from other XCDs'
  // write-throughs, i.e. from memory, and a loop with one slab row in flight (8 dependent round trips
  // of ~3 us) was HALF of this kernel's time.
  {
    const long long zstride = p.zstride;
    const float* src = p.ws + (long long)tile * (BM * BN) + r0 * BN + c4;
    f32x4 acc8[BM / 16];
#pragma unroll
    for (int i = 0; i < BM / 16; ++i) acc8[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int RZ = MFP_WGG_OCC >= 4 ? 2 : 4;      // slabs in flight per round (register budget)
    for (int z0 = 0; z0 < p.splitk; z0 += RZ) {
      f32x4 v[RZ][BM / 16];
#pragma unroll
      for (int u = 0; u < RZ; ++u)
#pragma unroll
        for (int i = 0; i < BM / 16; ++i)
          v[u][i] = z0 + u < p.splitk ? *reinterpret_cast<const f32x4*>(src + (z0 + u) * zstride + 16 * i * BN) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < RZ; ++u)
        if (z0 + u < p.splitk) {
#pragma unroll
          for (int i = 0; i < BM / 16; ++i) { acc8[i][0] += v[u][i][0]; acc8[i][1] += v[u][i][1]; acc8[i][2] += v[u][i][2]; acc8[i][3] += v[u][i][3]; }
        }
    }
#pragma unroll
    for (int i = 0; i < BM / 16; ++i) {
      const int row = r0 + 16 * i;
      if (m0 + row < M && n0 + c4 < N)
        *reinterpret_cast<f32x4*>(jb.C + (long long)(m0 + row) * jb.ldc + n0 + c4) = acc8[i];
    }
    if (do_colsum && tid < BM && m0 + tid < M) {
      float s = 0.f;
      for (int z = 0; z < p.splitk; ++z) s += p.ws_col[((long long)z * p.ntiles + tile) * BM + tid];
      jb.colsum[m0 + tid] = s;
    }
  }
#ifdef MFP_GEMM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WGG_STAMP();   // reduced (last arrivers only)
#endif
}

template <bool ROWSKIP, bool DEFER>
inline int launch_wgg_t(const WggParams& p, hipStream_t st) {
  constexpr int lds = 2 * (2 * 64 * 136 * 2);
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgg_kernel<WGG_XD, ROWSKIP, DEFER>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_wgrad_group: cannot raise dynamic LDS to %d: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_wgg_kernel<WGG_XD, ROWSKIP, DEFER>), dim3(p.splitk >= 8 ? p.ntiles * p.splitk : 8 * p.tpg), dim3(512), lds, st, p);
  return MFP_OK;
}

// ------------------------------------------------------------------ 256 x 128 macro tiles (many-tile groups: d_model 512)
// The same pipeline on a macro tile = two 128 x 128 tiles stacked in M (tiles t and t + tiles_n of one job: every job of the
// group must have an even number of tile rows).  At c5 a block's four products are 128 tiles; with 128 x 128 units and a
// split of 2 every workgroup walks 128 k-tiles at the pipeline's ~0.75 us per k-tile whatever the tile holds (100 us per
// group), and the 12 / 4 tiles of a row / column of tiles pull 1.07 GB out of the L2s for 201 MB of operands.  A macro
// tile does twice the products per k-tile step (A 64 x 256 + B 64 x 128 = 48 KB staged, 64 MFMAs per math wave and
// k-tile) over half as many steps per workgroup.  Slabs are written as the two standard tiles, so the workspace layout,
// mfp_wgrad_reduce and the column sums are those of gemm_wgg_kernel.  Deferred reduction only, no row masks.
template <int XD>
__global__ __launch_bounds__(512, 2) void gemm_wgt_kernel(WggParams p) {
  constexpr int BN = 128, BK = 64, LDA_S = 256 + 8, LDB_S = 128 + 8;
  constexpr int A_E = BK * LDA_S, B_E = BK * LDB_S;
  constexpr int STAGE_B = (A_E + B_E) * 2;
  constexpr int CS_LD = BN + 4;
  static_assert(128 * CS_LD * 4 <= 2 * STAGE_B, "output stage aliases the two operand stages");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* const colsum_s = reinterpret_cast<float*>(smem_raw + 2 * STAGE_B);      // [16][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int mtiles = p.ntiles >> 1;
  int kz, mt_i;
  if (p.splitk >= 8) { kz = (j / mtiles) * 8 + xcd; mt_i = j % mtiles; }
  else {
    kz = xcd % p.splitk; mt_i = (xcd / p.splitk) * p.tpg + j;       // (p.tpg: MACRO tiles per XCD group)
    if (mt_i >= mtiles) return;
  }
  int ji = 0;
  for (int q = 1; q < p.njobs; ++q) ji = mt_i >= (p.job[q].tile0 >> 1) ? q : ji;
  const WggJob& jb = p.job[ji];
  const int M = jb.M, N = jb.N, lda = jb.lda, ldb = jb.ldb;
  const int bid = mt_i - (jb.tile0 >> 1);
  const int tm2 = bid / jb.tiles_n, tn = bid % jb.tiles_n;
  const int tile_lo = jb.tile0 + (2 * tm2) * jb.tiles_n + tn;      // the standard tiles this macro tile is made of:
  const int m0 = tm2 * 256, n0 = tn * BN;                            // tile_lo (rows m0 ..) and tile_lo + tiles_n (rows m0 + 128 ..)
  const int ktiles = (p.K + BK - 1) / BK, kend = p.K;
  const int nk = kz < ktiles ? (ktiles - kz + p.splitk - 1) / p.splitk : 0;
  const bool do_colsum = jb.colsum != nullptr && tn == 0;
  f32x4 acc[8][4];

  if (wave < 4) {
    // ======================================================================== MATH waves: 128 x 64 of the macro tile each
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();   // prologue barrier (stage 0 filled)
    for (int t = 0; t < nk; ++t) {
      const unsigned short* As = reinterpret_cast<const unsigned short*>(smem_raw + (t & 1) * STAGE_B);
      const unsigned short* Bs = As + A_E;
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 xf[8], wf[4];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const unsigned short* ptr = &As[(ks * 32 + lg * 8 + (li >> 2)) * LDA_S + wm * 128 + a * 16 + (li & 3) * 4];
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 4 * LDA_S));
          xf[a] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const unsigned short* ptr = &Bs[(ks * 32 + lg * 8 + (li >> 2)) * LDB_S + wn * 64 + (li & 3) * 16 + b * 4];
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 4 * LDB_S));
          wf[b] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
      }
      __syncthreads();
    }
  } else {
    // ====================================================================== MEMORY waves
    const int mt = tid - 256;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(jb.A), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(jb.B), 0, 0x7FFFFFFF, 0x00020000);
    const int krow0 = mt >> 4, ccol = (mt & 15) * 8;
    const unsigned int abad0 = m0 + ccol < M ? 0u : 0xFFFFFFFFu, abad1 = m0 + 128 + ccol < M ? 0u : 0xFFFFFFFFu;
    const unsigned int bbad = n0 + ccol < N ? 0u : 0xFFFFFFFFu;
    const unsigned int voa0 = (unsigned int)((krow0 * lda + m0 + ccol) * 2);
    const unsigned int vob0 = (unsigned int)((krow0 * ldb + n0 + ccol) * 2);
    const int lsa0 = (krow0 * LDA_S + ccol) * 2, lsb0 = (krow0 * LDB_S + ccol) * 2;
    u32x4 ra[XD][2][4], rb[XD][4];
    float csum[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e) csum[h][e] = 0.f;
    auto gload = [&](int set, int t) {
      const int k0 = (t * p.splitk + kz) * BK;
      const int live = (t - nk) >> 31;                      // -1 while t < nk
      const int soa = (k0 * lda * 2) & live, sob = (k0 * ldb * 2) & live;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = k0 + krow0 + 16 * c;
        const unsigned int kbad = ~(unsigned int)(live & ((k - kend) >> 31));
        const unsigned int oa = voa0 + (unsigned int)(16 * c * lda * 2);
        ra[set][0][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, oa | abad0 | kbad, soa, 0));
        ra[set][1][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, (oa + 256u) | abad1 | kbad, soa, 0));
        rb[set][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rsb, (vob0 + (unsigned int)(16 * c * ldb * 2)) | bbad | kbad, sob, 0));
      }
    };
    auto lstore = [&](int set, int stage) {
      unsigned char* st = smem_raw + stage * STAGE_B;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        *reinterpret_cast<u32x4*>(st + lsa0 + 16 * c * LDA_S * 2) = ra[set][0][c];
        *reinterpret_cast<u32x4*>(st + lsa0 + 256 + 16 * c * LDA_S * 2) = ra[set][1][c];
        *reinterpret_cast<u32x4*>(st + A_E * 2 + lsb0 + 16 * c * LDB_S * 2) = rb[set][c];
        if (do_colsum) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned int w = ra[set][h][c][e];
              csum[h][2 * e] += bf16_to_f32((unsigned short)(w & 0xffffu));
              csum[h][2 * e + 1] += bf16_to_f32((unsigned short)(w >> 16));
            }
        }
      }
    };
#pragma unroll
    for (int i = 0; i < XD; ++i) gload(i, i);
    lstore(0, 0);
    gload(0, XD);
    __syncthreads();   // prologue barrier
    auto step = [&](auto tc, int t) {
      constexpr int xi = (decltype(tc)::value + 1) % XD;
      lstore(xi, (t + 1) & 1);
      gload(xi, t + 1 + XD);
      __syncthreads();
    };
    int t = 0;
    for (; t + XD - 1 < nk; t += XD) wgg_static_for<0, XD>([&](auto ic) { step(ic, t + decltype(ic)::value); });
    wgg_static_for<0, XD - 1>([&](auto ic) { if (t + decltype(ic)::value < nk) step(ic, t + decltype(ic)::value); });
    if (do_colsum) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) colsum_s[krow0 * 256 + h * 128 + ccol + e] = csum[h][e];
    }
  }
  // ---- publish: the two standard tiles one after the other through the output stage (128 x 128 f32, aliases the stages)
  const int r0 = tid >> 5, c4 = (tid & 31) * 4;
  float* Cs = reinterpret_cast<float*>(smem_raw);
  for (int hs = 0; hs < 2; ++hs) {
    __syncthreads();      // k-loop done (hs = 0) / the previous half has left the stage (hs = 1)
    if (wave < 4 && (wave >> 1) == hs) {
      const int wn = wave & 1;
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          *reinterpret_cast<f32x4*>(&Cs[(a * 16 + li) * CS_LD + wn * 64 + lg * 16 + b * 4]) = acc[a][b];
    }
    __syncthreads();
    float* slab = p.ws + kz * p.zstride + (long long)(tile_lo + hs * jb.tiles_n) * (128 * BN);
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc(slab, 0, 128 * BN * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = r0 + 16 * i;
      __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&Cs[row * CS_LD + c4]), rss,
                                             (unsigned int)((row * BN + c4) * 4), 0, 0);
    }
  }
  if (do_colsum && tid < 256) {
    float s = 0.f;
#pragma unroll
    for (int gI = 0; gI < 16; ++gI) s += colsum_s[gI * 256 + tid];
    p.ws_col[((long long)kz * p.ntiles + tile_lo + (tid >> 7) * jb.tiles_n) * 128 + (tid & 127)] = s;
  }
}

inline int launch_wgt(const WggParams& p, hipStream_t st) {
  constexpr int lds = 2 * ((64 * 264 + 64 * 136) * 2) + 16 * 256 * 4;
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgt_kernel<WGG_XD>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_wgrad_group_partial: cannot raise dynamic LDS to %d: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  const int mtiles = p.ntiles / 2;
  hipLaunchKernelGGL((gemm_wgt_kernel<WGG_XD>), dim3(p.splitk >= 8 ? mtiles * p.splitk : 8 * p.tpg), dim3(512), lds, st, p);
  return MFP_OK;
}

// ------------------------------------------------------------------ deferred split-K reduction, all groups of a step
constexpr int WGR_MAX_GROUPS = 8;
struct WgrJob { float* C; float* colsum; const float* nfix; int M, N, ldc, tiles_n, tile0, pad_; };
constexpr int WGR_MAX_JOBS = 56;      // jobs of all groups of one reduction launch (kernel arguments: 4 KB at most)
struct WgrGroup {
  const float* ws; const float* ws_col;
  long long zstride;
  int splitk, ntiles, unit0, njobs, job0, pad_;      // job0: the group's first entry of WgrParams::job
};
struct WgrParams { WgrGroup g[WGR_MAX_GROUPS]; WgrJob job[WGR_MAX_JOBS]; int ngroups, nunits; };
static_assert(sizeof(WgrParams) <= 4000, "kernel arguments");

// One workgroup per (tile, 16-row slice): 8 KB of the gradient = the sum of `splitk` slab pieces in FIXED order
// kz = 0 .. splitk-1 (bit-identical to the in-launch last-arriver sum), all of a thread's loads in flight.
__global__ __launch_bounds__(256) void wgg_reduce_kernel(WgrParams p) {
  const int unit = blockIdx.x, tid = threadIdx.x;
  int gi = 0;
  for (int q = 1; q < p.ngroups; ++q) gi = unit >= p.g[q].unit0 ? q : gi;
  const WgrGroup& G = p.g[gi];
  const int local = unit - G.unit0, tile = local >> 3, slice = local & 7;
  int ji = 0;
  for (int q = 1; q < G.njobs; ++q) ji = tile >= p.job[G.job0 + q].tile0 ? q : ji;
  const WgrJob& jb = p.job[G.job0 + ji];
  const int bid = tile - jb.tile0, tm = bid / jb.tiles_n, tn = bid % jb.tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  // this thread's two float4 of the tile: slab offsets (floats) and the (row, column) they belong to
  int off[2], rowh[2], colh[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    rowh[h] = slice * 16 + (tid >> 5) + 8 * h; colh[h] = (tid & 31) * 4;
    off[h] = rowh[h] * 128 + colh[h];
  }
  const float* src = G.ws + (long long)tile * (128 * 128);
  f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  for (int z0 = 0; z0 < G.splitk; z0 += 8) {
    f32x4 v[8][2];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        v[u][h] = z0 + u < G.splitk ? *reinterpret_cast<const f32x4*>(src + (z0 + u) * G.zstride + off[h]) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (z0 + u < G.splitk) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { acc[h][0] += v[u][h][0]; acc[h][1] += v[u][h][1]; acc[h][2] += v[u][h][2]; acc[h][3] += v[u][h][3]; }
      }
  }
  if (jb.nfix != nullptr && n0 + colh[0] < jb.N) {
    // the B operand was x-hat = (x - mean) rstd instead of y = x-hat gamma + beta (mfp_wgrad_job::n_affine): C[m][n] =
    // gamma[n] sum_t A[t][m] xhat[t][n] + beta[n] sum_t A[t][m]; the row sums are this job's bias-gradient partials
    const f32x4 gm = *reinterpret_cast<const f32x4*>(jb.nfix + n0 + colh[0]);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(jb.nfix + jb.N + n0 + colh[0]);
    const int tile_c = jb.tile0 + tm * jb.tiles_n;      // the tile of this row block that carries the column-sum partials
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float sr = 0.f;
      for (int z = 0; z < G.splitk; ++z) sr += G.ws_col[((long long)z * G.ntiles + tile_c) * 128 + rowh[h]];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[h][e] = gm[e] * acc[h][e] + bt[e] * sr;
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (m0 + rowh[h] < jb.M && n0 + colh[h] < jb.N)
      *reinterpret_cast<f32x4*>(jb.C + (long long)(m0 + rowh[h]) * jb.ldc + n0 + colh[h]) = acc[h];
  if (slice == 0 && jb.colsum != nullptr && tn == 0 && tid < 128 && m0 + tid < jb.M) {
    float s = 0.f;
    for (int z = 0; z < G.splitk; ++z) s += G.ws_col[((long long)z * G.ntiles + tile) * 128 + tid];
    jb.colsum[m0 + tid] = s;
  }
}
