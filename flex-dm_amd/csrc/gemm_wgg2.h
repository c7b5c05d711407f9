// Grouped weight-gradient GEMM, second generation: the protocol of gemm_wgg.h (jobs, 128 x 128 output tiles,
// cyclic k-slices, write-through partial slabs, tickets, last-arriver reduction in a fixed order) with a different
// engine inside the workgroup:
//
//   * a workgroup owns a MACRO tile of 256 x 128 (two vertically adjacent output tiles of one job): per 64-token
//     k-tile it ingests 48 KB for 2 x the products of a 128 x 128 tile (32 KB) -- the grouped launch of gemm_wgg.h
//     pulls 537 MB through the CUs for 217 MB of operands and sits at 34 GB/s per CU;
//   * ALL eight waves multiply (64 x 64 each, two per SIMD) AND stream: every wave keeps its share (6 x 16 B per
//     lane) of WGG2_XD k-tiles in flight in registers and drops one into the LDS stage of the next k-tile per
//     iteration.  What bounds this product is bytes in flight per CU against the memory latency under load
//     (~4-5 us): a first version that moved the k-tiles global -> LDS by LDS-DMA through a ring of three 48 KB stages
//     had only two k-tiles (96 KB) in flight per CU and ran at 2.3 us per k-tile (88 us per block group, no bank
//     conflicts, matrix pipe as busy as before); registers hold what LDS cannot;
//   * the LDS images are unpadded [64][256] / [64][128] bf16, XOR-swizzled in 32-byte groups by the token row: the
//     transposing fragment reads (ds_read_b64_tr_b16: 4 rows x 32 B per 16 lanes, 8 rows per half wave) hit 8
//     different bank groups (SQ_LDS_BANK_CONFLICT = 0; the padded images of gemm_wgg.h: 4.2 M per launch);
//   * bias gradients (column sums of A) come from the matrix pipe as well: one extra MFMA per A fragment against a
//     fragment of ones (there are no registers holding A to add up).
// Jobs that mask rows of A (rowcode: the encoder's Dense layers) stay on gemm_wgg.h.
//
// MEASURED (MI355X, block group of four products at T = 32 768, stand-alone): 81 us at splitk 16 (this file, 3 k-tiles
// in flight in registers) and 88 us (the LDS-DMA version) against 72 us for gemm_wgg.h at splitk 8 -- the larger tile
// halves neither the time per unit of work (2.2 us per 48 KB k-tile here, 0.96 us per 32 KB k-tile there) nor the
// HBM bytes (225 MB either way: the XCD-local L2 already merges the panels the 128 x 128 tiles share), and the
// split-K tail doubles (16 slabs per tile instead of 8).  Off by default (MFP_WGG2=1 selects it); what it
// established: the swizzled unpadded images are bank-conflict free, bias gradients can come from the matrix pipe,
// and neither LDS bank conflicts nor the number of math waves is what bounds the streaming weight gradient.
#pragma once

#ifndef MFP_WGG2_XD
#define MFP_WGG2_XD 3
#endif
constexpr int WGG2_XD = MFP_WGG2_XD;      // k-tiles in flight (registers) per wave

struct Wgg2Params {
  WggParams base;
  int mtile0[WGG_MAX_JOBS];     // first macro tile of each job
  int nmtiles;
};

__device__ __forceinline__ int wgg2_key(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

template <int DUMMY>
__global__ __launch_bounds__(512) void gemm_wgg2_kernel(Wgg2Params pp) {
  const WggParams& p = pp.base;
  constexpr int BK = 64, A_ROWB = 512, B_ROWB = 256;
  constexpr int A_TILE = BK * A_ROWB, STAGE = A_TILE + BK * B_ROWB;       // 32 KB + 16 KB
  constexpr int CS_LD = 128 + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int xcd = blockIdx.x & 7, jb_ = blockIdx.x >> 3;
  const int kz = (jb_ / pp.nmtiles) * 8 + xcd, mt = jb_ % pp.nmtiles;          // splitk % 8 == 0 (host)
  int ji = 0;
  for (int q = 1; q < p.njobs; ++q) ji = mt >= pp.mtile0[q] ? q : ji;
  const WggJob& jb = p.job[ji];
  const int M = jb.M, N = jb.N;
  const int mloc = mt - pp.mtile0[ji];
  const int tm2 = mloc / jb.tiles_n, tn = mloc % jb.tiles_n;
  const int m0 = tm2 * 256, n0 = tn * 128;
  const int tiles_m = (M + 127) / 128;
  const int tileA = jb.tile0 + (2 * tm2) * jb.tiles_n + tn;
  const bool haveB = 2 * tm2 + 1 < tiles_m;
  const int Kj = jb.k_dev != nullptr ? min(p.K, *jb.k_dev) : p.K;
  const int ktiles = (Kj + BK - 1) / BK;
  const int nk = kz < ktiles ? (ktiles - kz + p.splitk - 1) / p.splitk : 0;      // k-tiles kz, kz + splitk, ...
  const bool do_colsum = jb.colsum != nullptr && tn == 0;
  const int lda2 = jb.lda * 2, ldb2 = jb.ldb * 2;
  // rows >= Kj of the operands read as zero (buffer range); columns past M / N read neighbouring data whose
  // products are never stored
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(jb.A), 0, (unsigned int)Kj * (unsigned int)lda2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(jb.B), 0, (unsigned int)Kj * (unsigned int)ldb2, 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(wave);

  // ---- this wave's share of a k-tile: 4 x 16 B per lane of the A tile (2 token rows x 512 B per instruction), 2 x
  // 16 B of the B tile (4 rows x 256 B); the LDS slot of a piece is its column slot ^ (key(row) << 1)
  unsigned int aoff[4], boff[2];
  int alds[4], blds[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wv * 8 + 2 * i + (lane >> 5), slot = lane & 31;
    aoff[i] = (unsigned int)(row * lda2 + m0 * 2 + slot * 16);
    alds[i] = row * A_ROWB + ((slot ^ (wgg2_key(row) << 1)) << 4);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wv * 8 + 4 * i + (lane >> 4), slot = lane & 15;
    boff[i] = (unsigned int)(row * ldb2 + n0 * 2 + slot * 16);
    blds[i] = A_TILE + row * B_ROWB + ((slot ^ (wgg2_key(row) << 1)) << 4);
  }
  u32x4 ra[WGG2_XD][4], rb[WGG2_XD][2];
  auto gload = [&](auto sc, int t) {          // k-tile t of this workgroup's slice -> register set (compile-time index)
    constexpr int set = decltype(sc)::value;
    const int k0 = (t * p.splitk + kz) * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[set][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, aoff[i], k0 * lda2, 0));
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[set][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, boff[i], k0 * ldb2, 0));
  };
  auto lstore = [&](auto sc, int stage) {
    constexpr int set = decltype(sc)::value;
    unsigned char* st = smem_raw + stage * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(st + alds[i]) = ra[set][i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(st + blds[i]) = rb[set][i];
  };

  const int wm = wv & 3, wn = wv >> 2;
  f32x4 acc[4][4], csum[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    csum[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  // fragment addresses inside a stage: token row k = 32 ks + 8 lg + (li >> 2) (+ 4 for the second half), 32-byte
  // group (4 wm + a) resp. (4 wn + b), XOR key(k) on the group's low three bits, 8 bytes per lane
  int arow[2], brow[2], akey[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int k = ks * 32 + lg * 8 + (li >> 2);
    akey[ks] = wgg2_key(k);                       // (k + 4 has the same key)
    arow[ks] = k * A_ROWB + (li & 3) * 8;
    brow[ks] = A_TILE + k * B_ROWB + (li & 3) * 8;
  }

  auto compute = [&](int t) {
    const unsigned char* st = smem_raw + (t & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xf[4], wf[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const unsigned char* ptr = st + arow[ks] + ((((wm * 4 + a) & 8) | (((wm * 4 + a) ^ akey[ks]) & 7)) << 5);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 4 * A_ROWB));
        xf[a] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const unsigned char* ptr = st + brow[ks] + ((((wn * 4 + b) ^ akey[ks]) & 7) << 5);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 4 * B_ROWB));
        wf[b] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
      if (do_colsum && wn == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) csum[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xf[a], csum[a], 0, 0, 0);
      }
    }
  };
  // prologue: WGG2_XD k-tiles in flight, the first one into stage 0 (tiles past the slice read as zero: buffer range)
  wgg_static_for<0, WGG2_XD>([&](auto ic) { gload(ic, decltype(ic)::value); });
  lstore(std::integral_constant<int, 0>{}, 0);
  gload(std::integral_constant<int, 0>{}, WGG2_XD);
  __syncthreads();
  // iteration t: tile t + 1 (register set (t + 1) % XD, requested XD iterations ago) -> the other stage, which every
  // wave finished reading before the last barrier; its register set takes tile t + 1 + XD; multiply tile t
  auto step = [&](auto tc, int t) {
    constexpr int xi = (decltype(tc)::value + 1) % WGG2_XD;
    lstore(std::integral_constant<int, xi>{}, (t + 1) & 1);
    gload(std::integral_constant<int, xi>{}, t + 1 + WGG2_XD);
    compute(t);
    __syncthreads();
  };
  {
    int t = 0;
    for (; t + WGG2_XD - 1 < nk; t += WGG2_XD) wgg_static_for<0, WGG2_XD>([&](auto ic) { step(ic, t + decltype(ic)::value); });
    wgg_static_for<0, WGG2_XD - 1>([&](auto ic) { if (t + decltype(ic)::value < nk) step(ic, t + decltype(ic)::value); });
  }

  // ---- publish: the two 128 x 128 tiles of the macro tile, one after the other, exactly as gemm_wgg.h does
  float* Cs = reinterpret_cast<float*>(smem_raw);
  float* colsum_s = Cs + 128 * CS_LD + 4;                      // [128] behind the output stage and the flag
  volatile int* last_s = reinterpret_cast<volatile int*>(smem_raw + 128 * CS_LD * 4);
  const int r0 = tid >> 5, c4 = (tid & 31) * 4;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half == 1 && !haveB) break;
    const int tile = tileA + half * jb.tiles_n;
    const int mh = m0 + half * 128;
    if ((wm >> 1) == half) {
      const int wmh = wm & 1;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
          *reinterpret_cast<f32x4*>(&Cs[(wmh * 64 + a * 16 + li) * CS_LD + wn * 64 + b * 16 + lg * 4]) = acc[a][b];
        if (do_colsum && wn == 0 && lg == 0) colsum_s[wmh * 64 + a * 16 + li] = csum[a][0];
      }
    }
    __syncthreads();
    {
      float* slab = p.ws + kz * p.zstride + (long long)tile * (128 * 128);
      const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc(slab, 0, 128 * 128 * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = r0 + 16 * i;
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&Cs[row * CS_LD + c4]), rss,
                                               (unsigned int)((row * 128 + c4) * 4), 0, 16 /* sc1 */);
      }
      if (do_colsum && tid < 128)
        __hip_atomic_store(&p.ws_col[((long long)kz * p.ntiles + tile) * 128 + tid], colsum_s[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) {
      const unsigned int ticket = __hip_atomic_fetch_add(&p.tickets[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == (unsigned int)(p.splitk - 1);
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(&p.tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next launch
      }
      *last_s = last;
    }
    __syncthreads();
    if (*last_s) {
      // last arriver of this tile: C[m][n] = sum over kz (ascending) of the slabs; colsum likewise
      const long long zstride = p.zstride;
      const float* src = p.ws + (long long)tile * (128 * 128) + r0 * 128 + c4;
      f32x4 acc8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc8[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int z0 = 0; z0 < p.splitk; z0 += 2) {        // two slabs (16 loads per thread) in flight
        f32x4 v[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i)
            v[u][i] = z0 + u < p.splitk ? *reinterpret_cast<const f32x4*>(src + (z0 + u) * zstride + 16 * i * 128) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (z0 + u < p.splitk) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { acc8[i][0] += v[u][i][0]; acc8[i][1] += v[u][i][1]; acc8[i][2] += v[u][i][2]; acc8[i][3] += v[u][i][3]; }
          }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = r0 + 16 * i;
        if (mh + row < M && n0 + c4 < N)
          *reinterpret_cast<f32x4*>(jb.C + (long long)(mh + row) * jb.ldc + n0 + c4) = acc8[i];
      }
      if (do_colsum && tid < 128 && mh + tid < M) {
        float s = 0.f;
        for (int z = 0; z < p.splitk; ++z) s += p.ws_col[((long long)z * p.ntiles + tile) * 128 + tid];
        jb.colsum[mh + tid] = s;
      }
    }
    __syncthreads();      // Cs / the flag are rewritten by the second half
  }
}

inline int launch_wgg2(const Wgg2Params& pp, hipStream_t st) {
  constexpr int lds = 2 * (64 * 512 + 64 * 256);       // 96 KB: two stages; the output stage aliases them afterwards
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgg2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_wgrad_group: cannot raise dynamic LDS to %d: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_wgg2_kernel<0>, dim3(pp.nmtiles * pp.base.splitk), dim3(512), lds, st, pp);
  return MFP_OK;
}
