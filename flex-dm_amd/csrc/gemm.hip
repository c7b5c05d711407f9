// MFMA tile GEMM for the MFP hot path (gfx950).  One kernel template, three operand layouts:
//   forward Dense   C = X[M][K] * Wt[N][K]^T     (A k-major, B k-major; Wt = kernel stored [out][in])
//   dgrad           C = dY[M][K] * Wt[K][N]      (A k-major, B n-major)
//   wgrad           C = dY[K][M]^T * X[K][N]     (A m-major, B n-major, split-K over tokens)
// bf16 operands use v_mfma_f32_16x16x32_bf16; f32 operands use the exact-f32
// v_mfma_f32_16x16x4_f32 (parity path).  Operands that are not k-contiguous in memory are staged
// untransposed and read with ds_read_b64_tr_b16 (bf16) / a strided scalar read (f32).
//
// Shapes on this path are skinny (M = 32768 elements, N and K <= 1384): every product is
// HBM/latency-bound at 10-20 % of the MFMA peak, so the kernel is built for BYTES IN FLIGHT per CU
// = resident workgroups x prefetch registers, not for tile size.  Workgroup = 4 waves (2x2), wave
// tile = (16 MT) x (16 NQ); BK = 64 (bf16) / 16 (f32); the next k-tile's global loads are in
// flight in registers while the current one is multiplied.  Measured on MI355X (tools/
// bench_gemm.py): a 1-WG-per-CU A-panel-resident variant and a 2-tile-deep register prefetch were
// both SLOWER than more, smaller resident workgroups.
//
// The MFMA is issued TRANSPOSED (D^T = B^T A^T: the weight side is the MFMA "A" operand) and the
// weight rows are assigned to MFMA rows by the permutation n(b, 4q+e) = 4NQ q + 4b + e.  Result:
// lane (i = l&15, g = l>>4) owns, for each of its m-tiles a, the row m = 16a + i and the 4 NQ
// CONTIGUOUS columns n = 4NQ g .. (quad b supplies columns 4NQ g + 4b .. +3).  The whole epilogue
// (bias, ReLU, ReLU-mask, dropout, residual, accumulate, store) is 16-byte vector traffic.  For
// the k-major weight tile the permutation is applied when the tile is written to LDS, so fragment
// reads stay conflict-free rows; for the n-major tile the four lanes of a tr-read row supply the
// four column bases 4NQ q + 4b.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int NT = 256;

template <typename T> struct GemmCfg;
template <> struct GemmCfg<unsigned short> {  // bf16
  static constexpr int BK = 64, EPC = 8 /*elements per 16B chunk*/, PAD = 8;
};
template <> struct GemmCfg<float> {
  static constexpr int BK = 16, EPC = 4, PAD = 4;
};

struct GemmParams {
  const void* A; const void* B; void* C;
  const float* bias; const float* residual; const void* aux; const unsigned char* rowcode;
  float* ws;       // split-K partial C [splitk][M][N]
  float* ws_col;   // split-K partial colsum [splitk][M]
  int M, N, K, lda, ldb, ldc;
  int out_bf16, flags, kchunk;
  float dropout_p;
  unsigned long long seed, offset;
  const int* step_ptr;
  int tiles_m, tiles_n;
  int kz_xcd;      // 1: 1-D grid, k-chunks grouped per XCD (split-K with splitk % 8 == 0)
#ifdef MFP_GEMM_TRACE
  unsigned long long* trace;  // [workgroup][16] s_memtime stamps of wave 0
#endif
};

// weight-row permutation inside a (16 NQ)-row wave block: local n = 4NQ q + 4b + e -> LDS row 16b + 4q + e
template <int NQ>
__device__ __forceinline__ int perm_row(int n_local) {
  constexpr int WB = 16 * NQ;
  const int blk = (n_local / WB) * WB, r = n_local % WB;
  return blk + ((r >> 2) % NQ) * 16 + (r / (4 * NQ)) * 4 + (r & 3);
}

// Dense epilogue: lane (li, lg) owns rows mrow0 + 16a + li and the 4NQ contiguous columns
// nb .. nb + 4NQ - 1 (quad b = columns nb+4b .. nb+4b+3).
template <int MT, int NQ>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[MT][NQ], int mrow0, int nb, int li) {
  const int flags = p.flags;
  const float inv_keep = (flags & MFP_GEMM_DROPOUT) ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  const unsigned long long rng_off = p.offset + (p.step_ptr ? (unsigned long long)(*p.step_ptr) * MFP_RNG_STEP_STRIDE : 0ull);
  const unsigned int dkey = drop_key(p.seed, rng_off), dthr = drop_thr16(p.dropout_p);
  f32x4 bias4[NQ];
#pragma unroll
  for (int b = 0; b < NQ; ++b) {
    const int col = nb + b * 4;
    bias4[b] = ((flags & MFP_GEMM_BIAS) && col < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + col)
                                                      : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    const int row = mrow0 + a * 16 + li;
    if (row >= p.M) continue;
    const bool skip = (flags & MFP_GEMM_ROWSKIP) && p.rowcode[row] != 0;
    f32x4 v[NQ];
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
      const int col = nb + b * 4;
      if (col >= p.N) { v[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; continue; }
      const long long o = (long long)row * p.ldc + col;
      f32x4 x = acc[a][b] + bias4[b];
      if (flags & MFP_GEMM_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
      }
      if (flags & MFP_GEMM_RELU_BWD) {
        if (p.out_bf16) {
          const u32x2 h = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(p.aux) + o);
          x[0] = bf16_to_f32((unsigned short)(h[0] & 0xffff)) > 0.f ? x[0] : 0.f;
          x[1] = bf16_to_f32((unsigned short)(h[0] >> 16)) > 0.f ? x[1] : 0.f;
          x[2] = bf16_to_f32((unsigned short)(h[1] & 0xffff)) > 0.f ? x[2] : 0.f;
          x[3] = bf16_to_f32((unsigned short)(h[1] >> 16)) > 0.f ? x[3] : 0.f;
        } else {
          const f32x4 h = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.aux) + o);
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = h[r] > 0.f ? x[r] : 0.f;
        }
      }
      if (skip) x = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (flags & MFP_GEMM_DROPOUT) {
        bool keep[4];
        drop_keep4(drop_row(dkey, (unsigned int)row), (unsigned int)col, dthr, keep);
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = keep[r] ? x[r] * inv_keep : 0.f;
      }
      if (flags & MFP_GEMM_RESIDUAL) x += *reinterpret_cast<const f32x4*>(p.residual + o);
      if (!p.out_bf16 && (flags & MFP_GEMM_ACCUM)) x += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.C) + o);
      v[b] = x;
    }
    if (p.out_bf16) {
      unsigned short* c = reinterpret_cast<unsigned short*>(p.C) + (long long)row * p.ldc + nb;
#pragma unroll
      for (int b = 0; b < NQ; b += 2) {
        if (nb + b * 4 >= p.N) continue;
        if (nb + b * 4 + 4 < p.N) {
          u32x4 pk = {pack_bf16x2(v[b][0], v[b][1]), pack_bf16x2(v[b][2], v[b][3]),
                      pack_bf16x2(v[b + 1][0], v[b + 1][1]), pack_bf16x2(v[b + 1][2], v[b + 1][3])};
          *reinterpret_cast<u32x4*>(c + b * 4) = pk;
        } else {
          u32x2 pk = {pack_bf16x2(v[b][0], v[b][1]), pack_bf16x2(v[b][2], v[b][3])};
          *reinterpret_cast<u32x2*>(c + b * 4) = pk;
        }
      }
    } else {
      float* c = reinterpret_cast<float*>(p.C) + (long long)row * p.ldc + nb;
#pragma unroll
      for (int b = 0; b < NQ; ++b)
        if (nb + b * 4 < p.N) *reinterpret_cast<f32x4*>(c + b * 4) = v[b];
    }
  }
}

template <typename T, bool A_KMAJOR, bool B_KMAJOR, int MT, int NQ, int NBUF>
struct GemmLds {
  using Cfg = GemmCfg<T>;
  static constexpr int BM = 32 * MT, BN = 32 * NQ, BK = Cfg::BK, PAD = Cfg::PAD;
  static constexpr int A_ROWS = A_KMAJOR ? BM : BK, A_COLS = A_KMAJOR ? BK : BM, LDA_S = A_COLS + PAD;
  static constexpr int B_ROWS = B_KMAJOR ? BN : BK, B_COLS = B_KMAJOR ? BK : BN, LDB_S = B_COLS + PAD;
  static constexpr int A_ELEMS = A_ROWS * LDA_S, B_ELEMS = B_ROWS * LDB_S;
  static constexpr size_t BYTES = (size_t)NBUF * (A_ELEMS + B_ELEMS) * sizeof(T) + (A_KMAJOR ? (size_t)BM : (size_t)(256 / (A_COLS / GemmCfg<T>::EPC)) * BM) * sizeof(float);
};

template <typename T, bool A_KMAJOR, bool B_KMAJOR, int MT, int NQ, int NBUF>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(GemmParams p) {
  using Cfg = GemmCfg<T>;
  using L = GemmLds<T, A_KMAJOR, B_KMAJOR, MT, NQ, NBUF>;
  constexpr int BM = L::BM, BN = L::BN, BK = Cfg::BK, EPC = Cfg::EPC;
  constexpr int WBM = 16 * MT, WBN = 16 * NQ;   // wave tile
  constexpr bool IS_BF16 = sizeof(T) == 2;
  constexpr int A_ROWS = L::A_ROWS, A_COLS = L::A_COLS, LDA_S = L::LDA_S;
  constexpr int B_ROWS = L::B_ROWS, B_COLS = L::B_COLS, LDB_S = L::LDB_S;
  constexpr int A_CPR = A_COLS / EPC, B_CPR = B_COLS / EPC;  // chunks per row
  constexpr int A_CH = A_ROWS * A_CPR / NT, B_CH = B_ROWS * B_CPR / NT;
  static_assert(A_ROWS * A_CPR % NT == 0 && B_ROWS * B_CPR % NT == 0, "tile/thread mismatch");
  static_assert(NT % A_CPR == 0, "colsum ownership");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* As0 = reinterpret_cast<T*>(smem_raw);
  T* Bs0 = As0 + NBUF * L::A_ELEMS;
  float* colsum_s = reinterpret_cast<float*>(Bs0 + NBUF * L::B_ELEMS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
#ifdef MFP_GEMM_TRACE
  int trace_i = 0;
#define TRACE_STAMP() do { if (tid == 0 && trace_i < 20) p.trace[(long long)(blockIdx.x + gridDim.x * blockIdx.z) * 24 + trace_i++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TRACE_STAMP() do {} while (0)
#endif
  TRACE_STAMP();  // 0: start
#ifdef MFP_GEMM_TRACE
  if (tid == 0) p.trace[(long long)(blockIdx.x + gridDim.x * blockIdx.z) * 24 + 20] = __builtin_amdgcn_s_memrealtime();
#endif

  // XCD-aware block order (hardware: block b -> XCD b % 8).
  //  * no split-K: bijective remap so consecutive logical tiles stay on one XCD (shared A panel);
  //  * split-K (wgrad): grid is 1-D over (k-chunk, tile) and ALL tiles of a k-chunk run back to
  //    back on ONE XCD, so the dY / X row panels of that chunk are fetched from HBM once and
  //    re-read from that L2 by the other tiles (measured before: 2.3x the algorithmic reads).
  int bid, kz;
  if (p.kz_xcd) {
    const int tiles = p.tiles_m * p.tiles_n, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    kz = (j / tiles) * 8 + xcd;     // splitk % 8 == 0 (host)
    bid = j % tiles;
  } else {
    bid = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    kz = blockIdx.z;
  }
  const int tm = bid / p.tiles_n, tn = bid % p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = kz * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);

  const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B);

  f32x4 acc[MT][NQ];  // [m-tile a][n-quad b]
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NQ; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool do_colsum = (p.flags & MFP_GEMM_COLSUM_A) && tn == 0;
  const bool rowskip_a = (p.flags & MFP_GEMM_ROWSKIP_A) != 0;
  float csum[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) csum[e] = 0.f;

  // Branch-free staging loads: raw buffer loads through a wave-uniform descriptor.  Each chunk
  // has a FIXED 32-bit byte offset (voffset; 0xFFFFFFF0 = "out of range" -> the hardware returns
  // zeros, no exec-mask branch), the k-tile advance rides in the scalar soffset.  No per-tile
  // address VALU, and the whole k-loop body is one basic block so loads interleave with MFMAs.
  u32x4 ra[A_CH], rb[B_CH];
  unsigned int voa[A_CH], vob[B_CH];
  int kca[A_CH], kcb[B_CH];     // k offset of the chunk inside a tile (row for k-strided tiles)
  int lsa[A_CH], lsb[B_CH];     // LDS element offsets (weight-row permutation folded in)
  constexpr unsigned int OOB = 0xFFFFFFF0u;
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Ag), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Bg), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
  for (int c = 0; c < A_CH; ++c) {
    const int ch = tid + c * NT, row = ch / A_CPR, col = (ch % A_CPR) * EPC;
    lsa[c] = row * LDA_S + col;
    if (A_KMAJOR) {
      kca[c] = col;
      voa[c] = m0 + row < p.M ? (unsigned int)(((long long)(m0 + row) * p.lda + kbeg + col) * sizeof(T)) : OOB;
    } else {
      kca[c] = row;
      voa[c] = m0 + col < p.M ? (unsigned int)(((long long)(kbeg + row) * p.lda + m0 + col) * sizeof(T)) : OOB;
    }
  }
#pragma unroll
  for (int c = 0; c < B_CH; ++c) {
    const int ch = tid + c * NT, row = ch / B_CPR, col = (ch % B_CPR) * EPC;
    if (B_KMAJOR) {
      kcb[c] = col;
      vob[c] = n0 + row < p.N ? (unsigned int)(((long long)(n0 + row) * p.ldb + kbeg + col) * sizeof(T)) : OOB;
      lsb[c] = perm_row<NQ>(row) * LDB_S + col;
    } else {
      kcb[c] = row;
      vob[c] = n0 + col < p.N ? (unsigned int)(((long long)(kbeg + row) * p.ldb + n0 + col) * sizeof(T)) : OOB;
      lsb[c] = row * LDB_S + col;
    }
  }
  const int stepa = (A_KMAJOR ? BK : BK * p.lda) * (int)sizeof(T);   // bytes per k-tile
  const int stepb = (B_KMAJOR ? BK : BK * p.ldb) * (int)sizeof(T);

  auto gload = [&](int k0) {
    const bool full = k0 + BK <= kend;   // uniform: only a ragged / absent tile checks its k columns
    const int t = (k0 - kbeg) / BK;
    const int soa = t * stepa, sob = t * stepb;
#pragma unroll
    for (int c = 0; c < A_CH; ++c) {
      unsigned int vo = voa[c];
      if (!full) vo = k0 + kca[c] < kend ? vo : OOB;
      if (!A_KMAJOR && rowskip_a) vo = (k0 + kca[c] < kend && p.rowcode[k0 + kca[c]] != 0) ? OOB : vo;
      ra[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, vo, soa, 0));
    }
#pragma unroll
    for (int c = 0; c < B_CH; ++c) {
      unsigned int vo = vob[c];
      if (!full) vo = k0 + kcb[c] < kend ? vo : OOB;
      rb[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, vo, sob, 0));
    }
  };
  auto lstore = [&](int buf) {
    T* As = As0 + buf * L::A_ELEMS;
    T* Bs = Bs0 + buf * L::B_ELEMS;
#pragma unroll
    for (int c = 0; c < A_CH; ++c) {
      *reinterpret_cast<u32x4*>(&As[lsa[c]]) = ra[c];
      if (!A_KMAJOR && do_colsum) {
        // A_CPR chunks per row and NT % A_CPR == 0: a thread always owns the same columns.
        if (IS_BF16) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            unsigned int w = ra[c][e];
            csum[2 * e] += bf16_to_f32((unsigned short)(w & 0xffffu));
            csum[2 * e + 1] += bf16_to_f32((unsigned short)(w >> 16));
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) csum[e] += __uint_as_float(ra[c][e]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < B_CH; ++c) *reinterpret_cast<u32x4*>(&Bs[lsb[c]]) = rb[c];
  };
  auto compute = [&](int buf) {
    const T* As = As0 + buf * L::A_ELEMS;
    const T* Bs = Bs0 + buf * L::B_ELEMS;
    if constexpr (IS_BF16) {
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 xf[MT], wf[NQ];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          if (A_KMAJOR) {
            xf[t] = *reinterpret_cast<const bf16x8*>(&As[(wm * WBM + t * 16 + li) * LDA_S + ks * 32 + lg * 8]);
          } else {
            const T* ptr = &As[(ks * 32 + lg * 8 + (li >> 2)) * LDA_S + wm * WBM + t * 16 + (li & 3) * 4];
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 4 * LDA_S));
            xf[t] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
        }
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
          if (B_KMAJOR) {  // rows already permuted at staging: MFMA row li of quad t = LDS row 16t + li
            wf[t] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * WBN + t * 16 + li) * LDB_S + ks * 32 + lg * 8]);
          } else {         // tr-read: the 4 lanes of a k-row supply column bases 4NQ q + 4t
            const T* ptr = &Bs[(ks * 32 + lg * 8 + (li >> 2)) * LDB_S + wn * WBN + (li & 3) * (4 * NQ) + t * 4];
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 4 * LDB_S));
            wf[t] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
        }
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < NQ; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        float xf[MT], wf[NQ];
#pragma unroll
        for (int t = 0; t < MT; ++t)
          xf[t] = A_KMAJOR ? As[(wm * WBM + t * 16 + li) * LDA_S + ks * 4 + lg]
                           : As[(ks * 4 + lg) * LDA_S + wm * WBM + t * 16 + li];
#pragma unroll
        for (int t = 0; t < NQ; ++t)
          wf[t] = B_KMAJOR ? Bs[(wn * WBN + t * 16 + li) * LDB_S + ks * 4 + lg]
                           : Bs[(ks * 4 + lg) * LDB_S + wn * WBN + (li >> 2) * (4 * NQ) + t * 4 + (li & 3)];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < NQ; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[b], xf[a], acc[a][b], 0, 0, 0);
      }
    }
  };

  if (kbeg < kend) {
    gload(kbeg);
    TRACE_STAMP();  // 1: first loads issued
    lstore(0);
    TRACE_STAMP();  // 2: first tile landed + written to LDS
    __syncthreads();
    TRACE_STAMP();  // 3: barrier
    if constexpr (NBUF == 2) {      // one barrier per k-tile, 2x LDS
      int buf = 0;
      for (int k0 = kbeg; k0 < kend; k0 += BK) {
        // Unconditional: past the last tile every offset is out of range (zero-fill, no traffic),
        // so the loop body is ONE basic block and the scheduler may interleave loads and MFMAs.
        gload(k0 + BK);             // next tile: global -> registers, in flight during the MFMAs
        compute(buf);
        if constexpr (IS_BF16 && (A_CH + B_CH) * 2 <= MT * NQ * (BK / 32)) {
          // In-order issue: a wave's MFMAs would queue behind its 8 VMEM instructions, which the
          // address path throttles (1 KB per wave-load = 16 clk; profiles/r01_gemm_qkv_timeline.txt).
          // Trickle one load per two MFMAs so the load path and the matrix pipe run concurrently.
          __builtin_amdgcn_sched_group_barrier(0x100, MT + NQ, 0);     // fragment ds_reads of k-step 0
#pragma unroll
          for (int i = 0; i < A_CH + B_CH; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         // one global load
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);         // two MFMAs
          }
        }
        TRACE_STAMP();              // 4,7,10,13: MFMAs of the tile issued
        lstore(buf ^ 1);            // the other buffer was last read one barrier ago
        TRACE_STAMP();              // 5,8,11,14: next tile landed + written
        __syncthreads();
        TRACE_STAMP();              // 6,9,12,15: barrier
        buf ^= 1;
      }
    } else {                        // two barriers per k-tile, half the LDS -> more resident workgroups
      for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) gload(k0 + BK);
        compute(0);
        if (more) {
          __syncthreads();
          lstore(0);
          __syncthreads();
        }
      }
    }
  }

  // ---- bias-gradient column sums of A = dY (wgrad, n-tile 0 only)
  if (!A_KMAJOR && do_colsum) {
    // per-thread sums -> LDS [row group][column], summed in a FIXED order (float atomics made the
    // bias gradients differ in the last bit from run to run)
    constexpr int NG = NT / A_CPR;
    __syncthreads();
    const int col = (tid % A_CPR) * EPC, grp = tid / A_CPR;
#pragma unroll
    for (int e = 0; e < EPC; ++e) colsum_s[grp * BM + col + e] = csum[e];
    __syncthreads();
    if (tid < BM && m0 + tid < p.M) {
      float sacc = 0.f;
#pragma unroll 4
      for (int gI = 0; gI < NG; ++gI) sacc += colsum_s[gI * BM + tid];
      p.ws_col[(long long)kz * p.M + m0 + tid] = sacc;
    }
  }

  // ---- epilogue: lane owns rows .. + 16a + li and columns nb .. nb + 4NQ - 1
  const int nb = n0 + wn * WBN + lg * (4 * NQ);
  if (p.ws != nullptr) {  // split-K / wgrad path: raw partials, reduced by splitk_reduce_kernel
    float* ws = p.ws + (long long)kz * p.M * p.N;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int row = m0 + wm * WBM + a * 16 + li;
      if (row >= p.M) continue;
#pragma unroll
      for (int b = 0; b < NQ; ++b) {
        const int col = nb + b * 4;
        if (col < p.N) *reinterpret_cast<f32x4*>(ws + (long long)row * p.N + col) = acc[a][b];
      }
    }
    return;
  }
  TRACE_STAMP();  // epilogue start
  gemm_epilogue<MT, NQ>(p, acc, m0 + wm * WBM, nb, li);
  TRACE_STAMP();  // stores issued
#ifdef MFP_GEMM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  TRACE_STAMP();  // stores retired
#ifdef MFP_GEMM_TRACE
  if (tid == 0) p.trace[(long long)(blockIdx.x + gridDim.x * blockIdx.z) * 24 + 21] = __builtin_amdgcn_s_memrealtime();
#endif
}

// out[m][n] (+)= sum_z ws[z][m][n];  colsum[m] = sum_z ws_col[z][m].  N % 4 == 0.
// Block = 32 float4 outputs x 8 z-groups (the group sums meet in LDS): a 256x256 gradient gives
// 512 workgroups with splitk/8 loads per thread instead of 64 workgroups walking all splitk slabs.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws,
                                                            const float* __restrict__ ws_col,
                                                            float* __restrict__ C, float* __restrict__ colsum,
                                                            int M, int N, int ldc, int splitk, int accum) {
  __shared__ float4 red[8][32];
  const long long total4 = (long long)M * N / 4;
  const long long total = (long long)M * N;
  const int q = threadIdx.x & 31, zg = threadIdx.x >> 5;
  const long long i4 = (long long)blockIdx.x * 32 + q;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i4 < total4) {
    for (int z = zg; z < splitk; z += 8) {
      const float4 v = *reinterpret_cast<const float4*>(ws + (long long)z * total + i4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[zg][q] = s;
  __syncthreads();
  if (zg == 0 && i4 < total4) {
#pragma unroll
    for (int g = 1; g < 8; ++g) { const float4 v = red[g][q]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    const long long idx = i4 * 4, m = idx / N, n = idx % N;
    float4* c = reinterpret_cast<float4*>(C + m * ldc + n);
    if (accum) { const float4 o = *c; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    *c = s;
  }
  if (colsum != nullptr) {
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
      float t = 0.f;
      for (int z = 0; z < splitk; ++z) t += ws_col[(long long)z * M + m];
      colsum[m] = t;
    }
  }
}

#include "gemm_ws.h"
#include "gemm_wg.h"
#include "gemm_wgg.h"

int launch_ws_any(const mfp_gemm_args* a, const GemmParams& p, hipStream_t st) {
  const int ncu = mfp_ncu_launch();
  // narrow column slices (half the weight prologue per CU, twice the row tiles per workgroup) pay
  // off when a full-width slice would leave a workgroup only a handful of 32-row tiles
  // (narrow for N <= 256 is the measured best; the MFP_WS_NARROW experiment switch of rounds 2-4 is gone)
  const bool narrow = a->N <= 256;
  if (a->K == 256) {
    if (narrow) return launch_ws_epi<8, 2, true>(a, p, ncu, st);
    return launch_ws_epi<8, 2>(a, p, ncu, st);
  }
  if (a->K == 768)   // plain epilogue only (ws_eligible): the fused-QKV input gradient
    return a->out_dtype == MFP_BF16 ? launch_ws<24, 2, WS_EPI_PLAIN, false, true>(p, ncu, st)
                                    : launch_ws<24, 2, WS_EPI_PLAIN, false, false>(p, ncu, st);
  return launch_ws_epi<16, 2>(a, p, ncu, st);
}

template <typename T, bool AK, bool BK_, int MT, int NQ, int NBUF>
int launch_one(const GemmParams& p0, int M, int N, int splitk, hipStream_t st) {
  using L = GemmLds<T, AK, BK_, MT, NQ, NBUF>;
  constexpr size_t lds = L::BYTES;
  static bool attr_done[MFP_MAX_DEVICES] = {};  // benign cache: the attribute is a constant per (function, device)
  bool& attr_set = attr_done[mfp_device_slot()];
  if (lds > 64 * 1024 && !attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, AK, BK_, MT, NQ, NBUF>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_gemm: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  GemmParams p = p0;
  p.tiles_m = (M + L::BM - 1) / L::BM;
  p.tiles_n = (N + L::BN - 1) / L::BN;
  p.kz_xcd = (splitk > 1 && splitk % 8 == 0) ? 1 : 0;
  dim3 grid(p.tiles_m * p.tiles_n * (p.kz_xcd ? splitk : 1), 1, p.kz_xcd ? 1 : splitk);
  hipLaunchKernelGGL((gemm_kernel<T, AK, BK_, MT, NQ, NBUF>), grid, dim3(NT), lds, st, p);
  return MFP_OK;
}

// Tile choice per layout (measured with tools/bench_gemm.py on MI355X).
// (benchmarking only).
template <typename T, bool AK, bool BK_>
int launch_layout(const mfp_gemm_args* a, const GemmParams& p, int splitk, hipStream_t st, bool small, bool dbuf) {
  if (small) {
    return dbuf ? launch_one<T, AK, BK_, 2, 4, 2>(p, a->M, a->N, splitk, st)
                : launch_one<T, AK, BK_, 2, 4, 1>(p, a->M, a->N, splitk, st);
  }
  return dbuf ? launch_one<T, AK, BK_, 4, 4, 2>(p, a->M, a->N, splitk, st)
              : launch_one<T, AK, BK_, 4, 4, 1>(p, a->M, a->N, splitk, st);
}

template <typename T>
int launch_gemm(const mfp_gemm_args* a, const GemmParams& p, int splitk, hipStream_t st) {
  bool small = a->a_kmajor != 0, dbuf = !small;
  // long-K forward / input-gradient products (d_model 512: FFN2 forward and the FFN1 / Q|K|V input gradients, K = 1024 /
  // 1536): 128 x 128 tiles -- a third less operand traffic out of the L2s than 64 x 128 (c5: 4.64 -> 4.56 ms per step)
  if (a->a_kmajor && a->K >= 1024 && sizeof(T) == 2) { small = false; dbuf = false; }
  if (a->a_kmajor && !a->b_kmajor) return launch_layout<T, true, false>(a, p, splitk, st, small, dbuf);
  if (a->a_kmajor && a->b_kmajor) return launch_layout<T, true, true>(a, p, splitk, st, small, dbuf);
  if (!a->a_kmajor && !a->b_kmajor) return launch_layout<T, false, false>(a, p, splitk, st, small, dbuf);
  mfp_set_error("mfp_gemm: layout a_kmajor=0,b_kmajor=1 is not on the MFP path");
  return MFP_EINVAL;
}

#ifdef MFP_GEMM_TRACE
unsigned long long* g_trace = nullptr;
#endif

bool uses_workspace(const mfp_gemm_args* a) {
  return a->splitk > 1 || (a->flags & (MFP_GEMM_COLSUM_A | MFP_GEMM_ROWSKIP_A));
}

}  // namespace

#ifdef MFP_GEMM_TRACE
extern "C" void mfp_trace_buffer(void* p) { g_trace = reinterpret_cast<unsigned long long*>(p); }
#endif

extern "C" size_t mfp_gemm_workspace_bytes(const mfp_gemm_args* a) {
  if (!uses_workspace(a)) return 0;
  int sk = a->splitk < 1 ? 1 : a->splitk;
  return ((size_t)sk * a->M * a->N + (size_t)sk * a->M) * sizeof(float);
}


extern "C" const char* mfp_gemm_kernel_family(const mfp_gemm_args* a) {
  if (a == nullptr) return "";
  const int splitk = a->splitk < 1 ? 1 : a->splitk;
  if (ws_eligible(a, splitk)) return "gemm_ws_kernel";
  if (uses_workspace(a) && wg_eligible(a, splitk)) return "gemm_wg_kernel";
  return "gemm_kernel";
}

extern "C" int mfp_gemm(const mfp_gemm_args* a, mfp_stream_t stream) {
  MFP_CHECK_ARG(a != nullptr && a->A && a->B && a->C);
  MFP_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0);
  MFP_CHECK_ARG(a->in_dtype == MFP_F32 || a->in_dtype == MFP_BF16);
  MFP_CHECK_ARG(a->out_dtype == MFP_F32 || a->out_dtype == MFP_BF16);
  const int epc = a->in_dtype == MFP_BF16 ? 8 : 4;
  MFP_CHECK_ARG(a->lda % epc == 0 && a->ldb % epc == 0 && a->ldc % 4 == 0 && a->N % 4 == 0);
  MFP_CHECK_ARG(((uintptr_t)a->C % 16) == 0);
  MFP_CHECK_ARG(a->N % epc == 0 || a->b_kmajor);
  if (a->a_kmajor) MFP_CHECK_ARG(a->K % epc == 0); else MFP_CHECK_ARG(a->M % epc == 0);
  if (a->b_kmajor) MFP_CHECK_ARG(a->K % epc == 0);
  MFP_CHECK_ARG(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->B % 16) == 0);
  {  // staging loads use 32-bit byte offsets (raw buffer loads)
    const long long esz = a->in_dtype == MFP_BF16 ? 2 : 4;
    const long long rows_a = a->a_kmajor ? a->M : a->K, rows_b = a->b_kmajor ? a->N : a->K;
    MFP_CHECK_ARG(rows_a * a->lda * esz < 0x7FFFFFF0ll && rows_b * a->ldb * esz < 0x7FFFFFF0ll);
  }
  const int splitk = a->splitk < 1 ? 1 : a->splitk;
  const bool wgrad = !a->a_kmajor && !a->b_kmajor;
  MFP_CHECK_ARG(splitk == 1 || wgrad);
  if (a->flags & (MFP_GEMM_COLSUM_A | MFP_GEMM_ROWSKIP_A)) MFP_CHECK_ARG(wgrad);
  if (a->flags & MFP_GEMM_COLSUM_A) MFP_CHECK_ARG(a->colsum != nullptr);
  if (a->flags & (MFP_GEMM_ROWSKIP | MFP_GEMM_ROWSKIP_A)) MFP_CHECK_ARG(a->rowcode != nullptr);
  if (a->flags & MFP_GEMM_BIAS) MFP_CHECK_ARG(a->bias != nullptr);
  if (a->flags & MFP_GEMM_RESIDUAL) MFP_CHECK_ARG(a->residual != nullptr);
  if (a->flags & MFP_GEMM_RELU_BWD) MFP_CHECK_ARG(a->aux != nullptr);
  if (a->flags & MFP_GEMM_ACCUM) MFP_CHECK_ARG(a->out_dtype == MFP_F32);
  if (a->flags & MFP_GEMM_DROPOUT) MFP_CHECK_ARG(a->dropout_p >= 0.f && a->dropout_p < 1.f);
  if (a->out_dtype == MFP_BF16) MFP_CHECK_ARG(a->ldc % 8 == 0);
  const bool ws_path = uses_workspace(a);
  if (ws_path) {
    MFP_CHECK_ARG(a->out_dtype == MFP_F32);
    MFP_CHECK_ARG((a->flags & ~(MFP_GEMM_COLSUM_A | MFP_GEMM_ROWSKIP_A | MFP_GEMM_ACCUM)) == 0);
    if (a->workspace == nullptr || a->workspace_bytes < mfp_gemm_workspace_bytes(a)) {
      mfp_set_error("mfp_gemm: workspace too small (%zu < %zu)", a->workspace_bytes,
                    mfp_gemm_workspace_bytes(a));
      return MFP_EWORKSPACE;
    }
  }
  GemmParams p;
  p.A = a->A; p.B = a->B; p.C = a->C; p.bias = a->bias; p.residual = a->residual; p.aux = a->aux;
  p.rowcode = a->rowcode;
  p.ws = ws_path ? reinterpret_cast<float*>(a->workspace) : nullptr;
  p.ws_col = ws_path ? p.ws + (size_t)splitk * a->M * a->N : nullptr;
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc;
  p.out_bf16 = a->out_dtype == MFP_BF16; p.flags = a->flags;
  p.dropout_p = a->dropout_p; p.seed = a->seed; p.offset = a->offset; p.step_ptr = a->step_ptr;
  p.tiles_m = 0; p.tiles_n = 0; p.kz_xcd = 0;  // set per tile configuration in launch_one
#ifdef MFP_GEMM_TRACE
  p.trace = g_trace;
#endif
  const int bk = a->in_dtype == MFP_BF16 ? GemmCfg<unsigned short>::BK : GemmCfg<float>::BK;
  int kchunk = (a->K + splitk - 1) / splitk;
  kchunk = ((kchunk + bk - 1) / bk) * bk;
  p.kchunk = kchunk;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (ws_eligible(a, splitk)) {
    int rcw = launch_ws_any(a, p, st);
    if (rcw != MFP_OK) return rcw;
    MFP_CHECK_LAUNCH();
    return MFP_OK;
  }
  int rc;
  if (ws_path && wg_eligible(a, splitk)) rc = launch_wg(p, a->M, a->N, splitk, st);
  else rc = a->in_dtype == MFP_BF16 ? launch_gemm<unsigned short>(a, p, splitk, st)
                                    : launch_gemm<float>(a, p, splitk, st);
  if (rc != MFP_OK) return rc;
  MFP_CHECK_LAUNCH();
  if (ws_path) {
    long long total = (long long)a->M * a->N / 4;
    int blocks = (int)((total + 31) / 32);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.ws, p.ws_col,
                       reinterpret_cast<float*>(a->C),
                       (a->flags & MFP_GEMM_COLSUM_A) ? a->colsum : nullptr, a->M, a->N, a->ldc,
                       splitk, (a->flags & MFP_GEMM_ACCUM) ? 1 : 0);
    MFP_CHECK_LAUNCH();
  }
  return MFP_OK;
}

// ------------------------------------------------------------------ grouped weight gradients
namespace {
int wgg_tiles(const mfp_wgrad_job* jobs, int njobs) {
  int t = 0;
  for (int i = 0; i < njobs; ++i) t += ((jobs[i].M + 127) / 128) * ((jobs[i].N + 127) / 128);
  return t;
}
// 256 x 128 macro tiles (gemm_wgt_kernel): groups with many tiles whose jobs all have an even number of tile rows and
// no row masks (c5: the four products of a d_model-512 block, 128 tiles); "MFP_WGT=0" keeps 128 x 128 units (A/B switch)
bool wgg_macro_ok(const mfp_wgrad_job* jobs, int njobs) {
  static const bool off = getenv("MFP_WGT") != nullptr && atoi(getenv("MFP_WGT")) == 0;
  if (off) return false;
  int tiles = 0;
  for (int i = 0; i < njobs; ++i) {
    const int tm = (jobs[i].M + 127) / 128;
    if ((tm & 1) || jobs[i].rowcode != nullptr || jobs[i].M % 128 != 0) return false;
    tiles += tm * ((jobs[i].N + 127) / 128);
  }
  return tiles >= 64;
}
int wgg_ncu() { return mfp_ncu_launch(); }
}  // namespace

extern "C" int32_t mfp_wgrad_group_tiles(const mfp_wgrad_job* jobs, int32_t njobs) {
  if (jobs == nullptr || njobs < 1) return 0;
  return wgg_tiles(jobs, njobs);
}

// Split of the token dimension: a multiple of 8 (a k-slice's tiles share an XCD) that minimises
// waves-of-workgroups x k-tiles-per-workgroup on this device, smallest split on ties (fewer partial
// slabs); slices of at least 256 tokens, at most WG_MAX_KCHUNK when a job masks rows (codes in LDS).
extern "C" int32_t mfp_wgrad_group_splitk(const mfp_wgrad_job* jobs, int32_t njobs, int32_t K, int32_t deferred) {
  if (jobs == nullptr || njobs < 1 || K < 1) return 8;
  // (the 256 x 128 macro-tile kernel only exists in the deferred form: the in-place form launches one workgroup per 128 x 128 tile)
  const int tiles = (deferred && wgg_macro_ok(jobs, njobs)) ? wgg_tiles(jobs, njobs) / 2 : wgg_tiles(jobs, njobs), ncu = wgg_ncu();
  bool rowskip = false;
  for (int i = 0; i < njobs; ++i) rowskip |= jobs[i].rowcode != nullptr;
  int best = 8;
  long long best_cost = -1;
  for (int sk = 1; sk <= 512; sk = sk < 8 ? sk * 2 : sk + 8) {
    const int kchunk = (((K + 63) / 64 + sk - 1) / sk) * 64;      // tokens of the longest cyclic k-slice
    if (sk > 8 && kchunk < 256) break;
    if (rowskip && kchunk > WG_MAX_KCHUNK) continue;
    if (sk < 8 && (long long)tiles * sk < ncu) continue;          // fewer workgroups than CUs: split further
    const long long nwg = sk >= 8 ? (long long)tiles * sk : 8ll * ((tiles * sk + 7) / 8);
    const long long waves = (nwg + ncu - 1) / ncu;
    // measured (tools/bench_wgrad.py, block group at T = 32768): 1.1 us per k-tile + 11 us per workgroup
    // (pipeline fill, 64 KB slab store, ticket, the last arriver's read of `sk` slabs)
    const long long cost = waves * (kchunk / 64 + 10) + sk / 8;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = sk; }
  }
  return best;
}

extern "C" size_t mfp_wgrad_group_workspace_bytes(const mfp_wgrad_job* jobs, int32_t njobs, int32_t splitk) {
  if (jobs == nullptr || njobs < 1 || splitk < 1) return 0;
  return (size_t)splitk * ((size_t)wgg_tiles(jobs, njobs) * (128 * 128 + 128) + WGG_ZPAD) * sizeof(float);
}

namespace {
int wgrad_group_launch(const mfp_wgrad_job* jobs, int32_t njobs, int32_t K, int32_t splitk,
                       void* workspace, size_t workspace_bytes, uint32_t* tickets, bool defer, mfp_stream_t stream) {
  MFP_CHECK_ARG(jobs != nullptr && njobs >= 1 && njobs <= MFP_MAX_WGRAD_JOBS && K > 0);
  MFP_CHECK_ARG((splitk == 1 || splitk == 2 || splitk == 4 || (splitk >= 8 && splitk % 8 == 0)) && (defer || tickets != nullptr) && workspace != nullptr);
  WggParams p;
  int tile0 = 0;
  bool rowskip = false;
  for (int i = 0; i < njobs; ++i) {
    const mfp_wgrad_job& j = jobs[i];
    MFP_CHECK_ARG(j.A && j.B && j.C && j.M > 0 && j.N > 0);
    MFP_CHECK_ARG(defer || j.n_affine == nullptr);      // (the x-hat correction lives in mfp_wgrad_reduce)
    MFP_CHECK_ARG(j.M % 8 == 0 && j.N % 8 == 0 && j.lda % 8 == 0 && j.ldb % 8 == 0 && j.ldc % 4 == 0);
    MFP_CHECK_ARG(j.lda >= j.M && j.ldb >= j.N && j.ldc >= j.N);
    MFP_CHECK_ARG(((uintptr_t)j.A % 16) == 0 && ((uintptr_t)j.B % 16) == 0 && ((uintptr_t)j.C % 16) == 0);
    MFP_CHECK_ARG((long long)K * j.lda * 2 < 0x7FFFFFF0ll && (long long)K * j.ldb * 2 < 0x7FFFFFF0ll);   // 32-bit offsets
    WggJob& d = p.job[i];
    d.A = reinterpret_cast<const unsigned short*>(j.A);
    d.B = reinterpret_cast<const unsigned short*>(j.B);
    d.C = j.C; d.colsum = j.colsum; d.rowcode = j.rowcode;
    d.M = j.M; d.N = j.N; d.lda = j.lda; d.ldb = j.ldb; d.ldc = j.ldc;
    d.tiles_n = (j.N + 127) / 128;
    d.tile0 = tile0; d.pad_ = 0;
    tile0 += ((j.M + 127) / 128) * d.tiles_n;
    rowskip |= j.rowcode != nullptr;
  }
  for (int i = njobs; i < WGG_MAX_JOBS; ++i) { p.job[i] = p.job[0]; p.job[i].tile0 = 0x7FFFFFFF; }
  p.njobs = njobs; p.ntiles = tile0; p.K = K; p.splitk = splitk;
  p.nk_max = ((K + 63) / 64 + splitk - 1) / splitk;
  const bool macro = defer && wgg_macro_ok(jobs, njobs);
  p.tpg = splitk < 8 ? ((macro ? tile0 / 2 : tile0) * splitk + 7) / 8 : 0;
  if (rowskip) MFP_CHECK_ARG(p.nk_max * 64 <= WG_MAX_KCHUNK);
  const size_t need = mfp_wgrad_group_workspace_bytes(jobs, njobs, splitk);
  if (workspace_bytes < need) {
    mfp_set_error("mfp_wgrad_group: workspace too small (%zu < %zu)", workspace_bytes, need);
    return MFP_EWORKSPACE;
  }
  MFP_CHECK_ARG(((uintptr_t)workspace % 16) == 0);
  p.ws = reinterpret_cast<float*>(workspace);
  p.zstride = (long long)p.ntiles * 128 * 128 + WGG_ZPAD;
  p.ws_col = p.ws + (size_t)splitk * p.zstride;
  p.tickets = tickets;
#ifdef MFP_GEMM_TRACE
  p.trace = g_trace;
#endif
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (macro) {
    if (int rcm = launch_wgt(p, st)) return rcm;
    MFP_CHECK_LAUNCH();
    return MFP_OK;
  }
  const int rc = defer ? (rowskip ? launch_wgg_t<true, true>(p, st) : launch_wgg_t<false, true>(p, st))
                       : (rowskip ? launch_wgg_t<true, false>(p, st) : launch_wgg_t<false, false>(p, st));
  if (rc != MFP_OK) return rc;
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}
}  // namespace

extern "C" int mfp_wgrad_group(const mfp_wgrad_job* jobs, int32_t njobs, int32_t K, int32_t splitk,
                               void* workspace, size_t workspace_bytes, uint32_t* tickets, mfp_stream_t stream) {
  return wgrad_group_launch(jobs, njobs, K, splitk, workspace, workspace_bytes, tickets, false, stream);
}

extern "C" int mfp_wgrad_group_partial(const mfp_wgrad_job* jobs, int32_t njobs, int32_t K, int32_t splitk,
                                       void* workspace, size_t workspace_bytes, mfp_stream_t stream) {
  return wgrad_group_launch(jobs, njobs, K, splitk, workspace, workspace_bytes, nullptr, true, stream);
}

extern "C" int mfp_wgrad_reduce(const mfp_wgrad_pending* groups, int32_t ngroups, mfp_stream_t stream) {
  MFP_CHECK_ARG(groups != nullptr && ngroups >= 1 && ngroups <= MFP_MAX_WGRAD_PENDING);
  WgrParams p;
  int unit0 = 0, job0 = 0;
  for (int gi = 0; gi < ngroups; ++gi) job0 += groups[gi].njobs > 0 ? groups[gi].njobs : 0;
  MFP_CHECK_ARG(job0 <= WGR_MAX_JOBS);      // (all groups' jobs of one launch; the caller reduces earlier otherwise)
  job0 = 0;
  for (int gi = 0; gi < ngroups; ++gi) {
    const mfp_wgrad_pending& s = groups[gi];
    MFP_CHECK_ARG(s.jobs != nullptr && s.njobs >= 1 && s.njobs <= MFP_MAX_WGRAD_JOBS && (s.splitk == 1 || s.splitk == 2 || s.splitk == 4 || (s.splitk >= 8 && s.splitk % 8 == 0)));
    MFP_CHECK_ARG(s.workspace != nullptr && ((uintptr_t)s.workspace % 16) == 0);
    WgrGroup& G = p.g[gi];
    int tile0 = 0;
    for (int i = 0; i < s.njobs; ++i) {
      const mfp_wgrad_job& j = s.jobs[i];
      MFP_CHECK_ARG(j.C && j.M > 0 && j.N > 0 && j.N % 8 == 0 && j.ldc % 4 == 0 && j.ldc >= j.N && ((uintptr_t)j.C % 16) == 0);
      WgrJob& d = p.job[job0 + i];
      MFP_CHECK_ARG(j.n_affine == nullptr || (j.colsum != nullptr && j.N % 4 == 0 && ((uintptr_t)j.n_affine % 16) == 0));
      d.C = j.C; d.colsum = j.colsum; d.nfix = j.n_affine; d.M = j.M; d.N = j.N; d.ldc = j.ldc;
      d.tiles_n = (j.N + 127) / 128; d.tile0 = tile0; d.pad_ = 0;
      tile0 += ((j.M + 127) / 128) * d.tiles_n;
    }
    G.njobs = s.njobs; G.ntiles = tile0; G.splitk = s.splitk; G.unit0 = unit0; G.job0 = job0; G.pad_ = 0;
    job0 += s.njobs;
    G.zstride = (long long)tile0 * 128 * 128 + WGG_ZPAD;
    G.ws = reinterpret_cast<const float*>(s.workspace);
    G.ws_col = G.ws + (size_t)s.splitk * G.zstride;
    unit0 += tile0 * 8;
  }
  for (int gi = ngroups; gi < WGR_MAX_GROUPS; ++gi) { p.g[gi] = p.g[0]; p.g[gi].unit0 = 0x7FFFFFFF; }
  for (int i = job0; i < WGR_MAX_JOBS; ++i) p.job[i] = p.job[0];
  p.ngroups = ngroups; p.nunits = unit0;
  hipLaunchKernelGGL(wgg_reduce_kernel, dim3(unit0), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}
