// MFMA tile GEMM for the MFP hot path (gfx950).  One kernel template, three operand layouts:
//   forward Dense   C = X[M][K] * Wt[N][K]^T     (A k-major, B k-major; Wt = kernel stored [out][in])
//   dgrad           C = dY[M][K] * Wt[K][N]      (A k-major, B n-major)
//   wgrad           C = dY[K][M]^T * X[K][N]     (A m-major, B n-major, split-K over tokens)
// bf16 operands use v_mfma_f32_16x16x32_bf16; f32 operands use the exact-f32
// v_mfma_f32_16x16x4_f32 (parity path).  Operands that are not k-contiguous in memory are staged
// untransposed and read with ds_read_b64_tr_b16 (bf16) / a strided scalar read (f32).
//
// Tile: 128x128 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA tiles),
// BK = 64 (bf16) / 16 (f32), LDS double-buffered, ONE barrier per k-tile; the next tile's global
// loads are in flight in registers while the current tile is multiplied.
//
// The MFMA is issued TRANSPOSED (D^T = B^T A^T: the weight side is the MFMA "A" operand) and the
// weight rows are assigned to MFMA rows by the permutation n(b, 4q+e) = 16q + 4b + e.  Result:
// lane (i = l&15, g = l>>4) owns, for each of its 4 m-tiles a, the row m = 16a + i and the 16
// CONTIGUOUS columns n = 16g .. 16g+15 (tile b supplies columns 16g+4b .. 16g+4b+3).  The whole
// epilogue (bias, ReLU, ReLU-mask, dropout, residual, accumulate, store) is therefore 16-byte
// vector traffic.  For the k-major weight tile the permutation is applied when the tile is
// written to LDS (row n -> LDS row 16b + 4q + e), so fragment reads stay conflict-free rows; for
// the n-major tile the four lanes of a tr-read row supply the four column bases 16q + 4b.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, NT = 256;

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
};

template <typename T>
__device__ __forceinline__ u32x4 load_chunk(const T* base, long long off, bool ok) {
  u32x4 z = {0u, 0u, 0u, 0u};
  if (!ok) return z;
  return *reinterpret_cast<const u32x4*>(base + off);
}

// weight-row permutation inside a 64-row wave block: local n = 16q + 4b + e  ->  LDS row 16b + 4q + e
__device__ __forceinline__ int perm_row(int n_local) {
  const int blk = n_local & ~63, r = n_local & 63;
  return blk + ((r >> 2) & 3) * 16 + (r >> 4) * 4 + (r & 3);
}

// Dense epilogue for MT m-tiles: lane (li, lg) owns rows mrow0 + 16a + li and the 16 contiguous
// columns nb .. nb+15 (quad b = columns nb+4b .. nb+4b+3).
template <int MT>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[MT][4], int mrow0, int nb, int li) {
  const int flags = p.flags;
  const float inv_keep = (flags & MFP_GEMM_DROPOUT) ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  const unsigned long long rng_off = p.offset + (p.step_ptr ? (unsigned long long)(*p.step_ptr) * MFP_RNG_STEP_STRIDE : 0ull);
  f32x4 bias4[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int col = nb + b * 4;
    bias4[b] = ((flags & MFP_GEMM_BIAS) && col < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + col)
                                                      : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    const int row = mrow0 + a * 16 + li;
    if (row >= p.M) continue;
    const bool skip = (flags & MFP_GEMM_ROWSKIP) && p.rowcode[row] != 0;
    f32x4 v[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
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
        unsigned int rnd[4];
        philox4x32(p.seed, (unsigned int)row, (unsigned int)(col >> 2), rng_off, rnd);
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = philox_keep(rnd[r], p.dropout_p) ? x[r] * inv_keep : 0.f;
      }
      if (flags & MFP_GEMM_RESIDUAL) x += *reinterpret_cast<const f32x4*>(p.residual + o);
      if (!p.out_bf16 && (flags & MFP_GEMM_ACCUM)) x += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.C) + o);
      v[b] = x;
    }
    if (p.out_bf16) {
      unsigned short* c = reinterpret_cast<unsigned short*>(p.C) + (long long)row * p.ldc + nb;
#pragma unroll
      for (int b = 0; b < 4; b += 2) {
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
      for (int b = 0; b < 4; ++b)
        if (nb + b * 4 < p.N) *reinterpret_cast<f32x4*>(c + b * 4) = v[b];
    }
  }
}

template <typename T, bool A_KMAJOR, bool B_KMAJOR>
struct GemmLds {
  using Cfg = GemmCfg<T>;
  static constexpr int BK = Cfg::BK, PAD = Cfg::PAD;
  static constexpr int A_ROWS = A_KMAJOR ? BM : BK, A_COLS = A_KMAJOR ? BK : BM, LDA_S = A_COLS + PAD;
  static constexpr int B_ROWS = B_KMAJOR ? BN : BK, B_COLS = B_KMAJOR ? BK : BN, LDB_S = B_COLS + PAD;
  static constexpr int A_ELEMS = A_ROWS * LDA_S, B_ELEMS = B_ROWS * LDB_S;
  static constexpr size_t BYTES = (size_t)2 * (A_ELEMS + B_ELEMS) * sizeof(T) + BM * sizeof(float);
};

template <typename T, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmParams p) {
  using Cfg = GemmCfg<T>;
  using L = GemmLds<T, A_KMAJOR, B_KMAJOR>;
  constexpr int BK = Cfg::BK, EPC = Cfg::EPC;
  constexpr bool IS_BF16 = sizeof(T) == 2;
  constexpr int A_ROWS = L::A_ROWS, A_COLS = L::A_COLS, LDA_S = L::LDA_S;
  constexpr int B_ROWS = L::B_ROWS, B_COLS = L::B_COLS, LDB_S = L::LDB_S;
  constexpr int A_CPR = A_COLS / EPC, B_CPR = B_COLS / EPC;  // chunks per row
  constexpr int A_CH = A_ROWS * A_CPR / NT, B_CH = B_ROWS * B_CPR / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* As0 = reinterpret_cast<T*>(smem_raw);
  T* Bs0 = As0 + 2 * L::A_ELEMS;
  float* colsum_s = reinterpret_cast<float*>(Bs0 + 2 * L::B_ELEMS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware bijective remap: consecutive logical tiles stay on one XCD (shared A panel in L2)
  int nwg = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = bid / p.tiles_n, tn = bid % p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kz = blockIdx.z;
  const int kbeg = kz * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);

  const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B);

  f32x4 acc[4][4];  // [m-tile a][n-quad b]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool do_colsum = (p.flags & MFP_GEMM_COLSUM_A) && tn == 0;
  const bool rowskip_a = (p.flags & MFP_GEMM_ROWSKIP_A) != 0;
  float csum[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) csum[e] = 0.f;

  u32x4 ra[A_CH], rb[B_CH];

  auto gload = [&](int k0) {
#pragma unroll
    for (int c = 0; c < A_CH; ++c) {
      int ch = tid + c * NT, row = ch / A_CPR, col = (ch % A_CPR) * EPC;
      if (A_KMAJOR) {
        int m = m0 + row, k = k0 + col;
        ra[c] = load_chunk(Ag, (long long)m * p.lda + k, m < p.M && k < kend);
      } else {
        int k = k0 + row, m = m0 + col;
        bool ok = k < kend && m < p.M;
        if (rowskip_a && ok) ok = p.rowcode[k] == 0;
        ra[c] = load_chunk(Ag, (long long)k * p.lda + m, ok);
      }
    }
#pragma unroll
    for (int c = 0; c < B_CH; ++c) {
      int ch = tid + c * NT, row = ch / B_CPR, col = (ch % B_CPR) * EPC;
      if (B_KMAJOR) {
        int n = n0 + row, k = k0 + col;
        rb[c] = load_chunk(Bg, (long long)n * p.ldb + k, n < p.N && k < kend);
      } else {
        int k = k0 + row, n = n0 + col;
        rb[c] = load_chunk(Bg, (long long)k * p.ldb + n, k < kend && n < p.N);
      }
    }
  };
  auto lstore = [&](int buf) {
    T* As = As0 + buf * L::A_ELEMS;
    T* Bs = Bs0 + buf * L::B_ELEMS;
#pragma unroll
    for (int c = 0; c < A_CH; ++c) {
      int ch = tid + c * NT, row = ch / A_CPR, col = (ch % A_CPR) * EPC;
      *reinterpret_cast<u32x4*>(&As[row * LDA_S + col]) = ra[c];
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
    for (int c = 0; c < B_CH; ++c) {
      int ch = tid + c * NT, row = ch / B_CPR, col = (ch % B_CPR) * EPC;
      if (B_KMAJOR) row = perm_row(row);
      *reinterpret_cast<u32x4*>(&Bs[row * LDB_S + col]) = rb[c];
    }
  };
  auto compute = [&](int buf) {
    const T* As = As0 + buf * L::A_ELEMS;
    const T* Bs = Bs0 + buf * L::B_ELEMS;
    if constexpr (IS_BF16) {
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 xf[4], wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (A_KMAJOR) {
            xf[t] = *reinterpret_cast<const bf16x8*>(&As[(wm * 64 + t * 16 + li) * LDA_S + ks * 32 + lg * 8]);
          } else {
            const T* ptr = &As[(ks * 32 + lg * 8 + (li >> 2)) * LDA_S + wm * 64 + t * 16 + (li & 3) * 4];
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 4 * LDA_S));
            xf[t] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
          if (B_KMAJOR) {  // rows already permuted at staging: MFMA row li of quad t = LDS row 16t + li
            wf[t] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * 64 + t * 16 + li) * LDB_S + ks * 32 + lg * 8]);
          } else {         // tr-read: the 4 lanes of a k-row supply column bases 16q + 4t
            const T* ptr = &Bs[(ks * 32 + lg * 8 + (li >> 2)) * LDB_S + wn * 64 + (li & 3) * 16 + t * 4];
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 4 * LDB_S));
            wf[t] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        float xf[4], wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          xf[t] = A_KMAJOR ? As[(wm * 64 + t * 16 + li) * LDA_S + ks * 4 + lg]
                           : As[(ks * 4 + lg) * LDA_S + wm * 64 + t * 16 + li];
          wf[t] = B_KMAJOR ? Bs[(wn * 64 + t * 16 + li) * LDB_S + ks * 4 + lg]
                           : Bs[(ks * 4 + lg) * LDB_S + wn * 64 + (li >> 2) * 16 + t * 4 + (li & 3)];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[b], xf[a], acc[a][b], 0, 0, 0);
      }
    }
  };

  if (kbeg < kend) {
    gload(kbeg);
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      const bool more = k0 + BK < kend;
      if (more) gload(k0 + BK);   // next tile: global -> registers, in flight during the MFMAs
      compute(buf);
      if (more) lstore(buf ^ 1);  // the other buffer was last read one barrier ago
      __syncthreads();
      buf ^= 1;
    }
  }

  // ---- bias-gradient column sums of A = dY (wgrad, n-tile 0 only)
  if (!A_KMAJOR && do_colsum) {
    if (tid < BM) colsum_s[tid] = 0.f;
    __syncthreads();
    int col = (tid % A_CPR) * EPC;
#pragma unroll
    for (int e = 0; e < EPC; ++e) atomicAdd(&colsum_s[col + e], csum[e]);
    __syncthreads();
    if (tid < BM && m0 + tid < p.M) p.ws_col[(long long)kz * p.M + m0 + tid] = colsum_s[tid];
  }

  // ---- epilogue: lane owns rows m = .. + 16a + li and columns nb .. nb+15 (quad b = 4 columns)
  const int nb = n0 + wn * 64 + lg * 16;
  if (p.ws != nullptr) {  // split-K / wgrad path: raw partials, reduced by splitk_reduce_kernel
    float* ws = p.ws + (long long)kz * p.M * p.N;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int row = m0 + wm * 64 + a * 16 + li;
      if (row >= p.M) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int col = nb + b * 4;
        if (col < p.N) *reinterpret_cast<f32x4*>(ws + (long long)row * p.N + col) = acc[a][b];
      }
    }
    return;
  }

  gemm_epilogue<4>(p, acc, m0 + wm * 64, nb, li);
}

// ------------------------------------------------------------------------------------------
// A-panel-resident GEMM for the skinny forward / dgrad products of the MFP step: M = #elements
// (32768), N <= 1384, K <= 512.  One workgroup per CU owns BM rows of A: the [BM][K] panel is
// loaded ONCE with every 16-byte request in flight at the same time (at 1 WG/CU a wave may use
// the whole 512-register budget), then the workgroup sweeps all N tiles; the weight tiles
// ([128][64], L2-resident) are requested two steps ahead.  HBM sees A once and C once -- the
// algorithmic bytes -- instead of A once per N tile through L2 with one tile of latency exposed
// per k-step.  Same fragment / epilogue conventions as gemm_kernel.
template <typename T, bool B_KMAJOR, int MT /* m-tiles per wave: BM = 32*MT */>
__global__ __launch_bounds__(NT, 1) void gemm_apanel_kernel(GemmParams p, int kpad /*K rounded up to BK*/) {
  using Cfg = GemmCfg<T>;
  constexpr int BK = Cfg::BK, EPC = Cfg::EPC, PAD = Cfg::PAD;
  constexpr bool IS_BF16 = sizeof(T) == 2;
  constexpr int BMP = 32 * MT;
  constexpr int B_ROWS = B_KMAJOR ? BN : BK, B_COLS = B_KMAJOR ? BK : BN, LDB_S = B_COLS + PAD;
  constexpr int B_CPR = B_COLS / EPC, B_CH = B_ROWS * B_CPR / NT, B_ELEMS = B_ROWS * LDB_S;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lda_s = kpad + PAD;
  T* Ap = reinterpret_cast<T*>(smem_raw);
  T* Bs0 = Ap + BMP * lda_s;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BMP;
  const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B);
  const int nkt = kpad / BK, nnt = (p.N + BN - 1) / BN, nsteps = nkt * nnt;

  u32x4 rb0[B_CH], rb1[B_CH];
  auto gload_b = [&](int step, u32x4 (&rb)[B_CH]) {
    const int n0 = (step / nkt) * BN, k0 = (step % nkt) * BK;
#pragma unroll
    for (int c = 0; c < B_CH; ++c) {
      int ch = tid + c * NT, row = ch / B_CPR, col = (ch % B_CPR) * EPC;
      if (B_KMAJOR) {
        int n = n0 + row, k = k0 + col;
        rb[c] = load_chunk(Bg, (long long)n * p.ldb + k, n < p.N && k < p.K);
      } else {
        int k = k0 + row, n = n0 + col;
        rb[c] = load_chunk(Bg, (long long)k * p.ldb + n, k < p.K && n < p.N);
      }
    }
  };
  auto lstore_b = [&](int buf, u32x4 (&rb)[B_CH]) {
    T* Bs = Bs0 + buf * B_ELEMS;
#pragma unroll
    for (int c = 0; c < B_CH; ++c) {
      int ch = tid + c * NT, row = ch / B_CPR, col = (ch % B_CPR) * EPC;
      if (B_KMAJOR) row = perm_row(row);
      *reinterpret_cast<u32x4*>(&Bs[row * LDB_S + col]) = rb[c];
    }
  };

  // ---- weight tiles 0 and 1 first (short L2 latency), then the whole A panel in one burst
  gload_b(0, rb0);
  if (nsteps > 1) gload_b(1, rb1);
  {
    const int cpr = kpad / EPC;                 // chunks per panel row
    const int total = BMP * cpr;
    constexpr int BURST = 16;                   // 16 x 16 B per thread in flight per burst
    for (int base = 0; base < total; base += NT * BURST) {
      u32x4 r[BURST];
#pragma unroll
      for (int j = 0; j < BURST; ++j) {
        const int ch = base + j * NT + tid;
        const int row = ch / cpr, col = (ch % cpr) * EPC;
        const int m = m0 + row;
        r[j] = load_chunk(Ag, (long long)m * p.lda + col, ch < total && m < p.M && col < p.K);
      }
#pragma unroll
      for (int j = 0; j < BURST; ++j) {
        const int ch = base + j * NT + tid;
        if (ch < total) {
          const int row = ch / cpr, col = (ch % cpr) * EPC;
          *reinterpret_cast<u32x4*>(&Ap[row * lda_s + col]) = r[j];
        }
      }
    }
  }
  lstore_b(0, rb0);
  __syncthreads();

  f32x4 acc[MT][4];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf, int kt) {
    const T* Bs = Bs0 + buf * B_ELEMS;
    const T* As = Ap + kt * BK;   // column offset inside the panel
    if constexpr (IS_BF16) {
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 xf[MT], wf[4];
#pragma unroll
        for (int t = 0; t < MT; ++t)
          xf[t] = *reinterpret_cast<const bf16x8*>(&As[(wm * 16 * MT + t * 16 + li) * lda_s + ks * 32 + lg * 8]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (B_KMAJOR) {
            wf[t] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * 64 + t * 16 + li) * LDB_S + ks * 32 + lg * 8]);
          } else {
            const T* ptr = &Bs[(ks * 32 + lg * 8 + (li >> 2)) * LDB_S + wn * 64 + (li & 3) * 16 + t * 4];
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 4 * LDB_S));
            wf[t] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
        }
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        float xf[MT], wf[4];
#pragma unroll
        for (int t = 0; t < MT; ++t) xf[t] = As[(wm * 16 * MT + t * 16 + li) * lda_s + ks * 4 + lg];
#pragma unroll
        for (int t = 0; t < 4; ++t)
          wf[t] = B_KMAJOR ? Bs[(wn * 64 + t * 16 + li) * LDB_S + ks * 4 + lg]
                           : Bs[(ks * 4 + lg) * LDB_S + wn * 64 + (li >> 2) * 16 + t * 4 + (li & 3)];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[b], xf[a], acc[a][b], 0, 0, 0);
      }
    }
  };
  auto finish_tile = [&](int step) {   // after the last k-tile of an N tile: epilogue + reset
    if ((step + 1) % nkt != 0) return;
    const int n0 = (step / nkt) * BN;
    gemm_epilogue<MT>(p, acc, m0 + wm * 16 * MT, n0 + wn * 64 + lg * 16, li);
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  // step s: LDS buffer s&1 holds weight tile s; tile s+1 is in the other register set;
  // tile s+2 is requested now into the set tile s came from.
  for (int s0 = 0; s0 < nsteps; s0 += 2) {
    if (s0 + 2 < nsteps) gload_b(s0 + 2, rb0);
    compute(0, s0 % nkt);
    finish_tile(s0);
    if (s0 + 1 < nsteps) lstore_b(1, rb1);
    __syncthreads();
    if (s0 + 1 >= nsteps) break;
    if (s0 + 3 < nsteps) gload_b(s0 + 3, rb1);
    compute(1, (s0 + 1) % nkt);
    finish_tile(s0 + 1);
    if (s0 + 2 < nsteps) lstore_b(0, rb0);
    __syncthreads();
  }
}

// out[m][n] (+)= sum_z ws[z][m][n];  colsum[m] = sum_z ws_col[z][m].  N % 4 == 0.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws,
                                                            const float* __restrict__ ws_col,
                                                            float* __restrict__ C, float* __restrict__ colsum,
                                                            int M, int N, int ldc, int splitk, int accum) {
  const long long total4 = (long long)M * N / 4;
  const long long total = (long long)M * N;
  for (long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4;
       i4 += (long long)gridDim.x * blockDim.x) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int z = 0; z < splitk; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(ws + (long long)z * total + i4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const long long idx = i4 * 4, m = idx / N, n = idx % N;
    float4* c = reinterpret_cast<float4*>(C + m * ldc + n);
    if (accum) { const float4 o = *c; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    *c = s;
  }
  if (colsum != nullptr) {
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
      float s = 0.f;
      for (int z = 0; z < splitk; ++z) s += ws_col[(long long)z * M + m];
      colsum[m] = s;
    }
  }
}

template <typename T, bool AK, bool BK_>
int launch_one(const GemmParams& p, dim3 grid, hipStream_t st) {
  constexpr size_t lds = GemmLds<T, AK, BK_>::BYTES;
  static bool attr_set = false;  // benign cache: the attribute is a per-function constant
  if (lds > 64 * 1024 && !attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, AK, BK_>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_gemm: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_kernel<T, AK, BK_>), grid, dim3(NT), lds, st, p);
  return MFP_OK;
}

bool uses_workspace_fwd(const mfp_gemm_args* a);

template <typename T, bool BKM, int MT>
int launch_apanel(const GemmParams& p, int kpad, hipStream_t st) {
  using Cfg = GemmCfg<T>;
  constexpr int B_ROWS = BKM ? BN : Cfg::BK, B_COLS = BKM ? Cfg::BK : BN;
  const size_t lds = ((size_t)32 * MT * (kpad + Cfg::PAD) + (size_t)2 * B_ROWS * (B_COLS + Cfg::PAD)) * sizeof(T);
  static size_t attr_bytes = 0;  // benign cache (largest size requested so far)
  if (lds > 64 * 1024 && lds > attr_bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_apanel_kernel<T, BKM, MT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_gemm: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_bytes = lds;
  }
  dim3 grid((p.M + 32 * MT - 1) / (32 * MT));
  hipLaunchKernelGGL((gemm_apanel_kernel<T, BKM, MT>), grid, dim3(NT), lds, st, p, kpad);
  return MFP_OK;
}

// -> MT (m-tiles per wave) of the panel kernel that fits LDS, or 0 to use the tiled kernel
template <typename T>
int apanel_choice(const mfp_gemm_args* a, int* kpad_out) {
  using Cfg = GemmCfg<T>;
  if (!a->a_kmajor || uses_workspace_fwd(a) || a->M < 2048) return 0;
  const int kpad = ((a->K + Cfg::BK - 1) / Cfg::BK) * Cfg::BK;
  *kpad_out = kpad;
  const size_t b_bytes = (size_t)2 * 128 * (Cfg::BK + Cfg::PAD) * sizeof(T) + 4096;
  for (int mt = 4; mt >= 2; mt -= 2) {
    const size_t a_bytes = (size_t)32 * mt * (kpad + Cfg::PAD) * sizeof(T);
    if (a_bytes + b_bytes <= 150 * 1024) return mt;
  }
  return 0;
}

template <typename T>
int launch_gemm(const mfp_gemm_args* a, const GemmParams& p, dim3 grid, hipStream_t st) {
  if (a->a_kmajor && !a->b_kmajor) {
    return launch_one<T, true, false>(p, grid, st);
  } else if (a->a_kmajor && a->b_kmajor) {
    return launch_one<T, true, true>(p, grid, st);
  } else if (!a->a_kmajor && !a->b_kmajor) {
    return launch_one<T, false, false>(p, grid, st);
  } else {
    mfp_set_error("mfp_gemm: layout a_kmajor=0,b_kmajor=1 is not on the MFP path");
    return MFP_EINVAL;
  }
  return MFP_OK;
}

bool uses_workspace_fwd(const mfp_gemm_args* a) {
  return a->splitk > 1 || (a->flags & (MFP_GEMM_COLSUM_A | MFP_GEMM_ROWSKIP_A));
}

bool uses_workspace(const mfp_gemm_args* a) {
  return a->splitk > 1 || (a->flags & (MFP_GEMM_COLSUM_A | MFP_GEMM_ROWSKIP_A));
}

}  // namespace

extern "C" size_t mfp_gemm_workspace_bytes(const mfp_gemm_args* a) {
  if (!uses_workspace(a)) return 0;
  int sk = a->splitk < 1 ? 1 : a->splitk;
  return ((size_t)sk * a->M * a->N + (size_t)sk * a->M) * sizeof(float);
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
  const bool ws_path = uses_workspace(a);
  if (ws_path) {
    MFP_CHECK_ARG(a->out_dtype == MFP_F32 && a->N % 4 == 0 && a->ldc % 4 == 0);
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
  p.tiles_m = (a->M + BM - 1) / BM; p.tiles_n = (a->N + BN - 1) / BN;
  const int bk = a->in_dtype == MFP_BF16 ? GemmCfg<unsigned short>::BK : GemmCfg<float>::BK;
  int kchunk = (a->K + splitk - 1) / splitk;
  kchunk = ((kchunk + bk - 1) / bk) * bk;
  p.kchunk = kchunk;
  dim3 grid(p.tiles_m * p.tiles_n, 1, splitk);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc;
  int kpad = 0;
  const int mt = a->in_dtype == MFP_BF16 ? apanel_choice<unsigned short>(a, &kpad) : apanel_choice<float>(a, &kpad);
  if (mt && getenv("MFP_NO_APANEL") == nullptr) {
    if (a->in_dtype == MFP_BF16) {
      if (a->b_kmajor) rc = mt == 4 ? launch_apanel<unsigned short, true, 4>(p, kpad, st) : launch_apanel<unsigned short, true, 2>(p, kpad, st);
      else rc = mt == 4 ? launch_apanel<unsigned short, false, 4>(p, kpad, st) : launch_apanel<unsigned short, false, 2>(p, kpad, st);
    } else {
      if (a->b_kmajor) rc = mt == 4 ? launch_apanel<float, true, 4>(p, kpad, st) : launch_apanel<float, true, 2>(p, kpad, st);
      else rc = mt == 4 ? launch_apanel<float, false, 4>(p, kpad, st) : launch_apanel<float, false, 2>(p, kpad, st);
    }
  } else {
    rc = a->in_dtype == MFP_BF16 ? launch_gemm<unsigned short>(a, p, grid, st)
                                  : launch_gemm<float>(a, p, grid, st);
  }
  if (rc != MFP_OK) return rc;
  MFP_CHECK_LAUNCH();
  if (ws_path) {
    long long total = (long long)a->M * a->N / 4;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.ws, p.ws_col,
                       reinterpret_cast<float*>(a->C),
                       (a->flags & MFP_GEMM_COLSUM_A) ? a->colsum : nullptr, a->M, a->N, a->ldc,
                       splitk, (a->flags & MFP_GEMM_ACCUM) ? 1 : 0);
    MFP_CHECK_LAUNCH();
  }
  return MFP_OK;
}
