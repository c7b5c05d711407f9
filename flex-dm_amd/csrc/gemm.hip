// MFMA tile GEMM for the MFP hot path (gfx950).  One kernel template, three operand layouts:
//   forward Dense   C = A[M][K] * W[K][N]        (A k-major, B n-major)
//   dgrad           C = dY[M][K] * W[N][K]^T     (A k-major, B k-major)
//   wgrad           C = X[K][M]^T * dY[K][N]     (A m-major, B n-major, split-K over tokens)
// bf16 operands use v_mfma_f32_16x16x32_bf16; f32 operands use the exact-f32
// v_mfma_f32_16x16x4_f32 (parity path).  Operands that are not k-contiguous in memory are
// staged untransposed and read with ds_read_b64_tr_b16 (bf16) / a strided scalar read (f32),
// so no transposed weight copies exist anywhere.
//
// Tile: 128x128 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA tiles).
// Fragment conventions (lane l: i = l&15, g = l>>4):
//   A frag: row i of the 16-row tile, k = 8g..8g+7 (bf16) / k = g (f32)
//   B frag: col i of the 16-col tile, same k
//   C frag: col i, rows 4g..4g+3
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

// 16-byte global load of a chunk or zeros
template <typename T>
__device__ __forceinline__ u32x4 load_chunk(const T* base, long long off, bool ok) {
  u32x4 z = {0u, 0u, 0u, 0u};
  if (!ok) return z;
  return *reinterpret_cast<const u32x4*>(base + off);
}

template <typename T, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmParams p) {
  using Cfg = GemmCfg<T>;
  constexpr int BK = Cfg::BK, EPC = Cfg::EPC, PAD = Cfg::PAD;
  constexpr bool IS_BF16 = sizeof(T) == 2;
  // LDS tiles keep the operand's memory orientation.
  constexpr int A_ROWS = A_KMAJOR ? BM : BK, A_COLS = A_KMAJOR ? BK : BM, LDA_S = A_COLS + PAD;
  constexpr int B_ROWS = B_KMAJOR ? BN : BK, B_COLS = B_KMAJOR ? BK : BN, LDB_S = B_COLS + PAD;
  constexpr int A_CPR = A_COLS / EPC, B_CPR = B_COLS / EPC;  // chunks per row
  constexpr int A_CH = A_ROWS * A_CPR / NT, B_CH = B_ROWS * B_CPR / NT;
  __shared__ __attribute__((aligned(16))) T As[A_ROWS * LDA_S];
  __shared__ __attribute__((aligned(16))) T Bs[B_ROWS * LDB_S];
  __shared__ float colsum_s[BM];

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

  f32x4 acc[4][4];
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
  auto lstore = [&]() {
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
      *reinterpret_cast<u32x4*>(&Bs[row * LDB_S + col]) = rb[c];
    }
  };

  if (kbeg < kend) {
    gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      __syncthreads();  // previous tile fully consumed
      lstore();
      __syncthreads();
      if (k0 + BK < kend) gload(k0 + BK);  // prefetch next tile into registers

      if constexpr (IS_BF16) {
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
          bf16x8 af[4], bfr[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (A_KMAJOR) {
              af[t] = *reinterpret_cast<const bf16x8*>(
                  &As[(wm * 64 + t * 16 + li) * LDA_S + ks * 32 + lg * 8]);
            } else {
              const T* ptr = &As[(ks * 32 + lg * 8 + (li >> 2)) * LDA_S + wm * 64 + t * 16 + (li & 3) * 4];
              bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
              bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 4 * LDA_S));
              af[t] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            if (B_KMAJOR) {
              bfr[t] = *reinterpret_cast<const bf16x8*>(
                  &Bs[(wn * 64 + t * 16 + li) * LDB_S + ks * 32 + lg * 8]);
            } else {
              const T* ptr = &Bs[(ks * 32 + lg * 8 + (li >> 2)) * LDB_S + wn * 64 + t * 16 + (li & 3) * 4];
              bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
              bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 4 * LDB_S));
              bfr[t] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
          }
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
          float af[4], bfr[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            af[t] = A_KMAJOR ? As[(wm * 64 + t * 16 + li) * LDA_S + ks * 4 + lg]
                             : As[(ks * 4 + lg) * LDA_S + wm * 64 + t * 16 + li];
            bfr[t] = B_KMAJOR ? Bs[(wn * 64 + t * 16 + li) * LDB_S + ks * 4 + lg]
                              : Bs[(ks * 4 + lg) * LDB_S + wn * 64 + t * 16 + li];
          }
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
      }
    }
  }

  // ---- bias-gradient column sums of A = dY (wgrad, n-tile 0 only)
  if (!A_KMAJOR && do_colsum) {
    if (tid < BM) colsum_s[tid] = 0.f;
    __syncthreads();
    // every thread owns EPC columns starting at (tid % A_CPR) * EPC (same for all its chunks)
    int col = (tid % A_CPR) * EPC;
#pragma unroll
    for (int e = 0; e < EPC; ++e) atomicAdd(&colsum_s[col + e], csum[e]);
    __syncthreads();
    if (tid < BM && m0 + tid < p.M) p.ws_col[(long long)kz * p.M + m0 + tid] = colsum_s[tid];
  }

  // ---- epilogue
  const int flags = p.flags;
  if (p.ws != nullptr) {  // split-K / wgrad path: raw partials, reduced by splitk_reduce_kernel
    float* ws = p.ws + (long long)kz * p.M * p.N;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        int col = n0 + wn * 64 + b * 16 + li;
        int row = m0 + wm * 64 + a * 16 + lg * 4;
        if (col < p.N) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row + r < p.M) ws[(long long)(row + r) * p.N + col] = acc[a][b][r];
        }
      }
    return;
  }

  const float inv_keep = (flags & MFP_GEMM_DROPOUT) ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  const unsigned long long rng_off = p.offset + (p.step_ptr ? (unsigned long long)(*p.step_ptr) * MFP_RNG_STEP_STRIDE : 0ull);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int col = n0 + wn * 64 + b * 16 + li;
      int row = m0 + wm * 64 + a * 16 + lg * 4;
      if (col >= p.N || row >= p.M) continue;
      float bias = (flags & MFP_GEMM_BIAS) ? p.bias[col] : 0.f;
      unsigned int rnd[4] = {0u, 0u, 0u, 0u};
      if (flags & MFP_GEMM_DROPOUT) philox4x32(p.seed, (unsigned int)col, (unsigned int)(row >> 2), rng_off, rnd);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (row + r >= p.M) break;
        long long o = (long long)(row + r) * p.ldc + col;
        float v = acc[a][b][r] + bias;
        if (flags & MFP_GEMM_RELU) v = fmaxf(v, 0.f);
        if (flags & MFP_GEMM_RELU_BWD) {
          float h = p.out_bf16 ? bf16_to_f32(reinterpret_cast<const unsigned short*>(p.aux)[o])
                               : reinterpret_cast<const float*>(p.aux)[o];
          v = h > 0.f ? v : 0.f;
        }
        if ((flags & MFP_GEMM_ROWSKIP) && p.rowcode[row + r] != 0) v = 0.f;
        if (flags & MFP_GEMM_DROPOUT) v = philox_keep(rnd[r], p.dropout_p) ? v * inv_keep : 0.f;
        if (flags & MFP_GEMM_RESIDUAL) v += p.residual[o];
        if (p.out_bf16) {
          reinterpret_cast<unsigned short*>(p.C)[o] = f32_to_bf16(v);
        } else {
          float* c = reinterpret_cast<float*>(p.C) + o;
          if (flags & MFP_GEMM_ACCUM) v += *c;
          *c = v;
        }
      }
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

template <typename T>
int launch_gemm(const mfp_gemm_args* a, const GemmParams& p, dim3 grid, hipStream_t st) {
  if (a->a_kmajor && !a->b_kmajor) {
    hipLaunchKernelGGL((gemm_kernel<T, true, false>), grid, dim3(NT), 0, st, p);
  } else if (a->a_kmajor && a->b_kmajor) {
    hipLaunchKernelGGL((gemm_kernel<T, true, true>), grid, dim3(NT), 0, st, p);
  } else if (!a->a_kmajor && !a->b_kmajor) {
    hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, dim3(NT), 0, st, p);
  } else {
    mfp_set_error("mfp_gemm: layout a_kmajor=0,b_kmajor=1 is not on the MFP path");
    return MFP_EINVAL;
  }
  return MFP_OK;
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
  MFP_CHECK_ARG(a->lda % epc == 0 && a->ldb % epc == 0);
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
  int rc = a->in_dtype == MFP_BF16 ? launch_gemm<unsigned short>(a, p, grid, st)
                                    : launch_gemm<float>(a, p, grid, st);
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
