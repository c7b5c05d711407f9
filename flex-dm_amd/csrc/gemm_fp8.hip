// fp8 (OCP e4m3) forward Dense for BASELINE config c5 ("fp8 MFMA"): the QKV and FFN1 products of a
// DeepSVG block (reference architecture/transformer.py:85-90,161-166) with per-tensor scaling.
//
//   C[M][N] = relu?( (Xq Wq^T) / (sx * sw) + bias ),   Xq = e4m3(sx * X),  Wq = e4m3(sw * W)
//
//   * weights: quantised once per optimizer step from the f32 master copy (mfp_quantize_fp8: scale
//     sw = 448 / amax(W) per Keras variable, kept on the device);
//   * activations: X is the bf16 LayerNorm output; its amax comes from mfp_absmax (256 block maxima, no
//     atomics, no zero fill -- the consumer reduces them), sx = 448 / amax, and X is quantised ON THE FLY
//     while the tile is staged into LDS (16 k-values = 32 B of bf16 -> one 16-byte fp8 chunk);
//   * v_mfma_f32_16x16x32_fp8_fp8 (f32 accumulation).  A lane's 16-byte LDS fragment feeds TWO
//     consecutive k-steps (low / high 8 bytes): the induced permutation of k is the same for both
//     operands, so the contraction is unchanged and the fragment reads stay 16 bytes wide (8-byte LDS
//     reads from one wave per SIMD run at a fifth of the LDS rate);
//   * the backward pass is unchanged (bf16 operands from the saved bf16 activations: straight-through).
//
// This path exists for the c5 precision mode; it is an LDS-tiled kernel (128 x 128 x 128, 4 waves), not
// the weight-stationary design: non-scaled fp8 MFMA runs at the bf16 rate on gfx950 and these products
// are HBM-bound, so fp8 buys accuracy measurements here, not time.
#include "common.h"

namespace {

constexpr int ABSMAX_PARTS = 256;
constexpr float FP8_MAX = 448.0f;   // e4m3fn

__global__ __launch_bounds__(256) void absmax_kernel(const void* __restrict__ x, long long n, int is_bf16,
                                                     float* __restrict__ parts) {
  __shared__ float red[4];
  float m = 0.f;
  const long long stride = (long long)gridDim.x * 256 * 8;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
    if (is_bf16) {
      if (i + 8 <= n) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(x) + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          m = fmaxf(m, fabsf(bf16_to_f32((unsigned short)(v[e] & 0xffffu))));
          m = fmaxf(m, fabsf(bf16_to_f32((unsigned short)(v[e] >> 16))));
        }
      } else {
        for (long long j = i; j < n; ++j) m = fmaxf(m, fabsf(bf16_to_f32(reinterpret_cast<const unsigned short*>(x)[j])));
      }
    } else {
      for (long long j = i; j < n && j < i + 8; ++j) m = fmaxf(m, fabsf(reinterpret_cast<const float*>(x)[j]));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) parts[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ float scale_from_parts(const float* __restrict__ parts) {
  // every lane reduces the 256 block maxima (4 loads per lane + wave reduction)
  const int lane = threadIdx.x & 63;
  float m = fmaxf(fmaxf(parts[lane], parts[64 + lane]), fmaxf(parts[128 + lane], parts[192 + lane]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  return m > 0.f ? FP8_MAX / m : 1.0f;
}

// 8 floats -> 8 e4m3 bytes (saturating at +-448)
__device__ __forceinline__ u32x2 cvt8_fp8(const float (&v)[8]) {
  float c[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) c[e] = fminf(fmaxf(v[e], -FP8_MAX), FP8_MAX);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  return (u32x2){(unsigned int)lo, (unsigned int)hi};
}

__global__ __launch_bounds__(256) void quantize_fp8_kernel(const float* __restrict__ w, long long n,
                                                           const float* __restrict__ parts, unsigned char* __restrict__ out,
                                                           float* __restrict__ scale_out) {
  const float s = scale_from_parts(parts);
  if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = s;
  const long long stride = (long long)gridDim.x * 256 * 8;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8; i + 8 <= n; i += stride) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(w + i), b = *reinterpret_cast<const f32x4*>(w + i + 4);
    const float v[8] = {a[0] * s, a[1] * s, a[2] * s, a[3] * s, b[0] * s, b[1] * s, b[2] * s, b[3] * s};
    *reinterpret_cast<u32x2*>(out + i) = cvt8_fp8(v);
  }
}

// C[M][N] (bf16) = relu?((Xq Wq^T) / (sx sw) + bias); X bf16 [M][lda], Wq fp8 [N][K]
template <bool RELU>
__global__ __launch_bounds__(256) void gemm_fp8_kernel(const unsigned short* __restrict__ X, const unsigned char* __restrict__ Wq,
                                                       const float* __restrict__ x_parts, const float* __restrict__ w_scale,
                                                       const float* __restrict__ bias, unsigned short* __restrict__ C,
                                                       int M, int N, int K, int lda, int ldc) {
  constexpr int BM = 128, BN = 128, BK = 128, ROWB = BK + 16;   // LDS row: 128 fp8 + 16 B pad (36 words: 16 rows hit 16 bank quads)
  __shared__ __attribute__((aligned(16))) unsigned char As[BM * ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[BN * ROWB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (N + BN - 1) / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;      // the column tiles of a row tile are neighbours: X from L2
  const int m0 = tm * BM, n0 = tn * BN;
  const float sx = scale_from_parts(x_parts), sw = *w_scale;
  const float inv = 1.0f / (sx * sw);
  // staging plan: chunk = 16 k-values; thread -> chunk column q = tid & 7, rows (tid >> 3) + 32 c
  const int q = tid & 7, r0 = tid >> 3;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 xr[4][2], wr[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = r0 + 32 * c, k = k0 + q * 16;
      const bool xok = m0 + row < M && k < K, wok = n0 + row < N && k < K;
      const u32x4 z = {0u, 0u, 0u, 0u};
      xr[c][0] = xok ? *reinterpret_cast<const u32x4*>(X + (long long)(m0 + row) * lda + k) : z;
      xr[c][1] = xok ? *reinterpret_cast<const u32x4*>(X + (long long)(m0 + row) * lda + k + 8) : z;
      wr[c] = wok ? *reinterpret_cast<const u32x4*>(Wq + (long long)(n0 + row) * K + k) : z;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = r0 + 32 * c;
      float v[8];
      u32x4 o;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] = bf16_to_f32((unsigned short)(xr[c][h][e] & 0xffffu)) * sx;
          v[2 * e + 1] = bf16_to_f32((unsigned short)(xr[c][h][e] >> 16)) * sx;
        }
        const u32x2 pk = cvt8_fp8(v);
        o[2 * h] = pk[0]; o[2 * h + 1] = pk[1];
      }
      *reinterpret_cast<u32x4*>(As + row * ROWB + q * 16) = o;
      *reinterpret_cast<u32x4*>(Bs + row * ROWB + q * 16) = wr[c];
    }
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();          // previous tile fully consumed
    lstore();
    __syncthreads();
    if (k0 + BK < K) gload(k0 + BK);      // in flight while this tile is multiplied
#pragma unroll
    for (int sp = 0; sp < BK / 64; ++sp) {      // a 16-byte fragment = the operands of two k-steps
      u32x4 xf[4], wf[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        xf[a] = *reinterpret_cast<const u32x4*>(As + (wm * 64 + a * 16 + li) * ROWB + (sp * 4 + lg) * 16);
        wf[a] = *reinterpret_cast<const u32x4*>(Bs + (wn * 64 + a * 16 + li) * ROWB + (sp * 4 + lg) * 16);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const long wa = (long)(((unsigned long long)wf[b][2 * h + 1] << 32) | wf[b][2 * h]);
            const long xa = (long)(((unsigned long long)xf[a][2 * h + 1] << 32) | xf[a][2 * h]);
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wa, xa, acc[a][b], 0, 0, 0);
          }
    }
  }
  // acc[a][b][r] = C[m = 16 a + li][n = 16 b + 4 lg + r] of the wave's 64 x 64 block
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int m = m0 + wm * 64 + a * 16 + li;
    if (m >= M) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int n = n0 + wn * 64 + b * 16 + 4 * lg;
      if (n >= N) continue;      // N % 8 == 0: a 4-column group is in or out as a whole
      const f32x4 bb = bias ? *reinterpret_cast<const f32x4*>(bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[a][b][r] * inv + bb[r];
        if (RELU) v[r] = fmaxf(v[r], 0.f);
      }
      *reinterpret_cast<u32x2*>(C + (long long)m * ldc + n) = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
  }
}

}  // namespace

extern "C" int mfp_absmax(const void* x, int64_t n, int32_t dtype, float* parts, mfp_stream_t stream) {
  MFP_CHECK_ARG(x && parts && n > 0 && (dtype == MFP_F32 || dtype == MFP_BF16));
  MFP_CHECK_ARG(((uintptr_t)x % 16) == 0);
  hipLaunchKernelGGL(absmax_kernel, dim3(ABSMAX_PARTS), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     (long long)n, dtype == MFP_BF16 ? 1 : 0, parts);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_quantize_fp8(const float* w, int64_t n, const float* parts, uint8_t* out, float* scale_out,
                                mfp_stream_t stream) {
  MFP_CHECK_ARG(w && parts && out && scale_out && n > 0 && n % 8 == 0);
  MFP_CHECK_ARG(((uintptr_t)w % 16) == 0 && ((uintptr_t)out % 8) == 0);
  const int blocks = (int)((n / 8 + 255) / 256 < 1024 ? (n / 8 + 255) / 256 : 1024);
  hipLaunchKernelGGL(quantize_fp8_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
                     (long long)n, parts, out, scale_out);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_gemm_fp8(const void* X, const uint8_t* Wq, const float* x_parts, const float* w_scale,
                            const float* bias, void* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldc,
                            int32_t relu, mfp_stream_t stream) {
  MFP_CHECK_ARG(X && Wq && x_parts && w_scale && C && M > 0 && N > 0 && K > 0);
  MFP_CHECK_ARG(K % 16 == 0 && N % 8 == 0 && lda % 8 == 0 && ldc % 4 == 0 && lda >= K && ldc >= N);
  MFP_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)Wq % 16) == 0 && ((uintptr_t)C % 8) == 0);
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (relu)
    hipLaunchKernelGGL(gemm_fp8_kernel<true>, dim3(tiles), dim3(256), 0, st, (const unsigned short*)X, Wq, x_parts, w_scale,
                       bias, (unsigned short*)C, M, N, K, lda, ldc);
  else
    hipLaunchKernelGGL(gemm_fp8_kernel<false>, dim3(tiles), dim3(256), 0, st, (const unsigned short*)X, Wq, x_parts, w_scale,
                       bias, (unsigned short*)C, M, N, K, lda, ldc);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}
