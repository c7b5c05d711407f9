// Optimizer + small streaming kernels of the train step.
//  * mfp_adam_keras: Keras Adam + per-variable clipnorm + L2 regularisers fused over one flat
//    parameter buffer (reference train.py:71-77; architecture/utils.py:8-22; [TF-EXT] Keras
//    OptimizerV2 Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), eps=1e-7 outside the sqrt; clipnorm =
//    per-variable tf.clip_by_norm).  HBM-bound: 16 B read + 12 B (+2 B bf16 shadow) written per
//    parameter, plus an 8 B/parameter norm pass.
//  * mfp_dropout_bwd: regenerates the Philox keep mask of MFP_GEMM_DROPOUT and emits the
//    (cdt) gradient of the Dense output together with its column sums (= bias gradient).
#include "common.h"
#include "reduce.h"

namespace {

constexpr int ADAM_CHUNK = 4096;  // elements per workgroup

// partial[chunk] = { sum (g*gs + 2*l2*w)^2 , sum w^2 } over the chunk.  No float atomics: the
// per-variable totals are formed in a FIXED order by adam_update_kernel, so data-parallel replicas
// (which hold bit-identical all-reduced gradients) compute bit-identical clip factors and stay in
// lock-step -- atomics made the norm's last bit depend on arrival order and replicas drifted.
__global__ __launch_bounds__(256) void adam_norm_kernel(const float* __restrict__ w,
                                                        const float* __restrict__ g,
                                                        const int* __restrict__ chunk_seg,
                                                        const long long* __restrict__ chunk_beg,
                                                        const int* __restrict__ chunk_len,
                                                        const float* __restrict__ seg_l2,
                                                        float* __restrict__ partial, float grad_scale,
                                                        int* __restrict__ step_t) {
  __shared__ float red[2][4];
  const int seg = chunk_seg[blockIdx.x];
  const long long beg = chunk_beg[blockIdx.x];
  const int len = chunk_len[blockIdx.x];
  const float l2 = seg_l2[seg];
  float sg = 0.f, sw = 0.f;
  for (int i = threadIdx.x; i < len; i += 256) {
    const float wi = w[beg + i];
    const float gi = g[beg + i] * grad_scale + 2.f * l2 * wi;
    sg += gi * gi; sw += wi * wi;
  }
  sg = wave_sum(sg); sw = wave_sum(sw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = sg; red[1][wave] = sw; }
  __syncthreads();
  if (threadIdx.x < 2)
    partial[blockIdx.x * 2 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
  if (blockIdx.x == 0 && threadIdx.x == 0) *step_t += 1;   // the update kernel (next launch) reads it
}

__global__ __launch_bounds__(256) void adam_update_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          unsigned short* __restrict__ shadow,
                                                          const int* __restrict__ chunk_seg,
                                                          const long long* __restrict__ chunk_beg,
                                                          const int* __restrict__ chunk_len,
                                                          const int* __restrict__ seg_first,
                                                          const float* __restrict__ seg_l2,
                                                          const float* __restrict__ partial,
                                                          float* __restrict__ stats,
                                                          const int* __restrict__ step_t, float lr, float b1,
                                                          float b2, float eps, float clipnorm, float grad_scale) {
  __shared__ float tot[2];
  const int seg = chunk_seg[blockIdx.x];
  const long long beg = chunk_beg[blockIdx.x];
  const int len = chunk_len[blockIdx.x];
  const float l2 = seg_l2[seg];
  // fixed-order total over this variable's chunks: wave 0, lane-strided then xor-shuffle tree
  if (threadIdx.x < 64) {
    const int c0 = seg_first[seg], c1 = seg_first[seg + 1];
    float sg = 0.f, sw = 0.f;
    for (int c = c0 + (int)threadIdx.x; c < c1; c += 64) { sg += partial[c * 2]; sw += partial[c * 2 + 1]; }
    sg = wave_sum(sg); sw = wave_sum(sw);
    if (threadIdx.x == 0) {
      tot[0] = sg; tot[1] = sw;
      if ((int)blockIdx.x == c0) { stats[seg * 2] = sg; stats[seg * 2 + 1] = sw; }
    }
  }
  __syncthreads();
  const float t = (float)(*step_t);  // already incremented (1-based)
  const float lr_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
  float clip = 1.f;
  if (clipnorm > 0.f) clip = clipnorm / fmaxf(sqrtf(tot[0]), clipnorm);
  for (int i = threadIdx.x; i < len; i += 256) {
    const long long o = beg + i;
    const float wi = w[o];
    const float gi = (g[o] * grad_scale + 2.f * l2 * wi) * clip;
    const float mi = b1 * m[o] + (1.f - b1) * gi;
    const float vi = b2 * v[o] + (1.f - b2) * gi * gi;
    const float wn = wi - lr_t * mi / (sqrtf(vi) + eps);
    m[o] = mi; v[o] = vi; w[o] = wn;
    if (shadow) shadow[o] = f32_to_bf16(wn);
  }
}

__global__ void cast_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long long n) {
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    float4 v = *reinterpret_cast<const float4*>(src + i);
    u32x2 pk = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    *reinterpret_cast<u32x2*>(dst + i) = pk;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long j = n & ~3ll; j < n; ++j) dst[j] = f32_to_bf16(src[j]);
}

// thread = (row, 4 consecutive columns): the dropout keying of the GEMM epilogue (common.h).  Block = 256 threads = (N/4 column quads) x (1024/N rows per pass).
// TIN = float, or bf16 (unsigned short) when the step carries the residual gradient in the compute dtype (mfp_dropout_bwd_res16)
template <typename TOUT, typename TIN = float>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const TIN* __restrict__ dx, TOUT* __restrict__ dy,
                                                          float* __restrict__ colsum_part, int M, int N,
                                                          float p, unsigned long long seed,
                                                          unsigned long long offset0, const int* step_ptr,
                                                          int rows_per_block) {
  __shared__ float red[4][256];
  const unsigned long long offset = offset0 + (step_ptr ? (unsigned long long)(*step_ptr) * MFP_RNG_STEP_STRIDE : 0ull);
  const int nq = N >> 2;                      // column quads per row
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const unsigned int dkey = drop_key(seed, offset), dthr = drop_thr16(p);
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  for (int qb = blockIdx.x * 256; qb < nq; qb += gridDim.x * 256) {   // usually one pass
    // threads cover quads [qb, qb+256) of a row when nq >= 256, else several rows at once
    const int qpr = min(nq - qb, 256);           // quads of this row slice handled per row
    const int rows_at_once = 256 / qpr;
    const int q = qb + threadIdx.x % qpr, rsub = threadIdx.x / qpr;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rsub < rows_at_once) {
      for (int r = r0 + rsub; r < r1; r += rows_at_once) {
        const long long o = (long long)r * N + q * 4;
        float4 v;
        if constexpr (sizeof(TIN) == 4) {
          v = *reinterpret_cast<const float4*>(dx + o);
        } else {
          const u32x2 h = *reinterpret_cast<const u32x2*>(dx + o);
          v = make_float4(__uint_as_float(h[0] << 16), __uint_as_float(h[0] & 0xFFFF0000u), __uint_as_float(h[1] << 16),
                          __uint_as_float(h[1] & 0xFFFF0000u));
        }
        if (p > 0.f) {
          bool keep[4];
          drop_keep4(drop_row(dkey, (unsigned int)r), (unsigned int)(q * 4), dthr, keep);
          v.x = keep[0] ? v.x * inv_keep : 0.f;
          v.y = keep[1] ? v.y * inv_keep : 0.f;
          v.z = keep[2] ? v.z * inv_keep : 0.f;
          v.w = keep[3] ? v.w * inv_keep : 0.f;
        }
        if constexpr (sizeof(TOUT) == 4) {
          *reinterpret_cast<float4*>(dy + o) = v;
        } else {
          u32x2 pk = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
          *reinterpret_cast<u32x2*>(dy + o) = pk;
        }
        cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
      }
    }
    red[0][threadIdx.x] = cs.x; red[1][threadIdx.x] = cs.y; red[2][threadIdx.x] = cs.z; red[3][threadIdx.x] = cs.w;
    __syncthreads();
    if (threadIdx.x < qpr) {
      float s[4] = {0.f, 0.f, 0.f, 0.f};
      for (int rs = 0; rs < rows_at_once; ++rs)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += red[e][rs * qpr + threadIdx.x];
      float* out = colsum_part + (long long)blockIdx.y * N + (qb + threadIdx.x) * 4;
      out[0] = s[0]; out[1] = s[1]; out[2] = s[2]; out[3] = s[3];
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, float* __restrict__ part, int M,
                                                     int N, int ld, int rows_per_block) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  if (c >= N) return;
  float cs = 0.f;
  for (int r = r0; r < r1; ++r) cs += cdt_traits<T>::load(X + (long long)r * ld + c);
  part[(long long)blockIdx.y * N + c] = cs;
}

}  // namespace

// The chunk table (segment id, start, length per <=4096-element chunk) is static per model: it
// is built once on the host into caller memory and uploaded by the caller (no library state).
// out[ooff + c*old + r] = bf16(w[off + r*cols + c]) for each listed [rows][cols] matrix of the flat
// buffer: the transposed bf16 shadow that lets a dgrad (dY * W with W stored [out][in]) run as the
// k-major weight-stationary product (old > rows places several matrices side by side, e.g. the
// fused QKV transposed as one [D][3D]).  grid = (max tiles of 32x32, matrices).
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ w,
                                                             unsigned short* __restrict__ out,
                                                             const long long* __restrict__ seg_off,
                                                             const int* __restrict__ seg_rows,
                                                             const int* __restrict__ seg_cols,
                                                             const long long* __restrict__ seg_ooff,
                                                             const int* __restrict__ seg_old) {
  __shared__ float tile[32][33];
  const int sgi = blockIdx.y, rows = seg_rows[sgi], cols = seg_cols[sgi];
  const int tiles_c = (cols + 31) / 32, tiles_r = (rows + 31) / 32;
  if ((int)blockIdx.x >= tiles_c * tiles_r) return;
  const long long off = seg_off[sgi], ooff = seg_ooff[sgi];
  const int old = seg_old[sgi];
  const int r0 = (blockIdx.x / tiles_c) * 32, c0 = (blockIdx.x % tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < rows && c < cols) ? w[off + (long long)r * cols + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < rows && c < cols) out[ooff + (long long)c * old + r] = f32_to_bf16(tile[tx][ty + 8 * k]);
  }
}

extern "C" int64_t mfp_adam_num_chunks(const int32_t* seg_off_host, int32_t nseg) {
  int64_t n = 0;
  for (int s = 0; s < nseg; ++s) {
    int64_t len = (int64_t)seg_off_host[s + 1] - seg_off_host[s];
    n += (len + ADAM_CHUNK - 1) / ADAM_CHUNK;
  }
  return n;
}

// Fills host arrays (caller copies them to the device once): chunk_seg int32[nchunks],
// chunk_beg int64[nchunks], chunk_len int32[nchunks], seg_first int32[nseg+1] (first chunk of
// each variable).
extern "C" int mfp_adam_chunk_table(const int32_t* seg_off_host, int32_t nseg, int32_t* chunk_seg,
                                    int64_t* chunk_beg, int32_t* chunk_len, int32_t* seg_first) {
  MFP_CHECK_ARG(seg_off_host && chunk_seg && chunk_beg && chunk_len && seg_first && nseg > 0);
  int64_t k = 0;
  for (int s = 0; s < nseg; ++s) {
    seg_first[s] = (int32_t)k;
    int64_t beg = seg_off_host[s], end = seg_off_host[s + 1];
    for (int64_t b = beg; b < end; b += ADAM_CHUNK) {
      chunk_seg[k] = s; chunk_beg[k] = b;
      chunk_len[k] = (int32_t)((end - b) < ADAM_CHUNK ? (end - b) : ADAM_CHUNK);
      ++k;
    }
  }
  seg_first[nseg] = (int32_t)k;
  return MFP_OK;
}

extern "C" int mfp_adam_keras(float* w, const float* g, float* m, float* v, uint16_t* shadow,
                              const int32_t* chunk_seg, const int64_t* chunk_beg, const int32_t* chunk_len,
                              const int32_t* seg_first, int64_t nchunks, const float* seg_l2, float* partial,
                              float* stats, int32_t nseg, int32_t* step_t, float lr, float beta1, float beta2,
                              float eps, float clipnorm, float grad_scale, mfp_stream_t stream) {
  MFP_CHECK_ARG(w && g && m && v && chunk_seg && chunk_beg && chunk_len && seg_first && seg_l2 && partial && stats && step_t);
  MFP_CHECK_ARG(nchunks > 0 && nseg > 0);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(adam_norm_kernel, dim3((unsigned)nchunks), dim3(256), 0, st, w, g, chunk_seg,
                     reinterpret_cast<const long long*>(chunk_beg), chunk_len, seg_l2, partial, grad_scale, step_t);
  MFP_CHECK_LAUNCH();
  hipLaunchKernelGGL(adam_update_kernel, dim3((unsigned)nchunks), dim3(256), 0, st, w, g, m, v, shadow,
                     chunk_seg, reinterpret_cast<const long long*>(chunk_beg), chunk_len, seg_first, seg_l2, partial,
                     stats, step_t, lr, beta1, beta2, eps, clipnorm, grad_scale);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_cast_f32_bf16(const float* src, uint16_t* dst, int64_t n, mfp_stream_t stream) {
  MFP_CHECK_ARG(src && dst && n > 0);
  MFP_CHECK_ARG(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0);
  int blocks = (int)((n / 4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cast_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst,
                     (long long)n);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

static int rows_per_block_for(int M) {
  int rpb = (M + 1023) / 1024;  // <= 1024 row blocks
  if (rpb < 32) rpb = 32;
  return rpb;
}

extern "C" size_t mfp_colsum_workspace_bytes(int32_t M, int32_t N) {
  int rpb = rows_per_block_for(M);
  return (size_t)((M + rpb - 1) / rpb) * N * sizeof(float);
}

extern "C" int mfp_dropout_bwd(const float* dx, void* dy, float* colsum, void* workspace,
                               size_t workspace_bytes, int32_t M, int32_t N, float p, uint64_t seed,
                               uint64_t offset, const int32_t* step_ptr, int32_t out_dtype,
                               mfp_stream_t stream) {
  MFP_CHECK_ARG(dx && dy && colsum && M > 0 && N > 0 && N % 4 == 0 && p >= 0.f && p < 1.f);
  MFP_CHECK_ARG(out_dtype == MFP_F32 || out_dtype == MFP_BF16);
  if (!workspace || workspace_bytes < mfp_colsum_workspace_bytes(M, N)) {
    mfp_set_error("mfp_dropout_bwd: workspace too small");
    return MFP_EWORKSPACE;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rpb = rows_per_block_for(M), nrb = (M + rpb - 1) / rpb;
  dim3 grid(1, nrb);
  float* part = reinterpret_cast<float*>(workspace);
  if (out_dtype == MFP_F32)
    hipLaunchKernelGGL(dropout_bwd_kernel<float>, grid, dim3(256), 0, st, dx, (float*)dy, part, M, N, p, seed, offset, step_ptr, rpb);
  else
    hipLaunchKernelGGL(dropout_bwd_kernel<unsigned short>, grid, dim3(256), 0, st, dx, (unsigned short*)dy, part, M, N, p, seed, offset, step_ptr, rpb);
  MFP_CHECK_LAUNCH();
  launch_reduce_rows(part, colsum, colsum, N, nrb, N, N, st);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_dropout_bwd_res16(const void* dx, void* dy, float* colsum, void* workspace, size_t workspace_bytes, int32_t M,
                                     int32_t N, float p, uint64_t seed, uint64_t offset, const int32_t* step_ptr,
                                     mfp_stream_t stream) {
  MFP_CHECK_ARG(dx && dy && colsum && M > 0 && N > 0 && N % 4 == 0 && p >= 0.f && p < 1.f);
  if (!workspace || workspace_bytes < mfp_colsum_workspace_bytes(M, N)) {
    mfp_set_error("mfp_dropout_bwd_res16: workspace too small");
    return MFP_EWORKSPACE;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rpb = rows_per_block_for(M), nrb = (M + rpb - 1) / rpb;
  float* part = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL((dropout_bwd_kernel<unsigned short, unsigned short>), dim3(1, nrb), dim3(256), 0, st, (const unsigned short*)dx,
                     (unsigned short*)dy, part, M, N, p, seed, offset, step_ptr, rpb);
  MFP_CHECK_LAUNCH();
  launch_reduce_rows(part, colsum, colsum, N, nrb, N, N, st);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_colsum(const void* X, float* colsum, void* workspace, size_t workspace_bytes, int32_t M,
                          int32_t N, int32_t ld, int32_t dtype, mfp_stream_t stream) {
  MFP_CHECK_ARG(X && colsum && M > 0 && N > 0 && ld >= N);
  MFP_CHECK_ARG(dtype == MFP_F32 || dtype == MFP_BF16);
  if (!workspace || workspace_bytes < mfp_colsum_workspace_bytes(M, N)) {
    mfp_set_error("mfp_colsum: workspace too small");
    return MFP_EWORKSPACE;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rpb = rows_per_block_for(M), nrb = (M + rpb - 1) / rpb;
  dim3 grid((N + 255) / 256, nrb);
  float* part = reinterpret_cast<float*>(workspace);
  if (dtype == MFP_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)X, part, M, N, ld, rpb);
  else
    hipLaunchKernelGGL(colsum_kernel<unsigned short>, grid, dim3(256), 0, st, (const unsigned short*)X, part, M, N, ld, rpb);
  MFP_CHECK_LAUNCH();
  launch_reduce_rows(part, colsum, colsum, N, nrb, N, N, st);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_transpose_cast_bf16(const float* w, uint16_t* out, const int64_t* seg_off,
                                       const int32_t* seg_rows, const int32_t* seg_cols,
                                       const int64_t* seg_ooff, const int32_t* seg_old, int32_t nseg,
                                       int32_t max_tiles, mfp_stream_t stream) {
  MFP_CHECK_ARG(w && out && seg_off && seg_rows && seg_cols && seg_ooff && seg_old && nseg > 0 && max_tiles > 0);
  hipLaunchKernelGGL(transpose_cast_kernel, dim3((unsigned)max_tiles, (unsigned)nseg), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), w, out, reinterpret_cast<const long long*>(seg_off),
                     seg_rows, seg_cols, reinterpret_cast<const long long*>(seg_ooff), seg_old);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}
