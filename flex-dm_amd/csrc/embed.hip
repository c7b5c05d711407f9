// Fused multi-attribute embedding gather + element pooling (reference
// architecture/encoder.py:156-160,167-175,194-199): one pass reads NCOL int32 per element and
// sums the addressed rows of all (tiny, L2-resident) tables into the element's D-vector.
// HBM-bound: algorithmic bytes/element = NCOL*4 read + D*4 write (fwd).
#include "common.h"
#include "reduce.h"

namespace {

// Two dependent round trips per token instead of 2 * NCOL: all index loads are issued first
// (a token's D/4 threads read the same words: one broadcast transaction each), then all table rows
// (L2-resident: the tables are < 1 MB), then the sum.  MAXC is the compile-time column bound.
template <int MAXC>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int* __restrict__ idx,
                                                        const int* __restrict__ rowoff,
                                                        const float* __restrict__ tables,
                                                        float* __restrict__ out, int T, int NCOL, int D) {
  // thread = (token, 4 consecutive d); D/4 threads per token
  const int dv = D >> 2;
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)T * dv;
  int off[MAXC];
#pragma unroll
  for (int j = 0; j < MAXC; ++j) off[j] = j < NCOL ? rowoff[j] : 0;
  for (; gid < total; gid += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(gid / dv), c = (int)(gid % dv) * 4;
    const int* it = idx + (long long)t * NCOL;
    int r[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) r[j] = j < NCOL ? it[j] : -1;
    float4 v[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j)
      v[j] = r[j] >= 0 ? *reinterpret_cast<const float4*>(tables + (long long)(off[j] + r[j]) * D + c)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < MAXC; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }   // column order: as before
    *reinterpret_cast<float4*>(out + (long long)t * D + c) = acc;
  }
}

// The same sum with the tables in LDS: a workgroup owns a 64-column slice of ALL tables (ROWS x 256 B, 86 KB at
// Crello) for a strided set of tokens -- the gathers come from LDS (one ds_read_b128 per column and lane, a token's
// 16 lanes read one whole 256-byte row: conflict-free) instead of 12 KB of L2 reads per token (400 MB per step
// through the L1s: 21 us).  16 lanes per token, 16 tokens per pass; the sum order is the column order, as above.
template <int MAXC>
__global__ __launch_bounds__(1024) void embed_fwd_lds_kernel(const int* __restrict__ idx, const int* __restrict__ rowoff,
                                                             const float* __restrict__ tables, float* __restrict__ out,
                                                             int T, int NCOL, int ROWS, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tab_s[];      // [ROWS][64] f32
  const int c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < ROWS * 16; i += 1024) {
    const int r = i >> 4, q = i & 15;
    *reinterpret_cast<float4*>(tab_s + r * 256 + q * 16) = *reinterpret_cast<const float4*>(tables + (long long)r * D + c0 + q * 4);
  }
  int off[MAXC];
#pragma unroll
  for (int j = 0; j < MAXC; ++j) off[j] = j < NCOL ? rowoff[j] : 0;
  const int q = threadIdx.x & 15;
  const int stride = gridDim.x * 64;
  int t = blockIdx.x * 64 + (threadIdx.x >> 4);
  // the index words of the NEXT token of this lane group are in flight while the current one is summed
  int rn[MAXC];
#pragma unroll
  for (int j = 0; j < MAXC; ++j) rn[j] = (j < NCOL && t < T) ? idx[(long long)t * NCOL + j] : -1;
  __syncthreads();
  for (; t < T; t += stride) {
    int r[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) r[j] = rn[j];
    const int tn = t + stride;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) rn[j] = (j < NCOL && tn < T) ? idx[(long long)tn * NCOL + j] : -1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
      if (r[j] >= 0) {
        const float4 v = *reinterpret_cast<const float4*>(tab_s + (off[j] + r[j]) * 256 + q * 16);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    *reinterpret_cast<float4*>(out + (long long)t * D + c0 + q * 4) = acc;
  }
}

// Backward: dtables[rowoff[c] + idx[t][c]] += dout[t].  Workgroup = (token chunk, 16-wide d
// slice); the slice of ALL tables lives in LDS and is accumulated with 64-bit FIXED-POINT integer
// atomics: on gfx950 ds_add_f32 costs ~195 cycles per wave-op against ~19 for ds_add_u64
// (tools/ubench/lds_atomics.hip), and integer adds make the sum order-independent (deterministic).
// Scale 2^34: resolution 5.8e-11, range +-5.4e8 per (chunk, row, d) -- a chunk is <= 512 tokens.
// A 16-lane group owns a token (4 tokens per wave instruction); the NCOL <= 16 indices of a token
// come from ONE vector load and are broadcast by shuffle.  Partials [chunk][ROWS][D] (float) are
// summed by reduce_rows_kernel: no global atomics.
constexpr int EB_DSLICE = 16;
constexpr float EB_SCALE = 17179869184.0f;          // 2^34
constexpr float EB_INV_SCALE = 1.0f / 17179869184.0f;

__global__ __launch_bounds__(256) void embed_bwd_kernel(const int* __restrict__ idx,
                                                        const int* __restrict__ rowoff,
                                                        const float* __restrict__ dout,
                                                        float* __restrict__ part, int T, int NCOL,
                                                        int ROWS, int D, int tok_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];  // [ROWS][16]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, grp = lane >> 4;
  const int chunk = blockIdx.x, d0 = blockIdx.y * EB_DSLICE;
  for (int i = threadIdx.x; i < ROWS * EB_DSLICE; i += 256) tab[i] = 0ull;
  __syncthreads();
  const int myoff = l16 < NCOL ? rowoff[l16] : 0;
  const int t0 = chunk * tok_per_chunk, t1 = min(T, t0 + tok_per_chunk);
  for (int tb = t0 + wave * 4; tb < t1; tb += 16) {
    const int t = tb + grp;
    long long q = 0;
    int myrow = -1;
    if (t < t1) {
      q = (long long)(dout[(long long)t * D + d0 + l16] * EB_SCALE);
      if (l16 < NCOL) {
        const int r = idx[(long long)t * NCOL + l16];
        myrow = r < 0 ? -1 : myoff + r;
      }
    }
    for (int j = 0; j < NCOL; ++j) {
      const int row = __shfl(myrow, grp * 16 + j, 64);
      if (row >= 0) atomicAdd(&tab[row * EB_DSLICE + l16], (unsigned long long)q);
    }
  }
  __syncthreads();
  float* pout = part + (long long)chunk * ROWS * D;
  for (int i = threadIdx.x; i < ROWS * EB_DSLICE; i += 256) {
    int r = i / EB_DSLICE, c = i % EB_DSLICE;
    pout[(long long)r * D + d0 + c] = (float)(long long)tab[i] * EB_INV_SCALE;
  }
}

// P[t][r] = bf16(#index columns of token t addressing row r of the concatenated tables), r < ROWSP.
// With it the table gradient is the product P^T[ROWS][T] * dh[T][D] on the wgrad GEMM (exact one-hot
// counts 0..3 in bf16; dh in the compute dtype like every other weight gradient) -- off the
// critical path on the side stream, instead of the LDS-atomic scatter kernel above (106 us).
// One wave per token; a lane owns 32-bit words (two adjacent rows) l, l + 64, ...
constexpr int OH_TOK = 8;     // tokens per wave: their index words are loaded together (one round trip, not eight)
constexpr int OH_MAXW = 512;  // 32-bit words per row (ROWSP <= 1024)
__global__ __launch_bounds__(256) void embed_onehot_kernel(const int* __restrict__ idx,
                                                           const int* __restrict__ rowoff,
                                                           unsigned int* __restrict__ P, int T, int NCOL,
                                                           int ROWSP) {
  // counts per wave in LDS: the NCOL index lanes bump 16-bit halves of their row's word (the compare-and-count
  // walk over all columns for every output word was 180 VALU instructions per token: 20 us)
  __shared__ unsigned int cnt[4][OH_MAXW];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int t0 = (blockIdx.x * 4 + wv) * OH_TOK;
  if (t0 >= T) return;
  int myrow[OH_TOK];
  const int ro = lane < NCOL ? rowoff[lane] : 0;
#pragma unroll
  for (int u = 0; u < OH_TOK; ++u) {
    const int t = min(t0 + u, T - 1);
    const int r = lane < NCOL ? idx[(long long)t * NCOL + lane] : -1;
    myrow[u] = r < 0 ? -1 : ro + r;
  }
  const int words = ROWSP >> 1;
  unsigned int* c = cnt[wv];
#pragma unroll
  for (int u = 0; u < OH_TOK; ++u) {
    if (t0 + u >= T) break;
    for (int w = lane; w < words; w += 64) c[w] = 0u;
    __builtin_amdgcn_wave_barrier();
    if (myrow[u] >= 0) atomicAdd(&c[myrow[u] >> 1], (myrow[u] & 1) ? 0x10000u : 1u);
    __builtin_amdgcn_wave_barrier();
    for (int w = lane; w < words; w += 64) {
      const unsigned int v = c[w];
      // bf16 bit patterns of 0, 1, 2, 3, ... (small integers are exact)
      const unsigned int lo = f32_to_bf16((float)(v & 0xFFFFu)), hi = f32_to_bf16((float)(v >> 16));
      P[(long long)(t0 + u) * words + w] = lo | (hi << 16);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// rowcode: 1 = all == 10.0 (<MASK>), 2 = all == 0.0 (<UNUSED>); one wave per row of K floats.
__global__ __launch_bounds__(256) void row_flags_kernel(const float* __restrict__ x,
                                                        unsigned char* __restrict__ rowcode,
                                                        int* __restrict__ special_idx, int idx_stride,
                                                        int T, int K) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const float* xr = x + (long long)row * K;
  bool all10 = true, all0 = true;
  for (int c = lane * 4; c < K; c += 256) {
    float4 v = *reinterpret_cast<const float4*>(xr + c);
    all10 = all10 && v.x == 10.0f && v.y == 10.0f && v.z == 10.0f && v.w == 10.0f;
    all0 = all0 && v.x == 0.0f && v.y == 0.0f && v.z == 0.0f && v.w == 0.0f;
  }
  const bool m = __all(all10), u = __all(all0);
  if (lane == 0) {
    const int code = u ? 2 : (m ? 1 : 0);  // unused wins (encoder.py:174-175 order)
    rowcode[row] = (unsigned char)code;
    if (special_idx) special_idx[(long long)row * idx_stride] = code - 1;
  }
}

int embed_chunks(int T) {
  int chunks = (T + 255) / 256;  // >= 256 tokens per chunk
  if (chunks > 64) chunks = 64;
  if (chunks < 1) chunks = 1;
  return chunks;
}

}  // namespace

extern "C" int mfp_embed_pool_fwd(const int32_t* idx, const int32_t* rowoff, const float* tables,
                                  float* out, int32_t T, int32_t NCOL, int32_t ROWS, int32_t D,
                                  mfp_stream_t stream) {
  MFP_CHECK_ARG(idx && rowoff && tables && out);
  MFP_CHECK_ARG(T > 0 && NCOL > 0 && ROWS > 0 && D > 0 && D % 4 == 0);
  long long total = (long long)T * (D / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  MFP_CHECK_ARG(NCOL <= 32);
  const size_t lds_tab = (size_t)ROWS * 256;
  if (NCOL <= 16 && D % 64 == 0 && lds_tab <= 150 * 1024 && T >= 4096) {
    static bool attr_done[MFP_MAX_DEVICES] = {};
    bool& attr_set = attr_done[mfp_device_slot()];
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(embed_fwd_lds_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      if (e != hipSuccess) {
        mfp_set_error("mfp_embed_pool_fwd: cannot raise dynamic LDS: %s", hipGetErrorString(e));
        return MFP_ELAUNCH;
      }
      attr_set = true;
    }
    const int slices = D / 64;
    int per = 256 / slices;          // workgroups per slice: one per CU in total
    if (per < 1) per = 1;
    hipLaunchKernelGGL(embed_fwd_lds_kernel<16>, dim3(per, slices), dim3(1024), lds_tab, reinterpret_cast<hipStream_t>(stream),
                       idx, rowoff, tables, out, T, NCOL, ROWS, D);
    MFP_CHECK_LAUNCH();
    return MFP_OK;
  }
  if (NCOL <= 16)
    hipLaunchKernelGGL(embed_fwd_kernel<16>, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       idx, rowoff, tables, out, T, NCOL, D);
  else
    hipLaunchKernelGGL(embed_fwd_kernel<32>, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       idx, rowoff, tables, out, T, NCOL, D);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" size_t mfp_embed_pool_bwd_workspace_bytes(int32_t T, int32_t NCOL, int32_t ROWS, int32_t D) {
  (void)NCOL;
  return (size_t)embed_chunks(T) * ROWS * D * sizeof(float);
}

extern "C" int mfp_embed_pool_bwd(const int32_t* idx, const int32_t* rowoff, const float* dout,
                                  float* dtables, void* workspace, size_t workspace_bytes, int32_t T,
                                  int32_t NCOL, int32_t ROWS, int32_t D, mfp_stream_t stream) {
  MFP_CHECK_ARG(idx && rowoff && dout && dtables);
  MFP_CHECK_ARG(T > 0 && NCOL > 0 && NCOL <= 16 && ROWS > 0 && D > 0 && D % EB_DSLICE == 0);
  const size_t lds = (size_t)ROWS * EB_DSLICE * sizeof(unsigned long long);
  MFP_CHECK_ARG(lds <= 160 * 1024);
  if (!workspace || workspace_bytes < mfp_embed_pool_bwd_workspace_bytes(T, NCOL, ROWS, D)) {
    mfp_set_error("mfp_embed_pool_bwd: workspace too small");
    return MFP_EWORKSPACE;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int chunks = embed_chunks(T);
  const int tpc = (T + chunks - 1) / chunks;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(embed_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_embed_pool_bwd: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
  }
  float* part = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(chunks, D / EB_DSLICE), dim3(256), lds, st, idx, rowoff,
                     dout, part, T, NCOL, ROWS, D, tpc);
  MFP_CHECK_LAUNCH();
  long long n = (long long)ROWS * D;
  launch_reduce_rows(part, dtables, dtables, n, chunks, n, n, st);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_row_flags(const float* x, uint8_t* rowcode, int32_t* special_idx, int32_t idx_stride,
                             int32_t T, int32_t K, mfp_stream_t stream) {
  MFP_CHECK_ARG(x && rowcode && T > 0 && K > 0 && K % 4 == 0);
  hipLaunchKernelGGL(row_flags_kernel, dim3((T + 3) / 4), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, rowcode, special_idx, idx_stride, T, K);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_embed_onehot(const int32_t* idx, const int32_t* rowoff, uint16_t* P, int32_t T,
                                int32_t NCOL, int32_t ROWSP, mfp_stream_t stream) {
  MFP_CHECK_ARG(idx && rowoff && P && T > 0 && NCOL > 0 && NCOL <= 64 && ROWSP > 0 && ROWSP % 8 == 0 && ROWSP <= 2 * OH_MAXW);
  hipLaunchKernelGGL(embed_onehot_kernel, dim3((T + 4 * OH_TOK - 1) / (4 * OH_TOK)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     idx, rowoff, reinterpret_cast<unsigned int*>(P), T, NCOL, ROWSP);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}
