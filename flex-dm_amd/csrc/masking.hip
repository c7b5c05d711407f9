// Fused MFP input masking for the train step: filter_padding + random_masking + elem_masking +
// feat_masking + the per-document task select of preprocess_for_train (reference
// models/masking.py:24-53,68-155,227-269; models/mfp.py:95-138) in ONE launch.
//
// The reference computes all 7 task variants of every attribute with ~20 small eager ops each and
// tf.where-selects by task id; here one wave owns one element (token): its lanes decide every
// categorical (column, feature) at once and then stream the 512-wide numerical rows.  Outputs go
// straight into the buffers the encoder kernels read:
//   idx_all [T][NCOL] int32 : modified categorical indices (<MASK> = C, <UNUSED> = C+1) and, in
//                             the special columns, 0 / 1 / -1 for numerical <MASK>/<UNUSED>/none
//   rowcode [T] u8, x_out [T][W] cdt : numerical attributes (encoder.py:165-175 semantics)
//   mask_out [T] u8 : the MFP mask of each attribute (what LossLayer weighs by)
// HBM-bound: ~W*(4 + e) bytes per numerical attribute per element + O(100) bytes of indices.
//
// Randomness is counter-based Philox keyed by (seed, element, column, step): same distribution as
// the reference (MASK_PROB .15, 90 % changed of which 1/9 random token), not the same stream.
#include "common.h"

namespace {

struct MaskCols {
  mfp_mask_col c[MFP_MAX_MASK_COLS];
  int n;
};

__device__ __forceinline__ float u01(unsigned int r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

constexpr float MASK_PROB = 0.15f;          // masking.py:11
constexpr float CHANGE_PROB = 0.9f;         // masking.py:13-14
constexpr float THRESH = 0.1f / 0.9f;       // masking.py:15

// decision for (element t, column k): 0 keep, 1 <MASK>, 2 <UNUSED>, 3 random token; *mfp = mask bit
__device__ __forceinline__ int decide(const mfp_mask_col& col, int k, int t, int s, int nv, int task, int sel,
                                      unsigned long long seed, unsigned long long off, bool* mfp,
                                      unsigned int* rnd_extra) {
  const bool valid = s < nv;
  bool unused = !valid;
  if (col.cond_idx != nullptr && valid) {
    const int v = col.cond_idx[(long long)t * col.cond_stride];
    unused = !(v >= 0 && v < 32 && ((col.cond_bits >> v) & 1u));
  }
  int code = unused ? 2 : 0;
  bool m = false;
  if (task == 0) {
    unsigned int r[4];
    philox4x32(seed, (unsigned int)t, (unsigned int)k, off, r);
    m = valid && u01(r[0]) < MASK_PROB;
    const bool chg = m && u01(r[1]) < CHANGE_PROB;
    if (chg) code = u01(r[2]) >= THRESH ? 1 : 3;
    *rnd_extra = r[3];
  } else if (task == 1) {
    m = valid && s == sel;
    if (m) code = 1;
  } else {
    m = valid && col.group == task - 2;
    if (m) code = 1;
  }
  *mfp = m;
  return code;
}

// Round 5 (second half): SIXTEEN tokens per workgroup.  Phase 1: thread (token, column) decides its pair (one Philox), writes the
// mask bit, handles a categorical column's features, leaves the numerical columns' codes in LDS.  Phase 2: all 256 threads stream
// the workgroup's numerical rows (16 tokens x W floats per column), every load of a thread (eight float4 at W = 512) requested
// before its first store.  The wave-per-token form below walked a token's columns one after the other -- a chain of ~12
// dependent load -> store round trips with 32 tokens in flight per CU: 55 us at c2 for 45 MB read + 70 MB written.  Same draws
// (Philox is keyed by (seed, token, column / feature / float4 index, step)): outputs bit-identical to the wave-per-token form.
constexpr int MK_TOK = 16;
#ifndef MK_ABL
#define MK_ABL 0      // timing experiment for tools/bench_mask.py ONLY (the numerical rows stay stale): 1 = no numerical rows
#endif
template <typename TX>
__global__ __launch_bounds__(256) void mask_kernel(MaskCols cols, int* __restrict__ idx_all, int NCOL,
                                                   const int* __restrict__ nvalid, const int* __restrict__ tasks,
                                                   int T, int S, unsigned long long seed,
                                                   unsigned long long offset0, const int* __restrict__ step_ptr) {
  __shared__ unsigned char s_code[MK_TOK][MFP_MAX_MASK_COLS];
  const int tid = threadIdx.x;
  const unsigned long long off = offset0 + (step_ptr ? (unsigned long long)(*step_ptr) * MFP_RNG_STEP_STRIDE : 0ull);
  {
    const int tok = tid >> 4, k = tid & 15, t = blockIdx.x * MK_TOK + tok;
    if (t < T && k < cols.n) {
      const int b = t / S, s = t % S;
      const int nv = nvalid[b], task = tasks[b];
      int sel = -1;
      if (task == 1) {  // select_single_element (masking.py:98-113): floor(U * length)
        unsigned int r[4];
        philox4x32(seed, (unsigned int)b, 0xE1E0E1E0u, off, r);
        sel = min((int)(u01(r[0]) * (float)nv), nv - 1);
      }
      const mfp_mask_col& col = cols.c[k];
      bool mfp;
      unsigned int extra;
      const int code = decide(col, k, t, s, nv, task, sel, seed, off, &mfp, &extra);
      col.mask_out[t] = mfp ? 1 : 0;
      s_code[tok][k] = (unsigned char)code;
      if (!col.is_numerical) {
        const int* src = reinterpret_cast<const int*>(col.src) + (long long)t * col.n_feat;
        int* dst = idx_all + (long long)t * NCOL + col.idx_col;
        for (int f = 0; f < col.n_feat; ++f) {
          int v = src[f];
          if (code == 2) v = col.input_dim + 1;
          else if (code == 1) v = col.input_dim;
          else if (code == 3) {
            unsigned int r[4];
            philox4x32(seed, (unsigned int)t, (unsigned int)(k + 64 * (f + 1)), off, r);
            v = (int)(((unsigned long long)r[0] * (unsigned long long)col.input_dim) >> 32);
          }
          dst[f] = v;
        }
      } else {
        col.rowcode[t] = (unsigned char)((code == 1 || code == 2) ? code : 0);
        idx_all[(long long)t * NCOL + col.idx_col] = code == 1 ? 0 : (code == 2 ? 1 : -1);
      }
    }
  }
  __syncthreads();
  const int t0 = blockIdx.x * MK_TOK;
  if (MK_ABL == 1) return;
  for (int k = 0; k < cols.n; ++k) {
    const mfp_mask_col& col = cols.c[k];
    if (!col.is_numerical) continue;
    const int W = col.n_feat, q = W >> 3;                 // 8-float units per row (W % 8 == 0: host check)
    const int total = MK_TOK * q;                          // units of the workgroup's rows of this column
    const float* srcb = reinterpret_cast<const float*>(col.src);
    TX* dstb = reinterpret_cast<TX*>(col.x_out);
    // a thread's unit = 8 consecutive floats: two 16-byte loads, ONE 16-byte bf16 store (8-byte-per-lane stores run at about half
    // the rate of 16-byte ones: MI355X_MICROARCH.md)
    for (int base = 0; base < total; base += 256 * 4) {    // four units per thread and round (W = 512: one round)
      float4 v[4][2];
      int code[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = base + tid + 256 * i;
        const int tok = idx / q, t = t0 + tok;
        code[i] = (idx < total && t < T) ? (int)s_code[tok][k] : -1;
        if (code[i] == 0) {
          const float* sp = srcb + (long long)t * W + (idx - tok * q) * 8;
          v[i][0] = *reinterpret_cast<const float4*>(sp);
          v[i][1] = *reinterpret_cast<const float4*>(sp + 4);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (code[i] < 0) continue;
        const int idx = base + tid + 256 * i;
        const int tok = idx / q, t = t0 + tok, c = (idx - tok * q) * 8;
        float4 o[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (code[i] == 0) {
            o[h] = v[i][h];
          } else if (code[i] == 1) {
            o[h] = make_float4(10.f, 10.f, 10.f, 10.f);     // MASK_VALUE (masking.py:8)
          } else if (code[i] == 2) {
            o[h] = make_float4(0.f, 0.f, 0.f, 0.f);         // NULL_VALUE (masking.py:9)
          } else {  // tf.random.normal(stddev=0.1): Box-Muller on 4 Philox uniforms per float4
            unsigned int r[4];
            philox4x32(seed, (unsigned int)t, (unsigned int)(k + 64 * (1 + (c + 4 * h) / 4)), off ^ 0x5bd1e995ull, r);
            const float u0 = fmaxf(u01(r[0]), 1e-7f), u1 = u01(r[1]), u2 = fmaxf(u01(r[2]), 1e-7f), u3 = u01(r[3]);
            const float ra = 0.1f * sqrtf(-2.f * logf(u0)), rb = 0.1f * sqrtf(-2.f * logf(u2));
            o[h] = make_float4(ra * cosf(6.2831853f * u1), ra * sinf(6.2831853f * u1),
                               rb * cosf(6.2831853f * u3), rb * sinf(6.2831853f * u3));
          }
        }
        TX* dst = dstb + (long long)t * W + c;
        if constexpr (sizeof(TX) == 4) {
          *reinterpret_cast<float4*>(dst) = o[0];
          *reinterpret_cast<float4*>(dst + 4) = o[1];
        } else {
          const u32x4 pk = {pack_bf16x2(o[0].x, o[0].y), pack_bf16x2(o[0].z, o[0].w), pack_bf16x2(o[1].x, o[1].y), pack_bf16x2(o[1].z, o[1].w)};
          *reinterpret_cast<u32x4*>(dst) = pk;
        }
      }
    }
  }
}

// The wave-per-token form of rounds 1-5 (MFP_MASK_WAVE=1: A/B)
template <typename TX>
__global__ __launch_bounds__(256) void mask_wave_kernel(MaskCols cols, int* __restrict__ idx_all, int NCOL,
                                                   const int* __restrict__ nvalid, const int* __restrict__ tasks,
                                                   int T, int S, unsigned long long seed,
                                                   unsigned long long offset0, const int* __restrict__ step_ptr) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const unsigned long long off = offset0 + (step_ptr ? (unsigned long long)(*step_ptr) * MFP_RNG_STEP_STRIDE : 0ull);
  const int b = t / S, s = t % S;
  const int nv = nvalid[b], task = tasks[b];
  int sel = -1;
  if (task == 1) {  // select_single_element (masking.py:98-113): floor(U * length)
    unsigned int r[4];
    philox4x32(seed, (unsigned int)b, 0xE1E0E1E0u, off, r);
    sel = min((int)(u01(r[0]) * (float)nv), nv - 1);
  }
  // every column's decision at once: lane k decides column k (Philox4x32-10 is ~800 clk; one
  // after the other, wave-uniform, the 12 columns of Crello made this kernel VALU-bound at 75 us)
  int mycode = 0;
  bool mymfp = false;
  if (lane < cols.n) {
    unsigned int extra;
    mycode = decide(cols.c[lane], lane, t, s, nv, task, sel, seed, off, &mymfp, &extra);
  }
  for (int k = 0; k < cols.n; ++k) {
    const mfp_mask_col& col = cols.c[k];
    const int code = __shfl(mycode, k, 64);
    const bool mfp = __shfl((int)mymfp, k, 64) != 0;
    if (lane == 0) col.mask_out[t] = mfp ? 1 : 0;
    if (!col.is_numerical) {
      if (lane < col.n_feat) {
        int v = reinterpret_cast<const int*>(col.src)[(long long)t * col.n_feat + lane];
        if (code == 2) v = col.input_dim + 1;
        else if (code == 1) v = col.input_dim;
        else if (code == 3) {
          unsigned int r[4];
          philox4x32(seed, (unsigned int)t, (unsigned int)(k + 64 * (lane + 1)), off, r);
          v = (int)(((unsigned long long)r[0] * (unsigned long long)col.input_dim) >> 32);
        }
        idx_all[(long long)t * NCOL + col.idx_col + lane] = v;
      }
    } else {
      const int W = col.n_feat;
      if (lane == 0) {
        col.rowcode[t] = (unsigned char)((code == 1 || code == 2) ? code : 0);
        idx_all[(long long)t * NCOL + col.idx_col] = code == 1 ? 0 : (code == 2 ? 1 : -1);
      }
      const float* src = reinterpret_cast<const float*>(col.src) + (long long)t * W;
      TX* dst = reinterpret_cast<TX*>(col.x_out) + (long long)t * W;
      for (int c = lane * 4; c < W; c += 256) {
        float4 v;
        if (code == 0) {
          v = *reinterpret_cast<const float4*>(src + c);
        } else if (code == 1) {
          v = make_float4(10.f, 10.f, 10.f, 10.f);     // MASK_VALUE (masking.py:8)
        } else if (code == 2) {
          v = make_float4(0.f, 0.f, 0.f, 0.f);         // NULL_VALUE (masking.py:9)
        } else {  // tf.random.normal(stddev=0.1): Box-Muller on 4 Philox uniforms
          unsigned int r[4];
          philox4x32(seed, (unsigned int)t, (unsigned int)(k + 64 * (1 + c / 4)), off ^ 0x5bd1e995ull, r);
          const float u0 = fmaxf(u01(r[0]), 1e-7f), u1 = u01(r[1]), u2 = fmaxf(u01(r[2]), 1e-7f), u3 = u01(r[3]);
          const float ra = 0.1f * sqrtf(-2.f * logf(u0)), rb = 0.1f * sqrtf(-2.f * logf(u2));
          v = make_float4(ra * cosf(6.2831853f * u1), ra * sinf(6.2831853f * u1),
                          rb * cosf(6.2831853f * u3), rb * sinf(6.2831853f * u3));
        }
        if constexpr (sizeof(TX) == 4) {
          *reinterpret_cast<float4*>(dst + c) = v;
        } else {
          u32x2 pk = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
          *reinterpret_cast<u32x2*>(dst + c) = pk;
        }
      }
    }
  }
}

// tasks[b] ~ Categorical(probs) (reference mfp.py:34-43,301: tfp Categorical(logits=log probs)):
// inverse-CDF on one Philox uniform per document.
struct TaskProbs { float cdf[16]; int n; };
__global__ void sample_tasks_kernel(TaskProbs tp, int* __restrict__ tasks, int B, unsigned long long seed,
                                    unsigned long long offset0, const int* __restrict__ step_ptr) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const unsigned long long off = offset0 + (step_ptr ? (unsigned long long)(*step_ptr) * MFP_RNG_STEP_STRIDE : 0ull);
  unsigned int r[4];
  philox4x32(seed, (unsigned int)b, 0x7A5C7A5Cu, off, r);
  const float u = u01(r[0]) * tp.cdf[tp.n - 1];
  int k = 0;
  while (k < tp.n - 1 && u >= tp.cdf[k]) ++k;
  tasks[b] = k;
}

// The head of a train step in one launch: task ids (as sample_tasks_kernel, same stream), the number of valid
// positions per document (reference architecture/mask.py: get_seq_mask(length) counts length + 1 elements) and
// the zeroing of the step's loss accumulators -- three tiny launches of ~4.6 us each otherwise.
__global__ void step_prologue_kernel(TaskProbs tp, int* __restrict__ tasks, const int* __restrict__ length,
                                     int* __restrict__ nvalid, int B, unsigned long long seed, unsigned long long offset0,
                                     const int* __restrict__ step_ptr, float* __restrict__ zero, int nzero) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nzero) zero[b] = 0.f;
  if (b >= B) return;
  if (nvalid != nullptr) nvalid[b] = length[b] + 1;
  const unsigned long long off = offset0 + (step_ptr ? (unsigned long long)(*step_ptr) * MFP_RNG_STEP_STRIDE : 0ull);
  unsigned int r[4];
  philox4x32(seed, (unsigned int)b, 0x7A5C7A5Cu, off, r);
  const float u = u01(r[0]) * tp.cdf[tp.n - 1];
  int k = 0;
  while (k < tp.n - 1 && u >= tp.cdf[k]) ++k;
  tasks[b] = k;
}

}  // namespace

extern "C" int mfp_step_prologue(const float* probs, int32_t n, int32_t* tasks, const int32_t* length, int32_t* nvalid,
                                 int32_t B, uint64_t seed, uint64_t offset, const int32_t* step_ptr, float* zero,
                                 int32_t nzero, mfp_stream_t stream) {
  MFP_CHECK_ARG(probs && tasks && n > 0 && n <= 16 && B > 0 && nzero >= 0 && (nzero == 0 || zero != nullptr));
  MFP_CHECK_ARG((nvalid == nullptr) || (length != nullptr));
  TaskProbs tp;
  tp.n = n;
  float c = 0.f;
  for (int i = 0; i < n; ++i) {
    MFP_CHECK_ARG(probs[i] >= 0.f);
    c += probs[i];
    tp.cdf[i] = c;
  }
  MFP_CHECK_ARG(c > 0.f);
  const int nthr = B > nzero ? B : nzero;
  hipLaunchKernelGGL(step_prologue_kernel, dim3((nthr + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     tp, tasks, length, nvalid, B, seed, offset, step_ptr, zero, nzero);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_sample_tasks(const float* probs, int32_t n, int32_t* tasks, int32_t B, uint64_t seed,
                                uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream) {
  MFP_CHECK_ARG(probs && tasks && n > 0 && n <= 16 && B > 0);
  TaskProbs tp;
  tp.n = n;
  float c = 0.f;
  for (int i = 0; i < n; ++i) {
    MFP_CHECK_ARG(probs[i] >= 0.f);
    c += probs[i];
    tp.cdf[i] = c;
  }
  MFP_CHECK_ARG(c > 0.f);
  hipLaunchKernelGGL(sample_tasks_kernel, dim3((B + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     tp, tasks, B, seed, offset, step_ptr);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_mask_tokens(const mfp_mask_col* cols, int32_t ncols, int32_t* idx_all, int32_t NCOL,
                               const int32_t* nvalid, const int32_t* tasks, int32_t B, int32_t S, uint64_t seed,
                               uint64_t offset, const int32_t* step_ptr, int32_t x_dtype, mfp_stream_t stream) {
  MFP_CHECK_ARG(cols && idx_all && nvalid && tasks);
  MFP_CHECK_ARG(ncols > 0 && ncols <= MFP_MAX_MASK_COLS && B > 0 && S > 0 && NCOL > 0);
  MFP_CHECK_ARG(x_dtype == MFP_F32 || x_dtype == MFP_BF16);
  MaskCols mc;
  mc.n = ncols;
  for (int i = 0; i < ncols; ++i) {
    MFP_CHECK_ARG(cols[i].src && cols[i].mask_out);
    if (cols[i].is_numerical) MFP_CHECK_ARG(cols[i].x_out && cols[i].rowcode && cols[i].n_feat % 8 == 0 && ((uintptr_t)cols[i].x_out % 16) == 0);
    else MFP_CHECK_ARG(cols[i].n_feat >= 1 && cols[i].n_feat <= 64 && cols[i].input_dim > 0);
    mc.c[i] = cols[i];
  }
  const int T = B * S;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const char* env = getenv("MFP_MASK_WAVE");
  if (env != nullptr && env[0] == '1') {      // (A/B: the wave-per-token form)
    if (x_dtype == MFP_F32)
      hipLaunchKernelGGL(mask_wave_kernel<float>, dim3((T + 3) / 4), dim3(256), 0, st, mc, idx_all, NCOL, nvalid, tasks, T, S,
                         seed, offset, step_ptr);
    else
      hipLaunchKernelGGL(mask_wave_kernel<unsigned short>, dim3((T + 3) / 4), dim3(256), 0, st, mc, idx_all, NCOL, nvalid,
                         tasks, T, S, seed, offset, step_ptr);
  } else if (x_dtype == MFP_F32) {
    hipLaunchKernelGGL(mask_kernel<float>, dim3((T + MK_TOK - 1) / MK_TOK), dim3(256), 0, st, mc, idx_all, NCOL, nvalid, tasks, T, S,
                       seed, offset, step_ptr);
  } else {
    hipLaunchKernelGGL(mask_kernel<unsigned short>, dim3((T + MK_TOK - 1) / MK_TOK), dim3(256), 0, st, mc, idx_all, NCOL, nvalid,
                       tasks, T, S, seed, offset, step_ptr);
  }
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}
