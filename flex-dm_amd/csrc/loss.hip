// Fused masked losses of LossLayer (reference models/metrics.py:213-299): per attribute, one
// pass over the head's logits computes loss, score, denominator AND d(loss)/d(logits) (scaled
// by 1/B for the batch mean, metrics.py:277).  HBM-bound: logits read once (f32), dlogits
// written once (cdt).
//   categorical (metrics.py:36-49 + Keras CE-from-probabilities): p = softmax(z);
//     q = clip(p, 1e-7, 1-1e-7); loss = -log q_y + log sum_j q_j; score = [y == argmax p]
//     dz_i = p_i (g_i - sum_j g_j p_j),  g_j = [q_j unclipped] (-[j==y]/q_y + 1/sum q)
//   numerical (metrics.py:52-57, :247-248): loss = sum_j (p_j - y_j)^2 (= mse * width);
//     score = 0.5 cos(y, p) + 0.5;  dp_j = 2 (p_j - y_j)
//   weight(t) = mfp_mask[t] && cond(type[t]) && s < nvalid[b]   (metrics.py:251-267)
#include "common.h"

namespace {

struct LossKeys {
  mfp_loss_key k[MFP_MAX_LOSS_KEYS];
  int key_slot[MFP_MAX_LOSS_KEYS];  // row of `sums` this key accumulates into
  int n;
};

__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float token_weight(const mfp_loss_key& k, const int* nvalid, int t, int S) {
  const int b = t / S, s = t % S;
  bool w = k.mask[t] != 0 && s < nvalid[b];
  if (w && k.cond_idx != nullptr) {
    const int v = k.cond_idx[(long long)t * k.cond_stride];
    w = v >= 0 && v < 32 && ((k.cond_bits >> v) & 1u);
  }
  return w ? 1.f : 0.f;
}

constexpr int CE_MAXIT = 8;  // classes <= 128

template <typename TDL>
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, TDL* __restrict__ dlogits,
                                                 int ld, LossKeys keys, const int* __restrict__ nvalid,
                                                 float* __restrict__ sums, int T, int S, float inv_B) {
  __shared__ float red[3][16];
  const mfp_loss_key k = keys.k[blockIdx.y];
  const int C = k.n_class, NF = k.n_feat;
  const int grp = threadIdx.x >> 4, j16 = threadIdx.x & 15;
  const long long nitems = (long long)T * NF;
  float acc_loss = 0.f, acc_score = 0.f, acc_den = 0.f;
  for (long long item = (long long)blockIdx.x * 16 + grp; item < nitems; item += (long long)gridDim.x * 16) {
    const int t = (int)(item / NF), n = (int)(item % NF);
    const float w = token_weight(k, nvalid, t, S);
    const long long base = (long long)t * ld + k.col_off + n * C;
    if (w == 0.f) {  // group-uniform
      if (dlogits) {
        for (int j = j16; j < C; j += 16) cdt_traits<TDL>::store(dlogits + base + j, 0.f);
      }
      continue;
    }
    const int y = reinterpret_cast<const int*>(k.target)[(long long)t * NF + n];
    float z[CE_MAXIT];
    float m = -INFINITY;
#pragma unroll
    for (int it = 0; it < CE_MAXIT; ++it) {
      const int j = j16 + it * 16;
      z[it] = j < C ? logits[base + j] : -INFINITY;
      m = fmaxf(m, z[it]);
    }
    m = group16_max(m);
    // argmax: smallest index attaining the max
    int am = 1 << 30;
#pragma unroll
    for (int it = 0; it < CE_MAXIT; ++it) {
      const int j = j16 + it * 16;
      if (j < C && z[it] == m) am = min(am, j);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) am = min(am, __shfl_xor(am, o, 64));
    float e[CE_MAXIT];
    float se = 0.f;
#pragma unroll
    for (int it = 0; it < CE_MAXIT; ++it) {
      e[it] = (j16 + it * 16 < C) ? expf(z[it] - m) : 0.f;
      se += e[it];
    }
    se = group16_sum(se);
    const float inv = 1.f / se;
    float sq = 0.f, qy = 0.f;
    float g[CE_MAXIT];
#pragma unroll
    for (int it = 0; it < CE_MAXIT; ++it) {
      const int j = j16 + it * 16;
      const float p = e[it] * inv;
      e[it] = p;
      const float q = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
      g[it] = (j < C && p >= 1e-7f && p <= 1.f - 1e-7f) ? 1.f : 0.f;  // clip passes gradient
      if (j < C) sq += q;
      if (j == y) qy = q;
    }
    sq = group16_sum(sq);
    qy = group16_sum(qy);
    const float loss = -logf(qy) + logf(sq);
    const float inv_sq = 1.f / sq, inv_qy = 1.f / qy;
    float gp = 0.f;
#pragma unroll
    for (int it = 0; it < CE_MAXIT; ++it) {
      const int j = j16 + it * 16;
      g[it] = g[it] * ((j == y ? -inv_qy : 0.f) + inv_sq);
      gp += g[it] * e[it];
    }
    gp = group16_sum(gp);
    if (dlogits) {
#pragma unroll
      for (int it = 0; it < CE_MAXIT; ++it) {
        const int j = j16 + it * 16;
        if (j < C) cdt_traits<TDL>::store(dlogits + base + j, e[it] * (g[it] - gp) * inv_B);
      }
    }
    if (j16 == 0) {
      acc_loss += loss * inv_B;
      acc_score += (am == y) ? 1.f : 0.f;
      acc_den += 1.f;
    }
  }
  // block reduction: one value per 16-lane group leader
  if (j16 == 0) { red[0][grp] = acc_loss; red[1][grp] = acc_score; red[2][grp] = acc_den; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += red[threadIdx.x][i];
    if (s != 0.f) atomicAdd(&sums[keys.key_slot[blockIdx.y] * 3 + threadIdx.x], s);
  }
}

template <typename TDL>
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, TDL* __restrict__ dpred,
                                                  int ld, LossKeys keys, const int* __restrict__ nvalid,
                                                  float* __restrict__ sums, int T, int S, float inv_B) {
  __shared__ float red[3][4];
  const mfp_loss_key k = keys.k[blockIdx.y];
  const int W = k.n_class;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* target = reinterpret_cast<const float*>(k.target);
  float acc_loss = 0.f, acc_score = 0.f, acc_den = 0.f;
  for (int t = blockIdx.x * 4 + wave; t < T; t += gridDim.x * 4) {
    const float w = token_weight(k, nvalid, t, S);
    const long long base = (long long)t * ld + k.col_off;
    if (w == 0.f) {
      if (dpred) for (int j = lane; j < W; j += 64) cdt_traits<TDL>::store(dpred + base + j, 0.f);
      continue;
    }
    float sd = 0.f, sy = 0.f, sp = 0.f, syp = 0.f;
    for (int j = lane; j < W; j += 64) {
      const float p = pred[base + j], y = target[(long long)t * W + j];
      const float d = p - y;
      sd += d * d; sy += y * y; sp += p * p; syp += y * p;
      if (dpred) cdt_traits<TDL>::store(dpred + base + j, 2.f * d * inv_B);
    }
    sd = wave_sum(sd); sy = wave_sum(sy); sp = wave_sum(sp); syp = wave_sum(syp);
    if (lane == 0) {
      const float cosv = syp * rsqrtf(fmaxf(sy, 1e-12f)) * rsqrtf(fmaxf(sp, 1e-12f));
      acc_loss += sd * inv_B;
      acc_score += 0.5f * cosv + 0.5f;
      acc_den += 1.f;
    }
  }
  if (lane == 0) { red[0][wave] = acc_loss; red[1][wave] = acc_score; red[2][wave] = acc_den; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    if (s != 0.f) atomicAdd(&sums[keys.key_slot[blockIdx.y] * 3 + threadIdx.x], s);
  }
}

// hipMemsetAsync is avoided on purpose: captured into a hipGraph as a memset node it left
// stray words in the 120-byte `sums` buffer on ROCm 7.2 (replays only); a kernel node is exact.
__global__ void zero_sums_kernel(float* __restrict__ sums, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sums[i] = 0.f;
}

}  // namespace

extern "C" int mfp_loss_fwd_bwd(const float* logits, void* dlogits, int32_t ld, const mfp_loss_key* keys,
                                int32_t nkeys, const int32_t* nvalid, float* sums, int32_t B, int32_t S,
                                int32_t dl_dtype, mfp_stream_t stream) {
  MFP_CHECK_ARG(logits && keys && nvalid && sums);
  MFP_CHECK_ARG(nkeys > 0 && nkeys <= MFP_MAX_LOSS_KEYS && B > 0 && S > 0 && ld > 0);
  MFP_CHECK_ARG(dl_dtype == MFP_F32 || dl_dtype == MFP_BF16);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int T = B * S;
  LossKeys cat, num;
  cat.n = 0; num.n = 0;
  for (int i = 0; i < nkeys; ++i) {
    MFP_CHECK_ARG(keys[i].target && keys[i].mask && keys[i].n_class > 0 && keys[i].n_feat > 0);
    if (keys[i].is_numerical) {
      num.k[num.n] = keys[i]; num.key_slot[num.n] = i; num.n++;
    } else {
      MFP_CHECK_ARG(keys[i].n_class <= 16 * CE_MAXIT);
      cat.k[cat.n] = keys[i]; cat.key_slot[cat.n] = i; cat.n++;
    }
  }
  hipLaunchKernelGGL(zero_sums_kernel, dim3(1), dim3(64), 0, st, sums, nkeys * 3);
  MFP_CHECK_LAUNCH();
  const float inv_B = 1.0f / (float)B;
  if (cat.n > 0) {
    int bx = (T + 15) / 16;
    if (bx > 2048) bx = 2048;
    if (dl_dtype == MFP_F32)
      hipLaunchKernelGGL(ce_kernel<float>, dim3(bx, cat.n), dim3(256), 0, st, logits, (float*)dlogits, ld, cat, nvalid, sums, T, S, inv_B);
    else
      hipLaunchKernelGGL(ce_kernel<unsigned short>, dim3(bx, cat.n), dim3(256), 0, st, logits, (unsigned short*)dlogits, ld, cat, nvalid, sums, T, S, inv_B);
    MFP_CHECK_LAUNCH();
  }
  if (num.n > 0) {
    int bx = (T + 3) / 4;
    if (bx > 4096) bx = 4096;
    if (dl_dtype == MFP_F32)
      hipLaunchKernelGGL(mse_kernel<float>, dim3(bx, num.n), dim3(256), 0, st, logits, (float*)dlogits, ld, num, nvalid, sums, T, S, inv_B);
    else
      hipLaunchKernelGGL(mse_kernel<unsigned short>, dim3(bx, num.n), dim3(256), 0, st, logits, (unsigned short*)dlogits, ld, num, nvalid, sums, T, S, inv_B);
    MFP_CHECK_LAUNCH();
  }
  return MFP_OK;
}
