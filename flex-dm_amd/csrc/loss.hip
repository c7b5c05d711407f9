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

__device__ __forceinline__ float token_weight(const mfp_loss_key& k, const int* nvalid, int t, int S) {
  const int b = t / S, s = t % S;
  bool w = k.mask[t] != 0 && s < nvalid[b];
  if (w && k.cond_idx != nullptr) {
    const int v = k.cond_idx[(long long)t * k.cond_stride];
    w = v >= 0 && v < 32 && ((k.cond_bits >> v) & 1u);
  }
  return w ? 1.f : 0.f;
}

constexpr int CE_TOK = 32;        // token rows per workgroup
constexpr int CE_MAX_RANGES = 4;  // contiguous column ranges holding categorical heads

struct CeRanges {
  int beg[CE_MAX_RANGES], len[CE_MAX_RANGES], lds_off[CE_MAX_RANGES];
  int n, width;  // width = sum(len) (+1 pad) = LDS row stride in floats
};

// Categorical heads, tile-wise.  The old per-item kernel was a chain of dependent global loads
// (mask -> nvalid -> condition -> label -> logits) per 16-lane group: 165 us for ~70 MB.  Here a
// workgroup bulk-loads CE_TOK rows of every categorical column range into LDS (all requests in
// flight at once, row-contiguous), one THREAD then owns one (token, feature) item and walks its C
// classes in LDS (odd row stride: conflict-free), writing d(logits) back in place; the rows are
// finally stored coalesced.  Loss / score / denominator: block reduction + one atomic per key.
template <typename TDL>
__global__ __launch_bounds__(256) void ce_tile_kernel(const float* __restrict__ logits, TDL* __restrict__ dlogits,
                                                      int ld, LossKeys keys, CeRanges rg, const int* __restrict__ nvalid,
                                                      float* __restrict__ sums, int T, int S, float inv_B) {
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [CE_TOK][rg.width]
  __shared__ float red[MFP_MAX_LOSS_KEYS][3];
  const int t0 = blockIdx.x * CE_TOK;
  const int W = rg.width;
  if (threadIdx.x < MFP_MAX_LOSS_KEYS * 3) (&red[0][0])[threadIdx.x] = 0.f;
  // ---- stage: every (row, range) segment, coalesced along the row
  for (int r = 0; r < rg.n; ++r) {
    const int len = rg.len[r];
    for (int i = threadIdx.x; i < CE_TOK * len; i += 256) {
      const int row = i / len, c = i % len, t = t0 + row;
      tile[row * W + rg.lds_off[r] + c] = t < T ? logits[(long long)t * ld + rg.beg[r] + c] : 0.f;
    }
  }
  __syncthreads();
  // ---- phase A: thread = (token row, key, feature) item; inactive items (weight 0, ~85 % under the
  // 15 % masking rate) just zero their d(logits); active ones are COMPACTED so that phase B runs
  // without lane divergence.
  __shared__ int nactive;
  __shared__ unsigned short active[CE_TOK * 16];
  if (threadIdx.x == 0) nactive = 0;
  int nitem_per_tok = 0;
  for (int k = 0; k < keys.n; ++k) nitem_per_tok += keys.k[k].n_feat;
  __syncthreads();
  auto locate = [&](int it, int& row, int& k, int& f, int& pos) {
    row = it / nitem_per_tok;
    f = it % nitem_per_tok; k = 0;
    while (f >= keys.k[k].n_feat) { f -= keys.k[k].n_feat; ++k; }
    const int col = keys.k[k].col_off + f * keys.k[k].n_class;
    pos = -1;
    for (int r = 0; r < rg.n; ++r)
      if (col >= rg.beg[r] && col < rg.beg[r] + rg.len[r]) pos = rg.lds_off[r] + col - rg.beg[r];
  };
  for (int it = threadIdx.x; it < CE_TOK * nitem_per_tok; it += 256) {
    int row, k, f, pos;
    locate(it, row, k, f, pos);
    const int t = t0 + row;
    if (t >= T) continue;
    if (token_weight(keys.k[k], nvalid, t, S) == 0.f) {
      float* z = tile + row * W + pos;
      for (int j = 0; j < keys.k[k].n_class; ++j) z[j] = 0.f;
    } else {
      active[atomicAdd(&nactive, 1)] = (unsigned short)it;
    }
  }
  __syncthreads();
  // ---- phase B: active items spread round-robin over the 4 waves; one exp per class
  const int na = nactive;
  for (int a0 = 0; a0 < na; a0 += 256) {
    const int a = a0 + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6);
    if (a >= na) continue;
    int row, k, f, pos;
    locate(active[a], row, k, f, pos);
    const mfp_loss_key& key = keys.k[k];
    const int t = t0 + row, C = key.n_class;
    float* z = tile + row * W + pos;
    const int y = reinterpret_cast<const int*>(key.target)[(long long)t * key.n_feat + f];
    float m = -INFINITY;
    int am = 0;
    for (int j = 0; j < C; ++j) { const float v = z[j]; if (v > m) { m = v; am = j; } }
    float se = 0.f;
    for (int j = 0; j < C; ++j) { const float e = expf(z[j] - m); z[j] = e; se += e; }
    const float inv = 1.f / se;
    float sq = 0.f, qy = 0.f;
    for (int j = 0; j < C; ++j) {
      const float pj = z[j] * inv;
      const float q = fminf(fmaxf(pj, 1e-7f), 1.f - 1e-7f);
      sq += q;
      if (j == y) qy = q;
    }
    const float loss = -logf(qy) + logf(sq);
    const float inv_sq = 1.f / sq, inv_qy = 1.f / qy;
    float gp = 0.f;
    for (int j = 0; j < C; ++j) {
      const float pj = z[j] * inv;
      const float g = (pj >= 1e-7f && pj <= 1.f - 1e-7f) ? ((j == y ? -inv_qy : 0.f) + inv_sq) : 0.f;
      gp += g * pj;
    }
    for (int j = 0; j < C; ++j) {
      const float pj = z[j] * inv;
      const float g = (pj >= 1e-7f && pj <= 1.f - 1e-7f) ? ((j == y ? -inv_qy : 0.f) + inv_sq) : 0.f;
      z[j] = pj * (g - gp) * inv_B;
    }
    atomicAdd(&red[k][0], loss * inv_B);
    atomicAdd(&red[k][1], am == y ? 1.f : 0.f);
    atomicAdd(&red[k][2], 1.f);
  }
  __syncthreads();
  // ---- store d(logits) rows coalesced, publish the sums
  if (dlogits) {
    for (int r = 0; r < rg.n; ++r) {
      const int len = rg.len[r];
      for (int i = threadIdx.x; i < CE_TOK * len; i += 256) {
        const int row = i / len, c = i % len, t = t0 + row;
        if (t < T) cdt_traits<TDL>::store(dlogits + (long long)t * ld + rg.beg[r] + c, tile[row * W + rg.lds_off[r] + c]);
      }
    }
  }
  if (threadIdx.x < keys.n * 3) {
    const int k = threadIdx.x / 3, c = threadIdx.x % 3;
    const float v = red[k][c];
    if (v != 0.f) atomicAdd(&sums[keys.key_slot[k] * 3 + c], v);
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
      cat.k[cat.n] = keys[i]; cat.key_slot[cat.n] = i; cat.n++;
    }
  }
  hipLaunchKernelGGL(zero_sums_kernel, dim3(1), dim3(64), 0, st, sums, nkeys * 3);
  MFP_CHECK_LAUNCH();
  const float inv_B = 1.0f / (float)B;
  if (cat.n > 0) {
    // merge the categorical heads' columns into contiguous ranges (Crello: [0,319) and [1343,1378))
    CeRanges rg;
    rg.n = 0; rg.width = 0;
    for (int i = 0; i < cat.n; ++i) {
      const int beg = cat.k[i].col_off, len = cat.k[i].n_feat * cat.k[i].n_class;
      if (rg.n > 0 && rg.beg[rg.n - 1] + rg.len[rg.n - 1] == beg) {
        rg.len[rg.n - 1] += len;
      } else {
        MFP_CHECK_ARG(rg.n < CE_MAX_RANGES);
        rg.beg[rg.n] = beg; rg.len[rg.n] = len; rg.n++;
      }
    }
    for (int r = 0; r < rg.n; ++r) { rg.lds_off[r] = rg.width; rg.width += rg.len[r]; }
    rg.width |= 1;   // odd row stride: threads of different rows hit different banks
    const size_t lds = (size_t)CE_TOK * rg.width * sizeof(float);
    MFP_CHECK_ARG(lds <= 60 * 1024);
    {
      int items = 0;
      for (int i = 0; i < cat.n; ++i) items += cat.k[i].n_feat;
      MFP_CHECK_ARG(items <= 16);   // active[] capacity
    }
    const int bx = (T + CE_TOK - 1) / CE_TOK;
    if (dl_dtype == MFP_F32)
      hipLaunchKernelGGL(ce_tile_kernel<float>, dim3(bx), dim3(256), lds, st, logits, (float*)dlogits, ld, cat, rg, nvalid, sums, T, S, inv_B);
    else
      hipLaunchKernelGGL(ce_tile_kernel<unsigned short>, dim3(bx), dim3(256), lds, st, logits, (unsigned short*)dlogits, ld, cat, rg, nvalid, sums, T, S, inv_B);
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
