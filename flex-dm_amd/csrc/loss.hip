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

// The three loads are issued unconditionally and independently (one memory latency, not a chain of
// three): these kernels are latency-bound, their workgroups live for a handful of round trips.
// `tt` = row of the TARGET tensors that sits at position t (== t unless the loss is position-sorted:
// the mfp mask and the sequence mask stay positional, metrics.py:251,263).
__device__ __forceinline__ float token_weight(const mfp_loss_key& k, const int* nvalid, int t, int tt, int S) {
  const int b = t / S, s = t % S;
  const unsigned char m = k.mask[t];
  const int nv = nvalid[b];
  const int v = k.cond_idx != nullptr ? k.cond_idx[(long long)tt * k.cond_stride] : 0;
  const bool c = k.cond_idx == nullptr || (v >= 0 && v < 32 && ((k.cond_bits >> v) & 1u));
  return (m != 0 && s < nv && c) ? 1.f : 0.f;
}

// token rows per workgroup (16 lanes each): 32 when the staged rows fit 64 KB of LDS (fewer, fatter workgroups: the
// per-workgroup fixed costs -- barriers, compaction, sum atomics -- dominate; 8 rows +33 us, 4 rows +80 us, 32 rows
// -10 us per step against 16), else 16
constexpr int CE_MAX_RANGES = 4;  // contiguous column ranges holding categorical heads
constexpr int CE_MAX_ITEMS = 16;  // (key, feature) items per token
constexpr int CE_MAX_GAPS = 20;   // padding column runs (< 8 columns each) inside the ranges

struct CeRanges {
  int beg[CE_MAX_RANGES], len[CE_MAX_RANGES], lds_off[CE_MAX_RANGES];
  int n, width;     // width = LDS row stride in floats
  int vec;          // 1: every range is 8-column aligned (beg, len, ld): 16-byte global access
  int ngap, gap_pos[CE_MAX_GAPS], gap_len[CE_MAX_GAPS];   // LDS positions of padding columns
  int nitem;
  int item_key[CE_MAX_ITEMS], item_feat[CE_MAX_ITEMS], item_pos[CE_MAX_ITEMS], item_C[CE_MAX_ITEMS];
};

// Categorical heads, tile-wise: a workgroup owns CE_TOK token rows.
//  1. the rows of every categorical column range are bulk-loaded into LDS (16 lanes per row,
//     float4 when the head layout is 8-aligned -- ModelLayout pads every head to that);
//  2. one THREAD per (row, item) evaluates the weight (mask & condition & s < nvalid: a chain of
//     dependent loads, issued for all 256 items at once) and active items are compacted;
//  3. one 16-LANE GROUP per item: inactive items (~85 % under the 15 % masking rate) zero their
//     classes, active ones run softmax / clipped CE / gradient with 16-lane shuffles (the previous
//     kernel walked an item's classes serially in one thread: 102 us for 50 MB);
//  4. rows go back as d(logits) in the compute dtype, 16 bytes per lane.
// Loss / score / denominator: LDS accumulation + one global atomic per key and workgroup.
template <typename TDL, int CE_TOK>
__global__ __launch_bounds__(CE_TOK * 16) void ce_tile_kernel(const float* __restrict__ logits, TDL* __restrict__ dlogits,
                                                      int ld, LossKeys keys, CeRanges rg, const int* __restrict__ nvalid,
                                                      float* __restrict__ sums, int T, int S, float inv_B,
                                                      const int* __restrict__ pred_row, const int* __restrict__ true_row) {
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [CE_TOK][rg.width]
  __shared__ float red[MFP_MAX_LOSS_KEYS][3];
  __shared__ int nactive;
  __shared__ unsigned short active[CE_TOK * CE_MAX_ITEMS];   // compacted (row << 4 | item)
  __shared__ unsigned char isact[CE_TOK][CE_MAX_ITEMS];
  __shared__ int ylabel[CE_TOK][CE_MAX_ITEMS];
  const int t0 = blockIdx.x * CE_TOK;
  const int W = rg.width;
  const int row16 = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  if (threadIdx.x < MFP_MAX_LOSS_KEYS * 3) (&red[0][0])[threadIdx.x] = 0.f;
  if (threadIdx.x == 0) nactive = 0;
  // ---- weights and labels first: thread = (row, item); their (small, scattered) loads are in
  // flight together with the bulk staging loads below
  bool act = false;
  int ylab = 0;
  {
    const int item = l16, t = t0 + row16;
    if (item < rg.nitem && t < T) {
      const mfp_loss_key& key = keys.k[rg.item_key[item]];
      const int tt = true_row ? true_row[t] : t;
      act = token_weight(key, nvalid, t, tt, S) != 0.f;
      ylab = reinterpret_cast<const int*>(key.target)[(long long)tt * key.n_feat + rg.item_feat[item]];
    }
  }
  // ---- 1. stage (position t holds logits row pred_row[t] when the loss is position-sorted)
  const int prow = (pred_row && t0 + row16 < T) ? pred_row[t0 + row16] : t0 + row16;
  {
    const int t = t0 + row16;
    for (int r = 0; r < rg.n; ++r) {
      float* dst = tile + row16 * W + rg.lds_off[r];
      const float* src = logits + (long long)prow * ld + rg.beg[r];
      if (rg.vec) {
        for (int c4 = l16; c4 < rg.len[r] / 4; c4 += 16)
          *reinterpret_cast<float4*>(dst + 4 * c4) = t < T ? *reinterpret_cast<const float4*>(src + 4 * c4)
                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        for (int c = l16; c < rg.len[r]; c += 16) dst[c] = t < T ? src[c] : 0.f;
      }
    }
  }
  isact[row16][l16] = act ? 1 : 0;
  ylabel[row16][l16] = ylab;
  __syncthreads();
  for (int gI = 0; gI < rg.ngap; ++gI)      // padding columns: d(logits) = 0
    if (l16 < rg.gap_len[gI]) tile[row16 * W + rg.gap_pos[gI] + l16] = 0.f;
  __syncthreads();
  if (isact[row16][l16]) active[atomicAdd(&nactive, 1)] = (unsigned short)(threadIdx.x);
  // ---- 3a. inactive items zero their classes (group = 16 lanes, item a = group, group + 16, ...)
  for (int a = row16; a < CE_TOK * rg.nitem; a += CE_TOK) {
    const int row = a / rg.nitem, item = a % rg.nitem;
    if (isact[row][item]) continue;
    float* z = tile + row * W + rg.item_pos[item];
    for (int j = l16; j < rg.item_C[item]; j += 16) z[j] = 0.f;
  }
  __syncthreads();
  // ---- 3b. active items: one 16-lane group each
  const int na = nactive;
  for (int a = row16; a < na; a += CE_TOK) {
    const int code = active[a], row = code >> 4, item = code & 15;
    const int kidx = rg.item_key[item], C = rg.item_C[item];
    float* z = tile + row * W + rg.item_pos[item];
    const int y = ylabel[row][item];
    // max / argmax (first index on ties, as the serial reference walk)
    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int j = l16; j < C; j += 16) { const float v = z[j]; if (v > m) { m = v; am = j; } }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const float om = __shfl_xor(m, o, 64);
      const int oa = __shfl_xor(am, o, 64);
      if (om > m || (om == m && oa < am)) { m = om; am = oa; }
    }
    float se = 0.f;
    for (int j = l16; j < C; j += 16) { const float e = expf(z[j] - m); z[j] = e; se += e; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
    const float inv = 1.f / se;
    float sq = 0.f, qy = 0.f;
    for (int j = l16; j < C; j += 16) {
      const float pj = z[j] * inv;
      const float q = fminf(fmaxf(pj, 1e-7f), 1.f - 1e-7f);
      sq += q;
      if (j == y) qy = q;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { sq += __shfl_xor(sq, o, 64); qy += __shfl_xor(qy, o, 64); }
    const float loss = -logf(qy) + logf(sq);
    const float inv_sq = 1.f / sq, inv_qy = 1.f / qy;
    float gp = 0.f;
    for (int j = l16; j < C; j += 16) {
      const float pj = z[j] * inv;
      const float g = (pj >= 1e-7f && pj <= 1.f - 1e-7f) ? ((j == y ? -inv_qy : 0.f) + inv_sq) : 0.f;
      gp += g * pj;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) gp += __shfl_xor(gp, o, 64);
    for (int j = l16; j < C; j += 16) {
      const float pj = z[j] * inv;
      const float g = (pj >= 1e-7f && pj <= 1.f - 1e-7f) ? ((j == y ? -inv_qy : 0.f) + inv_sq) : 0.f;
      z[j] = pj * (g - gp) * inv_B;
    }
    if (l16 == 0) {
      atomicAdd(&red[kidx][0], loss * inv_B);
      atomicAdd(&red[kidx][1], am == y ? 1.f : 0.f);
      atomicAdd(&red[kidx][2], 1.f);
    }
  }
  __syncthreads();
  // ---- 4. store d(logits) rows, publish the sums
  if (dlogits) {
    const int t = t0 + row16;
    if (t < T) {
      for (int r = 0; r < rg.n; ++r) {
        const float* src = tile + row16 * W + rg.lds_off[r];
        TDL* dst = dlogits + (long long)prow * ld + rg.beg[r];
        if (rg.vec) {
          for (int c8 = l16; c8 < rg.len[r] / 8; c8 += 16) {
            const float4 a = *reinterpret_cast<const float4*>(src + 8 * c8);
            const float4 b = *reinterpret_cast<const float4*>(src + 8 * c8 + 4);
            if constexpr (sizeof(TDL) == 4) {
              *reinterpret_cast<float4*>(dst + 8 * c8) = a;
              *reinterpret_cast<float4*>(dst + 8 * c8 + 4) = b;
            } else {
              const u32x4 pk = {pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w)};
              *reinterpret_cast<u32x4*>(dst + 8 * c8) = pk;
            }
          }
        } else {
          for (int c = l16; c < rg.len[r]; c += 16) cdt_traits<TDL>::store(dst + c, src[c]);
        }
      }
    }
  }
  if (threadIdx.x < keys.n * 3) {
    const int k = threadIdx.x / 3, c = threadIdx.x % 3;
    const float v = red[k][c];
    if (v != 0.f) atomicAdd(&sums[keys.key_slot[k] * 3 + c], v);
  }
}

constexpr int MSE_TOK = 16;   // tokens per workgroup (4 per wave)

// Numerical heads.  VEC: the head is 8-column aligned and its width a multiple of 8 -> a lane owns
// 8 consecutive columns (two float4 of predictions and targets, one 16-byte d(pred) store); the
// token weights of the workgroup's MSE_TOK tokens are evaluated by its first lanes in one go.
template <typename TDL, bool VEC>
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, TDL* __restrict__ dpred,
                                                  int ld, LossKeys keys, const int* __restrict__ nvalid,
                                                  float* __restrict__ sums, int T, int S, float inv_B,
                                                  const int* __restrict__ pred_row, const int* __restrict__ true_row) {
  __shared__ float red[3][4];
  __shared__ float wt[MSE_TOK];
  __shared__ int prow_s[MSE_TOK], trow_s[MSE_TOK];
  const mfp_loss_key k = keys.k[blockIdx.y];
  const int W = k.n_class;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* target = reinterpret_cast<const float*>(k.target);
  const int t0 = blockIdx.x * MSE_TOK;
  if (threadIdx.x < MSE_TOK) {
    const int t = t0 + (int)threadIdx.x;
    const int tt = (true_row && t < T) ? true_row[t] : t;
    prow_s[threadIdx.x] = (pred_row && t < T) ? pred_row[t] : t;
    trow_s[threadIdx.x] = tt;
    wt[threadIdx.x] = t < T ? token_weight(k, nvalid, t, tt, S) : -1.f;
  }
  __syncthreads();
  float acc_loss = 0.f, acc_score = 0.f, acc_den = 0.f;
  for (int tt = wave; tt < MSE_TOK; tt += 4) {
    const float w = wt[tt];
    if (w < 0.f) break;
    const long long base = (long long)prow_s[tt] * ld + k.col_off;
    const long long tbase = (long long)trow_s[tt] * W;
    if (w == 0.f) {
      if (dpred) {
        if (VEC) {
          for (int c = lane * 8; c < W; c += 512) {
            if constexpr (sizeof(TDL) == 4) {
              *reinterpret_cast<float4*>(dpred + base + c) = make_float4(0.f, 0.f, 0.f, 0.f);
              *reinterpret_cast<float4*>(dpred + base + c + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
              *reinterpret_cast<u32x4*>(dpred + base + c) = (u32x4){0u, 0u, 0u, 0u};
            }
          }
        } else {
          for (int j = lane; j < W; j += 64) cdt_traits<TDL>::store(dpred + base + j, 0.f);
        }
      }
      continue;
    }
    float sd = 0.f, sy = 0.f, sp = 0.f, syp = 0.f;
    if (VEC) {
      for (int c = lane * 8; c < W; c += 512) {
        float p[8], y[8], d[8];
        *reinterpret_cast<float4*>(p) = *reinterpret_cast<const float4*>(pred + base + c);
        *reinterpret_cast<float4*>(p + 4) = *reinterpret_cast<const float4*>(pred + base + c + 4);
        *reinterpret_cast<float4*>(y) = *reinterpret_cast<const float4*>(target + tbase + c);
        *reinterpret_cast<float4*>(y + 4) = *reinterpret_cast<const float4*>(target + tbase + c + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          d[e] = p[e] - y[e];
          sd += d[e] * d[e]; sy += y[e] * y[e]; sp += p[e] * p[e]; syp += y[e] * p[e];
          d[e] *= 2.f * inv_B;
        }
        if (dpred) {
          if constexpr (sizeof(TDL) == 4) {
            *reinterpret_cast<float4*>(dpred + base + c) = *reinterpret_cast<const float4*>(d);
            *reinterpret_cast<float4*>(dpred + base + c + 4) = *reinterpret_cast<const float4*>(d + 4);
          } else {
            const u32x4 pk = {pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3]), pack_bf16x2(d[4], d[5]), pack_bf16x2(d[6], d[7])};
            *reinterpret_cast<u32x4*>(dpred + base + c) = pk;
          }
        }
      }
    } else {
      for (int j = lane; j < W; j += 64) {
        const float p = pred[base + j], y = target[tbase + j];
        const float d = p - y;
        sd += d * d; sy += y * y; sp += p * p; syp += y * p;
        if (dpred) cdt_traits<TDL>::store(dpred + base + j, 2.f * d * inv_B);
      }
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

// Position-sorted loss (reference models/tensor_utils.py:14-44): one workgroup per document.
//   priority(s) = sum_k v_k(s) 100^(4-k) + [s >= nvalid] 100^5,  v_k = label or first-index argmax
//   slot(s)     = #{j : p_j < p_s  or  (p_j == p_s and j < s)}          (stable ascending order)
//   row_map[b*S + slot(s)] = b*S + s     (identity for documents whose flag is 0)
// logits mode: one wave per position, lanes over the classes of each of the five heads.
struct SortSrc {
  const int* label[5];
  int label_stride[5];
  int col_off[5], n_class[5];
};

constexpr int SORT_MAX_S = 1024;

__global__ __launch_bounds__(256) void sort_positions_kernel(SortSrc src, const float* __restrict__ logits, int ld,
                                                             const int* __restrict__ nvalid,
                                                             const unsigned char* __restrict__ flag,
                                                             int* __restrict__ row_map, int S) {
  __shared__ long long prio[SORT_MAX_S];
  const int b = blockIdx.x;
  const int base = b * S;
  if (!flag[b]) {
    for (int s = threadIdx.x; s < S; s += blockDim.x) row_map[base + s] = base + s;
    return;
  }
  const int nv = nvalid[b];
  if (logits == nullptr) {
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
      long long p = 0;
#pragma unroll
      for (int k = 0; k < 5; ++k) p = p * 100 + src.label[k][(long long)(base + s) * src.label_stride[k]];
      prio[s] = p + (s >= nv ? 10000000000LL : 0LL);
    }
  } else {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int s = wave; s < S; s += nwave) {
      const float* row = logits + (long long)(base + s) * ld;
      long long p = 0;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        float m = -INFINITY;
        int am = 0x7fffffff;
        for (int j = lane; j < src.n_class[k]; j += 64) {
          const float v = row[src.col_off[k] + j];
          if (v > m) { m = v; am = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float om = __shfl_xor(m, o, 64);
          const int oa = __shfl_xor(am, o, 64);
          if (om > m || (om == m && oa < am)) { m = om; am = oa; }
        }
        if (am == 0x7fffffff) am = 0;   // a row of NaNs: argmax 0, as a serial first-maximum walk
        p = p * 100 + am;
      }
      if (lane == 0) prio[s] = p + (s >= nv ? 10000000000LL : 0LL);
    }
  }
  __syncthreads();
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    const long long p = prio[s];
    int slot = 0;
    for (int j = 0; j < S; ++j) {
      const long long q = prio[j];
      slot += (q < p || (q == p && j < s)) ? 1 : 0;
    }
    row_map[base + slot] = base + s;
  }
}

}  // namespace

extern "C" int mfp_sort_positions(const int32_t* const* labels, const int32_t* label_stride, const float* logits,
                                  int32_t ld, const int32_t* col_off, const int32_t* n_class, const int32_t* nvalid,
                                  const uint8_t* flag, int32_t* row_map, int32_t B, int32_t S, mfp_stream_t stream) {
  MFP_CHECK_ARG(nvalid && flag && row_map && B > 0 && S > 0 && S <= SORT_MAX_S);
  MFP_CHECK_ARG((labels && label_stride) || (logits && col_off && n_class && ld > 0));
  SortSrc src;
  for (int k = 0; k < 5; ++k) {
    src.label[k] = nullptr; src.label_stride[k] = 0; src.col_off[k] = 0; src.n_class[k] = 0;
    if (logits) {
      MFP_CHECK_ARG(n_class[k] > 0 && n_class[k] < 100 && col_off[k] >= 0 && col_off[k] + n_class[k] <= ld);
      src.col_off[k] = col_off[k]; src.n_class[k] = n_class[k];
    } else {
      MFP_CHECK_ARG(labels[k] && label_stride[k] > 0);
      src.label[k] = labels[k]; src.label_stride[k] = label_stride[k];
    }
  }
  hipLaunchKernelGGL(sort_positions_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, logits, ld,
                     nvalid, flag, row_map, S);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_loss_fwd_bwd(const float* logits, void* dlogits, int32_t ld, const mfp_loss_key* keys,
                                int32_t nkeys, const int32_t* nvalid, float* sums, int32_t B, int32_t S,
                                int32_t dl_dtype, mfp_stream_t stream) {
  return mfp_loss_fwd_bwd_sorted(logits, dlogits, ld, keys, nkeys, nvalid, sums, B, S, dl_dtype, nullptr, nullptr, stream);
}

static int loss_fwd_bwd_impl(const float* logits, void* dlogits, int32_t ld, const mfp_loss_key* keys,
                             int32_t nkeys, const int32_t* nvalid, float* sums, int32_t B, int32_t S,
                             int32_t dl_dtype, const int32_t* pred_row, const int32_t* true_row, bool prezeroed,
                             mfp_stream_t stream);

extern "C" int mfp_loss_fwd_bwd_sorted(const float* logits, void* dlogits, int32_t ld, const mfp_loss_key* keys,
                                       int32_t nkeys, const int32_t* nvalid, float* sums, int32_t B, int32_t S,
                                       int32_t dl_dtype, const int32_t* pred_row, const int32_t* true_row,
                                       mfp_stream_t stream) {
  return loss_fwd_bwd_impl(logits, dlogits, ld, keys, nkeys, nvalid, sums, B, S, dl_dtype, pred_row, true_row, false, stream);
}

extern "C" int mfp_loss_fwd_bwd_acc(const float* logits, void* dlogits, int32_t ld, const mfp_loss_key* keys,
                                    int32_t nkeys, const int32_t* nvalid, float* sums, int32_t B, int32_t S,
                                    int32_t dl_dtype, const int32_t* pred_row, const int32_t* true_row,
                                    mfp_stream_t stream) {
  return loss_fwd_bwd_impl(logits, dlogits, ld, keys, nkeys, nvalid, sums, B, S, dl_dtype, pred_row, true_row, true, stream);
}

static int loss_fwd_bwd_impl(const float* logits, void* dlogits, int32_t ld, const mfp_loss_key* keys,
                             int32_t nkeys, const int32_t* nvalid, float* sums, int32_t B, int32_t S,
                             int32_t dl_dtype, const int32_t* pred_row, const int32_t* true_row, bool prezeroed,
                             mfp_stream_t stream) {
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
  if (!prezeroed) {      // (mfp_loss_fwd_bwd_acc: the step prologue has zeroed sums)
    hipLaunchKernelGGL(zero_sums_kernel, dim3(1), dim3(64), 0, st, sums, nkeys * 3);
    MFP_CHECK_LAUNCH();
  }
  const float inv_B = 1.0f / (float)B;
  if (cat.n > 0) {
    // Merge the categorical heads' columns into contiguous ranges.  Gaps of < 8 columns between
    // heads (and behind the last head of a range, up to the next multiple of 8) are PADDING by
    // contract (include/mfp_hip.h): their logits are ignored and their d(logits) written as 0, so
    // they ride along and the ranges of an 8-aligned head layout stay 16-byte aligned
    // (Crello: [0,320) and [1344,1384)).
    struct Head { int beg, end, cat; };
    Head hs[MFP_MAX_LOSS_KEYS];
    for (int i = 0; i < nkeys; ++i)
      hs[i] = Head{keys[i].col_off, keys[i].col_off + keys[i].n_feat * keys[i].n_class, keys[i].is_numerical ? 0 : 1};
    for (int i = 1; i < nkeys; ++i)   // insertion sort by first column
      for (int j = i; j > 0 && hs[j].beg < hs[j - 1].beg; --j) { Head t = hs[j]; hs[j] = hs[j - 1]; hs[j - 1] = t; }
    CeRanges rg;
    rg.n = 0; rg.width = 0; rg.ngap = 0;
    auto add_gap = [&](int col, int len) {   // position fixed up below (needs lds_off)
      if (len > 0 && rg.ngap < CE_MAX_GAPS) { rg.gap_pos[rg.ngap] = col; rg.gap_len[rg.ngap] = len; rg.ngap++; }
    };
    for (int i = 0; i < nkeys; ++i) {
      MFP_CHECK_ARG(i == 0 || hs[i].beg >= hs[i - 1].end);   // heads must not overlap
      if (!hs[i].cat) continue;
      const bool extend = rg.n > 0 && i > 0 && hs[i - 1].cat && hs[i].beg - (rg.beg[rg.n - 1] + rg.len[rg.n - 1]) < 8
                          && hs[i].beg >= rg.beg[rg.n - 1] + rg.len[rg.n - 1];
      if (extend) {
        const int cur_end = rg.beg[rg.n - 1] + rg.len[rg.n - 1];
        add_gap(cur_end, hs[i].beg - cur_end);
        rg.len[rg.n - 1] = hs[i].end - rg.beg[rg.n - 1];
      } else {
        MFP_CHECK_ARG(rg.n < CE_MAX_RANGES);
        rg.beg[rg.n] = hs[i].beg; rg.len[rg.n] = hs[i].end - hs[i].beg; rg.n++;
      }
      // close the range at a multiple of 8 when the columns up to there are free
      const int next_beg = i + 1 < nkeys ? hs[i + 1].beg : ld;
      const int end = rg.beg[rg.n - 1] + rg.len[rg.n - 1], end8 = (end + 7) / 8 * 8;
      if ((i + 1 == nkeys || !hs[i + 1].cat || hs[i + 1].beg - end >= 8) && end8 <= next_beg && end8 <= ld) {
        add_gap(end, end8 - end);
        rg.len[rg.n - 1] = end8 - rg.beg[rg.n - 1];
      }
    }
    for (int r = 0; r < rg.n; ++r) { rg.lds_off[r] = rg.width; rg.width += rg.len[r]; }
    for (int gI = 0; gI < rg.ngap; ++gI) {     // global column -> LDS position
      for (int r = 0; r < rg.n; ++r)
        if (rg.gap_pos[gI] >= rg.beg[r] && rg.gap_pos[gI] < rg.beg[r] + rg.len[r]) {
          rg.gap_pos[gI] = rg.lds_off[r] + rg.gap_pos[gI] - rg.beg[r];
          break;
        }
    }
    rg.vec = (ld % 8 == 0) ? 1 : 0;
    for (int r = 0; r < rg.n; ++r)
      if (rg.beg[r] % 8 != 0 || rg.len[r] % 8 != 0) rg.vec = 0;
    rg.width = (rg.width + 3) / 4 * 4 + 4;   // 16-byte rows; +4 floats: 16-lane groups of different rows hit different banks
    // (the kernel also has ~4 KB of static LDS -- active / isact / ylabel / red -- inside the same 64 KB)
    const int ce_tok = (size_t)32 * rg.width * sizeof(float) + 4096 <= 64 * 1024 ? 32 : 16;
    const size_t lds = (size_t)ce_tok * rg.width * sizeof(float);
    MFP_CHECK_ARG(lds + 4096 <= 64 * 1024);
    rg.nitem = 0;
    for (int i = 0; i < cat.n; ++i) {
      for (int f = 0; f < cat.k[i].n_feat; ++f) {
        MFP_CHECK_ARG(rg.nitem < CE_MAX_ITEMS);
        const int col = cat.k[i].col_off + f * cat.k[i].n_class;
        int pos = -1;
        for (int r = 0; r < rg.n; ++r)
          if (col >= rg.beg[r] && col < rg.beg[r] + rg.len[r]) pos = rg.lds_off[r] + col - rg.beg[r];
        rg.item_key[rg.nitem] = i; rg.item_feat[rg.nitem] = f; rg.item_pos[rg.nitem] = pos;
        rg.item_C[rg.nitem] = cat.k[i].n_class;
        rg.nitem++;
      }
    }
    const int bx = (T + ce_tok - 1) / ce_tok;
#define CE_LAUNCH(TT, TOK) hipLaunchKernelGGL((ce_tile_kernel<TT, TOK>), dim3(bx), dim3(TOK * 16), lds, st, logits, (TT*)dlogits, ld, cat, rg, nvalid, sums, T, S, inv_B, pred_row, true_row)
    if (dl_dtype == MFP_F32) { if (ce_tok == 32) CE_LAUNCH(float, 32); else CE_LAUNCH(float, 16); }
    else { if (ce_tok == 32) CE_LAUNCH(unsigned short, 32); else CE_LAUNCH(unsigned short, 16); }
#undef CE_LAUNCH
    MFP_CHECK_LAUNCH();
  }
  if (num.n > 0) {
    const int bx = (T + MSE_TOK - 1) / MSE_TOK;
    bool vec = ld % 8 == 0;
    for (int i = 0; i < num.n; ++i) vec = vec && num.k[i].col_off % 8 == 0 && num.k[i].n_class % 8 == 0;
#define MSE_LAUNCH(TT, V) hipLaunchKernelGGL((mse_kernel<TT, V>), dim3(bx, num.n), dim3(256), 0, st, logits, (TT*)dlogits, ld, num, nvalid, sums, T, S, inv_B, pred_row, true_row)
    if (dl_dtype == MFP_F32) { if (vec) MSE_LAUNCH(float, true); else MSE_LAUNCH(float, false); }
    else { if (vec) MSE_LAUNCH(unsigned short, true); else MSE_LAUNCH(unsigned short, false); }
#undef MSE_LAUNCH
    MFP_CHECK_LAUNCH();
  }
  return MFP_OK;
}
