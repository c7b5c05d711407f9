// Fused multi-head self-attention, forward and backward, one workgroup per (document, head)
// (reference architecture/transformer.py:60-76: score = QK^T / sqrt(hd); score += -1e9 * (1 -
// keymask); softmax; .V).  The whole K/V of one (b,h) lives in LDS; no S x S matrix ever
// reaches HBM.  Two implementations behind one entry point:
//   * MFP_F32 : plain-VALU kernels (thread per query / per key), exact f32 -- the parity path.
//   * MFP_BF16: MFMA kernels (v_mfma_f32_16x16x32_bf16).  Scores are computed TRANSPOSED
//     (S^T = K Q^T) so that a lane owns one query column: the softmax is register-local plus two
//     cross-lane shuffles, and the probabilities feed the second MFMA (O^T = V^T P^T) straight
//     from registers as its B operand; V^T / K^T / Q^T / dO^T operands are produced from the
//     row-major LDS image with ds_read_b64_tr_b16.  The k-order of those MFMAs is the permuted
//     {4g+j, 16+4g+j} order the C-fragment dictates; both operands use the same order.
// Layout: qkv [B*S][3D] (q | k | v, head h at columns h*hd), out/dout [B*S][D], lse [B][H][S].
#include <stdlib.h>

#include "common.h"

namespace {

// ===================================================================================== f32
template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_f32(const float* __restrict__ qkv, const int* __restrict__ nvalid,
                                                    float* __restrict__ out, float* __restrict__ lse, int S,
                                                    int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int LD = HD + 1;
  float* Ks = sm;
  float* Vs = sm + S * LD;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);   // heads of one document -> one XCD (shared 128-B lines)
  const int b = bid / H, h = bid % H;
  const int D = H * HD, D3 = 3 * D;
  const int nv = nvalid[b];
  for (int i = threadIdx.x; i < S * HD; i += blockDim.x) {
    const int r = i / HD, d = i % HD;
    const float* row = qkv + (long long)(b * S + r) * D3 + h * HD + d;
    Ks[r * LD + d] = row[D];
    Vs[r * LD + d] = row[2 * D];
  }
  __syncthreads();
  for (int q = threadIdx.x; q < S; q += blockDim.x) {
    float qv[HD];
    const float* qrow = qkv + (long long)(b * S + q) * D3 + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) qv[d] = qrow[d];
    float m = -INFINITY;
    for (int j = 0; j < S; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) s += qv[d] * Ks[j * LD + d];
      s = s * scale + (j < nv ? 0.f : -1e9f);
      m = fmaxf(m, s);
    }
    float l = 0.f, o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    for (int j = 0; j < S; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) s += qv[d] * Ks[j * LD + d];
      s = s * scale + (j < nv ? 0.f : -1e9f);
      const float p = expf(s - m);
      l += p;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] += p * Vs[j * LD + d];
    }
    const float inv = 1.f / l;
    float* orow = out + (long long)(b * S + q) * D + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) orow[d] = o[d] * inv;
    lse[((long long)b * H + h) * S + q] = m + logf(l);
  }
}

// dQ: thread per query, K/V in LDS.
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_f32(const float* __restrict__ qkv, const int* __restrict__ nvalid,
                                                       const float* __restrict__ out, const float* __restrict__ dout,
                                                       const float* __restrict__ lse, float* __restrict__ dqkv,
                                                       int S, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int LD = HD + 1;
  float* Ks = sm;
  float* Vs = sm + S * LD;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);   // heads of one document -> one XCD (shared 128-B lines)
  const int b = bid / H, h = bid % H;
  const int D = H * HD, D3 = 3 * D;
  const int nv = nvalid[b];
  for (int i = threadIdx.x; i < S * HD; i += blockDim.x) {
    const int r = i / HD, d = i % HD;
    const float* row = qkv + (long long)(b * S + r) * D3 + h * HD + d;
    Ks[r * LD + d] = row[D];
    Vs[r * LD + d] = row[2 * D];
  }
  __syncthreads();
  for (int q = threadIdx.x; q < S; q += blockDim.x) {
    float qv[HD], dov[HD], dq[HD];
    const long long tok = (long long)(b * S + q);
    float delta = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      qv[d] = qkv[tok * D3 + h * HD + d];
      dov[d] = dout[tok * D + h * HD + d];
      delta += dov[d] * out[tok * D + h * HD + d];
      dq[d] = 0.f;
    }
    const float L = lse[((long long)b * H + h) * S + q];
    for (int j = 0; j < S; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) { s += qv[d] * Ks[j * LD + d]; dp += dov[d] * Vs[j * LD + d]; }
      s = s * scale + (j < nv ? 0.f : -1e9f);
      const float p = expf(s - L);
      const float ds = p * (dp - delta) * scale;
#pragma unroll
      for (int d = 0; d < HD; ++d) dq[d] += ds * Ks[j * LD + d];
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) dqkv[tok * D3 + h * HD + d] = dq[d];
  }
}

// dK, dV: thread per key, Q/dO (+ lse, delta) in LDS.
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_f32(const float* __restrict__ qkv, const int* __restrict__ nvalid,
                                                        const float* __restrict__ out, const float* __restrict__ dout,
                                                        const float* __restrict__ lse, float* __restrict__ dqkv,
                                                        int S, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int LD = HD + 1;
  float* Qs = sm;
  float* dOs = sm + S * LD;
  float* Ls = dOs + S * LD;
  float* Dl = Ls + S;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);   // heads of one document -> one XCD (shared 128-B lines)
  const int b = bid / H, h = bid % H;
  const int D = H * HD, D3 = 3 * D;
  const int nv = nvalid[b];
  for (int i = threadIdx.x; i < S * HD; i += blockDim.x) {
    const int r = i / HD, d = i % HD;
    Qs[r * LD + d] = qkv[(long long)(b * S + r) * D3 + h * HD + d];
    dOs[r * LD + d] = dout[(long long)(b * S + r) * D + h * HD + d];
  }
  for (int q = threadIdx.x; q < S; q += blockDim.x) {
    const long long tok = (long long)(b * S + q);
    float delta = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) delta += dout[tok * D + h * HD + d] * out[tok * D + h * HD + d];
    Dl[q] = delta;
    Ls[q] = lse[((long long)b * H + h) * S + q];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < S; j += blockDim.x) {
    float kv[HD], vv[HD], dk[HD], dv[HD];
    const long long tok = (long long)(b * S + j);
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      kv[d] = qkv[tok * D3 + D + h * HD + d];
      vv[d] = qkv[tok * D3 + 2 * D + h * HD + d];
      dk[d] = 0.f; dv[d] = 0.f;
    }
    const float madd = j < nv ? 0.f : -1e9f;
    for (int q = 0; q < S; ++q) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) { s += Qs[q * LD + d] * kv[d]; dp += dOs[q * LD + d] * vv[d]; }
      s = s * scale + madd;
      const float p = expf(s - Ls[q]);
      const float ds = p * (dp - Dl[q]) * scale;
#pragma unroll
      for (int d = 0; d < HD; ++d) { dv[d] += p * dOs[q * LD + d]; dk[d] += ds * Qs[q * LD + d]; }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      dqkv[tok * D3 + D + h * HD + d] = dk[d];
      dqkv[tok * D3 + 2 * D + h * HD + d] = dv[d];
    }
  }
}

// ===================================================================================== bf16
typedef unsigned short bf16_t;

// k-contiguous fragment: 8 consecutive k of row `row` starting at k0 + 8*lg
__device__ __forceinline__ bf16x8 frag_k(const bf16_t* tile, int ld, int row, int k0, int lg) {
  return *reinterpret_cast<const bf16x8*>(tile + row * ld + k0 + lg * 8);
}
// transposed fragment from a row-major [k][c] image: lane (li, lg) receives, for column
// c0 + li, the k rows {kb + 4lg + j} (j = 0..3) and {kb + 16 + 4lg + j}.
__device__ __forceinline__ bf16x8 frag_tr_perm(const bf16_t* tile, int ld, int kb, int c0, int li, int lg) {
  const bf16_t* ptr = tile + (kb + 4 * lg + (li >> 2)) * ld + c0 + (li & 3) * 4;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 16 * ld));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x8 pack_p(const f32x4& a, const f32x4& b) {
  const u32x4 r = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
  return __builtin_bit_cast(bf16x8, r);   // four v_cvt_pk_bf16_f32
}
__device__ __forceinline__ float xlane_max4(float v) {  // across the 4 lane groups (same li)
  v = fmaxf(v, lane_xor16(v));
  return fmaxf(v, lane_xor32(v));
}
__device__ __forceinline__ float xlane_sum4(float v) {
  v += lane_xor16(v);
  return v + lane_xor32(v);
}

// Stage the head slice of `nmat` matrices (each [S][HD] inside rows of `stride` elements) into
// LDS images [SP][LDH], zero-filling rows >= S and columns >= HD.
template <int HD, int HDP, int LDH>
__device__ __forceinline__ void stage_head(bf16_t* dst, const bf16_t* src, long long stride, int S, int SP) {
  constexpr int CPR = HDP / 8;  // 16-byte chunks per LDS row
  for (int i = threadIdx.x; i < SP * CPR; i += blockDim.x) {
    const int r = i / CPR, c = (i % CPR) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r < S && c < HD) v = *reinterpret_cast<const u32x4*>(src + (long long)r * stride + c);
    *reinterpret_cast<u32x4*>(dst + r * LDH + c) = v;
  }
}

// MAX_KT = key tiles held in registers: 8 (S <= 128) or 16 (S <= 256).
// QSPLIT (round 5, S > 128 / head width 64: BASELINE config 5): two workgroups per (document, head), each takes half of the
// query tiles (one per wave) and stages only K and V -- the query fragments come straight from global memory -- so an item's
// LDS drops from 111 KB to 75 KB and TWO workgroups share a CU: one's staging and stores run under the other's products
// (the one-workgroup form ran load -> compute -> store in sequence, 29.5 us per launch against a 13 us byte floor).
template <int HD, int MAX_KT, bool QSPLIT = false>
__global__ __launch_bounds__(512, 2) void attn_fwd_bf16(const bf16_t* __restrict__ qkv, const int* __restrict__ nvalid,
                                                     bf16_t* __restrict__ out, float* __restrict__ lse, int S,
                                                     int H, float scale) {
  constexpr int HDP = HD < 32 ? 32 : HD, LDH = HDP + 8, KS = HDP / 32, DT = HD / 16;
  extern __shared__ __attribute__((aligned(16))) bf16_t smb[];
  const int SP = (S + 31) & ~31;
  bf16_t* Qs = smb;
  bf16_t* Ks = QSPLIT ? smb : Qs + SP * LDH;
  bf16_t* Vs = Ks + SP * LDH;
  const int bid0 = xcd_remap(blockIdx.x, gridDim.x);   // heads of one document (and both halves of a head) -> one XCD
  const int bid = QSPLIT ? bid0 >> 1 : bid0, qhalf = QSPLIT ? bid0 & 1 : 0;
  const int b = bid / H, h = bid % H;
  const int D = H * HD, D3 = 3 * D;
  const int nv = nvalid[b];
  const bf16_t* base = qkv + (long long)b * S * D3 + h * HD;
  if (!QSPLIT) stage_head<HD, HDP, LDH>(Qs, base, D3, S, SP);
  stage_head<HD, HDP, LDH>(Ks, base + D, D3, S, SP);
  stage_head<HD, HDP, LDH>(Vs, base + 2 * D, D3, S, SP);
  // The score math is VALU-issue-bound (64 % of SIMD cycles issuing, 14 VALU per score): work in
  // the exp2 domain with the additive key term (0 / -1e9 log2 e for masked / -inf past S) read
  // from an LDS table: fma, max | sub, exp2, add per score.
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  float* Mb = reinterpret_cast<float*>(Vs + SP * LDH);
  for (int j = threadIdx.x; j < SP; j += blockDim.x) Mb[j] = j < S ? (j < nv ? 0.f : -1e9f * LOG2E) : -INFINITY;
  const float c2 = scale * LOG2E;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  const int nkt = SP / 16;
  const int nqh = QSPLIT ? (nkt + 1) >> 1 : nkt;                    // query tiles of this workgroup: [qhalf * nqh, + nqh)
  for (int qt = qhalf * nqh + wave; qt < min(nkt, (qhalf + 1) * nqh); qt += (int)(blockDim.x >> 6)) {
    bf16x8 bq[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (QSPLIT) {      // query row qt * 16 + li, k = 32 ks + 8 lg .. + 7, straight from global memory (zero past S)
        const int q = qt * 16 + li;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (q < S) v = *reinterpret_cast<const u32x4*>(base + (long long)q * D3 + ks * 32 + lg * 8);
        bq[ks] = __builtin_bit_cast(bf16x8, v);
      } else {
        bq[ks] = frag_k(Qs, LDH, qt * 16 + li, ks * 32, lg);
      }
    }
    f32x4 s[MAX_KT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < MAX_KT; ++kt) {
      if (kt < nkt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(Ks, LDH, kt * 16 + li, ks * 32, lg), bq[ks], acc, 0, 0, 0);
        const f32x4 mb4 = *reinterpret_cast<const f32x4*>(Mb + kt * 16 + 4 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[r] = __builtin_fmaf(acc[r], c2, mb4[r]);
          m = fmaxf(m, acc[r]);
        }
        s[kt] = acc;
      }
    }
    m = xlane_max4(m);
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < MAX_KT; ++kt) {
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[kt][r] - m);
          s[kt][r] = p;
          l += p;
        }
      }
    }
    l = xlane_sum4(l);
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < MAX_KT / 2; ++u) {
      if (2 * u < nkt) {
        const bf16x8 bp = pack_p(s[2 * u], s[2 * u + 1]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr_perm(Vs, LDH, u * 32, dt * 16, li, lg), bp, o[dt], 0, 0, 0);
      }
    }
    const int q = qt * 16 + li;
    if (q < S) {
      const float inv = 1.f / l;
      bf16_t* orow = out + (long long)(b * S + q) * D + h * HD;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        u32x2 pk = {pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv)};
        *reinterpret_cast<u32x2*>(orow + dt * 16 + 4 * lg) = pk;
      }
      if (lg == 0) lse[((long long)b * H + h) * S + q] = (m + __builtin_amdgcn_logf(l)) * LN2;   // natural-log lse
    }
  }
}

// Launched with 4 waves (S <= 128: two workgroups per CU) or 8 waves (S > 128: the four LDS images of one item take
// 147 KB at S = 256, head width 64 -- one workgroup per CU, so the second wave per SIMD has to come from this workgroup:
// with four waves a CU ran one wave per SIMD and an item took two rounds of 2 x 8 block pairs per wave).
template <int HD>
__global__ __launch_bounds__(512) void attn_bwd_bf16(const bf16_t* __restrict__ qkv, const int* __restrict__ nvalid,
                                                     const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout,
                                                     const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                     int S, int H, float scale) {
  constexpr int HDP = HD < 32 ? 32 : HD, LDH = HDP + 8, KS = HDP / 32, DT = HD / 16, CPR = HD / 8;
  extern __shared__ __attribute__((aligned(16))) bf16_t smb[];
  const int SP = (S + 31) & ~31;
  bf16_t* Qs = smb;
  bf16_t* Ks = Qs + SP * LDH;
  bf16_t* Vs = Ks + SP * LDH;
  bf16_t* dOs = Vs + SP * LDH;
  // The element-wise part is VALU-bound (measured: 49 of 64 us with the loads removed), so it
  // works in the exp2 domain with everything foldable folded: Ls holds lse * log2(e), Mb the
  // additive key term (0 / -1e9 log2(e) for masked / -inf past S), and the 1/sqrt(hd) factor of dS
  // is applied once to the dQ / dK accumulators: p = exp2(fma(s, c2, Mb[j]) - Ls[q]),
  // ds' = p * (dp - Dl[q])  -> 5 VALU per score instead of ~10.
  constexpr float LOG2E = 1.4426950408889634f;
  float* Ls = reinterpret_cast<float*>(dOs + SP * LDH);
  float* Dl = Ls + SP;
  float* Mb = Dl + SP;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);   // heads of one document -> one XCD (shared 128-B lines)
  const int b = bid / H, h = bid % H;
  const int D = H * HD, D3 = 3 * D;
  const int nv = nvalid[b];
  const float c2 = scale * LOG2E;
  const bf16_t* base = qkv + (long long)b * S * D3 + h * HD;
  for (int j = threadIdx.x; j < SP; j += blockDim.x) Mb[j] = j < S ? (j < nv ? 0.f : -1e9f * LOG2E) : -INFINITY;
  stage_head<HD, HDP, LDH>(Qs, base, D3, S, SP);
  stage_head<HD, HDP, LDH>(Ks, base + D, D3, S, SP);
  stage_head<HD, HDP, LDH>(Vs, base + 2 * D, D3, S, SP);
  stage_head<HD, HDP, LDH>(dOs, dout + (long long)b * S * D + h * HD, D, S, SP);
  // delta[q] = sum_d dO[q][d] * O[q][d]; CPR adjacent lanes share a row
  for (int i = threadIdx.x; i < SP * CPR; i += blockDim.x) {
    const int r = i / CPR, c = (i % CPR) * 8;
    float part = 0.f;
    if (r < S) {
      const long long o = (long long)(b * S + r) * D + h * HD + c;
      u32x4 a = *reinterpret_cast<const u32x4*>(dout + o);
      u32x4 bb = *reinterpret_cast<const u32x4*>(out + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        part += bf16_to_f32((bf16_t)(a[e] & 0xffff)) * bf16_to_f32((bf16_t)(bb[e] & 0xffff));
        part += bf16_to_f32((bf16_t)(a[e] >> 16)) * bf16_to_f32((bf16_t)(bb[e] >> 16));
      }
    }
#pragma unroll
    for (int o = CPR / 2; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((i % CPR) == 0) {
      Dl[r] = part;
      Ls[r] = r < S ? lse[((long long)b * H + h) * S + r] * LOG2E : 0.f;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  const int nblk = SP / 32, nw = blockDim.x >> 6;

  // ---------------- loop A: dQ for query block qb (S^T orientation: lane = query)
  for (int qb = wave; qb < nblk; qb += nw) {
    bf16x8 bq[2][KS], bdo[2][KS];
    float Lq[2], Dq[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int q = qb * 32 + t * 16 + li;
      Lq[t] = Ls[q]; Dq[t] = Dl[q];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bq[t][ks] = frag_k(Qs, LDH, q, ks * 32, lg);
        bdo[t][ks] = frag_k(dOs, LDH, q, ks * 32, lg);
      }
    }
    f32x4 dq[2][DT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dq[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < nblk; ++kb) {
      f32x4 ds[2][2];  // [key tile][query tile]
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        bf16x8 ak[KS], av[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          ak[ks] = frag_k(Ks, LDH, kb * 32 + kt * 16 + li, ks * 32, lg);
          av[ks] = frag_k(Vs, LDH, kb * 32 + kt * 16 + li, ks * 32, lg);
        }
        const f32x4 mb4 = *reinterpret_cast<const f32x4*>(Mb + kb * 32 + kt * 16 + 4 * lg);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak[ks], bq[t][ks], sacc, 0, 0, 0);
            dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ks], bdo[t][ks], dpacc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], c2, mb4[r]) - Lq[t]);
            ds[kt][t][r] = p * (dpacc[r] - Dq[t]);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 bds = pack_p(ds[0][t], ds[1][t]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          dq[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr_perm(Ks, LDH, kb * 32, dt * 16, li, lg), bds, dq[t][dt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int q = qb * 32 + t * 16 + li;
      if (q < S) {
        bf16_t* row = dqkv + (long long)(b * S + q) * D3 + h * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          dq[t][dt] *= scale;
          u32x2 pk = {pack_bf16x2(dq[t][dt][0], dq[t][dt][1]), pack_bf16x2(dq[t][dt][2], dq[t][dt][3])};
          *reinterpret_cast<u32x2*>(row + dt * 16 + 4 * lg) = pk;
        }
      }
    }
  }

  // ---------------- loop B: dK, dV for key block kb (S orientation: lane = key)
  for (int kb = wave; kb < nblk; kb += nw) {
    bf16x8 bk[2][KS], bv[2][KS];
    float madd[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = kb * 32 + t * 16 + li;
      madd[t] = Mb[j];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bk[t][ks] = frag_k(Ks, LDH, j, ks * 32, lg);
        bv[t][ks] = frag_k(Vs, LDH, j, ks * 32, lg);
      }
    }
    f32x4 dk[2][DT], dv[2][DT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) { dk[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    for (int qb = 0; qb < nblk; ++qb) {
      f32x4 pp[2][2], ds[2][2];  // [query tile][key tile]
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        bf16x8 aq[KS], ado[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          aq[ks] = frag_k(Qs, LDH, qb * 32 + qt * 16 + li, ks * 32, lg);
          ado[ks] = frag_k(dOs, LDH, qb * 32 + qt * 16 + li, ks * 32, lg);
        }
        const f32x4 Lr = *reinterpret_cast<const f32x4*>(Ls + qb * 32 + qt * 16 + 4 * lg);
        const f32x4 Dr = *reinterpret_cast<const f32x4*>(Dl + qb * 32 + qt * 16 + 4 * lg);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[ks], bk[t][ks], sacc, 0, 0, 0);
            dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ado[ks], bv[t][ks], dpacc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], c2, madd[t]) - Lr[r]);
            pp[qt][t][r] = p;
            ds[qt][t][r] = p * (dpacc[r] - Dr[r]);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 bp = pack_p(pp[0][t], pp[1][t]);
        const bf16x8 bds = pack_p(ds[0][t], ds[1][t]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          dv[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr_perm(dOs, LDH, qb * 32, dt * 16, li, lg), bp, dv[t][dt], 0, 0, 0);
          dk[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr_perm(Qs, LDH, qb * 32, dt * 16, li, lg), bds, dk[t][dt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = kb * 32 + t * 16 + li;
      if (j < S) {
        bf16_t* row = dqkv + (long long)(b * S + j) * D3 + h * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          dk[t][dt] *= scale;
          u32x2 pk = {pack_bf16x2(dk[t][dt][0], dk[t][dt][1]), pack_bf16x2(dk[t][dt][2], dk[t][dt][3])};
          u32x2 pv = {pack_bf16x2(dv[t][dt][0], dv[t][dt][1]), pack_bf16x2(dv[t][dt][2], dv[t][dt][3])};
          *reinterpret_cast<u32x2*>(row + D + dt * 16 + 4 * lg) = pk;
          *reinterpret_cast<u32x2*>(row + 2 * D + dt * 16 + 4 * lg) = pv;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Attention backward, single pass (round 3), head width 32, S <= 128.
//
// attn_bwd_bf16 above evaluates every score twice (once per orientation) and runs load -> compute -> store in sequence
// inside each workgroup: measured at the timed shape 16 us of staging + 25 us of compute + 12 us of stores with little
// overlap (49 us stand-alone, 57 us in the step, for 129 MB).  This kernel:
//   * is PERSISTENT: 2 workgroups per CU, each walks a contiguous run of (document, head) items; the Q / dO slices of
//     item i + 1 stream into the second LDS buffer and its K / V slices into theirs (free as soon as every wave holds its
//     K / V fragments in registers) by LDS-DMA (buffer_load ... lds, 16 B per lane, no registers) while item i is
//     computed, and the stores of item i drain under item i + 1;
//   * evaluates every score ONCE, in the orientation whose accumulator layout has lane = key (S = Q K^T as
//     D[m = query][n = key]): wave w owns keys 32 w .. + 31 and walks the four 32-query blocks; P and dS feed the
//     dV / dK products straight from registers (B operands);
//   * dQ: every wave drops its dS tile (bf16) into a shared [128 keys][32 queries] LDS image (two in rotation: one
//     barrier per query block), and wave (dt, qt) = (w & 1, w >> 1) computes the 16 x 16 tile dQ^T[d][q] over ALL keys
//     from transposing reads of that image -- no partial sums, no reduction, 16 accumulator registers;
//   * dK / dV / dQ leave through LDS as 64-byte row pieces (16 B per lane);
//   * LDS images are unpadded (64-byte rows: what LDS-DMA writes) with XOR-swizzled 16-byte slots (8-byte pieces in
//     the dS image) chosen so that the b128 fragment reads, the transposing reads and the 8-byte writes are all
//     bank-conflict-free (MI355X_MICROARCH.md, LDS lane groups).
// Rows >= S of an image are zero: their loads are sent out of the buffer's range (the DMA writes zeros).
namespace sp {

constexpr int ROWB = 64;                       // bytes per image row (32 bf16)
constexpr int IMG = 128 * ROWB;                // one matrix: 8 KB
constexpr int QD_OFF = 0;                      // [2] x (Q | dO)
constexpr int K_OFF = 4 * IMG, V_OFF = 5 * IMG;
constexpr int DS_OFF = 6 * IMG;                // [2] x dS image [128 keys][32 q]
constexpr int LSD_OFF = 8 * IMG;               // Ls[128] f32, Dl[128] f32
constexpr int LDS_BYTES = LSD_OFF + 1024;      // 66 560 B: two workgroups per CU

__device__ __forceinline__ int swz(int row) { return ((row >> 2) & 1) << 1; }      // 16-byte slot XOR of the Q/K/V/dO images
// 8-byte piece XOR of the dS image: bijective in (r3, r2, r1) for the writes, bit 2 = r2 for the transposing reads
__device__ __forceinline__ int hsw(int row) { return (((row >> 2) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 1) & 1); }

__device__ __forceinline__ bf16x8 fragk(const unsigned char* img, int row, int lg) {
  return *reinterpret_cast<const bf16x8*>(img + row * ROWB + ((lg ^ swz(row)) << 4));
}
// for column c0 + li: the rows {kb + 4 lg + j} and {kb + 16 + 4 lg + j}, j = 0..3 (kb a multiple of 32)
__device__ __forceinline__ bf16x8 fragtr(const unsigned char* img, int kb, int c0, int li, int lg) {
  const int row = kb + 4 * lg + (li >> 2);
  const int P = (c0 >> 2) + (li & 3);
  const unsigned char* ptr = img + row * ROWB + ((((P >> 1) ^ swz(row)) << 4) | ((P & 1) << 3));
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 16 * ROWB));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// the same out of a 32-row block of the dS image (8-byte piece swizzle)
__device__ __forceinline__ bf16x8 scrtr(const unsigned char* blk, int c0, int li, int lg) {
  const int row = 4 * lg + (li >> 2);
  const int P = (c0 >> 2) + (li & 3);
  const unsigned char* ptr = blk + row * ROWB + ((P ^ hsw(row)) << 3);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 16 * ROWB));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

typedef __attribute__((address_space(3))) unsigned char lds_u8;

__global__ __launch_bounds__(256, 2) void attn_bwd1_hd32(const bf16_t* __restrict__ qkv, const int* __restrict__ nvalid,
                                                         const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout,
                                                         const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                         int B, int S, int H, float scale, int items, int ipw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr float LOG2E = 1.4426950408889634f;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * 32, D3 = 3 * D;
  const float c2 = scale * LOG2E;
  const unsigned int qkv_bytes = (unsigned int)B * (unsigned int)S * (unsigned int)D3 * 2u;
  const unsigned int o_bytes = (unsigned int)B * (unsigned int)S * (unsigned int)D * 2u;
  const __amdgpu_buffer_rsrc_t rs_qkv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(qkv), 0, qkv_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_do = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dout), 0, o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(out), 0, o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dq = __builtin_amdgcn_make_buffer_rsrc(dqkv, 0, qkv_bytes, 0x00020000);
  constexpr unsigned int OOB = 0xFFFFFFF0u;
  const int item0 = xcd_remap(blockIdx.x, gridDim.x) * ipw;
  const int nit = min(ipw, items - item0);
  if (nit <= 0) return;

  // wave w stages matrix w of an item (0 Q, 1 K, 2 V from qkv; 3 dO): 8 instructions of 16 rows x 64 B
  auto issue_dma = [&](int item, int c) {
    const int b = item / H, h = item - b * H;
    unsigned char* dst = smem + (wave == 0 ? QD_OFF + c * 2 * IMG : wave == 1 ? K_OFF : wave == 2 ? V_OFF : QD_OFF + c * 2 * IMG + IMG);
    const unsigned int stride = wave < 3 ? D3 * 2 : D * 2;
    const unsigned int colb = wave < 3 ? (wave * D + h * 32) * 2 : h * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 16 * i + (lane >> 2);
      const unsigned int voff = row < S ? (unsigned int)(b * S + row) * stride + colb + (((lane & 3) ^ swz(row)) << 4) : OOB;
      if (wave < 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, (lds_u8*)(dst + i * 1024), 16, voff, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_do, (lds_u8*)(dst + i * 1024), 16, voff, 0, 0, 0);
    }
  };
  // the O row piece (32 B) of thread (row = tid >> 1, half = tid & 1) and the lse of row tid (< 128) for the delta / Ls prologue
  u32x4 o0, o1;
  float lse_r;
  auto issue_olse = [&](int item) {
    const int b = item / H, h = item - b * H;
    const int row = tid >> 1;
    const unsigned int voff = row < S ? (unsigned int)(b * S + row) * (D * 2) + h * 64 + (tid & 1) * 32 : OOB;
    o0 = __builtin_amdgcn_raw_buffer_load_b128(rs_o, voff, 0, 0);
    o1 = __builtin_amdgcn_raw_buffer_load_b128(rs_o, voff + 16, 0, 0);
    lse_r = (tid < S && tid < 128) ? lse[((long long)b * H + h) * S + tid] : 0.f;
  };
  issue_dma(item0, 0);
  issue_olse(item0);

  const int k0 = 32 * wave;
  const int dt_w = wave & 1, qt_w = wave >> 1;       // this wave's tile of every query block's dQ^T
  float* const Ls = reinterpret_cast<float*>(smem + LSD_OFF);
  float* const Dl = Ls + 128;
  unsigned char* const Ks = smem + K_OFF;
  unsigned char* const Vs = smem + V_OFF;

  for (int it = 0; it < nit; ++it) {
    const int c = it & 1;
    const int item = item0 + it;
    const int b = item / H, h = item - b * H;
    unsigned char* const Qs = smem + QD_OFF + c * 2 * IMG;
    unsigned char* const dOs = Qs + IMG;
    // ---- (A) this item's images have landed (memory operations retire in order: everything but the six result stores
    // of the previous item, which are younger than the loads); delta = rowsum(dO * O), Ls = lse * log2(e)
    if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __syncthreads();
    {
      const int row = tid >> 1, hf = tid & 1;
      const u32x4 d0 = *reinterpret_cast<const u32x4*>(dOs + row * ROWB + (((2 * hf) ^ swz(row)) << 4));
      const u32x4 d1 = *reinterpret_cast<const u32x4*>(dOs + row * ROWB + (((2 * hf + 1) ^ swz(row)) << 4));
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        part += bf16_to_f32((bf16_t)(d0[e] & 0xffff)) * bf16_to_f32((bf16_t)(o0[e] & 0xffff));
        part += bf16_to_f32((bf16_t)(d0[e] >> 16)) * bf16_to_f32((bf16_t)(o0[e] >> 16));
        part += bf16_to_f32((bf16_t)(d1[e] & 0xffff)) * bf16_to_f32((bf16_t)(o1[e] & 0xffff));
        part += bf16_to_f32((bf16_t)(d1[e] >> 16)) * bf16_to_f32((bf16_t)(o1[e] >> 16));
      }
      part += __shfl_xor(part, 1, 64);
      if (hf == 0) Dl[row] = part;
      if (tid < 128) Ls[tid] = lse_r * LOG2E;
    }
    // this wave's K / V fragments: keys k0 .. + 31 as B operands, and K^T (d tile dt_w) of EVERY key block for dQ
    const int nv = nvalid[b];
    bf16x8 bk[2], bv[2], kT[4];
    float madd[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = k0 + 16 * t + li;
      bk[t] = fragk(Ks, j, lg);
      bv[t] = fragk(Vs, j, lg);
      madd[t] = j < S ? (j < nv ? 0.f : -1e9f * LOG2E) : -INFINITY;
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) kT[kb] = fragtr(Ks, 32 * kb, 16 * dt_w, li, lg);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // K / V reads done before anyone refills those images
    __syncthreads();
    // ---- (B) next item on its way: Q / dO into the other buffer, K / V into theirs (every wave holds its fragments)
    if (it + 1 < nit) {
      issue_dma(item + 1, c ^ 1);
      issue_olse(item + 1);
    }
    // ---- (C)
    f32x4 dk[2][2], dv[2][2], dq[4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) { dk[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      unsigned char* const dsi = smem + DS_OFF + (qb & 1) * IMG;
      f32x4 pp[2][2], ds[2][2];     // [query tile][key tile]
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int q = qb * 32 + qt * 16;
        const bf16x8 aq = fragk(Qs, q + li, lg), ado = fragk(dOs, q + li, lg);
        const f32x4 Lr = *reinterpret_cast<const f32x4*>(Ls + q + 4 * lg);
        const f32x4 Dr = *reinterpret_cast<const f32x4*>(Dl + q + 4 * lg);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          const f32x4 sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, bk[t], z, 0, 0, 0);
          const f32x4 dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ado, bv[t], z, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], c2, madd[t]) - Lr[r]);
            pp[qt][t][r] = p;
            ds[qt][t][r] = p * (dpacc[r] - Dr[r]);
          }
          // dS tile -> image [key k0 + 16 t + li][query 16 qt + 4 lg .. + 3]
          const int row = k0 + 16 * t + li;
          const u32x2 pk = {pack_bf16x2(ds[qt][t][0], ds[qt][t][1]), pack_bf16x2(ds[qt][t][2], ds[qt][t][3])};
          *reinterpret_cast<u32x2*>(dsi + row * ROWB + (((4 * qt + lg) ^ hsw(row)) << 3)) = pk;
        }
      }
      const bf16x8 doT0 = fragtr(dOs, qb * 32, 0, li, lg), doT1 = fragtr(dOs, qb * 32, 16, li, lg);
      const bf16x8 qT0 = fragtr(Qs, qb * 32, 0, li, lg), qT1 = fragtr(Qs, qb * 32, 16, li, lg);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 bp = pack_p(pp[0][t], pp[1][t]);
        const bf16x8 bds = pack_p(ds[0][t], ds[1][t]);
        dv[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(doT0, bp, dv[t][0], 0, 0, 0);
        dv[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(doT1, bp, dv[t][1], 0, 0, 0);
        dk[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT0, bds, dk[t][0], 0, 0, 0);
        dk[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT1, bds, dk[t][1], 0, 0, 0);
      }
      // every wave's dS tile of this query block is in the image (the image of block qb - 2 was last read before the
      // barrier of block qb - 1): dQ^T tile (dt_w, qt_w) over all 128 keys
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT[kb], scrtr(dsi + kb * 2048, 16 * qt_w, li, lg), acc, 0, 0, 0);
      dq[qb] = acc;
    }
    // ---- dK, dV: through this wave's 32 rows of image 0 (last read before the barrier of query block 3) into 64-byte
    // row pieces (16 B per lane)
    const unsigned int rowbase = (unsigned int)(b * S) * (D3 * 2) + h * 64;
    unsigned char* const scr = smem + DS_OFF + wave * 2048;
#pragma unroll
    for (int m = 0; m < 2; ++m) {        // 0: dK (scaled), 1: dV
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const f32x4 v = m == 0 ? dk[t][dt] * scale : dv[t][dt];
          const int row = 16 * t + li;
          const u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *reinterpret_cast<u32x2*>(scr + row * ROWB + (((4 * dt + lg) ^ hsw(row)) << 3)) = pk;
        }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 16 * i + (lane >> 2), s4 = lane & 3;
        const u32x2 a = *reinterpret_cast<const u32x2*>(scr + row * ROWB + (((2 * s4) ^ hsw(row)) << 3));
        const u32x2 bb = *reinterpret_cast<const u32x2*>(scr + row * ROWB + (((2 * s4 + 1) ^ hsw(row)) << 3));
        const int j = k0 + row;
        const unsigned int voff = j < S ? rowbase + (unsigned int)j * (D3 * 2) + (m + 1) * D * 2 + s4 * 16 : OOB;
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){a[0], a[1], bb[0], bb[1]}, rs_dq, voff, 0, 0);
      }
    }
    // ---- dQ: tiles -> image 1 as [query][d] rows (every wave is past its reads of image 1), then 64-byte row pieces
    __syncthreads();
    {
      unsigned char* const dqi = smem + DS_OFF + IMG;
#pragma unroll
      for (int qb = 0; qb < 4; ++qb) {
        const int row = 32 * qb + 16 * qt_w + li;
        const f32x4 v = dq[qb] * scale;
        const u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(dqi + row * ROWB + (((4 * dt_w + lg) ^ hsw(row)) << 3)) = pk;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = k0 + 16 * i + (lane >> 2), s4 = lane & 3;
        const u32x2 a = *reinterpret_cast<const u32x2*>(dqi + row * ROWB + (((2 * s4) ^ hsw(row)) << 3));
        const u32x2 bb = *reinterpret_cast<const u32x2*>(dqi + row * ROWB + (((2 * s4 + 1) ^ hsw(row)) << 3));
        const unsigned int voff = row < S ? rowbase + (unsigned int)row * (D3 * 2) + s4 * 16 : OOB;
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){a[0], a[1], bb[0], bb[1]}, rs_dq, voff, 0, 0);
      }
    }
  }
}

}  // namespace sp

// ---------------------------------------------------------------------------------------------------------
// Attention backward, single pass, head width 64, S <= 256 (round 5: BASELINE config 5 = d_model 512, 8 heads, seq_len 256).
// The two-pass kernel above evaluates every score in both orientations and needs all four matrices of an item in LDS
// (147 KB at S = 256: one workgroup per CU, load -> loop A -> loop B -> store in sequence; 72 us per block at c5).  Here,
// attn_bwd1_hd32's scheme at twice the width: eight waves, wave w owns keys 32 w .. + 31 (K / V fragments and the dK / dV
// accumulators in registers, V straight from global memory: it is never needed transposed), walks the eight 32-query blocks
// evaluating every score ONCE (lane = key), drops its dS tile into a shared [256 keys][32 queries] image (two in rotation:
// one barrier per query block) and computes ONE 16 x 16 tile of the block's dQ^T (d tile w & 3, query tile w >> 2) over all
// 256 keys from transposing reads; dQ goes over the Q rows of the block (dead by then), dK over the K image, dV over the dO
// image, and the three images leave in 128-byte row pieces.  Images have 128-byte rows, 16-byte slot ^ ((row >> 1) & 7)
// (csrc/block_attn.hip's: fragment reads, transposing reads and the 8-byte tile writes are bank-conflict-free); the dS image
// is attn_bwd1_hd32's (64-byte rows, 8-byte piece swizzle).  LDS: Q 32 + dO 32 + K 32 + dS 2 x 16 + Ls / Dl 2 = 130 KB.
namespace sp64 {

constexpr int IMG = 256 * 128;                 // one matrix: [256][128 B]
constexpr int Q_OFF = 0, DO_OFF = IMG, K_OFF = 2 * IMG;
constexpr int DS_OFF = 3 * IMG;                // [2] x dS image [256 keys][64 B]
constexpr int DS_IMG = 256 * 64;
constexpr int LSD_OFF = DS_OFF + 2 * DS_IMG;   // Ls[256] f32, Dl[256] f32
constexpr int LDS_BYTES = LSD_OFF + 2048;      // 133 120 B

__device__ __forceinline__ int isw(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ bf16x8 fragk(const unsigned char* img, int row, int slot) {      // 16-byte slot `slot` of row `row`
  return *reinterpret_cast<const bf16x8*>(img + row * 128 + ((slot ^ isw(row)) << 4));
}
// for column c0 + li (c0 a multiple of 16): the rows {kb + 4 lg + j} and {kb + 16 + 4 lg + j}, j = 0..3 (kb a multiple of 32)
__device__ __forceinline__ bf16x8 fragtr(const unsigned char* img, int kb, int c0, int li, int lg) {
  const int row = kb + 4 * lg + (li >> 2);
  const int P = (c0 >> 2) + (li & 3);
  const unsigned char* ptr = img + row * 128 + ((((P >> 1) ^ isw(row)) << 4) | ((P & 1) << 3));
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 16 * 128));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

typedef __attribute__((address_space(3))) unsigned char lds_u8;

__global__ __launch_bounds__(512) void attn_bwd1_hd64(const bf16_t* __restrict__ qkv, const int* __restrict__ nvalid,
                                                      const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout,
                                                      const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                      int B, int S, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr float LOG2E = 1.4426950408889634f;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * 64, D3 = 3 * D;
  const float c2 = scale * LOG2E;
  const unsigned int qkv_bytes = (unsigned int)B * (unsigned int)S * (unsigned int)D3 * 2u;
  const unsigned int o_bytes = (unsigned int)B * (unsigned int)S * (unsigned int)D * 2u;
  const __amdgpu_buffer_rsrc_t rs_qkv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(qkv), 0, qkv_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_do = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dout), 0, o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(out), 0, o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dq = __builtin_amdgcn_make_buffer_rsrc(dqkv, 0, qkv_bytes, 0x00020000);
  constexpr unsigned int OOB = 0xFFFFFFF0u;
  const int item = xcd_remap(blockIdx.x, gridDim.x);      // heads of one document -> one XCD (shared 128-byte lines)
  const int b = item / H, h = item - b * H;
  unsigned char* const Qs = smem + Q_OFF;
  unsigned char* const dOs = smem + DO_OFF;
  unsigned char* const Ks = smem + K_OFF;
  float* const Ls = reinterpret_cast<float*>(smem + LSD_OFF);
  float* const Dl = Ls + 256;

  // ---- Q, dO, K -> LDS by LDS-DMA: wave w stages rows 32 w .. + 31 of each (4 pieces of 8 rows x 128 B per matrix); the source
  // slot of a lane is its destination slot ^ swizzle(row), swizzle = 4 (i & 1) + (lane >> 4) for row 32 w + 8 i + (lane >> 3)
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    unsigned char* dst = smem + m * IMG + wave * 4096;
    const unsigned int stride = m == 1 ? D * 2 : D3 * 2;
    const unsigned int colb = m == 0 ? h * 128 : m == 1 ? h * 128 : (D + h * 64) * 2;      // Q | dO | K
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 32 + 8 * i + (lane >> 3);
      const unsigned int voff = row < S ? (unsigned int)(b * S + row) * stride + colb + (((lane & 7) ^ ((i & 1) * 4 + (lane >> 4))) << 4) : OOB;
      if (m == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_do, (lds_u8*)(dst + i * 1024), 16, voff, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, (lds_u8*)(dst + i * 1024), 16, voff, 0, 0, 0);
    }
  }
  // this wave's V fragments (keys 32 w .. + 31 as B operands), straight from global memory
  const int k0 = 32 * wave;
  bf16x8 bv[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int j = k0 + 16 * t + li;
      bv[t][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
          rs_qkv, j < S ? (unsigned int)(b * S + j) * (D3 * 2) + (2 * D + h * 64) * 2 + (ks * 4 + lg) * 16 : OOB, 0, 0));
    }
  // the O row half (64 B) of thread (row = tid >> 1, half = tid & 1) and the lse of row tid (< 256) for the delta / Ls prologue
  u32x4 o4[4];
  {
    const int row = tid >> 1;
    const unsigned int voff = row < S ? (unsigned int)(b * S + row) * (D * 2) + h * 128 + (tid & 1) * 64 : OOB;
#pragma unroll
    for (int e = 0; e < 4; ++e) o4[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_o, voff + e * 16, 0, 0);
  }
  const float lse_r = (tid < S && tid < 256) ? lse[((long long)b * H + h) * S + tid] : 0.f;
  const int nv = nvalid[b];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    const int row = tid >> 1, hf = tid & 1;
    float part = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32x4 d = __builtin_bit_cast(u32x4, fragk(dOs, row, 4 * hf + e));
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        part += bf16_to_f32((bf16_t)(d[w] & 0xffff)) * bf16_to_f32((bf16_t)(o4[e][w] & 0xffff));
        part += bf16_to_f32((bf16_t)(d[w] >> 16)) * bf16_to_f32((bf16_t)(o4[e][w] >> 16));
      }
    }
    part += __shfl_xor(part, 1, 64);
    if (hf == 0) Dl[row] = part;
    if (tid < 256) Ls[tid] = lse_r * LOG2E;
  }
  // this wave's K fragments (B operands) and K^T (d tile dt_w) of EVERY key block for dQ
  const int dt_w = wave & 3, qt_w = wave >> 2;       // this wave's tile of every query block's dQ^T
  bf16x8 bk[2][2], kT[8];
  float madd[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int j = k0 + 16 * t + li;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) bk[t][ks] = fragk(Ks, j, ks * 4 + lg);
    madd[t] = j < S ? (j < nv ? 0.f : -1e9f * LOG2E) : -INFINITY;
  }
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) kT[kb] = fragtr(Ks, 32 * kb, 16 * dt_w, li, lg);
  f32x4 dk[2][4], dv[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();      // Ls / Dl are in place

  const int nqb = (S + 31) >> 5;
  for (int qb = 0; qb < nqb; ++qb) {
    unsigned char* const dsi = smem + DS_OFF + (qb & 1) * DS_IMG;
    u32x2 ppk[2][2], dsk[2][2];     // P and dS as bf16 pairs, [query tile][key tile]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int q = qb * 32 + qt * 16;
      const bf16x8 aq0 = fragk(Qs, q + li, lg), aq1 = fragk(Qs, q + li, 4 + lg);
      const bf16x8 ad0 = fragk(dOs, q + li, lg), ad1 = fragk(dOs, q + li, 4 + lg);
      const f32x4 Lr = *reinterpret_cast<const f32x4*>(Ls + q + 4 * lg);
      const f32x4 Dr = *reinterpret_cast<const f32x4*>(Dl + q + 4 * lg);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq0, bk[t][0], z, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq1, bk[t][1], sacc, 0, 0, 0);
        f32x4 dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad0, bv[t][0], z, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad1, bv[t][1], dpacc, 0, 0, 0);
        float pe[4], de[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pe[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], c2, madd[t]) - Lr[r]);
          de[r] = pe[r] * (dpacc[r] - Dr[r]);
        }
        ppk[qt][t] = (u32x2){pack_bf16x2(pe[0], pe[1]), pack_bf16x2(pe[2], pe[3])};
        dsk[qt][t] = (u32x2){pack_bf16x2(de[0], de[1]), pack_bf16x2(de[2], de[3])};
        // dS tile -> image [key k0 + 16 t + li][query 16 qt + 4 lg .. + 3]
        const int row = k0 + 16 * t + li;
        *reinterpret_cast<u32x2*>(dsi + row * 64 + (((4 * qt + lg) ^ sp::hsw(row)) << 3)) = dsk[qt][t];
      }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const bf16x8 doT = fragtr(dOs, qb * 32, 16 * dt, li, lg), qT = fragtr(Qs, qb * 32, 16 * dt, li, lg);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 bp = __builtin_bit_cast(bf16x8, (u32x4){ppk[0][t][0], ppk[0][t][1], ppk[1][t][0], ppk[1][t][1]});
        const bf16x8 bds = __builtin_bit_cast(bf16x8, (u32x4){dsk[0][t][0], dsk[0][t][1], dsk[1][t][0], dsk[1][t][1]});
        dv[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(doT, bp, dv[t][dt], 0, 0, 0);
        dk[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT, bds, dk[t][dt], 0, 0, 0);
      }
    }
    // every wave's dS tile of this query block is in the image (the image of block qb - 2 was last read before the barrier
    // of block qb - 1), and the block's Q rows have been read for the last time: dQ^T tile (dt_w, qt_w) over all 256 keys,
    // written (bf16, scaled) over those rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT[kb], sp::scrtr(dsi + kb * 2048, 16 * qt_w, li, lg), acc, 0, 0, 0);
    {
      const int row = 32 * qb + 16 * qt_w + li;
      const f32x4 v = acc * scale;
      *reinterpret_cast<u32x2*>(Qs + row * 128 + (((2 * dt_w + (lg >> 1)) ^ isw(row)) << 4) + (lg & 1) * 8) =
          (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
  }
  // ---- dK (scaled) over this wave's rows of the K image, dV over the same rows of the dO image (every wave is past its reads
  // of both after the barrier); then the three images leave in 128-byte row pieces
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int row = k0 + 16 * t + li;
      const int so = (((2 * dt + (lg >> 1)) ^ isw(row)) << 4) + (lg & 1) * 8;
      const f32x4 kv = dk[t][dt] * scale;
      *reinterpret_cast<u32x2*>(Ks + row * 128 + so) = (u32x2){pack_bf16x2(kv[0], kv[1]), pack_bf16x2(kv[2], kv[3])};
      *reinterpret_cast<u32x2*>(dOs + row * 128 + so) = (u32x2){pack_bf16x2(dv[t][dt][0], dv[t][dt][1]), pack_bf16x2(dv[t][dt][2], dv[t][dt][3])};
    }
  __syncthreads();
  const unsigned int rowbase = (unsigned int)(b * S) * (D3 * 2) + h * 128;
#pragma unroll
  for (int m = 0; m < 3; ++m) {        // 0: dQ (Q image), 1: dK (K image), 2: dV (dO image)
    const unsigned char* img = m == 0 ? Qs : m == 1 ? Ks : dOs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 512 * i, r = idx >> 3, c16 = idx & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(img + r * 128 + ((c16 ^ isw(r)) << 4));
      __builtin_amdgcn_raw_buffer_store_b128(v, rs_dq, r < S ? rowbase + (unsigned int)r * (D3 * 2) + m * D * 2 + c16 * 16 : OOB, 0, 0);
    }
  }
}

}  // namespace sp64

template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return MFP_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    mfp_set_error("attention: cannot raise dynamic LDS to %zu bytes: %s", bytes, hipGetErrorString(e));
    return MFP_ELAUNCH;
  }
  return MFP_OK;
}

int check_attn(int B, int S, int H, int hd, int dtype) {
  MFP_CHECK_ARG(B > 0 && S > 0 && S <= 256 && H > 0);
  MFP_CHECK_ARG(hd == 16 || hd == 32 || hd == 64);
  MFP_CHECK_ARG(dtype == MFP_F32 || dtype == MFP_BF16);
  return MFP_OK;
}

template <int HD>
int fwd_hd(const void* qkv, const int* nvalid, void* out, float* lse, int B, int S, int H, int dtype, hipStream_t st) {
  const float scale = 1.0f / sqrtf((float)HD);
  dim3 grid(B * H), block(256);
  if (dtype == MFP_F32) {
    size_t lds = (size_t)2 * S * (HD + 1) * sizeof(float);
    MFP_CHECK_ARG(lds <= 160 * 1024);
    if (int rc = set_lds(attn_fwd_f32<HD>, lds)) return rc;
    hipLaunchKernelGGL(attn_fwd_f32<HD>, grid, block, lds, st, (const float*)qkv, nvalid, (float*)out, lse, S, H, scale);
  } else {
    constexpr int LDH = (HD < 32 ? 32 : HD) + 8;
    const int SP = (S + 31) & ~31;
    size_t lds = (size_t)3 * SP * LDH * sizeof(bf16_t) + (size_t)SP * sizeof(float);
    block = dim3(512);   // 8 waves: one 16-query tile each at S = 128 (measured 20.0 vs 21.0 us with 4)
    static const bool qsplit = !(getenv("MFP_ATTN_FWD_QSPLIT") && atoi(getenv("MFP_ATTN_FWD_QSPLIT")) == 0);   // A/B switch
    if (SP <= 128) {
      if (int rc = set_lds(attn_fwd_bf16<HD, 8>, lds)) return rc;
      hipLaunchKernelGGL((attn_fwd_bf16<HD, 8>), grid, block, lds, st, (const bf16_t*)qkv, nvalid, (bf16_t*)out, lse, S, H, scale);
    } else if (HD == 64 && qsplit) {
      const size_t lds2 = (size_t)2 * SP * LDH * sizeof(bf16_t) + (size_t)SP * sizeof(float);
      if (int rc = set_lds(attn_fwd_bf16<HD, 16, true>, lds2)) return rc;
      hipLaunchKernelGGL((attn_fwd_bf16<HD, 16, true>), dim3(2 * B * H), block, lds2, st, (const bf16_t*)qkv, nvalid, (bf16_t*)out, lse, S, H, scale);
    } else {
      if (int rc = set_lds(attn_fwd_bf16<HD, 16>, lds)) return rc;
      hipLaunchKernelGGL((attn_fwd_bf16<HD, 16>), grid, block, lds, st, (const bf16_t*)qkv, nvalid, (bf16_t*)out, lse, S, H, scale);
    }
  }
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

template <int HD>
int bwd_hd(const void* qkv, const int* nvalid, const void* out, const void* dout, const float* lse, void* dqkv,
           int B, int S, int H, int dtype, hipStream_t st) {
  const float scale = 1.0f / sqrtf((float)HD);
  dim3 grid(B * H), block(256);
  if (dtype == MFP_F32) {
    size_t lds1 = (size_t)2 * S * (HD + 1) * sizeof(float);
    size_t lds2 = lds1 + (size_t)2 * S * sizeof(float);
    MFP_CHECK_ARG(lds2 <= 160 * 1024);
    if (int rc = set_lds(attn_bwd_dq_f32<HD>, lds1)) return rc;
    if (int rc = set_lds(attn_bwd_dkv_f32<HD>, lds2)) return rc;
    hipLaunchKernelGGL(attn_bwd_dq_f32<HD>, grid, block, lds1, st, (const float*)qkv, nvalid, (const float*)out,
                       (const float*)dout, lse, (float*)dqkv, S, H, scale);
    MFP_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_dkv_f32<HD>, grid, block, lds2, st, (const float*)qkv, nvalid, (const float*)out,
                       (const float*)dout, lse, (float*)dqkv, S, H, scale);
  } else {
    static const bool single_pass = !(getenv("MFP_ATTN_BWD_SINGLE") && atoi(getenv("MFP_ATTN_BWD_SINGLE")) == 0);   // A/B switch
    if (HD == 32 && S <= 128 && single_pass && (long long)B * S * H * 32 * 3 * 2 < 0xFFFFFF00LL) {
      // persistent single-pass kernel: two workgroups per CU, each a contiguous run of (document, head) items
      const int ncu = mfp_ncu_launch();
      const int items = B * H;
      const int ipw = (items + 2 * ncu - 1) / (2 * ncu);
      const int nwg = (items + ipw - 1) / ipw;
      if (int rc = set_lds(sp::attn_bwd1_hd32, sp::LDS_BYTES)) return rc;
      hipLaunchKernelGGL(sp::attn_bwd1_hd32, dim3(nwg), dim3(256), sp::LDS_BYTES, st, (const bf16_t*)qkv, nvalid, (const bf16_t*)out,
                         (const bf16_t*)dout, lse, (bf16_t*)dqkv, B, S, H, scale, items, ipw);
      MFP_CHECK_LAUNCH();
      return MFP_OK;
    }
    static const bool single64 = !(getenv("MFP_ATTN_BWD_SINGLE64") && atoi(getenv("MFP_ATTN_BWD_SINGLE64")) == 0);   // A/B switch
    if (HD == 64 && S > 128 && S <= 256 && single64 && (long long)B * S * H * 64 * 3 * 2 < 0xFFFFFF00LL) {
      // single pass, one workgroup of eight waves per (document, head)
      if (int rc = set_lds(sp64::attn_bwd1_hd64, sp64::LDS_BYTES)) return rc;
      hipLaunchKernelGGL(sp64::attn_bwd1_hd64, dim3(B * H), dim3(512), sp64::LDS_BYTES, st, (const bf16_t*)qkv, nvalid, (const bf16_t*)out,
                         (const bf16_t*)dout, lse, (bf16_t*)dqkv, B, S, H, scale);
      MFP_CHECK_LAUNCH();
      return MFP_OK;
    }
    constexpr int LDH = (HD < 32 ? 32 : HD) + 8;
    const int SP = (S + 31) & ~31;
    size_t lds = (size_t)4 * SP * LDH * sizeof(bf16_t) + (size_t)3 * SP * sizeof(float);
    MFP_CHECK_ARG(lds <= 160 * 1024);
    if (int rc = set_lds(attn_bwd_bf16<HD>, lds)) return rc;
    if (SP > 128) block = dim3(512);
    hipLaunchKernelGGL(attn_bwd_bf16<HD>, grid, block, lds, st, (const bf16_t*)qkv, nvalid, (const bf16_t*)out,
                       (const bf16_t*)dout, lse, (bf16_t*)dqkv, S, H, scale);
  }
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

}  // namespace

extern "C" int mfp_attention_fwd(const void* qkv, const int32_t* nvalid, void* out, float* lse, int32_t B,
                                 int32_t S, int32_t H, int32_t hd, int32_t dtype, mfp_stream_t stream) {
  MFP_CHECK_ARG(qkv && nvalid && out && lse);
  if (int rc = check_attn(B, S, H, hd, dtype)) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hd == 16) return fwd_hd<16>(qkv, nvalid, out, lse, B, S, H, dtype, st);
  if (hd == 32) return fwd_hd<32>(qkv, nvalid, out, lse, B, S, H, dtype, st);
  return fwd_hd<64>(qkv, nvalid, out, lse, B, S, H, dtype, st);
}

extern "C" int mfp_attention_bwd(const void* qkv, const int32_t* nvalid, const void* out, const void* dout,
                                 const float* lse, void* dqkv, int32_t B, int32_t S, int32_t H, int32_t hd,
                                 int32_t dtype, mfp_stream_t stream) {
  MFP_CHECK_ARG(qkv && nvalid && out && dout && lse && dqkv);
  if (int rc = check_attn(B, S, H, hd, dtype)) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hd == 16) return bwd_hd<16>(qkv, nvalid, out, dout, lse, dqkv, B, S, H, dtype, st);
  if (hd == 32) return bwd_hd<32>(qkv, nvalid, out, dout, lse, dqkv, B, S, H, dtype, st);
  return bwd_hd<64>(qkv, nvalid, out, dout, lse, dqkv, B, S, H, dtype, st);
}
