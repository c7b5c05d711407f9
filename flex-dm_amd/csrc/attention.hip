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
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xlane_sum4(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
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

// MAX_KT = key tiles held in registers: 8 (S <= 128) or 16 (S <= 256)
template <int HD, int MAX_KT>
__global__ __launch_bounds__(512, 2) void attn_fwd_bf16(const bf16_t* __restrict__ qkv, const int* __restrict__ nvalid,
                                                     bf16_t* __restrict__ out, float* __restrict__ lse, int S,
                                                     int H, float scale) {
  constexpr int HDP = HD < 32 ? 32 : HD, LDH = HDP + 8, KS = HDP / 32, DT = HD / 16;
  extern __shared__ __attribute__((aligned(16))) bf16_t smb[];
  const int SP = (S + 31) & ~31;
  bf16_t* Qs = smb;
  bf16_t* Ks = Qs + SP * LDH;
  bf16_t* Vs = Ks + SP * LDH;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);   // heads of one document -> one XCD (shared 128-B lines)
  const int b = bid / H, h = bid % H;
  const int D = H * HD, D3 = 3 * D;
  const int nv = nvalid[b];
  const bf16_t* base = qkv + (long long)b * S * D3 + h * HD;
  stage_head<HD, HDP, LDH>(Qs, base, D3, S, SP);
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
  for (int qt = wave; qt < nkt; qt += (int)(blockDim.x >> 6)) {
    bf16x8 bq[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bq[ks] = frag_k(Qs, LDH, qt * 16 + li, ks * 32, lg);
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

template <int HD>
__global__ __launch_bounds__(256, 2) void attn_bwd_bf16(const bf16_t* __restrict__ qkv, const int* __restrict__ nvalid,
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
  const int nblk = SP / 32;

  // ---------------- loop A: dQ for query block qb (S^T orientation: lane = query)
  for (int qb = wave; qb < nblk; qb += 4) {
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
  for (int kb = wave; kb < nblk; kb += 4) {
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
    static const int fwd_threads = getenv("MFP_ATTN_FWD_THREADS") ? atoi(getenv("MFP_ATTN_FWD_THREADS")) : 512;   // 8 waves: one 16-query tile each at S = 128 (measured 20.0 vs 21.0 us with 4)
    block = dim3(fwd_threads);
    if (SP <= 128) {
      if (int rc = set_lds(attn_fwd_bf16<HD, 8>, lds)) return rc;
      hipLaunchKernelGGL((attn_fwd_bf16<HD, 8>), grid, block, lds, st, (const bf16_t*)qkv, nvalid, (bf16_t*)out, lse, S, H, scale);
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
    constexpr int LDH = (HD < 32 ? 32 : HD) + 8;
    const int SP = (S + 31) & ~31;
    size_t lds = (size_t)4 * SP * LDH * sizeof(bf16_t) + (size_t)3 * SP * sizeof(float);
    MFP_CHECK_ARG(lds <= 160 * 1024);
    if (int rc = set_lds(attn_bwd_bf16<HD>, lds)) return rc;
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
