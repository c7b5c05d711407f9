// Attention half of a DeepSVG block, backward input-gradient chain, in ONE launch (Keras autodiff of reference
// architecture/transformer.py:216-221, 60-99), for d_model 256, 8 heads of 32 and documents of exactly 128 positions
// (a 128-row tile is a document: its attention is local to the workgroup that owns the tile):
//
//     da   = d_o1 Wo                      d_o1 = dropout-masked gradient of the attention branch output, bf16 [T,256]
//     dqkv = MHSA'(q, k, v, lse; da)      (softmax backward with delta = rowsum(da * a))
//     dy1  = dqkv Wqkv                    gradient of LN1's output, consumed by ln_bwd
//
// It replaces dgrad_qkv_kernel<256>, attn_bwd1_hd32 and dgrad_qkv_kernel<768>: da (1 KB per token, written and read)
// never leaves the chip, dqkv (1.5 KB) is written once for the weight-gradient launch and not read back here; two launch
// boundaries less.  Per head pair p = 0..3:
//   c0  da_p = d_o1 Wo^T rows 64 p .. + 63 (weight chunk [64][512 B] of the transposed shadow) -> LDS image [128][128 B];
//       meanwhile the q / k / v columns of the pair stream HBM -> LDS images by LDS-DMA;
//   attention backward of heads 2 p, 2 p + 1 out of the four images, the single-pass algorithm of attn_bwd1_hd32
//       (csrc/attention.hip) with four waves per head: wave (hh, w) owns keys 32 w .. + 31 of head hh, every score once
//       (lane = key), P and dS feed dV / dK from registers, dS goes through a shared [128 keys][32 queries] image per head
//       and wave (dt, qt) computes the tile dQ^T[d][q] over all keys;
//   dq_p, dk_p, dv_p (bf16) overwrite the q / k / v images; from there they leave for HBM (dqkv) and are the A operands
//   c1..c3  dy1 accumulators (64 registers) += dq_p Wq^T + dk_p Wk^T + dv_p Wv^T slices ([256][128 B] chunks of the
//       transposed fused kernel, K = 64 each).
// Weights stream through TWO 32 KB buffers (LDS: 4 pair images 64 KB, dS 16 KB, weights 64 KB, statistics 2 KB); the
// chunk behind the attention phase is loaded during it, so only c3's chunk has a prefetch distance of one chunk.
// Images with 128-byte rows: 16-byte slot ^ ((row >> 1) & 7); dS image: 8-byte piece swizzle of attn_bwd1_hd32.
#include "common.h"
#include <type_traits>
#ifndef BB_ABL
#define BB_ABL 0
#endif

namespace {

struct AttnBwdBlockParams {
  const unsigned short* d_o1;      // [T][256] bf16
  const unsigned short* Wot;       // [256][256] bf16: Wot[c][n] = Wo[n][c]
  const unsigned short* qkv;       // [T][768] bf16 (saved by the forward pass)
  const unsigned short* a;         // [T][256] bf16 (attention output, saved)
  const float* lse;                // [B][8][128]
  const int* nvalid;               // [B]
  const unsigned short* Wqkvt;     // [256][768] bf16: Wqkvt[c][n] = Wqkv[n][c]
  unsigned short* dqkv;            // [T][768] bf16
  unsigned short* dy1;             // [T][256] bf16
  int T, H; float scale;
};

constexpr int BB_ROWS = 128, BB_D = 256;
constexpr int BB_IMG = BB_ROWS * 128;                 // 16 KB
constexpr int BB_Q = 0, BB_K = BB_IMG, BB_V = 2 * BB_IMG, BB_O = 3 * BB_IMG;
constexpr int BB_DS = 4 * BB_IMG;                     // [2 heads][128 keys][64 B]
constexpr int BB_LSD = BB_DS + 2 * 8192;              // [2 heads]{Ls[128], Dl[128]} f32
constexpr int BB_WS = BB_LSD + 2048;                  // 2 x 32 KB weight ring
constexpr int BB_WS_B = 32768;
constexpr int BB_LDS = BB_WS + 2 * BB_WS_B;           // 149 504 B

typedef __attribute__((address_space(3))) unsigned char lds_u8;

template <int I, int N, typename F>
__device__ __forceinline__ void bb_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    bb_static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ int isw(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int hsw(int row) { return (((row >> 2) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 1) & 1); }

// head hh of a pair image: 8 consecutive d of row `row` starting at d = 8 lg
__device__ __forceinline__ bf16x8 pfragk(const unsigned char* img, int hh, int row, int lg) {
  return *reinterpret_cast<const bf16x8*>(img + row * 128 + (((hh * 4 + lg) ^ isw(row)) << 4));
}
// head hh, column c0 + li (of its 32): the rows {kb + 4 lg + j} and {kb + 16 + 4 lg + j}, j = 0..3 (kb a multiple of 32)
__device__ __forceinline__ bf16x8 pfragtr(const unsigned char* img, int hh, int kb, int c0, int li, int lg) {
  const int row = kb + 4 * lg + (li >> 2);
  const int P = hh * 8 + (c0 >> 2) + (li & 3);
  const unsigned char* ptr = img + row * 128 + ((((P >> 1) ^ isw(row)) << 4) | ((P & 1) << 3));
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 16 * 128));     // (same swizzle: row + 16)
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// the same out of a 32-row block of a dS image ([keys][64 B], 8-byte piece swizzle)
__device__ __forceinline__ bf16x8 dstr(const unsigned char* blk, int c0, int li, int lg) {
  const int row = 4 * lg + (li >> 2);
  const int P = (c0 >> 2) + (li & 3);
  const unsigned char* ptr = blk + row * 64 + ((P ^ hsw(row)) << 3);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 16 * 64));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x8 bb_pack(const f32x4& a, const f32x4& b) {
  const u32x4 r = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
  return __builtin_bit_cast(bf16x8, r);
}

__global__ __launch_bounds__(512) void attn_block_bwd_kernel(AttnBwdBlockParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Ws = smem + BB_WS;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rp = wave & 3, nh = wave >> 2;            // product role: rows 32 rp .. + 31, column half nh
  const int hh = wave >> 2, w4 = wave & 3;            // attention role: head hh of the pair, key block w4
  const int k0 = 32 * w4, dt_w = w4 & 1, qt_w = w4 >> 1;
  const int doc = blockIdx.x, row0 = doc * BB_ROWS;
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr unsigned int OOB = 0xFFFFFFF0u;
  const int nv = __builtin_amdgcn_readfirstlane(p.nvalid[doc]);
  const float c2 = p.scale * LOG2E;

  const unsigned int xbytes = (unsigned int)p.T * (BB_D * 2);
  const __amdgpu_buffer_rsrc_t rs_do = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.d_o1), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wo = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Wot), 0, BB_D * BB_D * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wq = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Wqkvt), 0, BB_D * 768 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_qkv = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.qkv), 0, (unsigned int)p.T * (768 * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.a), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.lse), 0, (unsigned int)(p.T / BB_ROWS) * (unsigned int)p.H * BB_ROWS * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dq = __builtin_amdgcn_make_buffer_rsrc(p.dqkv, 0, (unsigned int)p.T * (768 * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(p.dy1, 0, xbytes, 0x00020000);

  // ---- weight chunk cg = 4 pr + t into ring buffer cg & 1.  t = 0: Wot rows 64 pr .. + 63, all 256 k -> [64][512 B],
  // slot ^ (row & 15); t = 1, 2, 3: Wqkvt rows 0 .. 255, k = (t - 1) * 256 + 64 pr .. + 63 -> [256][128 B], slot ^ isw(row)
  const unsigned int w1off = (unsigned int)((wave * 8 + (lane >> 5)) * 512 + (((lane & 31) ^ ((wave & 1) * 8 + (lane >> 5))) << 4));
  auto wload = [&](int cg) {
    const int pr = cg >> 2, t = cg & 3;
    unsigned char* dst = Ws + (cg & 1) * BB_WS_B + wave * 4096;
    if (t == 0) {
      const int base = (pr * 64) * (BB_D * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wo, (lds_u8*)(dst + i * 1024), 16, w1off ^ (i << 5), base + i * 1024, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        const unsigned int vo = (unsigned int)(row * (768 * 2) + (((lane & 7) ^ isw(row)) << 4));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wq, (lds_u8*)(dst + i * 1024), 16, vo, ((t - 1) * 256 + pr * 64) * 2, 0, 0);
      }
    }
  };
  // ---- q / k / v columns of pair pr -> the three images: 48 instructions of 8 rows x 128 B, six per wave
  auto iload = [&](int pr) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int idx = wave * 6 + i, t = idx >> 4, row = (idx & 15) * 8 + (lane >> 3);
      const unsigned int vo = (unsigned int)((row0 + row) * (768 * 2) + (((lane & 7) ^ isw(row)) << 4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, (lds_u8*)(smem + t * BB_IMG + (idx & 15) * 1024), 16, vo, (t * 256 + pr * 64) * 2, 0, 0);
    }
  };
  iload(0);
  wload(0);

  f32x4 acc2[8][2];      // dy1 accumulators: column tile T = (ct >> 2) * 8 + nh * 4 + (ct & 3), rows 32 rp + 16 rt + li
  int xs[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xs[ks] = ((ks * 4 + g) ^ li) << 4;
  // an image's rows -> dqkv columns t * 256 + 64 pr .. (128-byte pieces)
  auto stash = [&](int pr, int t) {
    const unsigned char* img = smem + t * BB_IMG;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i, r = idx >> 3, c16 = idx & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(img + r * 128 + ((c16 ^ isw(r)) << 4));
      __builtin_amdgcn_raw_buffer_store_b128(v, rs_dq, (unsigned int)(row0 + r) * (768 * 2) + t * 512 + pr * 128 + c16 * 16, 0, 0);
    }
  };
  // K = 64 product: acc2 += image t (A operand) x ring buffer `buf` ([256][128 B] chunk)
  auto kprod = [&](int t, int buf, bool first) {
    if (BB_ABL == 2) { if (first) for (int ct = 0; ct < 8; ++ct) acc2[ct][0] = acc2[ct][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; return; }
    const unsigned char* ai = smem + t * BB_IMG;
    const unsigned char* wb = Ws + buf * BB_WS_B;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 hf[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int row = rp * 32 + rt * 16 + li;
        hf[rt] = *reinterpret_cast<const bf16x8*>(ai + row * 128 + (((ks * 4 + g) ^ isw(row)) << 4));
      }
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
        const int wrow = ((ct >> 2) * 8 + nh * 4 + (ct & 3)) * 16 + li;
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + wrow * 128 + (((ks * 4 + g) ^ isw(wrow)) << 4));
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          acc2[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hf[rt], (first && ks == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc2[ct][rt], 0, 0, 0);
      }
    }
  };

  auto pair = [&](auto pp_) {
    constexpr int pr = decltype(pp_)::value;
    constexpr int cg0 = 4 * pr;
    // ---- (a) d_o1 fragments of this wave's two row tiles (registers), (b) the a piece and lse for delta, (c) chunk c1
    bf16x8 xf[2][8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        xf[rt][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
            rs_do, (unsigned int)(row0 + rp * 32 + rt * 16 + li) * (BB_D * 2) + g * 16 + ks * 64, 0, 0));
    const int drow = tid >> 2, dq4 = tid & 3;          // delta: row, 16-column quarter of the pair (head dq4 >> 1)
    const unsigned int aoff = (unsigned int)(row0 + drow) * (BB_D * 2) + pr * 128 + dq4 * 32;
    const u32x4 a0 = __builtin_amdgcn_raw_buffer_load_b128(rs_a, aoff, 0, 0);
    const u32x4 a1 = __builtin_amdgcn_raw_buffer_load_b128(rs_a, aoff + 16, 0, 0);
    const float lse_r = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
        rs_l, tid < 256 ? (unsigned int)(((doc * p.H + 2 * pr + (tid >> 7)) * BB_ROWS + (tid & 127)) * 4) : OOB, 0, 0));
    wload(cg0 + 1);
    // ---- (d) chunk c0 has landed: memory operations retire in order; younger than its loads are (pair 0) the 16 + 3 + 4
    // operations above, (later pairs) also the two dv stores at the head of the previous c3 and the six image loads
    if (pr == 0) asm volatile("s_waitcnt vmcnt(23)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- (e) c0: da_pair = d_o1 Wot rows 64 pr .. + 63 -> image O
    {
      const unsigned char* wa = Ws + (cg0 & 1) * BB_WS_B + ((nh * 2) * 16 + li) * 512;
      f32x4 acc[2][2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[nt][0] = acc[nt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      bf16x8 wf[2][2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) wf[0][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[0]);
#pragma unroll
      for (int ks = 0; ks < (BB_ABL == 3 ? 1 : 8); ++ks) {
        if (ks + 1 < 8) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            wf[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[(ks + 1) & 3] + ((ks + 1) >> 2) * 256);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
            acc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][nt], xf[rt][ks], acc[nt][rt], 0, 0, 0);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const int row = rp * 32 + rt * 16 + li;
          const u32x2 pk = {pack_bf16x2(acc[nt][rt][0], acc[nt][rt][1]), pack_bf16x2(acc[nt][rt][2], acc[nt][rt][3])};
          *reinterpret_cast<u32x2*>(smem + BB_O + row * 128 + ((((nh * 2 + nt) * 2 + (g >> 1)) ^ isw(row)) << 4) + (g & 1) * 8) = pk;
        }
    }
    // ---- (f) the pair's q / k / v images have landed (younger: the a / lse loads and chunk c1), da is in its image
    asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    wload(cg0 + 2);        // into c0's buffer; lands under the attention phase
    // ---- (g) delta = rowsum(da * a) per head, Ls = lse * log2(e)
    {
      float* const LsD = reinterpret_cast<float*>(smem + BB_LSD);
      const u32x4 d0 = *reinterpret_cast<const u32x4*>(smem + BB_O + drow * 128 + (((2 * dq4) ^ isw(drow)) << 4));
      const u32x4 d1 = *reinterpret_cast<const u32x4*>(smem + BB_O + drow * 128 + (((2 * dq4 + 1) ^ isw(drow)) << 4));
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        part += bf16_to_f32((unsigned short)(d0[e] & 0xffff)) * bf16_to_f32((unsigned short)(a0[e] & 0xffff));
        part += bf16_to_f32((unsigned short)(d0[e] >> 16)) * bf16_to_f32((unsigned short)(a0[e] >> 16));
        part += bf16_to_f32((unsigned short)(d1[e] & 0xffff)) * bf16_to_f32((unsigned short)(a1[e] & 0xffff));
        part += bf16_to_f32((unsigned short)(d1[e] >> 16)) * bf16_to_f32((unsigned short)(a1[e] >> 16));
      }
      part += __shfl_xor(part, 1, 64);
      if ((dq4 & 1) == 0) LsD[(dq4 >> 1) * 256 + 128 + drow] = part;
      if (tid < 256) LsD[(tid >> 7) * 256 + (tid & 127)] = lse_r * LOG2E;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- (h) attention backward of head 2 pr + hh: this wave owns keys k0 .. + 31
    {
      const unsigned char* const Qi = smem + BB_Q;
      const unsigned char* const Ki = smem + BB_K;
      const unsigned char* const Vi = smem + BB_V;
      const unsigned char* const Oi = smem + BB_O;
      unsigned char* const dsi = smem + BB_DS + hh * 8192;
      const float* const Ls = reinterpret_cast<const float*>(smem + BB_LSD) + hh * 256;
      const float* const Dl = Ls + 128;
      bf16x8 bk[2], bv[2];
      float madd[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int j = k0 + 16 * t + li;
        bk[t] = pfragk(Ki, hh, j, g);
        bv[t] = pfragk(Vi, hh, j, g);
        madd[t] = j < nv ? 0.f : -1e9f * LOG2E;
      }
      f32x4 dk[2][2], dv[2][2], dq[4];
      if (BB_ABL == 1) dq[0] = dq[1] = dq[2] = dq[3] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) { dk[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int qb = 0; qb < (BB_ABL == 1 ? 0 : 4); ++qb) {
        f32x4 pp[2][2], ds[2][2];     // [query tile][key tile]
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const int q = qb * 32 + qt * 16;
          const bf16x8 aq = pfragk(Qi, hh, q + li, g), ado = pfragk(Oi, hh, q + li, g);
          const f32x4 Lr = *reinterpret_cast<const f32x4*>(Ls + q + 4 * g);
          const f32x4 Dr = *reinterpret_cast<const f32x4*>(Dl + q + 4 * g);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, bk[t], z, 0, 0, 0);
            const f32x4 dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ado, bv[t], z, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], c2, madd[t]) - Lr[r]);
              pp[qt][t][r] = pe;
              ds[qt][t][r] = pe * (dpacc[r] - Dr[r]);
            }
            const int row = k0 + 16 * t + li;
            const u32x2 pk = {pack_bf16x2(ds[qt][t][0], ds[qt][t][1]), pack_bf16x2(ds[qt][t][2], ds[qt][t][3])};
            *reinterpret_cast<u32x2*>(dsi + row * 64 + (((4 * qt + g) ^ hsw(row)) << 3)) = pk;
          }
        }
        const bf16x8 doT0 = pfragtr(Oi, hh, qb * 32, 0, li, g), doT1 = pfragtr(Oi, hh, qb * 32, 16, li, g);
        const bf16x8 qT0 = pfragtr(Qi, hh, qb * 32, 0, li, g), qT1 = pfragtr(Qi, hh, qb * 32, 16, li, g);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x8 bp = bb_pack(pp[0][t], pp[1][t]);
          const bf16x8 bds = bb_pack(ds[0][t], ds[1][t]);
          dv[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(doT0, bp, dv[t][0], 0, 0, 0);
          dv[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(doT1, bp, dv[t][1], 0, 0, 0);
          dk[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT0, bds, dk[t][0], 0, 0, 0);
          dk[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT1, bds, dk[t][1], 0, 0, 0);
        }
        // every wave's dS tile of this query block is in the head's image: dQ^T tile (dt_w, qt_w) over all 128 keys
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        f32x4 accq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
          accq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfragtr(Ki, hh, 32 * kb, 16 * dt_w, li, g), dstr(dsi + kb * 2048, 16 * qt_w, li, g), accq, 0, 0, 0);
        dq[qb] = accq;
        // (single dS image per head: everyone has read it before the next query block's tiles go in)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      // ---- (i) dq / dk / dv (bf16) over the q / k / v images: every wave is past its last read of them (the barrier above)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int row = k0 + 16 * t + li;
          const int so = (((hh * 4 + dt * 2 + (g >> 1)) ^ isw(row)) << 4) + (g & 1) * 8;
          const f32x4 kv = dk[t][dt] * p.scale;
          *reinterpret_cast<u32x2*>(smem + BB_K + row * 128 + so) = (u32x2){pack_bf16x2(kv[0], kv[1]), pack_bf16x2(kv[2], kv[3])};
          *reinterpret_cast<u32x2*>(smem + BB_V + row * 128 + so) = (u32x2){pack_bf16x2(dv[t][dt][0], dv[t][dt][1]), pack_bf16x2(dv[t][dt][2], dv[t][dt][3])};
        }
#pragma unroll
      for (int qb = 0; qb < 4; ++qb) {
        const int row = 32 * qb + 16 * qt_w + li;
        const f32x4 qv = dq[qb] * p.scale;
        *reinterpret_cast<u32x2*>(smem + BB_Q + row * 128 + (((hh * 4 + dt_w * 2 + (g >> 1)) ^ isw(row)) << 4) + (g & 1) * 8) =
            (u32x2){pack_bf16x2(qv[0], qv[1]), pack_bf16x2(qv[2], qv[3])};
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- (j) c1: dq -> HBM; dy1 += dq_pair Wq^T slice (chunk c1, buffer 1); then chunk c2 (loaded under the attention)
    stash(pr, 0);
    kprod(0, (cg0 + 1) & 1, pr == 0);
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");      // chunk c2 landed (younger: the two dq stores)
    __builtin_amdgcn_s_barrier();
    // ---- (k) c2
    wload(cg0 + 3);        // into c1's buffer (read by everyone before the barrier above)
    stash(pr, 1);
    kprod(1, (cg0 + 2) & 1, false);
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");      // chunk c3 landed (younger: the two dk stores)
    __builtin_amdgcn_s_barrier();
    // ---- (l) c3; the next pair's first chunk goes into c2's buffer
    if (pr < 3) wload(cg0 + 4);
    stash(pr, 2);
    kprod(2, (cg0 + 3) & 1, false);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- (m) the images are free: the next pair's q / k / v columns
    if (pr < 3) iload(pr + 1);
  };
  bb_static_for<0, 4>(pair);

  // ---- dy1 (bf16) -> [128][512 B] image over the four pair images (slot ^ (row & 15)) -> whole 512-byte rows
#pragma unroll
  for (int ct = 0; ct < 8; ++ct)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = rp * 32 + rt * 16 + li, tl = (ct >> 2) * 8 + nh * 4 + (ct & 3);
      const u32x2 pk = {pack_bf16x2(acc2[ct][rt][0], acc2[ct][rt][1]), pack_bf16x2(acc2[ct][rt][2], acc2[ct][rt][3])};
      *reinterpret_cast<u32x2*>(smem + row * 512 + (((tl * 2 + (g >> 1)) ^ (row & 15)) << 4) + (g & 1) * 8) = pk;
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 512 * i, r = idx >> 5, c16 = idx & 31;
    const u32x4 yv = *reinterpret_cast<const u32x4*>(smem + r * 512 + ((c16 ^ (r & 15)) << 4));
    __builtin_amdgcn_raw_buffer_store_b128(yv, rs_dy, (unsigned int)(row0 + r) * (BB_D * 2) + c16 * 16, 0, 0);
  }
}

}  // namespace

extern "C" int mfp_attn_block_bwd(const void* d_o1, const void* Wot, const void* qkv, const void* a, const float* lse,
                                  const int32_t* nvalid, const void* Wqkvt, void* dqkv, void* dy1, int32_t B, int32_t S,
                                  int32_t D, int32_t H, mfp_stream_t stream) {
  MFP_CHECK_ARG(d_o1 && Wot && qkv && a && lse && nvalid && Wqkvt && dqkv && dy1);
  MFP_CHECK_ARG(B > 0 && B <= 8192 && S == BB_ROWS && D == BB_D && H == 8);
  MFP_CHECK_ARG(((uintptr_t)d_o1 % 16) == 0 && ((uintptr_t)Wot % 16) == 0 && ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)a % 16) == 0 &&
                ((uintptr_t)Wqkvt % 16) == 0 && ((uintptr_t)dqkv % 16) == 0 && ((uintptr_t)dy1 % 16) == 0);
  AttnBwdBlockParams p;
  p.d_o1 = reinterpret_cast<const unsigned short*>(d_o1); p.Wot = reinterpret_cast<const unsigned short*>(Wot);
  p.qkv = reinterpret_cast<const unsigned short*>(qkv); p.a = reinterpret_cast<const unsigned short*>(a);
  p.lse = lse; p.nvalid = nvalid; p.Wqkvt = reinterpret_cast<const unsigned short*>(Wqkvt);
  p.dqkv = reinterpret_cast<unsigned short*>(dqkv); p.dy1 = reinterpret_cast<unsigned short*>(dy1);
  p.T = B * S; p.H = H; p.scale = 1.0f / sqrtf(32.0f);
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS);
    if (e != hipSuccess) {
      mfp_set_error("mfp_attn_block_bwd: cannot raise dynamic LDS to %d: %s", BB_LDS, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_block_bwd_kernel, dim3(B), dim3(512), BB_LDS, reinterpret_cast<hipStream_t>(stream), p);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}
