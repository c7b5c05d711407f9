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
// boundaries less.  LDS (all 160 KB of the CU): da image [128][512 B] (64 KB; at the end the dy1 image), the q | k | v
// images of one head pair [128][128 B] each (48 KB), a ring of three 16 KB weight chunks.
//   da product first: the d_o1 fragments of the tile sit in registers, eight chunks of 32 rows of the transposed
//       output-projection kernel stream through the ring, da (bf16) goes into its image -- d_o1 is read from HBM once.
//   Per head pair p = 0..3:
//     K / V fragments (keys 32 w .. + 31 of head hh as B operands) and K^T of every key block go into registers; from
//     then on the k image's place holds Ls / Dl (delta = rowsum(da * a) from the da image and the pair's a columns) and
//     the v image's place the dS exchange;
//     attention backward of heads 2 p, 2 p + 1: the single-pass algorithm of attn_bwd1_hd32 (csrc/attention.hip) with four
//     waves per head -- every score once (lane = key), P and dS (packed bf16) feed dV / dK from registers, dS goes through a
//     shared [128 keys][32 queries] image per head and wave (dt, qt) computes the tile dQ^T[d][q] over all keys; dq is
//     written over the q rows of a query block as soon as the block is done, dk / dv over the k / v images at the end;
//     six K = 32 steps: dy1 accumulators (64 registers) += d{q,k,v}_p times the matching [256][64 B] chunk of the transposed
//     fused Q|K|V kernel; the three images leave for HBM (dqkv) on the way;
//     the next pair's q | k | v columns, a columns and lse are prefetched into 33 REGISTERS behind the pair's last weight
//     chunk and placed into the images when the products are done with them.
// Images with 128-byte rows: 16-byte slot ^ ((row >> 1) & 7); da image (512-byte rows): slot ^ ((row & 7) << 1 | (row >> 3) & 1)
// -- conflict-free for the row-wise fragment reads and the transposing reads; dS image: 8-byte piece swizzle of
// attn_bwd1_hd32.  256 VGPRs, no spill (a spill reload is a scratch access: it would drain the weight ring).
#include "common.h"
#include "ln_bwd_tile.h"
#include <type_traits>
#ifndef BB_ABL
#define BB_ABL 0
#endif
// BB_ABL == 9: timing build (tools/trace_attn_bwd.py): every wave drops shader-clock stamps at its phase boundaries into
// the dy1 buffer (scalar stores: no vector-memory operation added) instead of the dy1 rows
#define BB_TR(i) do { if (BB_ABL == 9) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    asm volatile("s_store_dwordx2 %0, %1, %2 glc" :: "s"(t_), "s"(trbase), "n"((i) * 8) : "memory"); } } while (0)

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
  LnTileArgs ln;                   // LNB form (mfp_attn_block_bwd_ln): the backward of LN1 in the epilogue -- dy1 never leaves the CU
};

constexpr int BB_ROWS = 128, BB_D = 256;
constexpr int BB_IMG = BB_ROWS * 128;                 // 16 KB
constexpr int BB_DA = 0;                              // da [128][512 B]; at the end the dy1 image
constexpr int BB_Q = BB_ROWS * 512, BB_K = BB_Q + BB_IMG, BB_V = BB_K + BB_IMG;
constexpr int BB_LSD = BB_K;                          // during the attention: [2 heads]{Ls[128], Dl[128]} f32 over the k image
constexpr int BB_DS = BB_V;                           // during the attention: dS [2 heads][128 keys][64 B] over the v image
constexpr int BB_WS = BB_V + BB_IMG;                  // 3 x 16 KB weight ring
constexpr int BB_WS_B = 16384;
constexpr int BB_LDS = BB_WS + 3 * BB_WS_B;           // 163 840 B: all of a CU's LDS

typedef __attribute__((address_space(3))) unsigned char lds_u8;

template <int I, int N, typename F>
__device__ __forceinline__ void bb_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    bb_static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ int isw(int row) { return (row >> 1) & 7; }                             // 128-byte rows, 16-byte slots
__device__ __forceinline__ int dsw(int row) { return ((row & 7) << 1) | ((row >> 3) & 1); }        // 512-byte rows (da image)
__device__ __forceinline__ int swz64(int row) { return ((row >> 2) & 1) << 1; }                    // 64-byte rows
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
// the same two out of the da image: head h8 (0..7) of the 256 columns
__device__ __forceinline__ bf16x8 dafragk(const unsigned char* da, int h8, int row, int lg) {
  return *reinterpret_cast<const bf16x8*>(da + row * 512 + (((h8 * 4 + lg) ^ dsw(row)) << 4));
}
__device__ __forceinline__ bf16x8 dafragtr(const unsigned char* da, int h8, int kb, int c0, int li, int lg) {
  const int row = kb + 4 * lg + (li >> 2);
  const int P = h8 * 8 + (c0 >> 2) + (li & 3);
  const unsigned char* ptr = da + row * 512 + ((((P >> 1) ^ dsw(row)) << 4) | ((P & 1) << 3));
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 16 * 512));     // (same swizzle: row + 16)
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// out of a 32-row block of a dS image ([keys][64 B], 8-byte piece swizzle)
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
__device__ __forceinline__ float dot8(const u32x4& x, const u32x4& y) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s += bf16_to_f32((unsigned short)(x[e] & 0xffff)) * bf16_to_f32((unsigned short)(y[e] & 0xffff));
    s += bf16_to_f32((unsigned short)(x[e] >> 16)) * bf16_to_f32((unsigned short)(y[e] >> 16));
  }
  return s;
}

// Weight chunks, in consumption order: seq 0..7 = Wot rows 32 c .. + 31 ([32][512 B], slot ^ (row & 15)); seq 8 + 6 p + j =
// Wqkvt rows 0..255, k = (j >> 1) * 256 + 64 p + 32 (j & 1) .. + 31 ([256][64 B], slot ^ swz64(row)); ring slot = seq % 3.
// Memory operations retire in order, so "chunk seq has landed" = s_waitcnt vmcnt(number of operations issued after its two
// loads); the schedule below issues per wave, in this order (P = 9 prefetch loads of the next pair's q | k | v, a, lse; S = 2 stores):
//   prologue: ... A0 A1 | step c of the da product: wait(2) barrier, issue chunk c + 2 (A.., then pair 0's j = 0, 1)
//   after the K / V fragment reads of a pair: the pair's chunk j = 2
//   pair, step j: wait(N_j) barrier; j = 0: S(dq); j = 1: chunk 3; j = 2: S(dk) chunk 4; j = 3: chunk 5, P;
//                 j = 4: S(dv) next pair's chunk 0; j = 5: next pair's chunk 1
//   N_0 .. N_3 = 4   N_4 = 2 + P = 11   N_5 = P + S + 2 = 13   (last pair: no P, no next chunks: N_4 = N_5 = 2)
// P goes behind the pair's last weight chunk: loads return in order, a weight chunk (an L2 hit) issued behind the HBM loads
// of P would not count as landed before they are.
// SDOC = positions per document: 128 (a tile is a document) or 64 (two documents per tile, see csrc/block_attn.hip): a wave owns
// 32 keys of ONE document (key block w4: document w4 >> 1) and walks all four 32-query blocks -- the blocks of the other
// document get the padding term (P = dS = 0 exactly), so the barrier structure is the same for both forms.
// LNB: 0 = dy1 leaves as bf16 rows; 1 = LN1 backward in the epilogue from x (f32); 2 = from the x-hat stash (bf16)
template <int SDOC, int LNB = 0>
__global__ __launch_bounds__(512) void attn_block_bwd_kernel(AttnBwdBlockParams p) {
  static_assert(SDOC == 128 || SDOC == 64, "documents of 128 or 64 positions");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Ws = smem + BB_WS;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rp = wave & 3, nh = wave >> 2;            // product role: rows 32 rp .. + 31, column half nh
  const int hh = wave >> 2, w4 = wave & 3;            // attention role: head hh of the pair, key block w4
  const int k0 = 32 * w4, dt_w = w4 & 1, qt_w = w4 >> 1;
  const int doc = blockIdx.x, row0 = doc * BB_ROWS;
  constexpr float LOG2E = 1.4426950408889634f;
  const int kd = SDOC == 128 ? 0 : (w4 >> 1);          // this wave's keys' document inside the tile
  const int nv = __builtin_amdgcn_readfirstlane(p.nvalid[SDOC == 128 ? doc : 2 * doc + kd]);
  const float c2 = p.scale * LOG2E;
  const unsigned long long* trbase = reinterpret_cast<const unsigned long long*>(p.dy1) + (size_t)(doc * 8 + wave) * 64;
  BB_TR(0);

  const unsigned int xbytes = (unsigned int)p.T * (BB_D * 2);
  const __amdgpu_buffer_rsrc_t rs_do = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.d_o1), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wo = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Wot), 0, BB_D * BB_D * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wq = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Wqkvt), 0, BB_D * 768 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_qkv = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.qkv), 0, (unsigned int)p.T * (768 * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.a), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.lse), 0, (unsigned int)(p.T / BB_ROWS) * (unsigned int)p.H * BB_ROWS * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dq = __builtin_amdgcn_make_buffer_rsrc(p.dqkv, 0, (unsigned int)p.T * (768 * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(p.dy1, 0, xbytes, 0x00020000);

  auto wloadA = [&](int c) {           // seq c
    unsigned char* dst = Ws + (c % 3) * BB_WS_B + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wave * 4 + i * 2 + (lane >> 5);
      const unsigned int vo = (unsigned int)(row * 512 + (((lane & 31) ^ (row & 15)) << 4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wo, (lds_u8*)(dst + i * 1024), 16, vo, c * (32 * 512), 0, 0);
    }
  };
  auto wloadP = [&](int pr, int j) {   // seq 8 + 6 pr + j
    unsigned char* dst = Ws + ((8 + 6 * pr + j) % 3) * BB_WS_B + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wave * 32 + i * 16 + (lane >> 2);
      const unsigned int vo = (unsigned int)(row * (768 * 2) + (((lane & 3) ^ swz64(row)) << 4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wq, (lds_u8*)(dst + i * 1024), 16, vo, ((j >> 1) * 256 + pr * 64 + (j & 1) * 32) * 2, 0, 0);
    }
  };
  // the q | k | v columns of pair pr: six 16-byte pieces per thread (piece i: image i >> 1, row (tid + 512 (i & 1)) >> 3)
  u32x4 nq[6];
  auto qkv_fetch = [&](int pr) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int idx = tid + 512 * (i & 1), r = idx >> 3, c16 = idx & 7;
      nq[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_qkv, (unsigned int)(row0 + r) * (768 * 2) + (i >> 1) * 512 + pr * 128 + c16 * 16, 0, 0);
    }
  };
  auto qkv_place = [&]() {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int idx = tid + 512 * (i & 1), r = idx >> 3, c16 = idx & 7;
      *reinterpret_cast<u32x4*>(smem + BB_Q + (i >> 1) * BB_IMG + r * 128 + ((c16 ^ isw(r)) << 4)) = nq[i];
    }
  };
  // an image's rows -> dqkv columns t * 256 + 64 pr .. (128-byte pieces)
  auto stash = [&](int pr, int t) {
    const unsigned char* img = smem + BB_Q + t * BB_IMG;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i, r = idx >> 3, c16 = idx & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(img + r * 128 + ((c16 ^ isw(r)) << 4));
      __builtin_amdgcn_raw_buffer_store_b128(v, rs_dq, (unsigned int)(row0 + r) * (768 * 2) + t * 512 + pr * 128 + c16 * 16, 0, 0);
    }
  };

  // the a pieces / lse of pair pr for its deltas: thread (row tid >> 2, 16-column quarter tid & 3: head (tid & 3) >> 1)
  const int drow = tid >> 2, dq4 = tid & 3;
  u32x4 a0, a1;
  float lse_r;
  auto a_fetch = [&](int pr) {
    const unsigned int aoff = (unsigned int)(row0 + drow) * (BB_D * 2) + pr * 128 + dq4 * 32;
    a0 = __builtin_amdgcn_raw_buffer_load_b128(rs_a, aoff, 0, 0);
    a1 = __builtin_amdgcn_raw_buffer_load_b128(rs_a, aoff + 16, 0, 0);
    lse_r = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
        rs_l, tid >= 256 ? 0xFFFFFFF0u
              : SDOC == 128 ? (unsigned int)(((doc * p.H + 2 * pr + (tid >> 7)) * BB_ROWS + (tid & 127)) * 4)
                            : (unsigned int)((((2 * doc + ((tid >> 6) & 1)) * p.H + 2 * pr + (tid >> 7)) * 64 + (tid & 63)) * 4), 0, 0));
  };
  // ---- prologue: the d_o1 fragments and chunks 0, 1 first (the da product starts when they are in); behind them pair 0's
  // q | k | v and a / lse (nine loads: they land under the da product)
  {
    bf16x8 xf[2][8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        xf[rt][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
            rs_do, (unsigned int)(row0 + rp * 32 + rt * 16 + li) * (BB_D * 2) + g * 16 + ks * 64, 0, 0));
    wloadA(0);
    wloadA(1);
    qkv_fetch(0);
    a_fetch(0);
    BB_TR(1);
    int xs[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xs[ks] = ((ks * 4 + g) ^ li) << 4;
    // ---- da = d_o1 Wo: eight chunks of 32 output columns -> the da image
    bb_static_for<0, 8>([&](auto c_) {
      constexpr int c = decltype(c_)::value;
      if (c < 2) asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)" ::: "memory");      // younger: the next chunk + the nine
      else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (c == 0) BB_TR(3);
      if (c + 2 < 8) wloadA(c + 2);
      else wloadP(0, c + 2 - 8);
      const unsigned char* wa = Ws + (c % 3) * BB_WS_B + (nh * 16 + li) * 512;
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wa + xs[ks & 3] + (ks >> 2) * 256);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[rt][ks], acc[rt], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int row = rp * 32 + rt * 16 + li;
        const u32x2 pk = {pack_bf16x2(acc[rt][0], acc[rt][1]), pack_bf16x2(acc[rt][2], acc[rt][3])};
        *reinterpret_cast<u32x2*>(smem + BB_DA + row * 512 + (((4 * c + 2 * nh + (g >> 1)) ^ dsw(row)) << 4) + (g & 1) * 8) = pk;
      }
    });
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  BB_TR(4);
  qkv_place();
  BB_TR(2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x4 acc2[8][2];      // dy1 accumulators: column tile T = (ct >> 2) * 8 + nh * 4 + (ct & 3), rows 32 rp + 16 rt + li
  // one K = 32 step of the dy1 product: acc2 += image t, k half kh (A operand) x ring slot ([256][64 B] chunk)
  auto kprod = [&](int t, int kh, int slot, bool first) {
    const unsigned char* ai = smem + BB_Q + t * BB_IMG;
    const unsigned char* wb = Ws + slot * BB_WS_B;
    bf16x8 hf[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = rp * 32 + rt * 16 + li;
      hf[rt] = *reinterpret_cast<const bf16x8*>(ai + row * 128 + (((kh * 4 + g) ^ isw(row)) << 4));
    }
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const int wrow = ((ct >> 2) * 8 + nh * 4 + (ct & 3)) * 16 + li;
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + wrow * 64 + ((g ^ swz64(wrow)) << 4));
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
        acc2[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hf[rt], first ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc2[ct][rt], 0, 0, 0);
    }
  };

  auto pair = [&](auto pp_) {
    constexpr int pr = decltype(pp_)::value;
    constexpr bool last = pr == 3;
    BB_TR(5 + 12 * pr);
    // ---- delta = rowsum(da * a) of the pair's two heads (this thread: a row's 16 columns), Ls = lse * log2(e)
    float dpart;
    {
      const u32x4 d0 = *reinterpret_cast<const u32x4*>(smem + BB_DA + drow * 512 + (((pr * 8 + 2 * dq4) ^ dsw(drow)) << 4));
      const u32x4 d1 = *reinterpret_cast<const u32x4*>(smem + BB_DA + drow * 512 + (((pr * 8 + 2 * dq4 + 1) ^ dsw(drow)) << 4));
      dpart = dot8(d0, a0) + dot8(d1, a1);
      dpart += __shfl_xor(dpart, 1, 64);
    }
    const float ls_w = lse_r * LOG2E;
    // ---- this wave's K / V fragments (keys k0 .. + 31 as B operands) and K^T (d tile dt_w) of every key block; after the
    // barrier the k / v images are free: Ls / Dl go over the first, dS over the second
    int li_p = li, g_p = g;      // (opaque copies: the pair's LDS addresses are worked out here, not kept from kernel entry)
    asm volatile("" : "+v"(li_p), "+v"(g_p));
    bf16x8 bk[2], bv[2], kT[4];
    float madd[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = k0 + 16 * t + li_p;
      bk[t] = pfragk(smem + BB_K, hh, j, g_p);
      bv[t] = pfragk(smem + BB_V, hh, j, g_p);
      madd[t] = (SDOC == 128 ? j : (j & 63)) < nv ? 0.f : -1e9f * LOG2E;
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) kT[kb] = pfragtr(smem + BB_K, hh, 32 * kb, 16 * dt_w, li_p, g_p);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // (the compiler drains every LDS-DMA before a transposing LDS read: the pair's third chunk is issued behind those reads)
    wloadP(pr, 2);
    {
      float* const LsD = reinterpret_cast<float*>(smem + BB_LSD);
      if ((dq4 & 1) == 0) LsD[(dq4 >> 1) * 256 + 128 + drow] = dpart;
      if (tid < 256) LsD[(tid >> 7) * 256 + (tid & 127)] = ls_w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- attention backward of head 2 pr + hh: this wave owns keys k0 .. + 31 (see attn_bwd1_hd32, csrc/attention.hip)
    BB_TR(6 + 12 * pr);
    {
      const unsigned char* const Qi = smem + BB_Q;
      const unsigned char* const Da = smem + BB_DA;
      unsigned char* const dsi = smem + BB_DS + hh * 8192;
      const float* const Ls = reinterpret_cast<const float*>(smem + BB_LSD) + hh * 256;
      const float* const Dl = Ls + 128;
      const int h8 = 2 * pr + hh;
      f32x4 dk[2][2], dv[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) { dk[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int qb = 0; qb < (BB_ABL == 1 ? 0 : 4); ++qb) {
        u32x2 ppk[2][2], dsk[2][2];     // P and dS as bf16 pairs, [query tile][key tile]
        // SDOC = 64: queries 32 qb .. + 31 belong to document qb >> 1; the other document's keys weigh nothing
        const float mq[2] = {(SDOC == 64 && (qb >> 1) != kd) ? -1e9f * LOG2E : madd[0], (SDOC == 64 && (qb >> 1) != kd) ? -1e9f * LOG2E : madd[1]};
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const int q = qb * 32 + qt * 16;
          const bf16x8 aq = pfragk(Qi, hh, q + li_p, g_p), ado = dafragk(Da, h8, q + li_p, g_p);
          const f32x4 Lr = *reinterpret_cast<const f32x4*>(Ls + q + 4 * g_p);
          const f32x4 Dr = *reinterpret_cast<const f32x4*>(Dl + q + 4 * g_p);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, bk[t], z, 0, 0, 0);
            const f32x4 dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ado, bv[t], z, 0, 0, 0);
            float pe[4], de[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              pe[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], c2, mq[t]) - Lr[r]);
              de[r] = pe[r] * (dpacc[r] - Dr[r]);
            }
            ppk[qt][t] = (u32x2){pack_bf16x2(pe[0], pe[1]), pack_bf16x2(pe[2], pe[3])};
            dsk[qt][t] = (u32x2){pack_bf16x2(de[0], de[1]), pack_bf16x2(de[2], de[3])};
            const int row = k0 + 16 * t + li_p;
            *reinterpret_cast<u32x2*>(dsi + row * 64 + (((4 * qt + g_p) ^ hsw(row)) << 3)) = dsk[qt][t];
          }
        }
        const bf16x8 doT0 = dafragtr(Da, h8, qb * 32, 0, li_p, g_p), doT1 = dafragtr(Da, h8, qb * 32, 16, li_p, g_p);
        const bf16x8 qT0 = pfragtr(Qi, hh, qb * 32, 0, li_p, g_p), qT1 = pfragtr(Qi, hh, qb * 32, 16, li_p, g_p);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x8 bp = __builtin_bit_cast(bf16x8, (u32x4){ppk[0][t][0], ppk[0][t][1], ppk[1][t][0], ppk[1][t][1]});
          const bf16x8 bds = __builtin_bit_cast(bf16x8, (u32x4){dsk[0][t][0], dsk[0][t][1], dsk[1][t][0], dsk[1][t][1]});
          dv[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(doT0, bp, dv[t][0], 0, 0, 0);
          dv[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(doT1, bp, dv[t][1], 0, 0, 0);
          dk[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT0, bds, dk[t][0], 0, 0, 0);
          dk[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT1, bds, dk[t][1], 0, 0, 0);
        }
        // every wave's dS tile of this query block is in the head's image: dQ^T tile (dt_w, qt_w) over all 128 keys; the
        // q image's rows of this block have been read for the last time, dq (bf16, scaled) goes over them
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        f32x4 accq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
          accq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT[kb], dstr(dsi + kb * 2048, 16 * qt_w, li_p, g_p), accq, 0, 0, 0);
        {
          const int row = 32 * qb + 16 * qt_w + li_p;
          const f32x4 qv = accq * p.scale;
          *reinterpret_cast<u32x2*>(smem + BB_Q + row * 128 + (((hh * 4 + dt_w * 2 + (g_p >> 1)) ^ isw(row)) << 4) + (g_p & 1) * 8) =
              (u32x2){pack_bf16x2(qv[0], qv[1]), pack_bf16x2(qv[2], qv[3])};
        }
        // (one dS image per head: everyone has read it before the next query block's tiles go in)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      // ---- dk / dv (bf16) over the k / v images (Ls / Dl and dS are dead: the barrier above)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int row = k0 + 16 * t + li_p;
          const int so = (((hh * 4 + dt * 2 + (g_p >> 1)) ^ isw(row)) << 4) + (g_p & 1) * 8;
          const f32x4 kv = dk[t][dt] * p.scale;
          *reinterpret_cast<u32x2*>(smem + BB_K + row * 128 + so) = (u32x2){pack_bf16x2(kv[0], kv[1]), pack_bf16x2(kv[2], kv[3])};
          *reinterpret_cast<u32x2*>(smem + BB_V + row * 128 + so) = (u32x2){pack_bf16x2(dv[t][dt][0], dv[t][dt][1]), pack_bf16x2(dv[t][dt][2], dv[t][dt][3])};
        }
    }
    // ---- dy1 += dq_pair Wq^T + dk_pair Wk^T + dv_pair Wv^T slices: six K = 32 steps; dq | dk | dv leave for HBM on the way
    BB_TR(7 + 12 * pr);
    bb_static_for<0, 6>([&](auto j_) {
      constexpr int j = decltype(j_)::value;
      constexpr int seq = 8 + 6 * pr + j;
      if constexpr (j <= 3) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else if constexpr (j == 4) { if (last) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)" ::: "memory"); }
      else { if (last) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(13) lgkmcnt(0)" ::: "memory"); }
      __builtin_amdgcn_s_barrier();
      BB_TR(8 + 12 * pr + j);
      if constexpr ((j & 1) == 0) stash(pr, j >> 1);
      if constexpr (j >= 1 && j + 2 < 6) wloadP(pr, j + 2);
      if constexpr (j + 2 >= 6 && !last) wloadP(pr + 1, j + 2 - 6);
      if constexpr (j == 3 && !last) { qkv_fetch(pr + 1); a_fetch(pr + 1); }
      kprod(j >> 1, j & 1, seq % 3, pr == 0 && j == 0);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    BB_TR(14 + 12 * pr);
    if constexpr (!last) {
      qkv_place();
      BB_TR(15 + 12 * pr);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  };
  bb_static_for<0, 4>(pair);

  // ---- dy1 (bf16) -> [128][512 B] image over the da image (slot ^ (row & 15)) -> whole 512-byte rows
#pragma unroll
  for (int ct = 0; ct < 8; ++ct)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = rp * 32 + rt * 16 + li, tl = (ct >> 2) * 8 + nh * 4 + (ct & 3);
      const u32x2 pk = {pack_bf16x2(acc2[ct][rt][0], acc2[ct][rt][1]), pack_bf16x2(acc2[ct][rt][2], acc2[ct][rt][3])};
      *reinterpret_cast<u32x2*>(smem + row * 512 + (((tl * 2 + (g >> 1)) ^ (row & 15)) << 4) + (g & 1) * 8) = pk;
    }
  __syncthreads();
  BB_TR(53);
  if (BB_ABL == 9) { asm volatile("s_dcache_wb" ::: "memory"); return; }
  if constexpr (LNB) {
    // ---- backward of LN1 on the tile (ln_bwd_tile.h): dy1 is read from its image, the partial sums go through the q image
    // (dead: the barrier behind the last product)
    const int c16 = lane >> 1, sub = (lane & 1) * 8;
    auto dy_of = [&](int r) { return *reinterpret_cast<const u32x2*>(smem + r * 512 + ((c16 ^ (r & 15)) << 4) + sub); };
    if constexpr (LNB == 2 && SDOC == 128) {      // (16-byte accesses: 8 columns per lane, two rows per wave instruction; -6 us per c2 step)
      u32x4 xhv[8];
      ln_tile_load_xh16(ln_tile_xh_rsrc(p.ln, p.T), row0, wave, lane, xhv);
      auto dy_of8 = [&](int r, int s5) { return *reinterpret_cast<const u32x4*>(smem + r * 512 + ((s5 ^ (r & 15)) << 4)); };
      ln_bwd_tile16<8>(p.ln, p.T, row0, blockIdx.x, wave, lane, tid, xhv, dy_of8, reinterpret_cast<float*>(smem + BB_Q));
    } else if constexpr (LNB == 2) {      // (S = 64: the 16-byte form cost this instance eight more spilled registers and 4 us per step)
      u32x2 xhv[16];
      ln_tile_load_xh(ln_tile_xh_rsrc(p.ln, p.T), row0, wave, lane, xhv);
      ln_bwd_tile(p.ln, p.T, row0, blockIdx.x, wave, lane, tid, xhv, dy_of, reinterpret_cast<float*>(smem + BB_Q));
    } else {
      f32x4 xv[16];
      ln_tile_load_x(ln_tile_x_rsrc(p.ln, p.T), row0, wave, lane, xv);
      ln_bwd_tile(p.ln, p.T, row0, blockIdx.x, wave, lane, tid, xv, dy_of, reinterpret_cast<float*>(smem + BB_Q));
    }
    return;
  }
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
  MFP_CHECK_ARG(B > 0 && B <= 16384 && (S == BB_ROWS || (S == 64 && B % 2 == 0)) && D == BB_D && H == 8);
  MFP_CHECK_ARG(((uintptr_t)d_o1 % 16) == 0 && ((uintptr_t)Wot % 16) == 0 && ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)a % 16) == 0 &&
                ((uintptr_t)Wqkvt % 16) == 0 && ((uintptr_t)dqkv % 16) == 0 && ((uintptr_t)dy1 % 16) == 0);
  AttnBwdBlockParams p = {};
  p.d_o1 = reinterpret_cast<const unsigned short*>(d_o1); p.Wot = reinterpret_cast<const unsigned short*>(Wot);
  p.qkv = reinterpret_cast<const unsigned short*>(qkv); p.a = reinterpret_cast<const unsigned short*>(a);
  p.lse = lse; p.nvalid = nvalid; p.Wqkvt = reinterpret_cast<const unsigned short*>(Wqkvt);
  p.dqkv = reinterpret_cast<unsigned short*>(dqkv); p.dy1 = reinterpret_cast<unsigned short*>(dy1);
  p.T = B * S; p.H = H; p.scale = 1.0f / sqrtf(32.0f);
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_bwd_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_bwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS);
    if (e != hipSuccess) {
      mfp_set_error("mfp_attn_block_bwd: cannot raise dynamic LDS to %d: %s", BB_LDS, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  if (S == 64) hipLaunchKernelGGL(attn_block_bwd_kernel<64>, dim3(B / 2), dim3(512), BB_LDS, reinterpret_cast<hipStream_t>(stream), p);
  else hipLaunchKernelGGL(attn_block_bwd_kernel<128>, dim3(B), dim3(512), BB_LDS, reinterpret_cast<hipStream_t>(stream), p);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_attn_block_bwd_ln(const void* d_o1, const void* Wot, const void* qkv, const void* a, const float* lse,
                                     const int32_t* nvalid, const void* Wqkvt, void* dqkv, const float* x, const void* xhat, const float* gamma,
                                     const float* mean, const float* rstd, const void* dres, void* dx, void* ddrop, float* part,
                                     size_t part_bytes, int32_t B, int32_t S, int32_t D, int32_t H, float drop_p, uint64_t seed,
                                     uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream) {
  MFP_CHECK_ARG(d_o1 && Wot && qkv && a && lse && nvalid && Wqkvt && dqkv && gamma && rstd && dres && dx && part);
  MFP_CHECK_ARG((xhat != nullptr || (x != nullptr && mean != nullptr)) && ((uintptr_t)xhat % 16) == 0);
  MFP_CHECK_ARG(B > 0 && B <= 16384 && (S == BB_ROWS || (S == 64 && B % 2 == 0)) && D == BB_D && H == 8 && drop_p >= 0.f && drop_p < 1.f);
  MFP_CHECK_ARG(((uintptr_t)d_o1 % 16) == 0 && ((uintptr_t)Wot % 16) == 0 && ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)a % 16) == 0 &&
                ((uintptr_t)Wqkvt % 16) == 0 && ((uintptr_t)dqkv % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)gamma % 16) == 0 &&
                ((uintptr_t)dres % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)ddrop % 16) == 0);
  const int T = B * S;
  if (part_bytes < (size_t)(T / BB_ROWS) * 3 * BB_D * sizeof(float)) {
    mfp_set_error("mfp_attn_block_bwd_ln: partial-sum buffer too small");
    return MFP_EWORKSPACE;
  }
  AttnBwdBlockParams p = {};
  p.d_o1 = reinterpret_cast<const unsigned short*>(d_o1); p.Wot = reinterpret_cast<const unsigned short*>(Wot);
  p.qkv = reinterpret_cast<const unsigned short*>(qkv); p.a = reinterpret_cast<const unsigned short*>(a);
  p.lse = lse; p.nvalid = nvalid; p.Wqkvt = reinterpret_cast<const unsigned short*>(Wqkvt);
  p.dqkv = reinterpret_cast<unsigned short*>(dqkv); p.dy1 = nullptr;
  p.T = T; p.H = H; p.scale = 1.0f / sqrtf(32.0f);
  p.ln.x = x; p.ln.xhat = reinterpret_cast<const unsigned short*>(xhat); p.ln.gamma = gamma; p.ln.mean = mean; p.ln.rstd = rstd;
  p.ln.dres = reinterpret_cast<const unsigned short*>(dres);
  p.ln.dx = reinterpret_cast<unsigned short*>(dx); p.ln.ddrop = reinterpret_cast<unsigned short*>(ddrop); p.ln.part = part;
  p.ln.drop_p = drop_p; p.ln.seed = seed; p.ln.offset = offset; p.ln.step_ptr = step_ptr;
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_bwd_kernel<128, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_bwd_kernel<64, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_bwd_kernel<128, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_bwd_kernel<64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS);
    if (e != hipSuccess) {
      mfp_set_error("mfp_attn_block_bwd_ln: cannot raise dynamic LDS to %d: %s", BB_LDS, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (xhat != nullptr) {
    if (S == 64) hipLaunchKernelGGL((attn_block_bwd_kernel<64, 2>), dim3(B / 2), dim3(512), BB_LDS, st, p);
    else hipLaunchKernelGGL((attn_block_bwd_kernel<128, 2>), dim3(B), dim3(512), BB_LDS, st, p);
  } else {
    if (S == 64) hipLaunchKernelGGL((attn_block_bwd_kernel<64, 1>), dim3(B / 2), dim3(512), BB_LDS, st, p);
    else hipLaunchKernelGGL((attn_block_bwd_kernel<128, 1>), dim3(B), dim3(512), BB_LDS, st, p);
  }
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}
