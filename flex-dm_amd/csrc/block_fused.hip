// Fused MLP half of a DeepSVG block, forward (reference architecture/transformer.py:222-225,161-171):
//
//     x2 = x1 + Dropout( relu( LN2(x1) W1 + b1 ) W2 + b2 )
//
// in ONE launch, for d_model 256.  Saved for the backward pass as before: y2 = LN2(x1) (bf16), mean /
// rstd, h = relu(.) (bf16).  Unfused this is three launches (ln_fwd 12 us + FFN1 16.5 us + FFN2 27 us at
// T = 32 768) moving 6.1 KB per element; fused it is 4.6 KB per element: y2 and h are written once and
// never read back, x1 is read twice.
//
// Shape of the kernel (activation-stationary, all 8 waves compute; one workgroup per 128 rows = one per CU at
// 256 documents x 128 elements):
//   * LN: wave w normalises rows 16 w .. + 15 in the MFMA operand layout (lane (li, g) holds row li's columns
//     32 ks + 8 g .. + 7, statistics by two xor-shuffles); the bf16 result passes through a swizzled LDS image,
//     from which y2 leaves in whole rows and every wave picks up the B-operand fragments of ITS two row tiles;
//   * products are issued transposed (D^T = W X^T: the weight tile is the MFMA A operand), so a lane ends up
//     with 4 CONSECUTIVE output columns of its own row: bias / ReLU / dropout / residual are 16-byte vector
//     work and h goes to LDS as the next product's operand rows;
//   * wave (rp, nh) owns rows 32 rp .. + 31 and half of the output columns of every chunk: each weight fragment
//     read from LDS feeds two MFMAs (with one row tile per wave the kernel sits at one ds_read_b128 per MFMA,
//     100 % of the LDS read port -- measured 19 us of product time instead of 11);
//   * the 512 KB of W1 | W2 stream from L2 straight into LDS (global_load_lds, 16 B per lane, source columns
//     pre-swizzled) in 16 chunks of 32 KB through three buffers: chunk c + 2 is issued when chunk c starts and
//     only chunk c + 1 is waited for at the barrier that ends chunk c (counted vmcnt; loads return in order);
//   * the hidden layer is processed in quarters of 128 units (h quarter = 32 KB of LDS): FFN1 (2 chunks) ->
//     FFN2 partial sums into 64 accumulator registers (2 chunks) -> next quarter; the x2 epilogue of each
//     column half runs inside the last quarter's FFN2 chunks.
// Measured at T = 32 768 (MI355X): 43 us against 56-58 us for the three launches.  Timeline of one workgroup
// (s_memtime): LN prologue 8.5 us (33 MB of x1 from HBM), 14 product chunks of 1.05 us (0.45 us of MFMA each;
// h and y2 leave in the background), last two chunks 13 us (x1 again + x2: 66 MB).  150 MB in 38 us = 3.9 TB/s
// average: the kernel is within 25 % of what HBM delivers for its byte count; every workgroup is in the same
// phase at the same time (one per CU), which is what keeps it from overlapping the prologue / epilogue traffic
// with the products.  A first attempt in round 1 kept 64 rows per workgroup and lost to the three launches.
#include <stdlib.h>

#include "common.h"
#include "ln_bwd_tile.h"
#include <type_traits>

namespace {

struct MlpParams {
  const float* x1; const float* gamma; const float* beta;
  const unsigned short* W1; const float* b1;      // [512][256] bf16 (out, in), f32 [512]
  const unsigned short* W2; const float* b2;      // [256][512] bf16 (out, in), f32 [256]
  unsigned short* y2; float* mean; float* rstd;   // saved LN output / statistics
  unsigned short* h;                              // [T][512] bf16
  float* x2;                                      // [T][256] f32
  unsigned short* x2c;                            // [T][256] bf16 copy of x2 (the decoder heads' operand) or nullptr
  int T; float eps;
  float dropout_p; unsigned long long seed, offset; const int* step_ptr;
#ifdef MFP_GEMM_TRACE
  unsigned long long* trace;   // [workgroup][24] s_memrealtime stamps (100 MHz) of thread 0
#endif
};

template <int I, int N, typename F>
__device__ __forceinline__ void wgg_free_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    wgg_free_static_for<I + 1, N>(f);
  }
}

constexpr int MLP_D = 256, MLP_F = 512, MLP_ROWS = 128;
// LDS images have 512- or 256-byte rows with the 16-byte slot index XORed with (row & 15): a ds_read_b128 of 16
// rows x 4 slots (one MFMA operand fragment) then touches every bank once, and the image stays lane-linear for
// the direct global -> LDS loads (the same involution is applied to the SOURCE column of each lane).
constexpr int MLP_HS_B = MLP_ROWS * 256;            // h quarter: 128 rows x 128 hidden units, 32 KB
constexpr int MLP_WS_B = 32768;                     // one weight chunk; three buffers in rotation
constexpr int MLP_B1_OFF = MLP_HS_B + 3 * MLP_WS_B; // b1 (2 KB) | b2 (1 KB) | ...
constexpr int MLP_LDS = MLP_B1_OFF + (MLP_F + 3 * MLP_D) * 4;   // ... | gamma (1 KB) | beta (1 KB)
constexpr int MLP_CHUNKS = 16;

typedef __attribute__((address_space(3))) unsigned char lds_u8;

// Chunk c (32 KB of weights), hidden quarter q = c >> 2:
//   c & 3 = 0, 1: W1 rows q*128 + 64 j .. + 63, all 256 k      -> image [64][512 B]   (FFN1 of the quarter)
//   c & 3 = 2, 3: W2 rows 128 j .. + 127, k = q*128 .. + 127    -> image [128][256 B]  (FFN2 partial sums)
// Wave (rp, nh) = (wave & 3, wave >> 2) owns rows 32 rp .. + 31 (two MFMA row tiles) and half of every chunk's
// output columns: each weight fragment read from LDS feeds two products (LDS reads would otherwise cap the kernel
// at one ds_read_b128 per MFMA: 100 % of the LDS read port).
template <bool DROPOUT>
__global__ __launch_bounds__(512) void mlp_fused_kernel(MlpParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Hs = smem;
  unsigned char* const Ws = smem + MLP_HS_B;
  const float* const B1s = reinterpret_cast<const float*>(smem + MLP_B1_OFF);
  const float* const B2s = B1s + MLP_F;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int rp = wave & 3, nh = wave >> 2;
  const int row0 = blockIdx.x * MLP_ROWS;
#ifdef MFP_GEMM_TRACE
#define MLP_STAMP(i) do { if (tid == 0 && p.trace) p.trace[(long long)blockIdx.x * 24 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MLP_STAMP(i) do {} while (0)
#endif
  MLP_STAMP(0);    // start
  // (the step counter is read first: its load must not sit between the counted waits of the chunk loop)
  const int step_now = (DROPOUT && p.step_ptr) ? __builtin_amdgcn_readfirstlane(*p.step_ptr) : 0;

  // every global access goes through a buffer descriptor (SGPRs) + a 32-bit offset: no 64-bit address pairs
  // in VGPRs (the kernel runs at the 256-register limit; a spill is a scratch access, and scratch accesses retire
  // through the same in-order counter as the weight loads), and rows >= T need no predication (reads return 0,
  // writes are dropped)
  const unsigned int xbytes = (unsigned int)p.T * (MLP_D * 4), hbytes = (unsigned int)p.T * (MLP_F * 2);
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.W1), 0, MLP_F * MLP_D * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.W2), 0, MLP_F * MLP_D * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x1), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(p.x2, 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_y2 = __builtin_amdgcn_make_buffer_rsrc(p.y2, 0, xbytes / 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(p.h, 0, hbytes, 0x00020000);
  // (no copy wanted: a zero-sized buffer -- the stores are issued all the same, so the counted waits do not depend on it)
  const __amdgpu_buffer_rsrc_t rs_x2c = __builtin_amdgcn_make_buffer_rsrc(p.x2c ? p.x2c : p.y2, 0, p.x2c ? xbytes / 2 : 0u, 0x00020000);

  // weight piece i of a chunk (1 KB per wave instruction): W1 chunks are 64 rows x 512 B (2 rows per piece), W2
  // chunks 128 rows x 256 B out of 1 KB rows (4 rows per piece); the source column slot is the destination
  // slot ^ (row & 15).  Piece i differs from piece 0 by a row step and one XOR on the slot
  const unsigned int w1off = (unsigned int)((wave * 8 + (lane >> 5)) * 512 + (((lane & 31) ^ ((wave & 1) * 8 + (lane >> 5))) << 4));
  const unsigned int w2off = (unsigned int)((wave * 16 + (lane >> 4)) * 1024 + (((lane & 15) ^ (lane >> 4)) << 4));
  auto wload = [&](int c) {
    const int q = c >> 2, ffn2 = (c >> 1) & 1, j = c & 1;
    unsigned char* dst = Ws + ((c + 1) % 3) * MLP_WS_B + wave * 4096;
    if (!ffn2) {
      const int base = (q * 128 + j * 64) * (MLP_D * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (lds_u8*)(dst + i * 1024), 16, w1off ^ (i << 5), base + i * 1024, 0, 0);
    } else {
      const int base = (j * 128) * (MLP_F * 2) + q * 256;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, (lds_u8*)(dst + i * 1024), 16, w2off ^ (i << 6), base + i * 4096, 0, 0);
    }
  };
  wload(0);
  wload(1);
  if (tid < (MLP_F + 3 * MLP_D) / 4) {     // per-column vectors -> LDS (ds_read latency instead of L2 latency at every use)
    const float* src = tid < 128 ? p.b1 + tid * 4 : tid < 192 ? p.b2 + (tid - 128) * 4
                     : tid < 256 ? p.gamma + (tid - 192) * 4 : p.beta + (tid - 256) * 4;
    *reinterpret_cast<f32x4*>(smem + MLP_B1_OFF + tid * 16) = *reinterpret_cast<const f32x4*>(src);
  }
  const float* const Gs = B2s + MLP_D;
  const float* const Bs = Gs + MLP_D;

  // ---- LayerNorm: wave w normalises rows 16 w .. + 15 in the MFMA operand layout (lane (li, g) holds row li,
  // columns 32 ks + 8 g .. + 7: statistics by two xor-shuffles), and the bf16 result goes through a [128][512 B]
  // LDS image (Hs + the weight buffer chunk 2 will use) so that every wave can pick up the fragments of ITS
  // two row tiles and y2 leaves in whole rows
  bf16x8 xf[2][8];
  {
    const int lrow = wave * 16 + li, row = row0 + lrow;
    const bool rok = row < p.T;
    float v[8][8];
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const unsigned int vo = (unsigned int)row * (MLP_D * 4) + g * 32;
      const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x1, vo + ks * 128, 0, 0));
      const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x1, vo + ks * 128 + 16, 0, 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[ks][e] = a[e]; v[ks][4 + e] = b[e]; s += a[e] + b[e]; }
    }
    __syncthreads();      // gamma / beta are in LDS
    s += lane_xor16(s);
    s += lane_xor32(s);
    const float mu = s * (1.0f / MLP_D);
    float qq = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[ks][e] -= mu; qq += v[ks][e] * v[ks][e]; }
    qq += lane_xor16(qq);
    qq += lane_xor32(qq);
    const float rs = rsqrtf(qq * (1.0f / MLP_D) + p.eps);
    if (g == 0 && rok) { p.mean[row] = mu; p.rstd[row] = rs; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int col = ks * 32 + 8 * g;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(Gs + col), g1 = *reinterpret_cast<const f32x4*>(Gs + col + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + col), b1 = *reinterpret_cast<const f32x4*>(Bs + col + 4);
      float y[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { y[e] = v[ks][e] * rs * g0[e] + b0[e]; y[4 + e] = v[ks][4 + e] * rs * g1[e] + b1[e]; }
      const u32x4 pk = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7])};
      *reinterpret_cast<u32x4*>(smem + lrow * 512 + (((ks * 4 + g) ^ li) << 4)) = pk;
    }
  }
  __syncthreads();
  // y2 leaves in whole rows, straight from the image.  No wait for these stores before the barrier below: the
  // weight loads of chunks 0 and 1 were the first memory operations of the kernel and every x1 load issued
  // after them has been consumed, so (in-order retirement) they have landed
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 512 * i, r = idx >> 5, c16 = idx & 31;
    const u32x4 yv = *reinterpret_cast<const u32x4*>(smem + r * 512 + ((c16 ^ (r & 15)) << 4));
    __builtin_amdgcn_raw_buffer_store_b128(yv, rs_y2, (unsigned int)(row0 + r) * (MLP_D * 2) + c16 * 16, 0, 0);
  }
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      xf[rt][ks] = *reinterpret_cast<const bf16x8*>(smem + (rp * 32 + rt * 16 + li) * 512 + (((ks * 4 + g) ^ li) << 4));
  MLP_STAMP(1);  // LN done, fragments picked up
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();       // chunks 0, 1 are in LDS; the y2 image has been read by everyone
  MLP_STAMP(2);

  f32x4 acc2[8][2];      // first written by the first quarter's FFN2 chunks (not live before)
  bf16x8 hf[2][4];
  const float inv_keep = DROPOUT ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  const unsigned long long rng_off = p.offset + (unsigned long long)step_now * MFP_RNG_STEP_STRIDE;
  const unsigned int dthr = drop_thr16(p.dropout_p), dkey = drop_key(p.seed, rng_off);
  // slot term of a fragment address: (4 ks + g) ^ li, in bytes (ks < 4; ks >= 4 adds 256 in the 512-byte images)
  int xs[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xs[ks] = ((ks * 4 + g) ^ li) << 4;

  auto chunk = [&](auto cc_) {
    constexpr int c = decltype(cc_)::value;
    constexpr int q = c >> 2, ffn2 = (c >> 1) & 1, j = c & 1;
    if (c + 2 < MLP_CHUNKS) wload(c + 2);    // into the buffer the barrier that ended chunk c - 1 released
    f32x4 res[4][2];
    if (q == 3 && ffn2) {
      // residual rows for this chunk's epilogue, ahead of the stores below: a wait for a load also waits for
      // every older store
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const int row = row0 + rp * 32 + rt * 16 + li;
          res[nt][rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
              rs_x1, (unsigned int)row * (MLP_D * 4) + (nh * 4 * 16 + 4 * g) * 4 + (j * 8 + nt) * 64, 0, 0));
        }
    }
    const unsigned char* wb = Ws + ((c + 1) % 3) * MLP_WS_B;
    if (!ffn2) {
      // FFN1: h[row][q*128 + j*64 + (2 nh + nt)*16 + 4 g + r] = relu(sum_k W1[n][k] y2[row][k] + b1[n])
      const unsigned char* wa = wb + ((nh * 2) * 16 + li) * 512;
      f32x4 acc[2][2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[nt][0] = acc[nt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      bf16x8 wf[3][2];
#pragma unroll
      for (int pre = 0; pre < 2; ++pre)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) wf[pre][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[pre & 3] + (pre >> 2) * 256);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 2 < 8) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            wf[(ks + 2) % 3][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[(ks + 2) & 3] + ((ks + 2) >> 2) * 256);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
            acc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks % 3][nt], xf[rt][ks], acc[nt][rt], 0, 0, 0);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(B1s + q * 128 + j * 64 + (nh * 2 + nt) * 16 + 4 * g);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const u32x2 pk = {pack_bf16x2(fmaxf(acc[nt][rt][0] + bb[0], 0.f), fmaxf(acc[nt][rt][1] + bb[1], 0.f)),
                            pack_bf16x2(fmaxf(acc[nt][rt][2] + bb[2], 0.f), fmaxf(acc[nt][rt][3] + bb[3], 0.f))};
          *reinterpret_cast<u32x2*>(Hs + (rp * 32 + rt * 16 + li) * 256 + (((j * 8 + (nh * 2 + nt) * 2 + (g >> 1)) ^ li) << 4) + (g & 1) * 8) = pk;
        }
      }
    } else {
      // FFN2 partial sums over this quarter of the hidden units: column (8 j + 4 nh + nt)*16 + 4 g + r.
      // Last quarter: these columns are final after this chunk -- their epilogue (x2 = x1 + dropout(. + b2),
      // 4 consecutive columns per lane and tile) runs here, so half of the residual reads and x2 stores overlap
      // the last chunk's products
      constexpr bool last = q == 3;
      const unsigned char* wa = wb + ((nh * 4) * 16 + li) * 256;
      bf16x8 wf[2][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) wf[0][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 4096 + xs[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) wf[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 4096 + xs[ks + 1]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
            acc2[j * 4 + nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][nt], hf[rt][ks],
                                                                         (q == 0 && ks == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc2[j * 4 + nt][rt], 0, 0, 0);
      }
      if (last) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const int row = row0 + rp * 32 + rt * 16 + li;
          const unsigned int rowh = drop_row(dkey, (unsigned int)row);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const int n = (j * 8 + nh * 4 + nt) * 16 + 4 * g;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(B2s + n);
            bool keep[4] = {true, true, true, true};
            if (DROPOUT) drop_keep4(rowh, (unsigned int)n, dthr, keep);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = res[nt][rt][r] + (keep[r] ? (acc2[j * 4 + nt][rt][r] + bb[r]) * inv_keep : 0.f);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_x2,
                                                   (unsigned int)row * (MLP_D * 4) + (nh * 4 * 16 + 4 * g) * 4 + (j * 8 + nt) * 64, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])}, rs_x2c,
                                                  (unsigned int)row * (MLP_D * 2) + (nh * 4 * 16 + 4 * g) * 2 + (j * 8 + nt) * 32, 0, 0);
          }
        }
      }
    }
    // chunk c + 1 has landed: memory operations retire in order, so it is enough that no more operations are
    // outstanding than were issued AFTER its loads.  The h stores (4, issued behind the barrier of chunks 1, 5, 9,
    // 13) are younger than the loads of the chunk two ahead: the wait that includes them comes two chunks later
    // (a store acknowledged late by a busy memory system otherwise holds every wave at the barrier -- measured:
    // chunks behind a store burst took 3-6 us instead of 1).  This wave's LDS writes are done (lgkmcnt), and
    // after the barrier chunk c's buffer is free again
    {
      constexpr int st_prev = (c & 3) == 2 ? 4 : 0;                       // h stores of the previous chunk's tail
      constexpr int st_this = c == 0 ? 10 : c == 14 ? 24 : 0;             // prologue stores (chunk 1 landed long ago); residual loads + x2 stores (f32 + bf16 copy)
      constexpr int allowed = st_prev + (c + 2 < MLP_CHUNKS ? 4 : 0) + st_this;
      if (c + 1 < MLP_CHUNKS) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(allowed) : "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    MLP_STAMP(3 + c);   // chunk c done
    if (!ffn2 && j == 1) {
      // the h quarter is complete in LDS: write it out (256-byte row pieces, 16 B per lane) and pick up this
      // wave's rows as the next product's operand fragments
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + 512 * i, r = idx >> 4, c16 = idx & 15;
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(Hs + r * 256 + ((c16 ^ (r & 15)) << 4)), rs_h,
                                               (unsigned int)(row0 + r) * (MLP_F * 2) + c16 * 16 + q * 256, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          hf[rt][ks] = *reinterpret_cast<const bf16x8*>(Hs + (rp * 32 + rt * 16 + li) * 256 + xs[ks]);
    }
    // (the next quarter's first FFN1 chunk rewrites Hs: two barriers -- the ends of this quarter's FFN2 chunks --
    //  lie between every wave's hf reads above and those writes)
  };
  chunk(std::integral_constant<int, 0>{});  chunk(std::integral_constant<int, 1>{});
  chunk(std::integral_constant<int, 2>{});  chunk(std::integral_constant<int, 3>{});
  chunk(std::integral_constant<int, 4>{});  chunk(std::integral_constant<int, 5>{});
  chunk(std::integral_constant<int, 6>{});  chunk(std::integral_constant<int, 7>{});
  chunk(std::integral_constant<int, 8>{});  chunk(std::integral_constant<int, 9>{});
  chunk(std::integral_constant<int, 10>{}); chunk(std::integral_constant<int, 11>{});
  chunk(std::integral_constant<int, 12>{}); chunk(std::integral_constant<int, 13>{});
  chunk(std::integral_constant<int, 14>{}); chunk(std::integral_constant<int, 15>{});
  MLP_STAMP(19);
}

// ---------------------------------------------------------------------------------------------------------
// LayerNormalization + the fused Q | K | V Dense of a block in one launch (reference transformer.py:216-217,
// 85-90):  qkv = LN1(x) Wqkv^T + bqkv  (bf16 [T][768]), saving y1 = LN1(x) (bf16), mean, rstd for the backward
// pass.  The forward MLP kernel's machine without its second product: LN prologue through the swizzled LDS image,
// 12 weight chunks of 64 output columns through three buffers, and the bf16 result of every chunk leaving through
// one of two small LDS images in 128-byte row pieces while the next chunk multiplies.
// Unfused: ln_fwd 11.6 us + product 21.5 us, 115 MB; fused 99 MB.
struct QkvParams {
  const float* x; const float* gamma; const float* beta;
  const unsigned short* W; const float* bias;          // [768][256] bf16 (out, in), f32 [768]
  unsigned short* y1; float* mean; float* rstd; unsigned short* qkv;
  int T; float eps;
};

constexpr int QKV_N = 768, QKV_CHUNKS = QKV_N / 64;
constexpr int QKV_VEC_OFF = MLP_HS_B + 3 * MLP_WS_B;                 // bias (3 KB) | gamma (1 KB) | beta (1 KB)
constexpr int QKV_LDS = QKV_VEC_OFF + (QKV_N + 2 * MLP_D) * 4;

__global__ __launch_bounds__(512) void qkv_fused_kernel(QkvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Os = smem;                   // two [128][128 B] output images (16 KB each); the LN image before
  unsigned char* const Ws = smem + MLP_HS_B;
  const float* const Bq = reinterpret_cast<const float*>(smem + QKV_VEC_OFF);
  const float* const Gs = Bq + QKV_N;
  const float* const Bs = Gs + MLP_D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int rp = wave & 3, nh = wave >> 2;
  const int row0 = blockIdx.x * MLP_ROWS;
  const unsigned int xbytes = (unsigned int)p.T * (MLP_D * 4);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.W), 0, QKV_N * MLP_D * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y1, 0, xbytes / 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(p.qkv, 0, (unsigned int)p.T * (QKV_N * 2), 0x00020000);

  const unsigned int w1off = (unsigned int)((wave * 8 + (lane >> 5)) * 512 + (((lane & 31) ^ ((wave & 1) * 8 + (lane >> 5))) << 4));
  auto wload = [&](int c) {
    unsigned char* dst = Ws + ((c + 1) % 3) * MLP_WS_B + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_u8*)(dst + i * 1024), 16, w1off ^ (i << 5), c * 64 * (MLP_D * 2) + i * 1024, 0, 0);
  };
  wload(0);
  wload(1);
  if (tid < (QKV_N + 2 * MLP_D) / 4) {
    const float* src = tid < 192 ? p.bias + tid * 4 : tid < 256 ? p.gamma + (tid - 192) * 4 : p.beta + (tid - 256) * 4;
    *reinterpret_cast<f32x4*>(smem + QKV_VEC_OFF + tid * 16) = *reinterpret_cast<const f32x4*>(src);
  }
  bf16x8 xf[2][8];
  {
    const int lrow = wave * 16 + li, row = row0 + lrow;
    const bool rok = row < p.T;
    float v[8][8];
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const unsigned int vo = (unsigned int)row * (MLP_D * 4) + g * 32;
      const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo + ks * 128, 0, 0));
      const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo + ks * 128 + 16, 0, 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[ks][e] = a[e]; v[ks][4 + e] = b[e]; s += a[e] + b[e]; }
    }
    __syncthreads();      // gamma / beta are in LDS
    s += lane_xor16(s);
    s += lane_xor32(s);
    const float mu = s * (1.0f / MLP_D);
    float qq = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[ks][e] -= mu; qq += v[ks][e] * v[ks][e]; }
    qq += lane_xor16(qq);
    qq += lane_xor32(qq);
    const float rs = rsqrtf(qq * (1.0f / MLP_D) + p.eps);
    if (g == 0 && rok) { p.mean[row] = mu; p.rstd[row] = rs; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int col = ks * 32 + 8 * g;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(Gs + col), g1 = *reinterpret_cast<const f32x4*>(Gs + col + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + col), b1 = *reinterpret_cast<const f32x4*>(Bs + col + 4);
      float y[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { y[e] = v[ks][e] * rs * g0[e] + b0[e]; y[4 + e] = v[ks][4 + e] * rs * g1[e] + b1[e]; }
      const u32x4 pk = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7])};
      *reinterpret_cast<u32x4*>(smem + lrow * 512 + (((ks * 4 + g) ^ li) << 4)) = pk;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 512 * i, r = idx >> 5, c16 = idx & 31;
    const u32x4 yv = *reinterpret_cast<const u32x4*>(smem + r * 512 + ((c16 ^ (r & 15)) << 4));
    __builtin_amdgcn_raw_buffer_store_b128(yv, rs_y, (unsigned int)(row0 + r) * (MLP_D * 2) + c16 * 16, 0, 0);
  }
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      xf[rt][ks] = *reinterpret_cast<const bf16x8*>(smem + (rp * 32 + rt * 16 + li) * 512 + (((ks * 4 + g) ^ li) << 4));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();       // chunks 0, 1 are in LDS (first memory operations of the kernel); the LN image has been read

  int xs[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xs[ks] = ((ks * 4 + g) ^ li) << 4;
  // the output image of chunk c (columns 64 c .. + 63 of 128 rows, bf16): 128-byte rows, 16-byte slot ^ (row & 7)
  auto out_store = [&](int c) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i, r = idx >> 3, c16 = idx & 7;
      __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(Os + (c & 1) * 16384 + r * 128 + ((c16 ^ (r & 7)) << 4)), rs_q,
                                             (unsigned int)(row0 + r) * (QKV_N * 2) + c * 128 + c16 * 16, 0, 0);
    }
  };
  auto chunk = [&](auto cc_) {
    constexpr int c = decltype(cc_)::value;
    if (c + 2 < QKV_CHUNKS) wload(c + 2);
    if (c >= 1) out_store(c - 1);          // behind the barrier that ended chunk c - 1; AFTER the weight loads (counted waits)
    const unsigned char* wa = Ws + ((c + 1) % 3) * MLP_WS_B + ((nh * 2) * 16 + li) * 512;
    f32x4 acc[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[nt][0] = acc[nt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[3][2];
#pragma unroll
    for (int pre = 0; pre < 2; ++pre)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) wf[pre][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[pre & 3] + (pre >> 2) * 256);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 2 < 8) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          wf[(ks + 2) % 3][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[(ks + 2) & 3] + ((ks + 2) >> 2) * 256);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          acc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks % 3][nt], xf[rt][ks], acc[nt][rt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(Bq + c * 64 + (nh * 2 + nt) * 16 + 4 * g);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const u32x2 pk = {pack_bf16x2(acc[nt][rt][0] + bb[0], acc[nt][rt][1] + bb[1]), pack_bf16x2(acc[nt][rt][2] + bb[2], acc[nt][rt][3] + bb[3])};
        *reinterpret_cast<u32x2*>(Os + (c & 1) * 16384 + (rp * 32 + rt * 16 + li) * 128 + ((((nh * 2 + nt) * 2 + (g >> 1)) ^ (li & 7)) << 4) + (g & 1) * 8) = pk;
      }
    }
    // chunk c + 1 has landed when no more operations are outstanding than were issued after its loads: the image
    // stores of the previous and of this chunk head (2 each), the 4 loads of chunk c + 2; chunk 0: the 10 stores
    // of the prologue are older than nothing that is needed (chunk 1 landed there)
    {
      constexpr int allowed = c == 0 ? 14 : (c >= 2 ? 2 : 0) + (c + 2 < QKV_CHUNKS ? 4 : 0) + 2;
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(allowed) : "memory");
      __builtin_amdgcn_s_barrier();
    }
  };
  chunk(std::integral_constant<int, 0>{});  chunk(std::integral_constant<int, 1>{});
  chunk(std::integral_constant<int, 2>{});  chunk(std::integral_constant<int, 3>{});
  chunk(std::integral_constant<int, 4>{});  chunk(std::integral_constant<int, 5>{});
  chunk(std::integral_constant<int, 6>{});  chunk(std::integral_constant<int, 7>{});
  chunk(std::integral_constant<int, 8>{});  chunk(std::integral_constant<int, 9>{});
  chunk(std::integral_constant<int, 10>{}); chunk(std::integral_constant<int, 11>{});
  out_store(QKV_CHUNKS - 1);
}

// ---------------------------------------------------------------------------------------------------------
// Backward of the MLP half, input gradients (reference: Keras autodiff of transformer.py:161-171,224-225):
//
//     dh  = (d_o2 W2) * [h > 0]           d_o2 = dropout-masked gradient of the block output, bf16 [T,256]
//     dy2 = dh W1                          (gradient of LN2's output, consumed by ln_bwd)
//
// The same machine as the forward kernel with W2^T [512][256] in the place of W1 and W1^T [256][512] in the
// place of W2 (the transposed bf16 shadows): d_o2 fragments come straight from global memory (no LayerNorm),
// the ReLU mask comes from the saved h, whose quarter is loaded into its own LDS image two chunks ahead (every lane
// reads the 8 bytes that match the 4 values it holds), and dy2 leaves as bf16 through LDS in whole rows.
// Unfused: two launches, 23 + 17.5 us, 131 MB; fused 98 MB.
struct MlpBwdParams {
  const unsigned short* d_o2;      // [T][256] bf16
  const unsigned short* h;         // [T][512] bf16 (saved by the forward pass)
  const unsigned short* W2t;       // [512][256] bf16: W2t[k][n] = W2[n][k]
  const unsigned short* W1t;       // [256][512] bf16: W1t[c][k] = W1[k][c]
  unsigned short* dh;              // [T][512] bf16
  unsigned short* dy2;             // [T][256] bf16
  int T;
#ifdef MFP_GEMM_TRACE
  unsigned long long* trace;
#endif
  LnTileArgs ln;                   // LNB form (mfp_mlp_bwd_ln): the backward of LN2 in the epilogue -- dy2 never leaves the CU
};

// LNB: 0 = dy2 leaves as bf16 rows; 1 = LN2 backward in the epilogue from x1 (f32); 2 = from the x-hat stash (bf16)
// HT (round 5): HALF tiles -- two workgroups per 128-row tile, 64 rows each, a wave owns ONE 16-row tile (wave = (rp, nh, rtw): the
// two waves of a SIMD hold the two row tiles of a row pair), for batches with fewer 128-row tiles than CUs (BASELINE config c4);
// the same per-row arithmetic, per-thread store counts halved in the counted waits, the LayerNorm-backward epilogue walks 8 rows
// per wave and leaves one partial row per HALF tile (the csrc/block_attn.hip HALF == 2 form of this machine)
template <int LNB, bool HT = false>
__global__ __launch_bounds__(512) void mlp_bwd_kernel(MlpBwdParams p) {
  static_assert(!HT || LNB == 2, "half tiles: the x-hat form");
  constexpr int RT = HT ? 1 : 2;                 // row tiles per wave
  constexpr int ROWS = HT ? 64 : MLP_ROWS;
  constexpr int SI = ROWS * 16 / 512;            // 16-byte pieces per thread of a [ROWS][256 B] image
  constexpr int HP = HT ? 4 : 8;                 // 1 KB pieces per h wave and quarter
  constexpr int NR = ROWS / 8;                   // rows per wave in the LayerNorm-backward epilogue
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Hs = smem;
  unsigned char* const Ws = smem + MLP_HS_B;
  unsigned char* const Hm = smem + MLP_HS_B + 3 * MLP_WS_B;      // saved h of the quarter (ReLU mask), own 32 KB image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int rp = HT ? (wave & 1) : (wave & 3), nh = HT ? ((wave >> 1) & 1) : (wave >> 2);
  const int lb = rp * 32 + (HT ? (wave >> 2) * 16 : 0);      // first local row of this wave's row tile(s)
  const int row0 = blockIdx.x * ROWS;
  const unsigned int xbytes = (unsigned int)p.T * (MLP_D * 2), hbytes = (unsigned int)p.T * (MLP_F * 2);
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.W2t), 0, MLP_F * MLP_D * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.W1t), 0, MLP_F * MLP_D * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_do = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.d_o2), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.h), 0, hbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dh = __builtin_amdgcn_make_buffer_rsrc(p.dh, 0, hbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(p.dy2, 0, xbytes, 0x00020000);

  // Memory roles are split by latency class, because a wave's memory operations retire in order: waves 0-3 issue
  // ALL weight pieces (L2 hits, needed every chunk, counted waits), waves 4-7 issue all pieces of the saved h (first
  // touched since the forward pass: HBM latency under a burst of 8 MB per quarter across the chip).  With both in
  // one queue the weight loads queue up behind the h loads (measured: the chunks behind an h load took 2-4 us)
  const int wv = __builtin_amdgcn_readfirstlane(wave);       // scalar: the role branches below are uniform
  const int wl = wv & 3;
  // weight piece i (0..7) of this wave: W "1" chunks are 64 rows x 512 B (2 rows per 1 KB piece), W "2" chunks 128
  // rows x 256 B out of 1 KB rows (4 rows per piece); source column slot = destination slot ^ (row & 15)
  const unsigned int w1off = (unsigned int)((wl * 16 + (lane >> 5)) * 512 + (((lane & 31) ^ (lane >> 5)) << 4));
  const unsigned int w2off = (unsigned int)((wl * 32 + (lane >> 4)) * 1024 + (((lane & 15) ^ (lane >> 4)) << 4));
  auto wload = [&](int c) {
    if (wv >= 4) return;
    const int q = c >> 2, ffn2 = (c >> 1) & 1, j = c & 1;
    unsigned char* dst = Ws + ((c + 1) % 3) * MLP_WS_B + wl * 8192;
    if (!ffn2) {
      const int base = (q * 128 + j * 64) * (MLP_D * 2);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (lds_u8*)(dst + i * 1024), 16, w1off ^ (i << 5), base + i * 1024, 0, 0);
    } else {
      const int base = (j * 128) * (MLP_F * 2) + q * 256;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, (lds_u8*)(dst + i * 1024), 16, w2off ^ ((i & 3) << 6), base + i * 4096, 0, 0);
    }
  };
  // h quarter q -> image [128][256 B]: the row / slot pattern of a W "2" chunk, on the 1 KB rows of h
  const unsigned int hoff = (unsigned int)((row0 + wl * (HP * 4) + (lane >> 4)) * 1024 + (((lane & 15) ^ (lane >> 4)) << 4));
  auto hload = [&](int q) {
    if (wv < 4) return;
#ifndef MLPB_ABL_HP      // (timing experiment, tools/abl: how much of the launch is the saved h's read -- 1 = an eighth of its pieces; results wrong)
#define MLPB_ABL_HP HP
#endif
#pragma unroll
    for (int i = 0; i < MLPB_ABL_HP; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_h, (lds_u8*)(Hm + wl * (HP * 1024) + i * 1024), 16, hoff ^ ((i & 3) << 6), q * 256 + i * 4096, 0, 0);
  };
  MLP_STAMP(0);
  wload(0);
  wload(1);
  // operand fragments of d_o2: lane (li, g) holds row li of the tile, columns 32 ks + 8 g .. + 7
  bf16x8 xf[2][8];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      xf[rt][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
          rs_do, (unsigned int)(row0 + lb + rt * 16 + li) * (MLP_D * 2) + g * 16 + ks * 64, 0, 0));
  // the first h quarter is needed at the END of chunk 0 (its epilogue) only: it stays in flight across this
  // barrier (first touched since the forward pass: HBM latency) and is waited for in front of that epilogue
  hload(0);
  MLP_STAMP(1);
  if (wv < 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * RT) : "memory");      // chunks 0, 1 landed (older than the 8 RT fragment loads)
  __builtin_amdgcn_s_barrier();
  MLP_STAMP(2);

  f32x4 acc2[8][2];
  bf16x8 hf[2][4];
  int xs[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xs[ks] = ((ks * 4 + g) ^ li) << 4;

  // LNB: the x1 rows of the LN2-backward epilogue (wave w: rows 16 w .. + 15, a lane 4 columns) are requested at the head of
  // chunk 14 -- the d_o2 fragments are dead by then, the 64 registers change hands -- and cross the last two chunks in flight
  constexpr int LN_PF = LNB == 2 ? NR / 2 : LNB ? NR : 0;      // (x-hat form: 16-byte loads, two rows per wave instruction)
  f32x4 xv[NR];      // (dead registers in the other forms)
  u32x4 xhv[NR / 2];
  const __amdgpu_buffer_rsrc_t rs_x = LNB == 2 ? ln_tile_xh_rsrc(p.ln, p.T) : ln_tile_x_rsrc(p.ln, LNB ? p.T : 0);
  auto chunk = [&](auto cc_) {
    constexpr int c = decltype(cc_)::value;
    constexpr int q = c >> 2, ffn2 = (c >> 1) & 1, j = c & 1;
    if constexpr (LNB == 1 && c == MLP_CHUNKS - 2) ln_tile_load_x(rs_x, row0, wv, lane, xv);
    if constexpr (LNB == 2 && c == MLP_CHUNKS - 2) ln_tile_load_xh16(rs_x, row0, wv, lane, xhv);
    // the next quarter's h goes into its image while the two dy2 chunks of this quarter run (the masks of this
    // quarter were read before the barrier that ended the previous chunk); waited for at the end of the next chunk
    if (ffn2 && j == 0 && q < 3) hload(q + 1);
    if (c + 2 < MLP_CHUNKS) wload(c + 2);
    const unsigned char* wb = Ws + ((c + 1) % 3) * MLP_WS_B;
    if (!ffn2) {
      // dh[row][q*128 + j*64 + (2 nh + nt)*16 + 4 g + r] = [h > 0] * sum_n W2t[k][n] d_o2[row][n]
      const unsigned char* wa = wb + ((nh * 2) * 16 + li) * 512;
      f32x4 acc[2][2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[nt][0] = acc[nt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      bf16x8 wf[2][2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) wf[0][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[0]);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            wf[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[(ks + 1) & 3] + ((ks + 1) >> 2) * 256);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][nt], xf[rt][ks], acc[nt][rt], 0, 0, 0);
      }
      if (c == 0) {      // the first h quarter (waves 4-7 issued it in the prologue) must be in LDS now
        if (wv >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int slot = (lb + rt * 16 + li) * 256 + (((j * 8 + (nh * 2 + nt) * 2 + (g >> 1)) ^ li) << 4) + (g & 1) * 8;
          const u32x2 hv = *reinterpret_cast<const u32x2*>(Hm + slot);      // 4 bf16 of h: positive <=> nonzero, sign clear
          const bool m0 = (short)(hv[0] & 0xFFFFu) > 0, m1 = (int)hv[0] >= 0x10000;
          const bool m2 = (short)(hv[1] & 0xFFFFu) > 0, m3 = (int)hv[1] >= 0x10000;
          const u32x2 pk = {pack_bf16x2(m0 ? acc[nt][rt][0] : 0.f, m1 ? acc[nt][rt][1] : 0.f),
                            pack_bf16x2(m2 ? acc[nt][rt][2] : 0.f, m3 ? acc[nt][rt][3] : 0.f)};
          *reinterpret_cast<u32x2*>(Hs + slot) = pk;
        }
    } else {
      const unsigned char* wa = wb + ((nh * 4) * 16 + li) * 256;
      bf16x8 wf[2][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) wf[0][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 4096 + xs[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) wf[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 4096 + xs[ks + 1]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc2[j * 4 + nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][nt], hf[rt][ks],
                                                                         (q == 0 && ks == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc2[j * 4 + nt][rt], 0, 0, 0);
      }
    }
    if (c >= MLP_CHUNKS - 2) {
      // dy2 columns 128 j .. + 127 are final: bf16 into a free [128][256 B] image -- the weight buffer of the
      // chunk before this one (every wave finished reading it before the barrier that ended that chunk)
      unsigned char* img = Ws + (c % 3) * MLP_WS_B;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const u32x2 pk = {pack_bf16x2(acc2[j * 4 + nt][rt][0], acc2[j * 4 + nt][rt][1]), pack_bf16x2(acc2[j * 4 + nt][rt][2], acc2[j * 4 + nt][rt][3])};
          *reinterpret_cast<u32x2*>(img + (lb + rt * 16 + li) * 256 + ((((nh * 4 + nt) * 2 + (g >> 1)) ^ li) << 4) + (g & 1) * 8) = pk;
        }
    }
    // end of chunk.  Weight waves: chunk c + 1 has landed when at most the 8 loads of chunk c + 2 and the stores
    // issued behind the previous barrier (dh: 4, chunks 2, 6, 10, 14; dy2 of chunk 14: 4) are younger.  h waves: the
    // next h quarter must be there at the end of a quarter's last chunk; otherwise nothing to wait for
    {
      constexpr int st_prev = ((c & 3) == 2 || (c == MLP_CHUNKS - 1 && !LNB)) ? SI : 0;
      constexpr int allowed_w = st_prev + (c + 2 < MLP_CHUNKS ? 8 : 0) + (c >= MLP_CHUNKS - 2 ? LN_PF : 0);
      constexpr bool h_due = (c & 3) == 3 && q < 3;
      if (wv < 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(allowed_w) : "memory");
      else if (h_due) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    MLP_STAMP(3 + c);
    if (!ffn2 && j == 1) {
      // dh quarter complete in LDS: write it out in 256-byte row pieces and pick up this wave's operand fragments
#pragma unroll
      for (int i = 0; i < SI; ++i) {
        const int idx = tid + 512 * i, r = idx >> 4, c16 = idx & 15;
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(Hs + r * 256 + ((c16 ^ (r & 15)) << 4)), rs_dh,
                                               (unsigned int)(row0 + r) * (MLP_F * 2) + c16 * 16 + q * 256, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          hf[rt][ks] = *reinterpret_cast<const bf16x8*>(Hs + (lb + rt * 16 + li) * 256 + xs[ks]);
    }
    if (c >= MLP_CHUNKS - 2 && !LNB) {
      const unsigned char* img = Ws + (c % 3) * MLP_WS_B;
#pragma unroll
      for (int i = 0; i < SI; ++i) {
        const int idx = tid + 512 * i, r = idx >> 4, c16 = idx & 15;
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(img + r * 256 + ((c16 ^ (r & 15)) << 4)), rs_dy,
                                               (unsigned int)(row0 + r) * (MLP_D * 2) + c16 * 16 + j * 256, 0, 0);
      }
    }
  };
  chunk(std::integral_constant<int, 0>{});  chunk(std::integral_constant<int, 1>{});
  chunk(std::integral_constant<int, 2>{});  chunk(std::integral_constant<int, 3>{});
  chunk(std::integral_constant<int, 4>{});  chunk(std::integral_constant<int, 5>{});
  chunk(std::integral_constant<int, 6>{});  chunk(std::integral_constant<int, 7>{});
  chunk(std::integral_constant<int, 8>{});  chunk(std::integral_constant<int, 9>{});
  chunk(std::integral_constant<int, 10>{}); chunk(std::integral_constant<int, 11>{});
  chunk(std::integral_constant<int, 12>{}); chunk(std::integral_constant<int, 13>{});
  chunk(std::integral_constant<int, 14>{}); chunk(std::integral_constant<int, 15>{});
  MLP_STAMP(19);
  if constexpr (LNB) {
    // ---- backward of LN2 on the tile (ln_bwd_tile.h).  dy2 sits in LDS as two bf16 images (columns 0..127 in ring buffer 2,
    // 128..255 in ring buffer 0; every wave is past the barrier behind their last writes); the partial sums go through ring
    // buffer 1 (free since that barrier)
    const unsigned char* img = Ws + ((lane >> 5) ? 0 : 2) * MLP_WS_B;      // this lane's column half
    const int c16 = (lane & 31) >> 1, sub = (lane & 1) * 8;
    auto dy_of = [&](int r) { return *reinterpret_cast<const u32x2*>(img + r * 256 + ((c16 ^ (r & 15)) << 4) + sub); };
    // (x-hat form: 8 columns per lane -- the 16-byte slot s of tile row r: column half s >> 4 in its image)
    auto dy_of8 = [&](int r, int s5) {
      return *reinterpret_cast<const u32x4*>(Ws + ((s5 >> 4) ? 0 : 2) * MLP_WS_B + r * 256 + (((s5 & 15) ^ (r & 15)) << 4));
    };
    if constexpr (LNB == 2) ln_bwd_tile16<8>(p.ln, p.T, row0, blockIdx.x, wv, lane, tid, xhv, dy_of8, reinterpret_cast<float*>(Ws + 1 * MLP_WS_B));
    else ln_bwd_tile(p.ln, p.T, row0, blockIdx.x, wv, lane, tid, xv, dy_of, reinterpret_cast<float*>(Ws + 1 * MLP_WS_B));
  }
}

// ---------------------------------------------------------------------------------------------------------
// Input gradient of the fused Q | K | V Dense: dy1 = dqkv Wqkv  (bf16 [T][256], K = 768; Keras autodiff of
// transformer.py:85-90).  Activation-stationary like the kernels above: 128 rows per workgroup, the 64
// accumulator registers of a lane hold its share of the 128 x 256 result for the whole launch; dqkv streams through
// two LDS images in six 128-column pieces (loaded by waves 4-7, straight into LDS), the transposed weights [256][768]
// in twelve 32 KB chunks through three buffers (waves 0-3).  The weight-stationary kernel did this product at
// 2.6 TB/s (24-26 us for 66 MB).
struct DgradParams {
  const unsigned short* A;     // [T][768] bf16
  const unsigned short* Wt;    // [256][768] bf16: Wt[c][n] = W[n][c]
  unsigned short* C;           // [T][256] bf16
  int T;
  LnTileArgs ln;               // dgrad_half_kernel<768, true> (mfp_dgrad_qkv_ln_half): the backward of LN1 on the result tile
};
// (DG_K = 768: the fused Q | K | V; DG_K = 256: the attention output projection, da = d_o1 Wo)
template <int DG_K>
__global__ __launch_bounds__(512) void dgrad_qkv_kernel(DgradParams p) {
  constexpr int DG_KQ = DG_K / 128, DG_CHUNKS = 2 * DG_KQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const As = smem;                       // two [128][256 B] images of dqkv pieces
  unsigned char* const Ws = smem + 2 * MLP_HS_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int rp = wave & 3, nh = wave >> 2;
  const int row0 = blockIdx.x * MLP_ROWS;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Wt), 0, MLP_D * DG_K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.A), 0, (unsigned int)p.T * (DG_K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (unsigned int)p.T * (MLP_D * 2), 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(wave), wl = wv & 3;
  // both streams: [128 rows][256 B] pieces out of 1536-byte rows, 4 rows per 1 KB instruction, source slot =
  // destination slot ^ (row & 15); 8 instructions per wave
  const unsigned int poff = (unsigned int)((wl * 32 + (lane >> 4)) * (DG_K * 2) + (((lane & 15) ^ (lane >> 4)) << 4));
  auto wload = [&](int c) {          // chunk c = (kq, j): Wt rows 128 j .. + 127, columns 128 kq .. + 127
    if (wv >= 4) return;
    unsigned char* dst = Ws + (c % 3) * MLP_WS_B + wl * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_u8*)(dst + i * 1024), 16, poff ^ ((i & 3) << 6),
                                               (c & 1) * 128 * (DG_K * 2) + (c >> 1) * 256 + i * 4 * (DG_K * 2), 0, 0);
  };
  auto aload = [&](int kq) {
    if (wv < 4) return;
    unsigned char* dst = As + (kq & 1) * MLP_HS_B + wl * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_u8*)(dst + i * 1024), 16, poff ^ ((i & 3) << 6),
                                               row0 * (DG_K * 2) + kq * 256 + i * 4 * (DG_K * 2), 0, 0);
  };
  wload(0);
  wload(1);
  aload(0);
  aload(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x4 acc2[8][2];
  bf16x8 hf[2][4];
  int xs[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xs[ks] = ((ks * 4 + g) ^ li) << 4;

  auto chunk = [&](auto cc_) {
    constexpr int c = decltype(cc_)::value;
    constexpr int kq = c >> 1, j = c & 1;
    if (c + 2 < DG_CHUNKS) wload(c + 2);
    // piece kq + 1 goes into the image piece kq - 1 was read from (every wave picked up its fragments of it before
    // the barrier that ended chunk (kq - 1, 0)); waited for at the end of chunk (kq, 1): two chunk periods
    if (j == 0 && kq >= 1 && kq + 1 < DG_KQ) aload(kq + 1);
    if (j == 0) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          hf[rt][ks] = *reinterpret_cast<const bf16x8*>(As + (kq & 1) * MLP_HS_B + (rp * 32 + rt * 16 + li) * 256 + xs[ks]);
    }
    const unsigned char* wa = Ws + (c % 3) * MLP_WS_B + ((nh * 4) * 16 + li) * 256;
    bf16x8 wf[2][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[0][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 4096 + xs[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 4096 + xs[ks + 1]);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          acc2[j * 4 + nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][nt], hf[rt][ks],
                                                                       (kq == 0 && ks == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc2[j * 4 + nt][rt], 0, 0, 0);
    }
    if (c >= DG_CHUNKS - 2) {
      // columns 128 j .. + 127 of the result are final: bf16 into a free [128][256 B] image (the weight buffer of
      // the chunk before this one), stored in whole 256-byte row pieces behind the barrier
      unsigned char* img = Ws + ((c + 2) % 3) * MLP_WS_B;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const u32x2 pk = {pack_bf16x2(acc2[j * 4 + nt][rt][0], acc2[j * 4 + nt][rt][1]), pack_bf16x2(acc2[j * 4 + nt][rt][2], acc2[j * 4 + nt][rt][3])};
          *reinterpret_cast<u32x2*>(img + (rp * 32 + rt * 16 + li) * 256 + ((((nh * 4 + nt) * 2 + (g >> 1)) ^ li) << 4) + (g & 1) * 8) = pk;
        }
    }
    {
      // weight waves: chunk c + 1 landed when at most the 8 loads of chunk c + 2 (and, in the last chunk, the 4 result
      // stores of the chunk before) are outstanding; activation waves: piece kq + 1 is due at the end of (kq, 1)
      constexpr int allowed_w = (c + 2 < DG_CHUNKS ? 8 : 0) + (c == DG_CHUNKS - 1 ? 4 : 0);
      if (wv < 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(allowed_w) : "memory");
      else if (j == 1 && kq + 1 < DG_KQ) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (c >= DG_CHUNKS - 2) {
      const unsigned char* img = Ws + ((c + 2) % 3) * MLP_WS_B;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + 512 * i, r = idx >> 4, c16 = idx & 15;
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(img + r * 256 + ((c16 ^ (r & 15)) << 4)), rs_c,
                                               (unsigned int)(row0 + r) * (MLP_D * 2) + c16 * 16 + j * 256, 0, 0);
      }
    }
  };
  chunk(std::integral_constant<int, 0>{});  chunk(std::integral_constant<int, 1>{});
  chunk(std::integral_constant<int, 2>{});  chunk(std::integral_constant<int, 3>{});
  if constexpr (DG_CHUNKS > 4) {
    chunk(std::integral_constant<int, 4>{});  chunk(std::integral_constant<int, 5>{});
    chunk(std::integral_constant<int, 6>{});  chunk(std::integral_constant<int, 7>{});
    chunk(std::integral_constant<int, 8>{});  chunk(std::integral_constant<int, 9>{});
    chunk(std::integral_constant<int, 10>{}); chunk(std::integral_constant<int, 11>{});
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same product with HALF-SIZE workgroups: 64 rows, 4 waves, 80 KB of LDS -- two workgroups per CU -- for launches
// whose 128-row grid would leave CUs idle: a batch of 128 documents x 128 elements (BASELINE config c4 per GPU) is 256
// workgroups instead of 128.  Measured (MI355X, stand-alone): T = 16 384: 13.7 vs 15.4 us (K = 768), 7.3 vs 7.9 us
// (K = 256); T = 32 768: 20.8 vs 19.7 us -- two co-resident workgroups in different phases do NOT lift the full-size
// launch: what bounds these kernels is the weight stream out of L2 (two 32 KB chunks in flight per CU against ~2 us of
// loaded L2 latency = 32 GB/s per CU, about a chunk per us whatever the split), so the half-size form is selected only
// when T / 128 < #CUs.
// Wave (rp, nh) = (wave & 1, wave >> 1) owns rows 32 rp .. + 31 and half of the 64 output columns of every weight
// chunk (16 KB: Wt rows 64 j .. + 63, columns 128 kq .. + 127); waves 0-1 stream the weights, waves 2-3 the activation
// pieces; the bf16 result leaves through the activation images (free during the last piece) in whole 256-byte rows.
constexpr int H_ROWS = 64;
constexpr int H_AS_B = H_ROWS * 256;     // one [64][256 B] activation piece
constexpr int H_WS_B = 16384;            // one weight chunk
constexpr int H_LDS = 2 * H_AS_B + 3 * H_WS_B;      // 80 KB

// LNB (round 5): the x-hat LayerNorm backward on the 64 x 256 result tile (whole rows: no exchange), ln_bwd_tile with four waves
// of 16 rows -- dy1 never reaches HBM and the stand-alone ln_bwd launch of the three-launch attention route is gone
template <int DG_K, bool LNB = false>
__global__ __launch_bounds__(256, 2) void dgrad_half_kernel(DgradParams p) {
  constexpr int DG_KQ = DG_K / 128, DG_CHUNKS = 4 * DG_KQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const As = smem;
  unsigned char* const Ws = smem + 2 * H_AS_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int rp = wave & 1, nh = wave >> 1;
  const int row0 = blockIdx.x * H_ROWS;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Wt), 0, MLP_D * DG_K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.A), 0, (unsigned int)p.T * (DG_K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (unsigned int)p.T * (MLP_D * 2), 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(wave), wl = wv & 1;
  // both streams: 32 rows of 256 B per wave out of (2 DG_K)-byte rows, 4 rows per 1 KB instruction, source slot =
  // destination slot ^ (row & 15); 8 instructions per wave
  const unsigned int poff = (unsigned int)((wl * 32 + (lane >> 4)) * (DG_K * 2) + (((lane & 15) ^ (lane >> 4)) << 4));
  auto wload = [&](int c) {          // chunk c = (kq, j): Wt rows 64 j .. + 63, columns 128 kq .. + 127
    if (wv >= 2) return;
    unsigned char* dst = Ws + (c % 3) * H_WS_B + wl * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_u8*)(dst + i * 1024), 16, poff ^ ((i & 3) << 6),
                                               (c & 3) * 64 * (DG_K * 2) + (c >> 2) * 256 + i * 4 * (DG_K * 2), 0, 0);
  };
  auto aload = [&](int kq) {
    if (wv < 2) return;
    unsigned char* dst = As + (kq & 1) * H_AS_B + wl * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_u8*)(dst + i * 1024), 16, poff ^ ((i & 3) << 6),
                                               row0 * (DG_K * 2) + kq * 256 + i * 4 * (DG_K * 2), 0, 0);
  };
  wload(0);
  wload(1);
  aload(0);
  if (DG_KQ > 1) aload(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x4 acc2[4][2][2];
  bf16x8 hf[2][4];
  u32x4 xhv[8];       // (LNB: the tile's x-hat rows, wave w rows 16 w .. + 15, two rows per 16-byte-per-lane load)
  int xs[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xs[ks] = ((ks * 4 + g) ^ li) << 4;
  // result image of column half jh (columns 128 jh .. + 127 of 64 rows, bf16): [64][256 B], slot ^ (row & 15); half 0
  // in the activation buffer the last piece does not use, half 1 in the last piece's own buffer (its fragments are in
  // registers two barriers earlier)
  auto out_store = [&](int jh) {
    const unsigned char* img = As + ((DG_KQ + jh) & 1) * H_AS_B;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, r = idx >> 4, c16 = idx & 15;
      __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(img + r * 256 + ((c16 ^ (r & 15)) << 4)), rs_c,
                                             (unsigned int)(row0 + r) * (MLP_D * 2) + c16 * 16 + jh * 256, 0, 0);
    }
  };

  auto chunk = [&](auto cc_) {
    constexpr int c = decltype(cc_)::value;
    constexpr int kq = c >> 2, j = c & 3;
    constexpr bool lastq = kq == DG_KQ - 1;
    if (c + 2 < DG_CHUNKS) wload(c + 2);
    if (!LNB && lastq && j == 2) out_store(0);        // behind the barrier that ended (last, 1); AFTER the weight loads (counted waits)
    if constexpr (LNB && c == DG_CHUNKS - 1) ln_tile_load_xh16(ln_tile_xh_rsrc(p.ln, p.T), row0, wv, lane, xhv);      // 8 loads cross the last chunk
    if (j == 0 && kq >= 1 && kq + 1 < DG_KQ) aload(kq + 1);
    if (j == 0) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          hf[rt][ks] = *reinterpret_cast<const bf16x8*>(As + (kq & 1) * H_AS_B + (rp * 32 + rt * 16 + li) * 256 + xs[ks]);
    }
    const unsigned char* wa = Ws + (c % 3) * H_WS_B + ((nh * 2) * 16 + li) * 256;
    bf16x8 wf[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) wf[0][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 4096 + xs[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) wf[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 4096 + xs[ks + 1]);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          acc2[j][nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][nt], hf[rt][ks],
                                                                    (kq == 0 && ks == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc2[j][nt][rt], 0, 0, 0);
    }
    if (lastq) {
      // columns 64 j .. + 63 of the result are final: bf16 into the image of their column half
      unsigned char* img = As + ((DG_KQ + (j >> 1)) & 1) * H_AS_B;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const u32x2 pk = {pack_bf16x2(acc2[j][nt][rt][0], acc2[j][nt][rt][1]), pack_bf16x2(acc2[j][nt][rt][2], acc2[j][nt][rt][3])};
          *reinterpret_cast<u32x2*>(img + (rp * 32 + rt * 16 + li) * 256 +
                                    (((((j & 1) * 4 + nh * 2 + nt) * 2 + (g >> 1)) ^ li) << 4) + (g & 1) * 8) = pk;
        }
    }
    {
      // weight waves: chunk c + 1 landed when at most the 8 loads of chunk c + 2 (and the 4 stores of result half 0,
      // issued at the head of (last, 2)) are younger; activation waves: piece kq + 1 is due at the end of (kq, 3)
      constexpr int allowed_w = (c + 2 < DG_CHUNKS ? 8 : 0) + ((!LNB && lastq && j == 2) ? 4 : 0) + ((LNB && c == DG_CHUNKS - 1) ? 8 : 0);
      if (wv < 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(allowed_w) : "memory");
      else if (j == 3 && kq + 1 < DG_KQ) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  };
  wgg_free_static_for<0, DG_CHUNKS>(chunk);
  if constexpr (LNB) {
    // dy sits in the two activation buffers as bf16 images of its column halves; the partial sums go through the weight ring
    auto dy_of8 = [&](int r, int s5) {      // the 16-byte slot s5 of tile row r: column half s5 >> 4 in its image
      return *reinterpret_cast<const u32x4*>(As + ((DG_KQ + (s5 >> 4)) & 1) * H_AS_B + r * 256 + (((s5 & 15) ^ (r & 15)) << 4));
    };
    ln_bwd_tile16<4>(p.ln, p.T, row0, blockIdx.x, wv, lane, tid, xhv, dy_of8, reinterpret_cast<float*>(Ws));
  } else {
    out_store(1);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Encoder: both numerical-attribute Dense layers in one launch (reference encoder.py:156-160,174-175,194-198):
//
//     h[t] += sum_j [code_j[t] == 0] * (x_j[t] W_j^T + b_j)        j = image embedding, text embedding (512-wide)
//
// (a masked / padded attribute contributes its special-token embedding instead, which the embedding kernel has
// already put into h).  The same activation-stationary machine as dgrad_qkv_kernel with two activation sources:
// K = 2 x 512 in eight 128-column pieces, the Dense kernels [256][512] are k-major as stored; rows whose code is
// non-zero get zero fragments and no bias; h is read and rewritten in the epilogue.  Two weight-stationary launches
// did this in 45 us for 198 MB (h read and rewritten twice); one pass moves 132 MB.
struct EncParams {
  const unsigned short* X[2];        // [T][512] bf16
  const unsigned short* W[2];        // [256][512] bf16 (out, in)
  const float* bias[2];              // f32 [256]
  const unsigned char* code[2];      // u8 [T]
  float* h;                          // [T][256] f32, accumulated in place
  int T;
};
constexpr int ENC_K = 512, ENC_KQ = 2 * ENC_K / 128, ENC_CHUNKS = 2 * ENC_KQ;

__global__ __launch_bounds__(512) void enc_dense_kernel(EncParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const As = smem;
  unsigned char* const Ws = smem + 2 * MLP_HS_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int rp = wave & 3, nh = wave >> 2;
  const int row0 = blockIdx.x * MLP_ROWS;
  const unsigned int xbytes = (unsigned int)p.T * (ENC_K * 2);
  const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.W[0]), 0, MLP_D * ENC_K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.W[1]), 0, MLP_D * ENC_K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.X[0]), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.X[1]), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(p.h, 0, (unsigned int)p.T * (MLP_D * 4), 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(wave), wl = wv & 3;
  const unsigned int poff = (unsigned int)((wl * 32 + (lane >> 4)) * (ENC_K * 2) + (((lane & 15) ^ (lane >> 4)) << 4));
  auto wload = [&](int c) {          // chunk c = (kq, j): W_{kq >> 2} rows 128 j .. + 127, columns 128 (kq & 3) .. + 127
    if (wv >= 4) return;
    unsigned char* dst = Ws + (c % 3) * MLP_WS_B + wl * 8192;
    const int so = (c & 1) * 128 * (ENC_K * 2) + ((c >> 1) & 3) * 256;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if ((c >> 3) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w0, (lds_u8*)(dst + i * 1024), 16, poff ^ ((i & 3) << 6), so + i * 4 * (ENC_K * 2), 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (lds_u8*)(dst + i * 1024), 16, poff ^ ((i & 3) << 6), so + i * 4 * (ENC_K * 2), 0, 0);
    }
  };
  auto aload = [&](int kq) {
    if (wv < 4) return;
    unsigned char* dst = As + (kq & 1) * MLP_HS_B + wl * 8192;
    const int so = row0 * (ENC_K * 2) + (kq & 3) * 256;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if ((kq >> 2) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x0, (lds_u8*)(dst + i * 1024), 16, poff ^ ((i & 3) << 6), so + i * 4 * (ENC_K * 2), 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x1, (lds_u8*)(dst + i * 1024), 16, poff ^ ((i & 3) << 6), so + i * 4 * (ENC_K * 2), 0, 0);
    }
  };
  wload(0);
  wload(1);
  aload(0);
  aload(1);
  // row codes of this lane's two rows, per source (rows >= T: skipped)
  bool live[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = row0 + rp * 32 + rt * 16 + li;
      live[j][rt] = row < p.T && p.code[j][row] == 0;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x4 acc2[8][2];
  bf16x8 hf[2][4];
  int xs[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xs[ks] = ((ks * 4 + g) ^ li) << 4;

  auto chunk = [&](auto cc_) {
    constexpr int c = decltype(cc_)::value;
    constexpr int kq = c >> 1, j = c & 1, src = kq >> 2;
    if (c + 2 < ENC_CHUNKS) wload(c + 2);
    if (j == 0 && kq >= 1 && kq + 1 < ENC_KQ) aload(kq + 1);
    if (j == 0) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(As + (kq & 1) * MLP_HS_B + (rp * 32 + rt * 16 + li) * 256 + xs[ks]);
          hf[rt][ks] = live[src][rt] ? v : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    const unsigned char* wa = Ws + (c % 3) * MLP_WS_B + ((nh * 4) * 16 + li) * 256;
    bf16x8 wf[2][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[0][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 4096 + xs[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 4096 + xs[ks + 1]);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          acc2[j * 4 + nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][nt], hf[rt][ks],
                                                                       (kq == 0 && ks == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc2[j * 4 + nt][rt], 0, 0, 0);
    }
    if (c >= ENC_CHUNKS - 2) {
      // columns 128 j .. + 127 are final: h += acc + the biases of the live sources, 4 consecutive columns per lane
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = (j * 8 + nh * 4 + nt) * 16 + 4 * g;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias[0] + n), b1 = *reinterpret_cast<const f32x4*>(p.bias[1] + n);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const unsigned int off = (unsigned int)(row0 + rp * 32 + rt * 16 + li) * (MLP_D * 4) + n * 4;
          f32x4 o = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_h, off, 0, 0));
#pragma unroll
          for (int r = 0; r < 4; ++r)
            o[r] += acc2[j * 4 + nt][rt][r] + (live[0][rt] ? b0[r] : 0.f) + (live[1][rt] ? b1[r] : 0.f);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_h, off, 0, 0);
        }
      }
    }
    if (c + 1 < ENC_CHUNKS) {
      // (chunk 14: its 8 epilogue loads were consumed, so chunk 15 -- older -- has landed; the 8 stores may stay)
      constexpr int allowed_w = (c + 2 < ENC_CHUNKS ? 8 : 0) + (c == ENC_CHUNKS - 2 ? 8 : 0);
      if (wv < 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(allowed_w) : "memory");
      else if (j == 1 && kq + 1 < ENC_KQ) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  };
  chunk(std::integral_constant<int, 0>{});  chunk(std::integral_constant<int, 1>{});
  chunk(std::integral_constant<int, 2>{});  chunk(std::integral_constant<int, 3>{});
  chunk(std::integral_constant<int, 4>{});  chunk(std::integral_constant<int, 5>{});
  chunk(std::integral_constant<int, 6>{});  chunk(std::integral_constant<int, 7>{});
  chunk(std::integral_constant<int, 8>{});  chunk(std::integral_constant<int, 9>{});
  chunk(std::integral_constant<int, 10>{}); chunk(std::integral_constant<int, 11>{});
  chunk(std::integral_constant<int, 12>{}); chunk(std::integral_constant<int, 13>{});
  chunk(std::integral_constant<int, 14>{}); chunk(std::integral_constant<int, 15>{});
}

// ---------------------------------------------------------------------------------------------------------
// Input gradient of the concatenated decoder heads: dh f32 [T][256] = dlogits[T][U] Wheads (decoder.py:39-43 under
// Keras autodiff), U = 1384 at Crello.  dgrad_qkv_kernel's machine with the piece count as a run-time loop:
// dlogits in 128-column pieces (the last one reaches past the row end: those columns meet ZERO weights -- the
// transposed weight copy [256][ldw] is zero-padded to a multiple of 128 columns), f32 result straight from the
// accumulators.  The LDS-tiled kernel took 44 us (2 us per 64-wide k-tile and workgroup, 22 of them).
struct DgradRowsParams {
  const unsigned short* A;     // [T][lda] bf16
  const unsigned short* Wt;    // [256][ldw] bf16, zero beyond the real columns
  float* C;                    // [T][256] f32
  int T, KQ, lda, ldw;         // lda, ldw in ELEMENTS; KQ = 128-column pieces (ldw >= 128 KQ)
  // optional second result: the dropout-masked, 1 / keep-scaled bf16 copy of C -- what the backward pass of the Dense
  // in front of a Dropout consumes (mfp_dropout_bwd fused; same mask as the forward epilogue: seed, offset, step)
  unsigned short* Cd;
  float dropout_p; unsigned long long seed, offset; const int* step_ptr;
};

__global__ __launch_bounds__(512) void dgrad_rows_kernel(DgradRowsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const As = smem;
  unsigned char* const Ws = smem + 2 * MLP_HS_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int rp = wave & 3, nh = wave >> 2;
  const int row0 = blockIdx.x * MLP_ROWS;
  const int lda2 = p.lda * 2, ldw2 = p.ldw * 2, KQ = p.KQ;
  // (read first: this load must not sit between the counted waits of the chunk loop)
  const int step_now = (p.Cd && p.step_ptr) ? __builtin_amdgcn_readfirstlane(*p.step_ptr) : 0;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Wt), 0, (unsigned int)(MLP_D * ldw2), 0x00020000);
  // (no masked copy wanted: a zero-sized buffer; its stores are issued all the same -- the counted waits stay fixed)
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(p.Cd ? (void*)p.Cd : (void*)p.C, 0, p.Cd ? (unsigned int)p.T * (MLP_D * 2) : 0u, 0x00020000);
  const float inv_keep = 1.0f / (1.0f - p.dropout_p);
  const unsigned int dthr = drop_thr16(p.dropout_p);
  const unsigned int dkey = drop_key(p.seed, p.offset + (unsigned long long)step_now * MFP_RNG_STEP_STRIDE);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.A), 0, (unsigned int)p.T * (unsigned int)lda2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (unsigned int)p.T * (MLP_D * 4), 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(wave), wl = wv & 3;
  const unsigned int slot0 = (unsigned int)(((lane & 15) ^ (lane >> 4)) << 4);
  // (row strides are arbitrary multiples of 16 bytes here: the slot XOR is applied to the slot term alone)
  const unsigned int woff = (unsigned int)((wl * 32 + (lane >> 4)) * ldw2);
  const unsigned int aoff = (unsigned int)((row0 + wl * 32 + (lane >> 4)) * lda2);
  auto wload = [&](int c) {          // chunk c = (kq, j): Wt rows 128 j .. + 127, columns 128 kq .. + 127
    if (wv >= 4) return;
    unsigned char* dst = Ws + (c % 3) * MLP_WS_B + wl * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_u8*)(dst + i * 1024), 16, woff + (slot0 ^ ((i & 3) << 6)),
                                               (c & 1) * 128 * ldw2 + (c >> 1) * 256 + i * 4 * ldw2, 0, 0);
  };
  auto aload = [&](int kq) {
    if (wv < 4) return;
    unsigned char* dst = As + (kq & 1) * MLP_HS_B + wl * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_u8*)(dst + i * 1024), 16, aoff + (slot0 ^ ((i & 3) << 6)), kq * 256 + i * 4 * lda2, 0, 0);
  };
  wload(0);
  wload(1);
  aload(0);
  if (KQ > 1) aload(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x4 acc2[8][2];
#pragma unroll
  for (int a = 0; a < 8; ++a) acc2[a][0] = acc2[a][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 hf[2][4];
  int xs[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xs[ks] = ((ks * 4 + g) ^ li) << 4;
  const int nchunks = 2 * KQ;
  for (int kq = 0; kq < KQ; ++kq) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = 2 * kq + j;
      if (c + 2 < nchunks) wload(c + 2);
      if (j == 0 && kq >= 1 && kq + 1 < KQ) aload(kq + 1);
      if (j == 0) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            hf[rt][ks] = *reinterpret_cast<const bf16x8*>(As + (kq & 1) * MLP_HS_B + (rp * 32 + rt * 16 + li) * 256 + xs[ks]);
      }
      const unsigned char* wa = Ws + (c % 3) * MLP_WS_B + ((nh * 4) * 16 + li) * 256;
      bf16x8 wf[2][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) wf[0][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 4096 + xs[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) wf[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 4096 + xs[ks + 1]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
            acc2[j * 4 + nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][nt], hf[rt][ks], acc2[j * 4 + nt][rt], 0, 0, 0);
      }
      if (kq == KQ - 1) {
        // columns 128 j .. + 127 of the result are final: 4 consecutive f32 per lane and tile, straight out
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            const int row = row0 + rp * 32 + rt * 16 + li, n = (j * 8 + nh * 4 + nt) * 16 + 4 * g;
            const f32x4 v = acc2[j * 4 + nt][rt];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_c, (unsigned int)row * (MLP_D * 4) + n * 4, 0, 0);
            bool keep[4] = {true, true, true, true};
            if (p.dropout_p > 0.f) drop_keep4(drop_row(dkey, (unsigned int)row), (unsigned int)n, dthr, keep);
            const u32x2 pk = {pack_bf16x2(keep[0] ? v[0] * inv_keep : 0.f, keep[1] ? v[1] * inv_keep : 0.f),
                              pack_bf16x2(keep[2] ? v[2] * inv_keep : 0.f, keep[3] ? v[3] * inv_keep : 0.f)};
            __builtin_amdgcn_raw_buffer_store_b64(pk, rs_d, (unsigned int)row * (MLP_D * 2) + n * 2, 0, 0);
          }
      }
      if (c + 1 < nchunks) {
        // weight waves: chunk c + 1 landed when at most the 8 loads of chunk c + 2 (and the 8 result stores of the
        // second to last chunk) are outstanding; activation waves: piece kq + 1 is due at the end of (kq, 1)
        if (wv < 4) {
          if (c + 2 < nchunks) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");      // (here: the 8 + 8 result stores of this chunk)
        } else if (j == 1 && kq + 1 < KQ) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }
}

}  // namespace

#ifdef MFP_GEMM_TRACE
static unsigned long long* g_mlp_trace = nullptr;
extern "C" void mfp_mlp_trace_buffer(void* ptr) { g_mlp_trace = reinterpret_cast<unsigned long long*>(ptr); }
#endif

extern "C" int mfp_mlp_fused_fwd(const float* x1, const float* gamma, const float* beta, const void* W1, const float* b1,
                                 const void* W2, const float* b2, void* y2, float* mean, float* rstd, void* h, float* x2,
                                 void* x2c, int32_t T, int32_t D, float eps, float dropout_p, uint64_t seed, uint64_t offset,
                                 const int32_t* step_ptr, mfp_stream_t stream) {
  MFP_CHECK_ARG(x1 && gamma && beta && W1 && b1 && W2 && b2 && y2 && mean && rstd && h && x2);
  MFP_CHECK_ARG(T > 0 && T <= (1 << 21) && D == MLP_D && eps > 0.f && dropout_p >= 0.f && dropout_p < 1.f);
  MFP_CHECK_ARG(((uintptr_t)x1 % 16) == 0 && ((uintptr_t)W1 % 16) == 0 && ((uintptr_t)W2 % 16) == 0 &&
                ((uintptr_t)y2 % 16) == 0 && ((uintptr_t)h % 16) == 0 && ((uintptr_t)x2 % 16) == 0 && ((uintptr_t)x2c % 16) == 0);
  MlpParams p;
  p.x1 = x1; p.gamma = gamma; p.beta = beta;
  p.W1 = reinterpret_cast<const unsigned short*>(W1); p.b1 = b1;
  p.W2 = reinterpret_cast<const unsigned short*>(W2); p.b2 = b2;
  p.y2 = reinterpret_cast<unsigned short*>(y2); p.mean = mean; p.rstd = rstd;
  p.h = reinterpret_cast<unsigned short*>(h); p.x2 = x2; p.x2c = reinterpret_cast<unsigned short*>(x2c);
  p.T = T; p.eps = eps; p.dropout_p = dropout_p; p.seed = seed; p.offset = offset; p.step_ptr = step_ptr;
#ifdef MFP_GEMM_TRACE
  p.trace = g_mlp_trace;
#endif
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS);
    if (e != hipSuccess) {
      mfp_set_error("mfp_mlp_fused_fwd: cannot raise dynamic LDS to %d: %s", MLP_LDS, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  const int blocks = (T + MLP_ROWS - 1) / MLP_ROWS;
  if (dropout_p > 0.f)
    hipLaunchKernelGGL(mlp_fused_kernel<true>, dim3(blocks), dim3(512), MLP_LDS, st, p);
  else
    hipLaunchKernelGGL(mlp_fused_kernel<false>, dim3(blocks), dim3(512), MLP_LDS, st, p);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_mlp_fused_bwd(const void* d_o2, const void* h, const void* W2t, const void* W1t, void* dh, void* dy2,
                                 int32_t T, int32_t D, mfp_stream_t stream) {
  MFP_CHECK_ARG(d_o2 && h && W2t && W1t && dh && dy2);
  MFP_CHECK_ARG(T > 0 && T <= (1 << 21) && D == MLP_D);
  MFP_CHECK_ARG(((uintptr_t)d_o2 % 16) == 0 && ((uintptr_t)h % 16) == 0 && ((uintptr_t)W2t % 16) == 0 &&
                ((uintptr_t)W1t % 16) == 0 && ((uintptr_t)dh % 16) == 0 && ((uintptr_t)dy2 % 16) == 0);
  MlpBwdParams p = {};
  p.d_o2 = reinterpret_cast<const unsigned short*>(d_o2); p.h = reinterpret_cast<const unsigned short*>(h);
  p.W2t = reinterpret_cast<const unsigned short*>(W2t); p.W1t = reinterpret_cast<const unsigned short*>(W1t);
  p.dh = reinterpret_cast<unsigned short*>(dh); p.dy2 = reinterpret_cast<unsigned short*>(dy2);
  p.T = T;
#ifdef MFP_GEMM_TRACE
  p.trace = g_mlp_trace;
#endif
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  constexpr int lds = 2 * MLP_HS_B + 3 * MLP_WS_B;      // 160 KB: all of a CU's LDS
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_mlp_fused_bwd: cannot raise dynamic LDS to %d: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(mlp_bwd_kernel<0>, dim3((T + MLP_ROWS - 1) / MLP_ROWS), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), p);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_mlp_bwd_ln(const void* d_o2, const void* h, const void* W2t, const void* W1t, void* dh, const float* x,
                              const void* xhat, const float* gamma, const float* mean, const float* rstd, const void* dres, void* dx, void* ddrop,
                              float* part, size_t part_bytes, int32_t T, int32_t D, float drop_p, uint64_t seed, uint64_t offset,
                              const int32_t* step_ptr, mfp_stream_t stream) {
  MFP_CHECK_ARG(d_o2 && h && W2t && W1t && dh && gamma && rstd && dres && dx && ddrop && part);
  MFP_CHECK_ARG(xhat != nullptr || (x != nullptr && mean != nullptr));
  MFP_CHECK_ARG(((uintptr_t)xhat % 16) == 0);
  MFP_CHECK_ARG(T > 0 && T % MLP_ROWS == 0 && T <= (1 << 21) && D == MLP_D && drop_p >= 0.f && drop_p < 1.f);
  MFP_CHECK_ARG(((uintptr_t)d_o2 % 16) == 0 && ((uintptr_t)h % 16) == 0 && ((uintptr_t)W2t % 16) == 0 && ((uintptr_t)W1t % 16) == 0 &&
                ((uintptr_t)dh % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)dres % 16) == 0 &&
                ((uintptr_t)dx % 16) == 0 && ((uintptr_t)ddrop % 16) == 0);
  if (part_bytes < (size_t)(T / MLP_ROWS) * 3 * MLP_D * sizeof(float)) {
    mfp_set_error("mfp_mlp_bwd_ln: partial-sum buffer too small");
    return MFP_EWORKSPACE;
  }
  MlpBwdParams p = {};
  p.d_o2 = reinterpret_cast<const unsigned short*>(d_o2); p.h = reinterpret_cast<const unsigned short*>(h);
  p.W2t = reinterpret_cast<const unsigned short*>(W2t); p.W1t = reinterpret_cast<const unsigned short*>(W1t);
  p.dh = reinterpret_cast<unsigned short*>(dh); p.dy2 = nullptr;
  p.T = T;
#ifdef MFP_GEMM_TRACE
  p.trace = g_mlp_trace;
#endif
  p.ln.x = x; p.ln.xhat = reinterpret_cast<const unsigned short*>(xhat); p.ln.gamma = gamma; p.ln.mean = mean; p.ln.rstd = rstd;
  p.ln.dres = reinterpret_cast<const unsigned short*>(dres);
  p.ln.dx = reinterpret_cast<unsigned short*>(dx); p.ln.ddrop = reinterpret_cast<unsigned short*>(ddrop); p.ln.part = part;
  p.ln.drop_p = drop_p; p.ln.seed = seed; p.ln.offset = offset; p.ln.step_ptr = step_ptr;
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  constexpr int lds = 2 * MLP_HS_B + 3 * MLP_WS_B;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_mlp_bwd_ln: cannot raise dynamic LDS to %d: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  if (xhat != nullptr) hipLaunchKernelGGL(mlp_bwd_kernel<2>, dim3(T / MLP_ROWS), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), p);
  else hipLaunchKernelGGL(mlp_bwd_kernel<1>, dim3(T / MLP_ROWS), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), p);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

// mfp_mlp_bwd_ln (x-hat form) on HALF tiles: two workgroups per 128-row tile (64 rows each, one 16-row tile per wave), for
// batches with fewer 128-row tiles than CUs (BASELINE config c4).  `part` holds T / 64 rows of [3][256].  dh bit-identical to
// mfp_mlp_bwd_ln, dx / ddrop too; the partial rows sum to the same parameter gradients in another grouping.
extern "C" int mfp_mlp_bwd_ln_half(const void* d_o2, const void* h, const void* W2t, const void* W1t, void* dh, const void* xhat,
                                   const float* gamma, const float* rstd, const void* dres, void* dx, void* ddrop, float* part,
                                   size_t part_bytes, int32_t T, int32_t D, float drop_p, uint64_t seed, uint64_t offset,
                                   const int32_t* step_ptr, mfp_stream_t stream) {
  MFP_CHECK_ARG(d_o2 && h && W2t && W1t && dh && xhat && gamma && rstd && dres && dx && ddrop && part);
  MFP_CHECK_ARG(T > 0 && T % MLP_ROWS == 0 && T <= (1 << 21) && D == MLP_D && drop_p >= 0.f && drop_p < 1.f);
  MFP_CHECK_ARG(((uintptr_t)d_o2 % 16) == 0 && ((uintptr_t)h % 16) == 0 && ((uintptr_t)W2t % 16) == 0 && ((uintptr_t)W1t % 16) == 0 &&
                ((uintptr_t)dh % 16) == 0 && ((uintptr_t)xhat % 16) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)dres % 16) == 0 &&
                ((uintptr_t)dx % 16) == 0 && ((uintptr_t)ddrop % 16) == 0);
  if (part_bytes < (size_t)(T / 64) * 3 * MLP_D * sizeof(float)) {
    mfp_set_error("mfp_mlp_bwd_ln_half: partial-sum buffer too small");
    return MFP_EWORKSPACE;
  }
  MlpBwdParams p = {};
  p.d_o2 = reinterpret_cast<const unsigned short*>(d_o2); p.h = reinterpret_cast<const unsigned short*>(h);
  p.W2t = reinterpret_cast<const unsigned short*>(W2t); p.W1t = reinterpret_cast<const unsigned short*>(W1t);
  p.dh = reinterpret_cast<unsigned short*>(dh); p.dy2 = nullptr;
  p.T = T;
#ifdef MFP_GEMM_TRACE
  p.trace = g_mlp_trace;
#endif
  p.ln.x = nullptr; p.ln.xhat = reinterpret_cast<const unsigned short*>(xhat); p.ln.gamma = gamma; p.ln.mean = nullptr; p.ln.rstd = rstd;
  p.ln.dres = reinterpret_cast<const unsigned short*>(dres);
  p.ln.dx = reinterpret_cast<unsigned short*>(dx); p.ln.ddrop = reinterpret_cast<unsigned short*>(ddrop); p.ln.part = part;
  p.ln.drop_p = drop_p; p.ln.seed = seed; p.ln.offset = offset; p.ln.step_ptr = step_ptr;
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  constexpr int lds = 2 * MLP_HS_B + 3 * MLP_WS_B;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_mlp_bwd_ln_half: cannot raise dynamic LDS to %d: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((mlp_bwd_kernel<2, true>), dim3(T / 64), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), p);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_qkv_fused_fwd(const float* x, const float* gamma, const float* beta, const void* W, const float* bias,
                                 void* y1, float* mean, float* rstd, void* qkv, int32_t T, int32_t D, float eps,
                                 mfp_stream_t stream) {
  MFP_CHECK_ARG(x && gamma && beta && W && bias && y1 && mean && rstd && qkv);
  MFP_CHECK_ARG(T > 0 && T <= (1 << 20) && D == MLP_D && eps > 0.f);
  MFP_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)y1 % 16) == 0 && ((uintptr_t)qkv % 16) == 0 &&
                ((uintptr_t)bias % 16) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0);
  QkvParams p;
  p.x = x; p.gamma = gamma; p.beta = beta; p.W = reinterpret_cast<const unsigned short*>(W); p.bias = bias;
  p.y1 = reinterpret_cast<unsigned short*>(y1); p.mean = mean; p.rstd = rstd; p.qkv = reinterpret_cast<unsigned short*>(qkv);
  p.T = T; p.eps = eps;
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qkv_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, QKV_LDS);
    if (e != hipSuccess) {
      mfp_set_error("mfp_qkv_fused_fwd: cannot raise dynamic LDS to %d: %s", QKV_LDS, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(qkv_fused_kernel, dim3((T + MLP_ROWS - 1) / MLP_ROWS), dim3(512), QKV_LDS, reinterpret_cast<hipStream_t>(stream), p);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

// 1: the half-size workgroups (two per CU); 0: 128-row workgroups; unset: half-size when the 128-row grid would leave
// CUs idle (T / 128 < #CUs)
static int half_mode(int T) {
  const char* env = getenv("MFP_FUSED_HALF");      // (read per call: the tests flip it)
  if (env != nullptr && env[0] != 0) return atoi(env) != 0;
  const int ncu = mfp_ncu_physical();
  return (T + MLP_ROWS - 1) / MLP_ROWS < ncu;
}

template <int K>
static int launch_dgrad_k(const DgradParams& p, hipStream_t st) {
  if (half_mode(p.T)) {
    static bool attr_done_h[MFP_MAX_DEVICES] = {};
    bool& attr_set_h = attr_done_h[mfp_device_slot()];
    if (!attr_set_h) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_half_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS);
      if (e != hipSuccess) {
        mfp_set_error("mfp_dgrad_qkv: cannot raise dynamic LDS to %d: %s", H_LDS, hipGetErrorString(e));
        return MFP_ELAUNCH;
      }
      attr_set_h = true;
    }
    hipLaunchKernelGGL(dgrad_half_kernel<K>, dim3((p.T + H_ROWS - 1) / H_ROWS), dim3(256), H_LDS, st, p);
    return MFP_OK;
  }
  constexpr int lds = 2 * MLP_HS_B + 3 * MLP_WS_B;
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_qkv_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_dgrad_qkv: cannot raise dynamic LDS to %d: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(dgrad_qkv_kernel<K>, dim3((p.T + MLP_ROWS - 1) / MLP_ROWS), dim3(512), lds, st, p);
  return MFP_OK;
}

extern "C" int mfp_dgrad_qkv(const void* dqkv, const void* Wt, void* dy, int32_t T, int32_t D, mfp_stream_t stream) {
  MFP_CHECK_ARG(dqkv && Wt && dy && T > 0 && T <= (1 << 20) && D == MLP_D);
  MFP_CHECK_ARG(((uintptr_t)dqkv % 16) == 0 && ((uintptr_t)Wt % 16) == 0 && ((uintptr_t)dy % 16) == 0);
  DgradParams p;
  p.A = reinterpret_cast<const unsigned short*>(dqkv); p.Wt = reinterpret_cast<const unsigned short*>(Wt);
  p.C = reinterpret_cast<unsigned short*>(dy); p.T = T;
  const int rc = launch_dgrad_k<768>(p, reinterpret_cast<hipStream_t>(stream));
  if (rc != MFP_OK) return rc;
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

// mfp_dgrad_qkv on 64-row tiles with the x-hat backward of LN1 on its result (dgrad_half_kernel<768, true>): what
// mfp_dgrad_qkv + mfp_layernorm_bwd_xhat compute for batches with fewer 128-row tiles than CUs (BASELINE config c4: the
// three-launch attention route), dy1 never written.  ddrop may be NULL (block 0); part: T / 64 rows of [3][256].  T % 64 == 0.
extern "C" int mfp_dgrad_qkv_ln_half(const void* dqkv, const void* Wt, const void* xhat, const float* gamma, const float* rstd,
                                     const void* dres, void* dx, void* ddrop, float* part, size_t part_bytes, int32_t T, int32_t D,
                                     float drop_p, uint64_t seed, uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream) {
  MFP_CHECK_ARG(dqkv && Wt && xhat && gamma && rstd && dres && dx && part && T > 0 && T <= (1 << 20) && T % H_ROWS == 0 && D == MLP_D);
  MFP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
  MFP_CHECK_ARG(((uintptr_t)dqkv % 16) == 0 && ((uintptr_t)Wt % 16) == 0 && ((uintptr_t)xhat % 16) == 0 && ((uintptr_t)gamma % 16) == 0 &&
                ((uintptr_t)dres % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)ddrop % 16) == 0);
  if (part_bytes < (size_t)(T / H_ROWS) * 3 * MLP_D * sizeof(float)) {
    mfp_set_error("mfp_dgrad_qkv_ln_half: partial-sum buffer too small");
    return MFP_EWORKSPACE;
  }
  DgradParams p = {};
  p.A = reinterpret_cast<const unsigned short*>(dqkv); p.Wt = reinterpret_cast<const unsigned short*>(Wt);
  p.C = nullptr; p.T = T;
  p.ln.x = nullptr; p.ln.xhat = reinterpret_cast<const unsigned short*>(xhat); p.ln.gamma = gamma; p.ln.mean = nullptr; p.ln.rstd = rstd;
  p.ln.dres = reinterpret_cast<const unsigned short*>(dres);
  p.ln.dx = reinterpret_cast<unsigned short*>(dx); p.ln.ddrop = reinterpret_cast<unsigned short*>(ddrop); p.ln.part = part;
  p.ln.drop_p = drop_p; p.ln.seed = seed; p.ln.offset = offset; p.ln.step_ptr = step_ptr;
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_half_kernel<768, true>), hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS);
    if (e != hipSuccess) {
      mfp_set_error("mfp_dgrad_qkv_ln_half: cannot raise dynamic LDS to %d: %s", H_LDS, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((dgrad_half_kernel<768, true>), dim3(T / H_ROWS), dim3(256), H_LDS, reinterpret_cast<hipStream_t>(stream), p);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_dgrad_d256(const void* dy, const void* Wt, void* dx, int32_t T, int32_t D, mfp_stream_t stream) {
  MFP_CHECK_ARG(dy && Wt && dx && T > 0 && T <= (1 << 20) && D == MLP_D);
  MFP_CHECK_ARG(((uintptr_t)dy % 16) == 0 && ((uintptr_t)Wt % 16) == 0 && ((uintptr_t)dx % 16) == 0);
  DgradParams p;
  p.A = reinterpret_cast<const unsigned short*>(dy); p.Wt = reinterpret_cast<const unsigned short*>(Wt);
  p.C = reinterpret_cast<unsigned short*>(dx); p.T = T;
  const int rc = launch_dgrad_k<256>(p, reinterpret_cast<hipStream_t>(stream));
  if (rc != MFP_OK) return rc;
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_encoder_dense2(const void* x0, const void* x1, const void* W0, const void* W1, const float* b0, const float* b1,
                                  const uint8_t* code0, const uint8_t* code1, float* h, int32_t T, int32_t D, int32_t K,
                                  mfp_stream_t stream) {
  MFP_CHECK_ARG(x0 && x1 && W0 && W1 && b0 && b1 && code0 && code1 && h);
  MFP_CHECK_ARG(T > 0 && T <= (1 << 20) && D == MLP_D && K == ENC_K);
  MFP_CHECK_ARG(((uintptr_t)x0 % 16) == 0 && ((uintptr_t)x1 % 16) == 0 && ((uintptr_t)W0 % 16) == 0 && ((uintptr_t)W1 % 16) == 0 &&
                ((uintptr_t)b0 % 16) == 0 && ((uintptr_t)b1 % 16) == 0 && ((uintptr_t)h % 16) == 0);
  EncParams p;
  p.X[0] = reinterpret_cast<const unsigned short*>(x0); p.X[1] = reinterpret_cast<const unsigned short*>(x1);
  p.W[0] = reinterpret_cast<const unsigned short*>(W0); p.W[1] = reinterpret_cast<const unsigned short*>(W1);
  p.bias[0] = b0; p.bias[1] = b1; p.code[0] = code0; p.code[1] = code1; p.h = h; p.T = T;
  constexpr int lds = 2 * MLP_HS_B + 3 * MLP_WS_B;
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc_dense_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_encoder_dense2: cannot raise dynamic LDS to %d: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(enc_dense_kernel, dim3((T + MLP_ROWS - 1) / MLP_ROWS), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), p);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_dgrad_rows(const void* A, int32_t lda, const void* Wt, int32_t ldw, float* C, int32_t T, int32_t D, int32_t K,
                              void* C_drop, float dropout_p, uint64_t seed, uint64_t offset, const int32_t* step_ptr,
                              mfp_stream_t stream) {
  MFP_CHECK_ARG(A && Wt && C && T > 0 && T <= (1 << 19) && D == MLP_D && K > 0 && lda >= K && lda % 8 == 0 && ldw % 128 == 0 && ldw >= K);
  MFP_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)Wt % 16) == 0 && ((uintptr_t)C % 16) == 0 && ((uintptr_t)C_drop % 16) == 0);
  MFP_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f);
  DgradRowsParams p;
  p.A = reinterpret_cast<const unsigned short*>(A); p.Wt = reinterpret_cast<const unsigned short*>(Wt); p.C = C;
  p.T = T; p.KQ = (K + 127) / 128; p.lda = lda; p.ldw = ldw;
  p.Cd = reinterpret_cast<unsigned short*>(C_drop); p.dropout_p = C_drop ? dropout_p : 0.f; p.seed = seed; p.offset = offset; p.step_ptr = step_ptr;
  constexpr int lds = 2 * MLP_HS_B + 3 * MLP_WS_B;
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      mfp_set_error("mfp_dgrad_rows: cannot raise dynamic LDS to %d: %s", lds, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(dgrad_rows_kernel, dim3((T + MLP_ROWS - 1) / MLP_ROWS), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), p);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}
