// Decoder heads + LossLayer + the heads' input gradient in ONE launch (reference models/architecture/decoder.py:95-111,
// models/metrics.py:213-299, and their autodiff), d_model 256, bf16 operands:
//
//     logits = x W^T + b                      x bf16 [T,256], W bf16 [U][256] (all heads side by side, 8-aligned columns)
//     per head: masked clipped-softmax CE (categorical) / sum of squares + cosine score (numerical), their per-key
//               sums, and dlogits = d(sum of losses / B) / d(logits)             (csrc/loss.hip states the formulas)
//     dx = dlogits W                          f32 [T,256] (+ the dropout-masked bf16 copy the last block starts from)
//
// It replaces gemm_ws_kernel (heads forward) + ce_tile_kernel + mse_kernel + dgrad_rows_kernel: the f32 logits (5.5 KB
// per element: 181 MB at c2) are never written unless the caller wants them, dlogits (bf16, the weight-gradient launch
// needs it) is written once and not read back here, and W streams through LDS ONCE for both products.
//
// One 8-wave workgroup per 128 rows; x fragments stay in registers (64), dx accumulates in 64 registers.  The columns are
// walked in CHUNKS of <= 64 (host-built table): a categorical chunk holds whole items (softmax groups), a numerical head
// is cut into 64-column chunks.  Per chunk c (32 KB of W = rows col0 .. col0 + 63 as a [64][512 B] LDS image, 3-buffer
// LDS-DMA ring, counted waits):
//   1. logits tile = W_c x^T (transposed product: a lane holds 4 consecutive columns of its row) + bias
//   2. categorical: tile -> LDS (f32), the dl image zeroed, one 16-lane group per ACTIVE (row, item) runs softmax / clipped CE /
//      gradient (ce_tile_kernel's walk) and writes the item's d(logits) into the dl image as bf16 (inactive items / padding stay 0);
//      numerical: in the accumulator layout -- targets (loaded one chunk ahead, only rows that carry a loss) ->
//      2 (p - y) / B -> dl image; sum of squares / |y|^2 / |p|^2 / y.p per row carried in registers across the head's chunks
//   3. dl image -> HBM (dlogits) and, as the B operand, dx += dl_c W_c with W_c read TRANSPOSED from the same LDS image
//      (ds_read_b64_tr_b16; row swizzle ((row & 7) << 1 | (row >> 3) & 1 keeps both read patterns conflict-free).
// Per-key sums leave as per-workgroup partials (no global atomics): the caller reduces them (mfp_reduce_partials).
// Every iteration issues the same vector-memory operations (out-of-range offsets / zero-sized buffers where a chunk has
// nothing to load or store), so "chunk c has landed" is s_waitcnt vmcnt(24) in every iteration (16 in the variant
// that does not write the logits).
#include "common.h"
#ifndef HL_ABL
#define HL_ABL 0
#endif
// HL_ABL == 9: timing build (tools/trace_heads.py): every wave drops shader-clock stamps (scalar stores, no vector-memory
// operation added) into the logits buffer, which is then not written: stamp 4 c + {0: chunk landed, 1: logits tile,
// 2: dl image complete, 3: second product} of chunk c, 254 = start, 255 = end
#define HL_TR(i) do { if (HL_ABL == 9) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); const unsigned int o_ = (unsigned int)(i) * 8u; \
    asm volatile("s_store_dwordx2 %0, %1, %2 glc" :: "s"(t_), "s"(trbase), "s"(o_) : "memory"); } } while (0)

namespace {

constexpr int HL_ROWS = 128, HL_D = 256, HL_MAXCH = 40, HL_MAXIT = 16, HL_MAXU = 1536;
constexpr int HL_WS_B = 32768;
constexpr int HL_TILE = 3 * HL_WS_B, HL_TW = 68;             // f32 [128][68] staging of a categorical chunk
constexpr int HL_DLI = HL_TILE + HL_ROWS * HL_TW * 4;        // bf16 dl image [128][128 B]
constexpr int HL_WROW = HL_DLI + 16384;                      // u8 [16 keys][128 rows]: the row carries a loss for the key
constexpr int HL_YLAB = HL_WROW + 2048;                      // u8 [128 rows][16 items]: label
constexpr int HL_BIAS = HL_YLAB + 2048;                      // f32 [HL_MAXU]
constexpr int HL_RED = HL_BIAS + HL_MAXU * 4;                // f32 [16][3] (+ pad)
constexpr int HL_ACT = HL_RED + 256;                         // u16 [1024] compacted active (row, item-in-chunk)
constexpr int HL_NACT = HL_ACT + 2048;
constexpr int HL_ITAB = HL_NACT + 16;                        // int [3][16]: item -> key, first column, classes
constexpr int HL_LDS = HL_ITAB + 192;                        // 162 256 B

struct HlParams {
  const unsigned short* X;
  const unsigned short* W;
  const float* bias;
  const int* nvalid;
  float* part;               // [workgroups][48]
  unsigned short* dlogits;   // [T][ld] bf16
  float* logits;             // [T][ld] f32 or nullptr
  float* dX;                 // [T][256] f32 or nullptr
  unsigned short* dXb;       // [T][256] bf16 (the same, unmasked, in the compute dtype) or nullptr
  unsigned short* dXd;       // [T][256] bf16 dropout-masked copy or nullptr
  const int* step_ptr;
  unsigned long long seed, offset;
  mfp_loss_key key[MFP_MAX_LOSS_KEYS];
  int nkeys, nitem, nch, T, S, U, ld;
  float inv_B, dropout_p;
  int item_key[HL_MAXIT], item_feat[HL_MAXIT], item_col[HL_MAXIT], item_C[HL_MAXIT];
  // chunk table, 32-bit words only (a byte / short array indexed by the loop counter becomes a VECTOR-memory load inside
  // the counted-wait schedule): [0] col0 | ncols << 16, [1] kind | key << 8 | first << 16 | last << 24 (kind 0 categorical,
  // 1 numerical), [2] item0 | nitem << 8, [3] / [4] 8-column piece -> item (one byte each, 0xFF: padding / another chunk's)
  int ch[HL_MAXCH][8];
};

typedef __attribute__((address_space(3))) unsigned char lds_u8;

// All-reduce over a 16-lane row with DPP moves (quad xor 1, quad xor 2, half-row mirror, row mirror): a `__shfl_xor` is a
// ds_bpermute round trip, and the softmax walk below is a chain of ~28 of them per item.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
  return v;
}
__device__ __forceinline__ int row16_min(int v) {
  v = min(v, dpp_i<0xB1>(v)); v = min(v, dpp_i<0x4E>(v)); v = min(v, dpp_i<0x141>(v)); v = min(v, dpp_i<0x140>(v));
  return v;
}

__device__ __forceinline__ int isw(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int dsw(int row) { return ((row & 7) << 1) | ((row >> 3) & 1); }

// LOGITS: the f32 logits are written (2 RT more stores per thread and chunk in the counted schedule).
// RT = 16-row tiles per wave: 2 = one workgroup per 128 rows; 1 (round 6) = per 64 rows, for batches with fewer 128-row tiles than
// CUs (BASELINE config c4: 128 documents per GPU = 128 tiles on 256 CUs): the same eight waves, wave (rp, nh) owns rows 16 rp ..
// + 15 -- every row walks the same instructions on the same operands as in the 128-row form: logits, d(logits) and dx are
// bit-identical to it; the per-key sums are the same terms in twice as many partial rows.
template <bool LOGITS, int RT = 2>
__global__ __launch_bounds__(512) void heads_loss_kernel(HlParams p) {
  constexpr int RS = 64 * RT;      // rows of the workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Ws = smem;
  float* const Tile = reinterpret_cast<float*>(smem + HL_TILE);
  unsigned char* const Dli = smem + HL_DLI;
  unsigned char* const Wrow = smem + HL_WROW;
  unsigned char* const Ylab = smem + HL_YLAB;
  float* const Bias = reinterpret_cast<float*>(smem + HL_BIAS);
  float* const Red = reinterpret_cast<float*>(smem + HL_RED);
  unsigned short* const Act = reinterpret_cast<unsigned short*>(smem + HL_ACT);
  int* const Nact = reinterpret_cast<int*>(smem + HL_NACT);
  int* const Itab = reinterpret_cast<int*>(smem + HL_ITAB);      // (per-lane item lookups go to LDS, never to the kernel arguments)
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rp = wave & 3, nh = wave >> 2;
  const int row0 = blockIdx.x * RS;
  const int nch = p.nch;
  constexpr unsigned int OOB = 0xFFFFFFF0u;
  const int step_now = (p.dXd && p.step_ptr) ? __builtin_amdgcn_readfirstlane(*p.step_ptr) : 0;
  const unsigned long long* trbase = reinterpret_cast<const unsigned long long*>(p.logits) + (size_t)(blockIdx.x * 8 + wave) * 256;
  HL_TR(254);

  const unsigned int xbytes = (unsigned int)p.T * (HL_D * 2);
  const unsigned int ldb2 = (unsigned int)p.ld * 2u, ldb4 = (unsigned int)p.ld * 4u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.X), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.W), 0, (unsigned int)p.U * (HL_D * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dl = __builtin_amdgcn_make_buffer_rsrc(p.dlogits, 0, (unsigned int)p.T * ldb2, 0x00020000);
  // (no logits / no masked copy wanted: zero-sized buffers; the stores are issued all the same -- the counted waits stay fixed)
  const __amdgpu_buffer_rsrc_t rs_lg = __builtin_amdgcn_make_buffer_rsrc(p.logits ? (void*)p.logits : (void*)p.dlogits, 0, (p.logits && HL_ABL != 9) ? (unsigned int)p.T * ldb4 : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.dX ? (void*)p.dX : (void*)p.dlogits, 0, p.dX ? (unsigned int)p.T * (HL_D * 4) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cb = __builtin_amdgcn_make_buffer_rsrc(p.dXb ? (void*)p.dXb : (void*)p.dlogits, 0, p.dXb ? xbytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(p.dXd ? (void*)p.dXd : (void*)p.dlogits, 0, p.dXd ? xbytes : 0u, 0x00020000);

  // ---- weight chunk c -> ring buffer c % 3: W rows col0 .. col0 + 63 ([64][512 B], slot ^ dsw(row)); rows past U read zero
  auto wload = [&](int c) {
    const int col0 = c < nch ? (p.ch[c][0] & 0xffff) : p.U;
    unsigned char* dst = Ws + (c % 3) * HL_WS_B + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 8 + i * 2 + (lane >> 5);
      const unsigned int vo = (unsigned int)(row * 512 + (((lane & 31) ^ dsw(row)) << 4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_u8*)(dst + i * 1024), 16, vo, col0 * 512, 0, 0);
    }
  };
  unsigned int onm[2] = {0u, 0u};      // bit k: this lane's row (of row tile rt) carries a loss for key k (filled after the prologue)
  // ---- targets of a numerical chunk for this lane's (row tile, column tile) pairs: only rows that carry a loss
  auto tload = [&](int c, f32x4 (&tg)[2][RT]) {
    const int w1 = c < nch ? p.ch[c][1] : 0;
    const bool num = (w1 & 0xff) == 1;
    const int k = num ? ((w1 >> 8) & 0xff) : 0;
    const mfp_loss_key& key = p.key[k];
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(key.target), 0, num ? (unsigned int)p.T * (unsigned int)key.n_class * 4u : 0u, 0x00020000);
    const int cb = num ? (p.ch[c < nch ? c : 0][0] & 0xffff) - key.col_off : 0;
    // (branch-free: offsets first, then the four loads back to back)
    unsigned int off[2][RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = rp * (16 * RT) + rt * 16 + li;
      const bool on = num & (((onm[rt] >> k) & 1u) != 0u);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        off[nt][rt] = on ? ((unsigned int)(row0 + row) * (unsigned int)key.n_class + cb + (nh * 2 + nt) * 16 + 4 * g) * 4u : OOB;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        tg[nt][rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_t, off[nt][rt], 0, 0));
  };

  // ---- prologue: x fragments, chunks 0 and 1; per (key, row) weights, per (row, item) labels, bias -> LDS
  bf16x8 xf[RT][8];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      xf[rt][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
          rs_x, (unsigned int)(row0 + rp * (16 * RT) + rt * 16 + li) * (HL_D * 2) + g * 16 + ks * 64, 0, 0));
  wload(0);
  wload(1);
  {
    // (four keys / four items per thread (RT = 1: two), all their loads issued before the first is used: one memory round trip
    //  each, not a chain per key; the key index is wave-uniform, so the key records are scalar loads)
    constexpr int KG = 512 / RS, KI = 16 / KG;      // key groups of RS threads, keys per thread
    const int row = tid & (RS - 1), t = row0 + row;
    const int kq = __builtin_amdgcn_readfirstlane(tid / RS);
    const bool live = t < p.T;
    const int b = live ? t / p.S : 0, s = t - b * p.S;
    const int nv = p.nvalid[b];
    unsigned char m[KI]; int v[KI]; int y[KI];
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int k = kq + KG * i;
      const mfp_loss_key& key = p.key[k < p.nkeys ? k : 0];
      m[i] = (live && k < p.nkeys) ? key.mask[t] : 0;
      v[i] = (live && k < p.nkeys && key.cond_idx != nullptr) ? key.cond_idx[(long long)t * key.cond_stride] : 0;
      const int it = kq + KG * i;
      const int itc = it < p.nitem ? it : 0;
      const mfp_loss_key& ikey = p.key[p.item_key[itc]];
      y[i] = (live && it < p.nitem) ? reinterpret_cast<const int*>(ikey.target)[(long long)t * ikey.n_feat + p.item_feat[itc]] : 0;
    }
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int k = kq + KG * i;
      if (k < p.nkeys) {
        const mfp_loss_key& key = p.key[k];
        const bool cnd = key.cond_idx == nullptr || (v[i] >= 0 && v[i] < 32 && ((key.cond_bits >> v[i]) & 1u));
        Wrow[k * 128 + row] = (m[i] != 0 && s < nv && cnd && live) ? 1 : 0;
      }
      if (k < p.nitem) Ylab[row * 16 + k] = (unsigned char)(y[i] < 0 ? 255 : (y[i] > 254 ? 255 : y[i]));
    }
    for (int i = tid; i < HL_MAXU; i += 512) Bias[i] = i < p.U ? p.bias[i] : 0.f;
    if (tid < 48) Red[tid] = 0.f;
    if (tid == 0) *Nact = 0;
    if (tid < 16) { Itab[tid] = p.item_key[tid]; Itab[16 + tid] = p.item_col[tid]; Itab[32 + tid] = p.item_C[tid]; }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
    for (int k = 0; k < p.nkeys; ++k) onm[rt] |= (unsigned int)(Wrow[k * 128 + rp * (16 * RT) + rt * 16 + li] != 0) << k;
  int xs[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xs[ks] = ((ks * 4 + g) ^ dsw(li)) << 4;
  f32x4 acc2[8][RT];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc2[a][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float ms[RT][4];      // numerical head in flight: per row tile {sum d^2, sum y^2, sum p^2, sum y p} over this lane's columns
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) ms[rt][0] = ms[rt][1] = ms[rt][2] = ms[rt][3] = 0.f;
  f32x4 tg[2][RT];
  tload(0, tg);

  for (int c = 0; c < nch; ++c) {
    // chunk c has landed: younger than its four loads are the 10 (6) other operations of iteration c - 2 and the 14 (10) of c - 1
    // (per iteration: 4 weight loads, 2 RT logits stores, 2 RT target loads, RT d(logits) stores)
    constexpr int PER_IT = 4 + (LOGITS ? 2 * RT : 0) + 2 * RT + RT;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PER_IT - 4) : "memory");
    __builtin_amdgcn_s_barrier();                                                    // B1
    HL_TR(4 * c);
    // (opaque copies per iteration: loop-invariant address arithmetic is not hoisted out of the loop and kept in registers)
    int li_c = li, g_c = g, tid_c = tid;
    asm volatile("" : "+v"(li_c), "+v"(g_c), "+v"(tid_c));
    const int cw0 = p.ch[c][0], cw1 = p.ch[c][1], cw2 = p.ch[c][2];
    const int col0 = cw0 & 0xffff, ncols = cw0 >> 16, kind = cw1 & 0xff, ckey = (cw1 >> 8) & 0xff;
    const bool cfirst = (cw1 >> 16) & 1, clast = (cw1 >> 24) & 1;
    wload(c + 2);                                                                    // 4 loads
    const unsigned char* wb = Ws + (c % 3) * HL_WS_B;
    // ---- 1. logits tile: acc[nt][rt] = rows 32 rp + 16 rt + li_c, columns col0 + (2 nh + nt) 16 + 4 g_c .. + 3
    f32x4 acc[2][RT];
    {
      const unsigned char* wa = wb + ((nh * 2) * 16 + li_c) * 512;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[nt][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[ks & 3] + (ks >> 2) * 256);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[rt][ks], acc[nt][rt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int cc = (nh * 2 + nt) * 16 + 4 * g_c;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(Bias + col0 + cc);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          acc[nt][rt] += bb;
          const int row = rp * (16 * RT) + rt * 16 + li_c;
          if (LOGITS) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[nt][rt]), rs_lg,
                                                 cc < ncols ? (unsigned int)(row0 + row) * ldb4 + (unsigned int)(col0 + cc) * 4u : OOB, 0, 0);   // 4 stores
        }
      }
    }
    HL_TR(4 * c + 1);
    if (kind == 0) {
      // ---- 2a. categorical chunk: logits -> LDS, active (row, item) pairs compacted; the dl image starts as zeros (every wave is
      // past B1, i.e. past the previous chunk's reads of it) -- the walk below writes the gradients of the ACTIVE items straight
      // into it as bf16 (round 5: the f32 write-back + barrier + tile -> image conversion pass cost ~0.9 us per categorical chunk)
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const int idx = tid_c + 512 * i;
        *reinterpret_cast<u32x4*>(Dli + (idx >> 3) * 128 + ((idx & 7) << 4)) = (u32x4){0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          *reinterpret_cast<f32x4*>(Tile + (rp * (16 * RT) + rt * 16 + li_c) * HL_TW + (nh * 2 + nt) * 16 + 4 * g_c) = acc[nt][rt];
      const int item0 = cw2 & 0xff, ni = (cw2 >> 8) & 0xff;
      for (int idx = tid_c; idx < RS * ni; idx += 512) {      // (RS ni is a multiple of the wave size: whole waves)
        const int row = idx & (RS - 1), j = idx / RS;
        const bool on = Wrow[Itab[item0 + j] * 128 + row] != 0;
        // one LDS atomic per wave (per-item atomics on the one counter were most of this phase)
        const unsigned long long bm = __ballot(on);
        int base = 0;
        if (lane == 0 && bm != 0ull) base = atomicAdd(Nact, __popcll(bm));
        base = __shfl(base, 0, 64);
        if (on) Act[base + __popcll(bm & ((1ull << lane) - 1ull))] = (unsigned short)(row | (j << 7));
      }
      if (c < 8) HL_TR(128 + 8 * c);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                                  // B2
      if (c < 8) HL_TR(128 + 8 * c + 1);
      const int na = *Nact;
      const int l16 = tid_c & 15;
      for (int a = tid_c >> 4; a < na; a += 32) {
        const int code = Act[a], row = code & 127, item = item0 + (code >> 7);
        const int kidx = Itab[item], C = Itab[32 + item];
        float* z = Tile + row * HL_TW + (Itab[16 + item] - col0);
        const int y = Ylab[row * 16 + item];
        // the item's classes (<= 64: four per lane) stay in registers from the one LDS read to the one LDS write
        float zv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) zv[q] = l16 + 16 * q < C ? z[l16 + 16 * q] : -INFINITY;
        float m = -INFINITY;
        int am = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (zv[q] > m) { m = zv[q]; am = l16 + 16 * q; }
        {      // max, and the FIRST index that holds it (as the serial reference walk)
          const float mm = row16_max(m);
          am = row16_min(m == mm ? am : 0x7fffffff);
          m = mm;
        }
        float se = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { zv[q] = l16 + 16 * q < C ? __builtin_amdgcn_exp2f((zv[q] - m) * 1.4426950408889634f) : 0.f; se += zv[q]; }
        se = row16_sum(se);
        const float inv = __builtin_amdgcn_rcpf(se);
        float sq = 0.f, qy = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = l16 + 16 * q;
          zv[q] *= inv;                                        // p_j
          const float qq = fminf(fmaxf(zv[q], 1e-7f), 1.f - 1e-7f);
          if (j < C) sq += qq;
          if (j == y) qy = qq;
        }
        sq = row16_sum(sq); qy = row16_sum(qy);
        const float loss = (__builtin_amdgcn_logf(sq) - __builtin_amdgcn_logf(qy)) * 0.6931471805599453f;
        const float inv_sq = __builtin_amdgcn_rcpf(sq), inv_qy = __builtin_amdgcn_rcpf(qy);
        float gg[4], gp = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = l16 + 16 * q;
          gg[q] = (j < C && zv[q] >= 1e-7f && zv[q] <= 1.f - 1e-7f) ? ((j == y ? -inv_qy : 0.f) + inv_sq) : 0.f;
          gp += gg[q] * zv[q];
        }
        gp = row16_sum(gp);
        // d(logits) of the item's classes -> the dl image (bf16; column cc of the chunk sits in 16-byte slot (cc >> 3) ^ isw(row))
        const int cc0 = Itab[16 + item] - col0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (l16 + 16 * q < C) {
            const int cc = cc0 + l16 + 16 * q;
            *reinterpret_cast<unsigned short*>(Dli + row * 128 + (((cc >> 3) ^ isw(row)) << 4) + (cc & 7) * 2) = f32_to_bf16(zv[q] * (gg[q] - gp) * p.inv_B);
          }
        if (l16 == 0) {
          atomicAdd(&Red[kidx * 3 + 0], loss * p.inv_B);
          atomicAdd(&Red[kidx * 3 + 1], am == y ? 1.f : 0.f);
          atomicAdd(&Red[kidx * 3 + 2], 1.f);
        }
      }
      if (c < 8) { HL_TR(128 + 8 * c + 2); HL_TR(128 + 8 * c + 3); }
    } else {
      // ---- 2b. numerical chunk, in the accumulator layout
      const int k = ckey;
      if (cfirst) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) ms[rt][0] = ms[rt][1] = ms[rt][2] = ms[rt][3] = 0.f;
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int row = rp * (16 * RT) + rt * 16 + li_c;
        const bool on = ((onm[rt] >> k) & 1u) != 0u;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          u32x2 pk = {0u, 0u};
          if (on) {
            float d[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pv = acc[nt][rt][r], yv = tg[nt][rt][r];
              d[r] = pv - yv;
              ms[rt][0] += d[r] * d[r]; ms[rt][1] += yv * yv; ms[rt][2] += pv * pv; ms[rt][3] += yv * pv;
              d[r] *= 2.f * p.inv_B;
            }
            pk = (u32x2){pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3])};
          }
          *reinterpret_cast<u32x2*>(Dli + row * 128 + ((((nh * 2 + nt) * 2 + (g_c >> 1)) ^ isw(row)) << 4) + (g_c & 1) * 8) = pk;
        }
      }
      if (clast) {
        // row sums: over the 4 lanes of a row in this wave, then the two column halves (waves nh = 0, 1) through LDS
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            ms[rt][q] += lane_xor16(ms[rt][q]);
            ms[rt][q] += lane_xor32(ms[rt][q]);
          }
        if (nh == 1 && g_c == 0) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            *reinterpret_cast<f32x4*>(Tile + (rp * (16 * RT) + rt * 16 + li_c) * 4) = (f32x4){ms[rt][0], ms[rt][1], ms[rt][2], ms[rt][3]};
        }
      }
    }
    if (c < 8) HL_TR(128 + 8 * c + 4);
    tload(c + 1, tg);                                                                // 4 loads: the next chunk's targets
    if (c < 8) HL_TR(128 + 8 * c + 5);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                                    // B4: the dl image is complete
    HL_TR(4 * c + 2);
    if (kind == 0 && tid_c == 0) *Nact = 0;      // (every thread read the count behind B2; the next compaction is behind B1)
    if (kind == 1 && clast && nh == 0) {
      const int k = ckey;
      float l0 = 0.f, l1 = 0.f, l2 = 0.f;
      if (g_c == 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row = rp * (16 * RT) + rt * 16 + li_c;
          if (Wrow[k * 128 + row]) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(Tile + row * 4);
            const float sd = ms[rt][0] + o[0], sy = ms[rt][1] + o[1], sp = ms[rt][2] + o[2], syp = ms[rt][3] + o[3];
            const float cosv = syp * rsqrtf(fmaxf(sy, 1e-12f)) * rsqrtf(fmaxf(sp, 1e-12f));
            l0 += sd * p.inv_B; l1 += 0.5f * cosv + 0.5f; l2 += 1.f;
          }
        }
      }
      l0 = wave_sum(l0); l1 = wave_sum(l1); l2 = wave_sum(l2);
      if (lane == 0 && l2 != 0.f) { atomicAdd(&Red[k * 3 + 0], l0); atomicAdd(&Red[k * 3 + 1], l1); atomicAdd(&Red[k * 3 + 2], l2); }
    }
    // ---- 3. dl image -> dlogits (16-byte pieces of the chunk's own columns), dx += dl_c W_c
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int idx = tid_c + 512 * i, row = idx >> 3, c8 = idx & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(Dli + row * 128 + ((c8 ^ isw(row)) << 4));
      __builtin_amdgcn_raw_buffer_store_b128(v, rs_dl, c8 * 8 < ncols ? (unsigned int)(row0 + row) * ldb2 + (unsigned int)(col0 + c8 * 8) * 2u : OOB, 0, 0);   // 2 stores
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 hf[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int row = rp * (16 * RT) + rt * 16 + li_c;
        // (k order of a transposing read: lane group g holds u = 32 ks + 4 g + j and 32 ks + 16 + 4 g + j, j = 0..3 --
        //  the dl operand is picked up in the same order: two 8-byte pieces)
        const int p0 = 8 * ks + g_c, p1 = p0 + 4;
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(Dli + row * 128 + (((p0 >> 1) ^ isw(row)) << 4) + (p0 & 1) * 8);
        const bf16x4 hi = *reinterpret_cast<const bf16x4*>(Dli + row * 128 + (((p1 >> 1) ^ isw(row)) << 4) + (p1 & 1) * 8);
        hf[rt] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
        // W_c^T fragment: for column n = tile * 16 + li_c the chunk rows {32 ks + 4 g_c + j} and {32 ks + 16 + 4 g_c + j}, j = 0..3
        const int tile = (ct >> 2) * 8 + nh * 4 + (ct & 3);
        const int wrow = 32 * ks + 4 * g_c + (li_c >> 2);
        const int P = tile * 4 + (li_c & 3);                        // 8-byte piece of the 512-byte row
        const unsigned char* ptr = wb + wrow * 512 + ((((P >> 1) ^ dsw(wrow)) << 4) | ((P & 1) << 3));
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)(ptr + 16 * 512));     // (row + 16: same swizzle)
        const bf16x8 wt = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc2[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt, hf[rt], acc2[ct][rt], 0, 0, 0);
      }
    }
    HL_TR(4 * c + 3);
  }

  HL_TR(253);
  // ---- dx (f32) and its dropout-masked, 1/keep-scaled bf16 copy (dgrad_rows_kernel's epilogue); the per-key partial sums
  {
    const float inv_keep = 1.0f / (1.0f - p.dropout_p);
    const unsigned int dthr = drop_thr16(p.dropout_p);
    const unsigned int dkey = drop_key(p.seed, p.offset + (unsigned long long)step_now * MFP_RNG_STEP_STRIDE);
    // The two bf16 copies leave in 16-BYTE pieces (8-byte stores run at about half the rate: MI355X_MICROARCH.md): the lanes (li, g)
    // and (li, g ^ 1) hold adjacent 4-column groups of the same two rows -- the even-g lane takes row tile 0 of both (its own four
    // columns + its neighbour's), the odd-g lane row tile 1; one row swap (v_permlane16_swap) per dword on the way.
    const bool odd = (g & 1) != 0;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const int n0c = ((ct >> 2) * 8 + nh * 4 + (ct & 3)) * 16;
      u32x2 pb[RT], pd[RT];      // [rt]: unmasked / dropout-masked bf16 of this lane's 4 columns
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int row = row0 + rp * (16 * RT) + rt * 16 + li, n = n0c + 4 * g;
        const f32x4 v = acc2[ct][rt];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_c, (unsigned int)row * (HL_D * 4) + n * 4, 0, 0);
        pb[rt] = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        bool keep[4] = {true, true, true, true};
        if (p.dropout_p > 0.f) drop_keep4(drop_row(dkey, (unsigned int)row), (unsigned int)n, dthr, keep);
        pd[rt] = (u32x2){pack_bf16x2(keep[0] ? v[0] * inv_keep : 0.f, keep[1] ? v[1] * inv_keep : 0.f),
                         pack_bf16x2(keep[2] ? v[2] * inv_keep : 0.f, keep[3] ? v[3] * inv_keep : 0.f)};
      }
      // send the row tile this lane does NOT store, receive the neighbour's share of the one it does
      // (RT = 1: one row tile -- the even-g lane stores its four columns + its neighbour's, the odd-g lane's stores go out of range)
      const u32x2 sb = (RT == 1 || odd) ? pb[0] : pb[RT - 1], sd = (RT == 1 || odd) ? pd[0] : pd[RT - 1];
      const u32x2 rb = {__float_as_uint(lane_xor16(__uint_as_float(sb[0]))), __float_as_uint(lane_xor16(__uint_as_float(sb[1])))};
      const u32x2 rd = {__float_as_uint(lane_xor16(__uint_as_float(sd[0]))), __float_as_uint(lane_xor16(__uint_as_float(sd[1])))};
      const u32x2 mb = (RT == 2 && odd) ? pb[RT - 1] : pb[0], md = (RT == 2 && odd) ? pd[RT - 1] : pd[0];
      const int row = row0 + rp * (16 * RT) + ((RT == 2 && odd) ? 16 : 0) + li, n8 = n0c + 8 * (g >> 1);
      const bool second = RT == 2 && odd;      // (this lane's own columns are the piece's second half)
      const u32x4 ob = second ? (u32x4){rb[0], rb[1], mb[0], mb[1]} : (u32x4){mb[0], mb[1], rb[0], rb[1]};
      const u32x4 od = second ? (u32x4){rd[0], rd[1], md[0], md[1]} : (u32x4){md[0], md[1], rd[0], rd[1]};
      const unsigned int o16 = (RT == 1 && odd) ? OOB : (unsigned int)row * (HL_D * 2) + n8 * 2;
      __builtin_amdgcn_raw_buffer_store_b128(ob, rs_cb, o16, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(od, rs_d, o16, 0, 0);
    }
  }
  __syncthreads();
  HL_TR(255);
  if (HL_ABL == 9) asm volatile("s_dcache_wb" ::: "memory");
  if (tid < 48) p.part[(long long)blockIdx.x * 48 + tid] = tid < p.nkeys * 3 ? Red[tid] : 0.f;
}

}  // namespace

extern "C" size_t mfp_heads_loss_partials(int32_t T) { return (size_t)((T + HL_ROWS - 1) / HL_ROWS); }
extern "C" size_t mfp_heads_loss_partials_half(int32_t T) { return (size_t)((T + 63) / 64); }

static int heads_loss_impl(int rows, const void* x, const void* W, const float* bias, int32_t U, const mfp_loss_key* keys,
                           int32_t nkeys, const int32_t* nvalid, float* part, void* dlogits, float* logits,
                           float* dx, void* dx_bf16, void* dx_drop, int32_t B, int32_t S, int32_t D, float dropout_p,
                           uint64_t seed, uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream) {
  MFP_CHECK_ARG(x && W && bias && keys && nvalid && part && dlogits && (dx || dx_bf16));
  MFP_CHECK_ARG(B > 0 && S > 0 && D == HL_D && U > 0 && U % 8 == 0 && U <= HL_MAXU && nkeys > 0 && nkeys <= MFP_MAX_LOSS_KEYS);
  MFP_CHECK_ARG((long long)B * S <= (1 << 20) && dropout_p >= 0.f && dropout_p < 1.f);
  // the kernel addresses logits / dlogits / targets with 32-bit byte offsets and buffer sizes (num_records)
  MFP_CHECK_ARG((long long)B * S * U * (logits != nullptr ? 4 : 2) < 0xFFFFFFF0ll);
  for (int i = 0; i < nkeys; ++i)
    MFP_CHECK_ARG(!keys[i].is_numerical || (long long)B * S * keys[i].n_class * 4 < 0xFFFFFFF0ll);
  MFP_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)dlogits % 16) == 0 && ((uintptr_t)logits % 16) == 0 &&
                ((uintptr_t)dx % 16) == 0 && ((uintptr_t)dx_bf16 % 16) == 0 && ((uintptr_t)dx_drop % 16) == 0 && ((uintptr_t)bias % 16) == 0);
  HlParams p;
  p.X = reinterpret_cast<const unsigned short*>(x); p.W = reinterpret_cast<const unsigned short*>(W); p.bias = bias;
  p.nvalid = nvalid; p.part = part; p.dlogits = reinterpret_cast<unsigned short*>(dlogits); p.logits = logits;
  p.dX = dx; p.dXb = reinterpret_cast<unsigned short*>(dx_bf16); p.dXd = reinterpret_cast<unsigned short*>(dx_drop); p.step_ptr = step_ptr; p.seed = seed; p.offset = offset;
  p.nkeys = nkeys; p.T = B * S; p.S = S; p.U = U; p.ld = U; p.inv_B = 1.0f / (float)B; p.dropout_p = dropout_p;
  // heads in column order; categorical items (key, feature) and the chunk table
  int order[MFP_MAX_LOSS_KEYS];
  for (int i = 0; i < nkeys; ++i) { p.key[i] = keys[i]; order[i] = i; }
  for (int i = nkeys; i < MFP_MAX_LOSS_KEYS; ++i) p.key[i] = keys[0];
  for (int i = 1; i < nkeys; ++i)
    for (int j = i; j > 0 && keys[order[j]].col_off < keys[order[j - 1]].col_off; --j) { int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  p.nitem = 0; p.nch = 0;
  int prev_end = 0;
  short ncols[HL_MAXCH]; unsigned char nitem_c[HL_MAXCH]; unsigned char pm[HL_MAXCH][8];
  auto new_chunk = [&](int col0, int kind, int key) -> int {
    if (p.nch >= HL_MAXCH) return -1;
    const int c = p.nch++;
    p.ch[c][0] = col0; p.ch[c][1] = kind | (key << 8); p.ch[c][2] = p.nitem;
    ncols[c] = 0; nitem_c[c] = 0;
    for (int e = 0; e < 8; ++e) pm[c][e] = 0xFF;
    return c;
  };
  int cur = -1;      // open categorical chunk
  for (int oi = 0; oi < nkeys; ++oi) {
    const int ki = order[oi];
    const mfp_loss_key& k = keys[ki];
    MFP_CHECK_ARG(k.col_off >= prev_end && k.col_off % 8 == 0 && k.n_class > 0 && k.n_feat > 0);   // heads must not overlap
    prev_end = k.col_off + k.n_feat * k.n_class;
    MFP_CHECK_ARG(prev_end <= U);
    if (k.is_numerical) {
      MFP_CHECK_ARG(k.n_feat == 1 && k.n_class % 8 == 0);
      cur = -1;
      for (int c0 = 0; c0 < k.n_class; c0 += 64) {
        const int c = new_chunk(k.col_off + c0, 1, ki);
        MFP_CHECK_ARG(c >= 0);
        ncols[c] = (short)(k.n_class - c0 < 64 ? k.n_class - c0 : 64);
        if (c0 == 0) p.ch[c][1] |= 1 << 16;
        if (c0 + 64 >= k.n_class) p.ch[c][1] |= 1 << 24;
      }
      continue;
    }
    for (int f = 0; f < k.n_feat; ++f) {
      const int col = k.col_off + f * k.n_class;
      MFP_CHECK_ARG(col % 8 == 0 && k.n_class <= 64 && p.nitem < HL_MAXIT);     // items sit on 8-column boundaries (ModelLayout pads)
      const int end8 = (col + k.n_class + 7) / 8 * 8;
      if (cur < 0 || end8 - p.ch[cur][0] > 64) {
        cur = new_chunk(col, 0, 0);
        MFP_CHECK_ARG(cur >= 0);
      }
      const int it = p.nitem++;
      p.item_key[it] = ki; p.item_feat[it] = f; p.item_col[it] = col; p.item_C[it] = k.n_class;
      nitem_c[cur]++;
      ncols[cur] = (short)((end8 < U ? end8 : U) - p.ch[cur][0]);
      for (int c8 = (col - p.ch[cur][0]) / 8; c8 < (end8 - p.ch[cur][0]) / 8 && c8 < 8; ++c8) pm[cur][c8] = (unsigned char)it;
    }
  }
  for (int it = p.nitem; it < HL_MAXIT; ++it) { p.item_key[it] = 0; p.item_feat[it] = 0; p.item_col[it] = 0; p.item_C[it] = 0; }
  for (int c = 0; c < HL_MAXCH; ++c) {
    if (c >= p.nch) { p.ch[c][0] = U; p.ch[c][1] = 0; p.ch[c][2] = 0; ncols[c] = 0; nitem_c[c] = 0; for (int e = 0; e < 8; ++e) pm[c][e] = 0xFF; }
    p.ch[c][0] |= (int)ncols[c] << 16;
    p.ch[c][2] |= (int)nitem_c[c] << 8;
    p.ch[c][3] = (int)((unsigned)pm[c][0] | ((unsigned)pm[c][1] << 8) | ((unsigned)pm[c][2] << 16) | ((unsigned)pm[c][3] << 24));
    p.ch[c][4] = (int)((unsigned)pm[c][4] | ((unsigned)pm[c][5] << 8) | ((unsigned)pm[c][6] << 16) | ((unsigned)pm[c][7] << 24));
    p.ch[c][5] = p.ch[c][6] = p.ch[c][7] = 0;
  }
  MFP_CHECK_ARG(p.nch > 0);
  static bool attr_done[MFP_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mfp_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(heads_loss_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, HL_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(heads_loss_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, HL_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(heads_loss_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, HL_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(heads_loss_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, HL_LDS);
    if (e != hipSuccess) {
      mfp_set_error("mfp_heads_loss_fwd_bwd: cannot raise dynamic LDS to %d: %s", HL_LDS, hipGetErrorString(e));
      return MFP_ELAUNCH;
    }
    attr_set = true;
  }
  const dim3 grid((p.T + rows - 1) / rows);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (rows == 64) {
    if (logits != nullptr) hipLaunchKernelGGL((heads_loss_kernel<true, 1>), grid, dim3(512), HL_LDS, st, p);
    else hipLaunchKernelGGL((heads_loss_kernel<false, 1>), grid, dim3(512), HL_LDS, st, p);
  } else {
    if (logits != nullptr) hipLaunchKernelGGL((heads_loss_kernel<true, 2>), grid, dim3(512), HL_LDS, st, p);
    else hipLaunchKernelGGL((heads_loss_kernel<false, 2>), grid, dim3(512), HL_LDS, st, p);
  }
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_heads_loss_fwd_bwd(const void* x, const void* W, const float* bias, int32_t U, const mfp_loss_key* keys,
                                      int32_t nkeys, const int32_t* nvalid, float* part, void* dlogits, float* logits,
                                      float* dx, void* dx_bf16, void* dx_drop, int32_t B, int32_t S, int32_t D, float dropout_p,
                                      uint64_t seed, uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream) {
  return heads_loss_impl(HL_ROWS, x, W, bias, U, keys, nkeys, nvalid, part, dlogits, logits, dx, dx_bf16, dx_drop, B, S, D, dropout_p, seed,
                         offset, step_ptr, stream);
}

// The same launch on 64-row tiles (one eight-wave workgroup per 64 rows; `part` holds mfp_heads_loss_partials_half(B * S) rows):
// batches with fewer 128-row tiles than CUs (BASELINE config c4: 128 documents per GPU).  logits / dlogits / dx bit-identical.
extern "C" int mfp_heads_loss_fwd_bwd_half(const void* x, const void* W, const float* bias, int32_t U, const mfp_loss_key* keys,
                                           int32_t nkeys, const int32_t* nvalid, float* part, void* dlogits, float* logits,
                                           float* dx, void* dx_bf16, void* dx_drop, int32_t B, int32_t S, int32_t D, float dropout_p,
                                           uint64_t seed, uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream) {
  return heads_loss_impl(64, x, W, bias, U, keys, nkeys, nvalid, part, dlogits, logits, dx, dx_bf16, dx_drop, B, S, D, dropout_p, seed,
                         offset, step_ptr, stream);
}
