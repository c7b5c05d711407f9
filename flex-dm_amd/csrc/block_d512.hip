// Dense layers of a DeepSVG block at d_model 512 (BASELINE config c5: Crello Ours-EXP-FT, 8 blocks, S = 256; reference
// architecture/transformer.py:211-229, 60-99, 161-171 and their autodiff).  Round 5: c5 ran every product on the generic
// kernels (ln_fwd + weight-stationary / LDS-tiled GEMMs); the d_model-256 machine of csrc/block_fused.hip does not carry
// over as it is -- a 128-row tile's operand fragments for K = 512 take 128 registers per lane, and 16 384 tokens are only
// 128 such tiles.  Two kernels, duals of each other, both streaming the weights L2 -> LDS by LDS-DMA:
//
//  as512_kernel  ACTIVATION-stationary, contraction 512 (the block's products whose INPUT is d_model wide):
//      out[T][N] = epi( A'[T][512] W[N][512]^T ),   A' = LayerNorm(x) (x f32; y, mean, rstd saved) or a bf16 matrix
//      LN1 + Q|K|V (N = 1536), LN2 + FFN1 + ReLU (N = 1024), dh = (d_o2 W2) * [h > 0] (N = 1024).
//      One 8-wave workgroup per (128-row tile, column range): grid (T / 128, NSPLIT) -- the column split is what fills the
//      chip at 128 tiles; both column halves of a tile sit on one XCD (x re-read from its L2), LayerNorm is recomputed.
//      Wave (rp, nh) keeps the fragments of its 32 rows for all 512 k in 128 registers; a weight chunk is 64 output
//      columns x 256 k (32 KB, [64][512 B] image, slot ^ (row & 15)): two chunks per column group, accumulators carried
//      across the two; three ring buffers, chunk c + 2 in flight while chunk c multiplies; the bf16 result of a column
//      group leaves through a [128][128 B] LDS image in 128-byte row pieces at the head of the next chunk.
//
//  os512_kernel  OUTPUT-stationary, 512 output columns (the products whose OUTPUT is d_model wide):
//      out[T][512] = epi( A[T][K] W[512][K]^T ),   K = 512, 1024, 1536
//      attention output projection and FFN2 (f32 out = residual + Dropout(. + bias), optional bf16 copy), and the input
//      gradients da = d_o1 Wo, dy2 = dh W1, dy1 = dqkv Wqkv (bf16 out).
//      One 8-wave workgroup per 128 rows x 256 columns (grid (T / 128, 2): 256 workgroups at 16 384 tokens): wave (rp, nh)
//      holds 64 rows x 64 columns in 64 accumulator registers; A and W stream through three 48 KB LDS stages of 64 k
//      ([128][128 B] + [256][128 B] images, slot ^ ((row >> 1) & 7)), two stages ahead; the result goes through an f32 LDS
//      image and leaves in whole row pieces (1 KB contiguous per wave instruction).
//
// Every global access is a buffer instruction with a 32-bit offset (rows >= T read zeros and drop their writes); waits
// for LDS-DMA are counted (memory operations of a wave retire in order, csrc/block_fused.hip).
#include "common.h"
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) unsigned char lds_u8;

// ===================================================================================================== as512
constexpr int A5_K = 512, A5_ROWS = 128;
constexpr int A5_WS_B = 32768;                       // one weight chunk: [64 columns][512 B = 256 k]; three ring buffers
constexpr int A5_AUX_OFF = 3 * A5_WS_B;              // two [128][128 B] images of the mask operand (column group parity)
constexpr int A5_OUT_B = A5_ROWS * 128;              // out image of a column group: [128][128 B]
constexpr int A5_OUT_OFF = A5_AUX_OFF + 2 * A5_OUT_B;      // = 128 KB: the A' image [128][1024 B] occupies ring + aux before the first chunk
constexpr int A5_VEC_OFF = A5_OUT_OFF + A5_OUT_B;    // bias of this workgroup's columns (<= 768) | gamma (512) | beta (512)
constexpr int A5_MAXCOLS = 768;
constexpr int A5_LDS = A5_VEC_OFF + (A5_MAXCOLS + 2 * A5_K) * 4;      // 154 624 B

enum { A5_SRC_LN = 0, A5_SRC_BF16 = 1 };

// D5_TRACE build (tools/abl/build_abl.sh block_d512 D5_TRACE 1; tools/trace_d512.py): every wave drops shader-clock stamps at its
// phase boundaries through SCALAR stores (no vector-memory operation added: the counted waits are untouched)
#ifndef D5_TRACE
#define D5_TRACE 0
#endif
// D5_ABL (timing by elimination, results WRONG): 1 no epilogue pieces, 2 no image stores, 3 no products, 4 no y stores
#ifndef D5_ABL
#define D5_ABL 0
#endif
#if D5_TRACE
static unsigned long long* g_d5_trace = nullptr;
#define D5_TR(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); const unsigned int o_ = (unsigned int)(i) * 8u; \
    asm volatile("s_store_dwordx2 %0, %1, %2 glc" :: "s"(t_), "s"(trbase), "s"(o_) : "memory"); } while (0)
#define D5_TR_END() asm volatile("s_dcache_wb" ::: "memory")
#else
#define D5_TR(i) do {} while (0)
#define D5_TR_END() do {} while (0)
#endif

enum { A5_EPI_BIAS = 0, A5_EPI_RELU = 1, A5_EPI_MASK = 2 };

struct As512Params {
  const float* x; const float* gamma; const float* beta;       // A5_SRC_LN: x f32 [T][512]
  const unsigned short* A; int lda;                             // A5_SRC_BF16: bf16 [T][lda], 512 columns used
  const unsigned short* W;                                      // [N][512] bf16 (out, in)
  const float* bias;                                            // [N] or nullptr
  const unsigned short* aux; int ldaux;                         // A5_EPI_MASK: bf16 [T][ldaux] (the saved ReLU output)
  unsigned short* y; float* mean; float* rstd;                  // A5_SRC_LN: LN(x) bf16 [T][512], statistics (column range 0 writes them)
  unsigned short* out; int ldo;                                 // bf16 [T][ldo]
  int T, N, ncols;                                              // ncols = columns per blockIdx.y (a multiple of 128, <= 768)
  float eps;
  unsigned long long* trace;                                    // D5_TRACE builds: [workgroup][8 waves][64] clock stamps
};

// XH (LayerNorm form only, round 5): the y buffer receives x-hat = (x - mean) rstd (bf16) instead of LN(x) -- straight from the
// operand layout, BEFORE the first weight chunk is requested (so the stores are older than every counted load and P_STORES = 0)
template <int ASRC, int EPI, bool XH = false>
__global__ __launch_bounds__(512) void as512_kernel(As512Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Ws = smem;
  unsigned char* const Os = smem + A5_OUT_OFF;
  float* const Bv = reinterpret_cast<float*>(smem + A5_VEC_OFF);
  const float* const Gs = Bv + A5_MAXCOLS;
  const float* const Bs = Gs + A5_K;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rp = wave & 3, nh = wave >> 2;
  const int row0 = blockIdx.x * A5_ROWS, n0 = blockIdx.y * p.ncols;
  const int nch = p.ncols >> 5;                      // chunks: 2 per 64-column group
#if D5_TRACE
  const unsigned long long* trbase = p.trace + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 64;
#endif
  D5_TR(0);
  constexpr unsigned int OOB = 0x40000000u;          // added to a 32-bit offset: beyond every buffer here (all < 1 GB)

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.W), 0, (unsigned int)p.N * (A5_K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (unsigned int)p.T * (unsigned int)(p.ldo * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ASRC == A5_SRC_LN ? p.x : p.gamma), 0,
                                                                        ASRC == A5_SRC_LN ? (unsigned int)p.T * (A5_K * 4) : 0u, 0x00020000);
  // (only column range 0 writes y: the others issue the same stores into a zero-sized buffer -- the counted waits stay uniform)
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(ASRC == A5_SRC_LN ? p.y : p.out, 0,
                                                                        (ASRC == A5_SRC_LN && blockIdx.y == 0) ? (unsigned int)p.T * (A5_K * 2) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(ASRC == A5_SRC_BF16 ? p.A : p.W), 0,
                                                                        ASRC == A5_SRC_BF16 ? (unsigned int)p.T * (unsigned int)(p.lda * 2) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_aux = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(EPI == A5_EPI_MASK ? p.aux : p.W), 0,
                                                                          EPI == A5_EPI_MASK ? (unsigned int)p.T * (unsigned int)(p.ldaux * 2) : 0u, 0x00020000);

  // ---- weight chunk c = 2 cg + kh -> ring buffer c % 3: W rows n0 + 64 cg .. + 63, k = 256 kh .. + 255 as a [64][512 B] image,
  // 16-byte slot ^ (row & 15).  Four 1 KB pieces per wave (2 rows each); the source slot of a lane is its destination slot ^ row.
  const unsigned int w_off0 = (unsigned int)((wave * 8 + (lane >> 5)) * (A5_K * 2) + (((lane & 31) ^ ((wave & 1) * 8 + (lane >> 5))) << 4));
  auto wload = [&](int c, int buf) {
    const unsigned int so = (unsigned int)((n0 + (c >> 1) * 64) * (A5_K * 2) + (c & 1) * 512);
    const unsigned int oob = c < nch ? 0u : OOB;       // past the last chunk: zeros into a buffer nobody reads (uniform op count)
    unsigned char* dst = Ws + buf * A5_WS_B + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_u8*)(dst + i * 1024), 16, ((w_off0 ^ (i << 5)) + i * (2 * A5_K * 2)) + oob, so, 0, 0);
  };
  // ---- mask form: the aux piece of column group cg (128 rows x 128 B) -> aux image cg & 1, laid out like the out image
  // (slot ^ (row & 7)): two 1 KB pieces per wave (8 rows each).  In LDS, not in registers: a loaded VALUE makes the compiler
  // place its own waits (vmcnt(0) behind the ring's loads), an LDS-DMA piece is counted by ours.
  const unsigned int x_off0 = (unsigned int)(row0 + wave * 16 + (lane >> 3)) * (unsigned int)(p.ldaux * 2) +
                              (unsigned int)(n0 * 2 + (((lane & 7) ^ (lane >> 3)) << 4));
  auto xload = [&](int cg) {
    const unsigned int oob = cg < (nch >> 1) ? 0u : OOB;
    unsigned char* dst = smem + A5_AUX_OFF + (cg & 1) * A5_OUT_B + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_aux, (lds_u8*)(dst + i * 1024), 16, x_off0 + (unsigned int)i * 8u * (unsigned int)(p.ldaux * 2) + oob,
                                               (unsigned int)cg * 128u, 0, 0);
  };

  // ---- per-column vectors -> LDS
  if (tid < p.ncols / 4) {
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) b = *reinterpret_cast<const f32x4*>(p.bias + n0 + tid * 4);
    *reinterpret_cast<f32x4*>(Bv + tid * 4) = b;
  }
  if (ASRC == A5_SRC_LN && tid < 256) {
    const float* src = tid < 128 ? p.gamma + tid * 4 : p.beta + (tid - 128) * 4;
    *reinterpret_cast<f32x4*>(smem + A5_VEC_OFF + A5_MAXCOLS * 4 + tid * 16) = *reinterpret_cast<const f32x4*>(src);
  }

  bf16x8 xf[2][16];
  constexpr int P_STORES = (ASRC == A5_SRC_LN && !XH) ? 16 : 0;      // y row stores issued behind the first three weight chunks
  if constexpr (ASRC == A5_SRC_LN) {
    // LayerNorm in the MFMA operand layout: wave w normalises rows 16 w .. + 15, lane (li, g) holds row li's columns
    // 32 ks + 8 g .. + 7 (statistics = two cross-lane steps); the bf16 result passes through a [128][1024 B] image
    {
      const int lrow = wave * 16 + li, row = row0 + lrow;
      float v[16][8];
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const unsigned int vo = (unsigned int)row * (A5_K * 4) + ks * 128 + g * 32;
        const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, 0, 0));
        const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo + 16, 0, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[ks][e] = a[e]; v[ks][4 + e] = b[e]; s += a[e] + b[e]; }
      }
      __syncthreads();      // gamma / beta are in LDS
      s += lane_xor16(s);
      s += lane_xor32(s);
      const float mu = s * (1.0f / A5_K);
      float qq = 0.f;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[ks][e] -= mu; qq += v[ks][e] * v[ks][e]; }
      qq += lane_xor16(qq);
      qq += lane_xor32(qq);
      const float rs = rsqrtf(qq * (1.0f / A5_K) + p.eps);
      if (g == 0 && row < p.T && blockIdx.y == 0) { p.mean[row] = mu; p.rstd[row] = rs; }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int col = ks * 32 + 8 * g;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(Gs + col), g1 = *reinterpret_cast<const f32x4*>(Gs + col + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + col), b1 = *reinterpret_cast<const f32x4*>(Bs + col + 4);
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = v[ks][e] * rs;      // x-hat
        if constexpr (XH) {      // a lane's 8 columns = 16 bytes; the four g-lanes of a row cover 64 contiguous bytes
          const u32x4 px = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7])};
          __builtin_amdgcn_raw_buffer_store_b128(px, rs_y, (unsigned int)row * (A5_K * 2) + ks * 64 + g * 16, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { y[e] = y[e] * g0[e] + b0[e]; y[4 + e] = y[4 + e] * g1[e] + b1[e]; }
        const u32x4 pk = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7])};
        *reinterpret_cast<u32x4*>(smem + lrow * 1024 + (((ks * 4 + g) ^ li) << 4)) = pk;
      }
    }
    __syncthreads();
  } else {
    // bf16 rows straight into the image by LDS-DMA: one 1 KB row per wave instruction (lane = destination slot)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = wave * 16 + i;
      const unsigned int vo = (unsigned int)(row0 + r) * (unsigned int)(p.lda * 2) + (unsigned int)((lane ^ i) << 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_u8*)(smem + r * 1024), 16, vo, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // this wave's operand fragments (rows 32 rp .. + 31, all 512 k) and, for the LayerNorm form, the y rows it will store
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      xf[rt][ks] = *reinterpret_cast<const bf16x8*>(smem + (rp * 32 + rt * 16 + li) * 1024 + (((ks * 4 + g) ^ li) << 4));
  u32x4 yv[P_STORES ? P_STORES : 1];
  if constexpr (ASRC == A5_SRC_LN && !XH) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = wave + 8 * i;                      // (tid + 512 i) >> 6
      yv[i] = *reinterpret_cast<const u32x4*>(smem + r * 1024 + ((lane ^ (r & 15)) << 4));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();       // the image has been read by everyone: the ring may be filled
  D5_TR(1);
  wload(0, 0);
  wload(1, 1);
  if (EPI == A5_EPI_MASK) xload(0);
  // (compiler fence: the counted waits below assume this ISSUE order -- without it hipcc moved 12 of the 16 y stores in front
  //  of the weight loads, and the first group's wait let chunk 1 be read before its last pieces had landed)
  asm volatile("" ::: "memory");
  if constexpr (ASRC == A5_SRC_LN && !XH) {
    // y leaves in whole rows (1 KB per wave instruction), BEHIND the first weight chunks
#pragma unroll
    for (int i = 0; i < 16; ++i)
      __builtin_amdgcn_raw_buffer_store_b128(yv[i], rs_y, D5_ABL == 4 ? OOB : (unsigned int)(row0 + wave + 8 * i) * (A5_K * 2) + lane * 16, 0, 0);
  }
  asm volatile("" ::: "memory");
  constexpr int EA = EPI == A5_EPI_MASK ? 2 : 0;
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + EA + P_STORES) : "memory");      // chunk 0 has landed (issued before chunk 1, the aux piece and the y stores)
  __builtin_amdgcn_s_barrier();

  int xs[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xs[ks] = ((ks * 4 + g) ^ li) << 4;
  auto drain = [&](int cgp) {      // the out image's rows -> HBM in 128-byte pieces (cgp < 0: out of range, dropped)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i, r = idx >> 3, c16 = idx & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(Os + r * 128 + ((c16 ^ (r & 7)) << 4));
      const unsigned int off = (unsigned int)(row0 + r) * (unsigned int)(p.ldo * 2) + (unsigned int)((n0 + cgp * 64) * 2 + c16 * 16);
      __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, (cgp >= 0 && D5_ABL != 2) ? off : OOB, 0, 0);
    }
  };
  // one (nt, rt) piece of a finished column group: + bias (ReLU | mask) -> bf16 -> out image (4 consecutive columns per lane)
  auto epi_piece = [&](int cgp, const f32x4 (&accp)[2][2], int nt, int rt) {
    f32x4 o = accp[nt][rt];
    if (EPI == A5_EPI_RELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
    }
    u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    const int io = (rp * 32 + rt * 16 + li) * 128 + ((((nh * 2 + nt) * 2 + (g >> 1)) ^ (li & 7)) << 4) + (g & 1) * 8;
    if (EPI == A5_EPI_MASK) {      // the lane's 4 values against the matching 8 bytes of the aux image (> 0 <=> bits != 0: a ReLU output)
      const u32x2 a = *reinterpret_cast<const u32x2*>(smem + A5_AUX_OFF + (cgp & 1) * A5_OUT_B + io);
#pragma unroll
      for (int e = 0; e < 2; ++e) pk[e] &= ((a[e] & 0xFFFFu) ? 0xFFFFu : 0u) | ((a[e] >> 16) ? 0xFFFF0000u : 0u);
    }
    *reinterpret_cast<u32x2*>(Os + io) = pk;
  };
  // 32 products of a chunk; EPI_PREV: the four epilogue pieces of the PREVIOUS column group are issued between the k-steps
  // (trace, first version: the epilogue behind the last product cost 0.4 us per column group -- ~50 VALU instructions with the
  // matrix pipe idle, both waves of a SIMD in the same phase behind the barrier)
  auto product = [&](auto kh_, auto prev_, int buf, f32x4 (&acc)[2][2], int cgp, const f32x4 (&accp)[2][2]) {
    constexpr int kh = decltype(kh_)::value;
    constexpr bool EPI_PREV = decltype(prev_)::value;
    const unsigned char* wa = Ws + buf * A5_WS_B + ((nh * 2) * 16 + li) * 512;
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
          if (D5_ABL != 3) acc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks % 3][nt], xf[rt][kh * 8 + ks], acc[nt][rt], 0, 0, 0);
      if (EPI_PREV && (ks & 1) && D5_ABL != 1) epi_piece(cgp, accp, ks >> 2, (ks >> 1) & 1);
    }
  };
  // Column group cg: chunks 2 cg (ring buffer rb) and 2 cg + 1 (ring buffer rb + 1 mod 3).  Vector-memory operations in issue
  // order (all LDS-DMA or stores: no loaded value is in flight, so every wait is one of ours):
  //   chunk 2 cg     (kh = 0): the 4 weight pieces of chunk 2 cg + 2            [+ the epilogue of group cg - 1 -> out image]
  //   chunk 2 cg + 1 (kh = 1): the 4 weight pieces of chunk 2 cg + 3, [mask form: the 2 aux pieces of group cg + 1], the two
  //                            image stores of group cg - 1
  // At the end of a chunk the next chunk's weights must have landed -- no more operations outstanding than were issued behind
  // them (in-order retirement); at the end of chunk 2 cg also the aux pieces of cg (issued in chunk 2 cg - 1, read by the
  // epilogue of cg in chunk 2 cg + 2).
  f32x4 accp[2][2];
  auto group = [&](auto first_, int cg, int rb) {
    constexpr bool FIRST = decltype(first_)::value;
    const int rb1 = rb == 2 ? 0 : rb + 1, rb2 = rb1 == 2 ? 0 : rb1 + 1;
    // (the accumulators start from the bias: the epilogue's ~20 VALU instructions per piece do not overlap the products -- timing by
    //  elimination, D5_ABL: 0.35 us per column group with both waves of a SIMD in it -- so every instruction taken out of it counts)
    f32x4 acc[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[nt][0] = acc[nt][1] = *reinterpret_cast<const f32x4*>(Bv + cg * 64 + (nh * 2 + nt) * 16 + 4 * g);
    // ---- kh = 0
    wload(2 * cg + 2, rb2);
    asm volatile("" ::: "memory");
    product(std::integral_constant<int, 0>{}, std::integral_constant<bool, !FIRST>{}, rb, acc, cg - 1, accp);
    // chunk 2 cg + 1 and the aux pieces of cg: behind them the 2 image stores of chunk 2 cg - 1 and this chunk's 4 pieces
    // (first group: the prologue's y stores and this chunk's 4)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(FIRST ? 4 + P_STORES : 6) : "memory");
    __builtin_amdgcn_s_barrier();
    D5_TR(2 + 2 * (cg < 14 ? cg : 14));
    // ---- kh = 1
    wload(2 * cg + 3, rb);
    if (EPI == A5_EPI_MASK) xload(cg + 1);
    asm volatile("" ::: "memory");      // (issue order: the image stores BEHIND the weight / aux pieces -- the waits count them)
    drain(cg - 1);
    asm volatile("" ::: "memory");
    product(std::integral_constant<int, 1>{}, std::integral_constant<bool, false>{}, rb1, acc, 0, accp);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) accp[nt][rt] = acc[nt][rt];
    // chunk 2 cg + 2 (issued in chunk 2 cg): behind it this chunk's 4 weight pieces, aux pieces and 2 image stores
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(6 + EA) : "memory");
    __builtin_amdgcn_s_barrier();
    D5_TR(3 + 2 * (cg < 14 ? cg : 14));
  };
  group(std::integral_constant<bool, true>{}, 0, 0);
  int rb = 2;
  for (int cg = 1; cg < (nch >> 1); ++cg) {
    group(std::integral_constant<bool, false>{}, cg, rb);
    rb = rb == 0 ? 2 : rb - 1;      // (+ 2 mod 3)
  }
  // the last column group's epilogue and image stores
#pragma unroll
  for (int q = 0; q < 4; ++q) epi_piece((nch >> 1) - 1, accp, q >> 1, q & 1);
  __syncthreads();
  drain((nch >> 1) - 1);
  // (the out-of-range weight / aux pieces of the last group are LDS-DMA too: none may be in flight when the workgroup's LDS is
  //  handed on -- everything but the two stores just issued has retired after this)
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  D5_TR(40);
  D5_TR_END();
}

// ===================================================================================================== os512
// Output tile 128 rows x 256 columns (half of the 512 output columns: grid (T / 128, 2), both halves of a row tile on one XCD).
// First version (round 5, measured): 64 rows x all 512 columns per workgroup -- 64 KB of W per 64 k and workgroup made the
// stage L2-stream-bound (1.03-1.2 us per stage against 0.62 us of products, tools/trace_d512.py) and the step SLOWER than the
// generic kernels (4.68 vs 4.49 ms).  128 x 256 moves 48 KB per stage (A 16 KB + W 32 KB) for the same products, and three
// stages fit the LDS: stage s + 2 is in flight while stage s multiplies.
constexpr int O5_N = 512, O5_ROWS = 128, O5_COLS = 256;
constexpr int O5_A_B = O5_ROWS * 128, O5_W_B = O5_COLS * 128, O5_STAGE = O5_A_B + O5_W_B;      // 16 KB + 32 KB per 64 k
constexpr int O5_VEC_OFF = 3 * O5_STAGE;            // bias of this column half (1 KB)
constexpr int O5_LDS = O5_VEC_OFF + O5_COLS * 4;    // 148 480 B
constexpr int O5_EROW = O5_COLS * 4 + 16;           // f32 result image: 1 KB rows + 16 B (the accumulator layout's writes then spread over all banks)

// O5_EPI_LNB (round 5): the product is the gradient of a LayerNorm OUTPUT (dy2 = dh W1, dy1 = dqkv Wqkv) and the LayerNorm
// backward runs on the f32 result image -- dy never reaches HBM and the stand-alone ln_bwd launch behind it is gone.  A tile owns
// HALF of a row's 512 columns, so the two row statistics (mean_c(dy gamma), mean_c(dy gamma xhat)) are exchanged between the
// two workgroups of a row tile: 128 x 2 partial sums each way through a global scratch (write-through stores, drained, then a
// relaxed agent-scope flag; the reader spins on the partner's flag, does one agent acquire and resets it -- the hand-off form of
// gemm_wgg.h).  The pair sits 8 apart in a group of 16 consecutive workgroups (same XCD, adjacent in dispatch order).
enum { O5_EPI_RES = 0, O5_EPI_BF16 = 1, O5_EPI_LNB = 2 };

struct Os512Params {
  const unsigned short* A; int lda; int K;          // bf16 [T][lda], K columns used (a multiple of 64)
  const unsigned short* W;                          // [512][K] bf16
  const float* bias;                                // [512] or nullptr
  const float* res; float* out; unsigned short* outc;      // O5_EPI_RES: out = res + Dropout(acc + bias) f32 [T][512] (+ bf16 copy or nullptr)
  unsigned short* outb;                             // O5_EPI_BF16: bf16 [T][512]
  // O5_EPI_LNB: x-hat form of the LayerNorm backward on the result (ln_bwd_tile.h states the arithmetic), bf16 residual-gradient stream
  const unsigned short* xhat; const float* gamma; const float* rstd;      // bf16 [T][512] = (x - mean) rstd, f32 [512], f32 [T]
  const unsigned short* dres; unsigned short* dx; unsigned short* ddrop;  // bf16 [T][512]: residual gradient in, dx out, masked copy out (or nullptr)
  float* part;                                      // f32 [T / 128][3][512]: per-tile sums of dy xhat | dy | ddrop
  float* exch; int* flags;                          // f32 [T / 128][2][128][2] scratch; int [T / 128][2], zero between launches
  int T;
  float dropout_p; unsigned long long seed, offset; const int* step_ptr;
  unsigned long long* trace;                        // D5_TRACE builds
};

template <int EPI, bool DROPOUT>
__global__ __launch_bounds__(512) void os512_kernel(Os512Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const Bv = reinterpret_cast<float*>(smem + O5_VEC_OFF);
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave (rp, nh) owns 64 rows x 64 columns (4 x 4 MFMA tiles): 8 fragment reads per 16 products -- at 32 x 128 (10 reads) the
  // stage was bound by the LDS port (208 KB of reads + LDS-DMA writes per stage = 0.78 us against 0.62 us of products)
  const int rp = wave & 1, nh = wave >> 1;
  // (O5_EPI_LNB: the two column halves of a row tile are workgroups b and b + 8 of a group of 16 in dispatch order)
  const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
  const bool paired = EPI == O5_EPI_LNB && (gridDim.x & 7) == 0;
  const int tile = EPI != O5_EPI_LNB ? (int)blockIdx.x : paired ? (lin >> 4) * 8 + (lin & 7) : lin >> 1;
  const int half = EPI != O5_EPI_LNB ? (int)blockIdx.y : paired ? (lin >> 3) & 1 : lin & 1;
  const int row0 = tile * O5_ROWS, n0 = half * O5_COLS;
  const int step_now = (DROPOUT && p.step_ptr) ? __builtin_amdgcn_readfirstlane(*p.step_ptr) : 0;
  const unsigned int kb2 = (unsigned int)p.K * 2u;
  constexpr unsigned int OOB = 0x40000000u;
#if D5_TRACE
  const unsigned long long* trbase = p.trace + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 64;
#endif
  D5_TR(0);

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.W), 0, (unsigned int)O5_N * kb2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.A), 0, (unsigned int)p.T * (unsigned int)(p.lda * 2), 0x00020000);

  // ---- stage s (k = 64 s .. + 63) -> buffer s % 3.  Images have 128-byte rows, 16-byte slot ^ ((row >> 1) & 7).  A: 2 pieces of
  // 1 KB per wave (8 rows each), W: 4 pieces.  The source slot of a lane is its destination slot ^ swizzle(row); a piece's rows
  // are 8 i + (lane >> 3) behind a multiple of 16, so swizzle = 4 (i & 1) + (lane >> 4): piece i differs from piece 0 by one XOR.
  const unsigned int sl0 = (unsigned int)(((lane & 7) ^ (lane >> 4)) << 4);
  // (the slot XOR is applied to the slot, not to the whole offset: an A row need not be a multiple of 128 bytes long --
  //  mfp_dense_n512_lda)
  const unsigned int a_row0 = (unsigned int)(row0 + wave * 16 + (lane >> 3)) * (unsigned int)(p.lda * 2);
  const unsigned int w_off0 = (unsigned int)(n0 + wave * 32 + (lane >> 3)) * kb2 + sl0;
  const int nst = p.K >> 6;
  auto sload = [&](int s, int buf) {
    unsigned char* base = smem + buf * O5_STAGE;
    const unsigned int so = (unsigned int)s * 128u;
    const unsigned int oob = s < nst ? 0u : OOB;       // past the last stage: zeros into a buffer nobody reads (uniform op count)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_u8*)(base + wave * 2048 + i * 1024), 16,
                                               (a_row0 + (sl0 ^ ((i & 1) << 6)) + (unsigned int)i * 8u * (unsigned int)(p.lda * 2)) + oob, so, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_u8*)(base + O5_A_B + wave * 4096 + i * 1024), 16,
                                               ((w_off0 ^ ((i & 1) << 6)) + (unsigned int)i * 8u * kb2) + oob, so, 0, 0);
  };
  // (the bias is loaded BEFORE the first stages are requested: the wait for its value then does not include them)
  f32x4 bias_v = {0.f, 0.f, 0.f, 0.f};
  if (tid < O5_COLS / 4 && p.bias != nullptr) bias_v = *reinterpret_cast<const f32x4*>(p.bias + n0 + tid * 4);
  asm volatile("" ::: "memory");
  sload(0, 0);
  sload(1, 1);
  if (tid < O5_COLS / 4) *reinterpret_cast<f32x4*>(Bv + tid * 4) = bias_v;
  f32x4 acc[4][4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[ct][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");      // stage 0 has landed (stage 1's 6 pieces are behind it)
  __builtin_amdgcn_s_barrier();

  const int fo = li * 128;                             // row li of a 16-row tile; slot swizzle of rows 16 t + li = li >> 1
  int buf = 0;
  for (int s = 0; s < nst; ++s) {
    const int b2 = buf == 0 ? 2 : buf - 1;             // (s + 2) % 3
    sload(s + 2, b2);
    const unsigned char* Ab = smem + buf * O5_STAGE + (rp * 64) * 128 + fo;
    const unsigned char* Wb = smem + buf * O5_STAGE + O5_A_B + (nh * 64) * 128 + fo;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int so = ((ks * 4 + g) ^ (li >> 1)) << 4;
      bf16x8 af[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) af[rt] = *reinterpret_cast<const bf16x8*>(Ab + rt * 2048 + so);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Wb + ct * 2048 + so);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[rt], acc[ct][rt], 0, 0, 0);
      }
    }
    // stage s + 1 has landed when only the 6 pieces of stage s + 2 are outstanding (in-order retirement); this stage has been read
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    D5_TR(2 + (s < 30 ? s : 30));
    buf = buf == 2 ? 0 : buf + 1;
  }
  // (the two out-of-range stages behind the last one are LDS-DMA into buffers the result image is about to overwrite)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  D5_TR(1);

  // ---- the accumulators -> f32 image [128][1 KB + 16] (over the stage buffers)
  unsigned char* const E = smem;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
      *reinterpret_cast<f32x4*>(E + (rp * 64 + rt * 16 + li) * O5_EROW + (nh * 64 + ct * 16 + 4 * g) * 4) = acc[ct][rt];
  __syncthreads();
  D5_TR(40);
  if constexpr (EPI == O5_EPI_RES) {
    // out = res + Dropout(acc + bias): wave w owns rows 16 w .. + 15, a lane 4 consecutive columns -- 1 KB contiguous per wave
    // instruction for the residual loads and the result stores
    const unsigned int obytes = (unsigned int)p.T * (O5_N * 4);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_oc = __builtin_amdgcn_make_buffer_rsrc(p.outc ? p.outc : reinterpret_cast<unsigned short*>(p.out), 0, p.outc ? obytes / 2 : 0u, 0x00020000);
    const float inv_keep = DROPOUT ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
    const unsigned int dthr = drop_thr16(p.dropout_p);
    const unsigned int dkey = drop_key(p.seed, p.offset + (unsigned long long)step_now * MFP_RNG_STEP_STRIDE);
    const int n = n0 + lane * 4;
    const f32x4 bb = *reinterpret_cast<const f32x4*>(Bv + lane * 4);
    // (requesting the 16 residual rows of a wave before the main loop was measured and dropped: in-order retirement puts them in
    //  front of the first stage -- 7.2 instead of 2.2 us to the first product -- and the stages ran at 0.95 instead of 0.83 us)
#pragma unroll
    for (int hr = 0; hr < 2; ++hr) {      // two rounds of eight rows: 8 residual loads in flight
      f32x4 res[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        res[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, (unsigned int)(row0 + wave * 16 + hr * 8 + i) * (O5_N * 4) + n * 4, 0, 0));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int lr = wave * 16 + hr * 8 + i, row = row0 + lr;
        const f32x4 a = *reinterpret_cast<const f32x4*>(E + lr * O5_EROW + lane * 16);
        bool keep[4] = {true, true, true, true};
        if (DROPOUT) drop_keep4(drop_row(dkey, (unsigned int)row), (unsigned int)n, dthr, keep);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = res[i][r] + (keep[r] ? (a[r] + bb[r]) * inv_keep : 0.f);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_out, (unsigned int)row * (O5_N * 4) + n * 4, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])}, rs_oc, (unsigned int)row * (O5_N * 2) + n * 2, 0, 0);
      }
    }
  } else if constexpr (EPI == O5_EPI_LNB) {
    // wave w owns rows 16 w .. + 15 of the tile, a lane 4 consecutive columns of this half (ln_bwd_tile's walk on 512-byte row halves)
    const unsigned int hb = (unsigned int)p.T * (O5_N * 2);
    const __amdgpu_buffer_rsrc_t rs_xh = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.xhat), 0, hb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.dres), 0, hb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dx = __builtin_amdgcn_make_buffer_rsrc(p.dx, 0, hb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dd = __builtin_amdgcn_make_buffer_rsrc(p.ddrop ? p.ddrop : p.dx, 0, p.ddrop ? hb : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ex = __builtin_amdgcn_make_buffer_rsrc(p.exch, 0, (unsigned int)(p.T / O5_ROWS) * (2 * O5_ROWS * 8), 0x00020000);
    const int r0 = wave * 16;
    const unsigned int cb = (unsigned int)(n0 + lane * 4) * 2u;
    u32x2 xv[16], rv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      xv[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_xh, (unsigned int)(row0 + r0 + i) * (O5_N * 2) + cb, 0, 0));
#pragma unroll
    for (int i = 0; i < 16; ++i)
      rv[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_dr, (unsigned int)(row0 + r0 + i) * (O5_N * 2) + cb, 0, 0));
    const f32x4 gam = *reinterpret_cast<const f32x4*>(p.gamma + n0 + lane * 4);
    const float rs_l = lane < 16 ? p.rstd[row0 + r0 + lane] : 0.f;
    // ---- pass 1: this half's share of the two row sums (f32 dy straight from the result image), parameter-gradient sums
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f}, dc[4] = {0.f, 0.f, 0.f, 0.f};
    float my1 = 0.f, my2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(E + (r0 + i) * O5_EROW + lane * 16);
      const float xh[4] = {__uint_as_float(xv[i][0] << 16), __uint_as_float(xv[i][0] & 0xFFFF0000u), __uint_as_float(xv[i][1] << 16),
                           __uint_as_float(xv[i][1] & 0xFFFF0000u)};
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dg[e] += d[e] * xh[e];
        db[e] += d[e];
        const float gy = d[e] * gam[e];
        s1 += gy;
        s2 += gy * xh[e];
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (lane == i) { my1 = s1; my2 = s2; }
    }
    // ---- exchange with the other column half of the row tile
    const unsigned int exo = (unsigned int)(((tile * 2 + half) * O5_ROWS + r0 + lane) * 8);
    const unsigned int exp_ = (unsigned int)(((tile * 2 + (half ^ 1)) * O5_ROWS + r0 + lane) * 8);
    __builtin_amdgcn_raw_buffer_store_b64((u32x2){__float_as_uint(my1), __float_as_uint(my2)}, rs_ex, lane < 16 ? exo : OOB, 0, 16 /* sc1 */);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(&p.flags[tile * 2 + half], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (bounded: HIP promises no forward progress between workgroups of a plain launch -- the partner sits 8 workgroups away in
      //  dispatch order, so it is resident whenever 16 slots are, but a masked-off / preempted partner must end in a loud
      //  launch failure, not in a hung GPU: ~0.5 s of polling, then trap)
      unsigned int spins = 0;
      while (__hip_atomic_load(&p.flags[tile * 2 + (half ^ 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 24)) __builtin_trap();
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(&p.flags[tile * 2 + (half ^ 1)], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero again for the next launch
    }
    __syncthreads();
    // (sc1: the partner may sit on another XCD when the grid is not a multiple of 16 -- a coherent load beside the acquire fence)
    const u32x2 ov = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_ex, lane < 16 ? exp_ : OOB, 0, 16 /* sc1 */));
    const float t1 = (my1 + __uint_as_float(ov[0])) * (1.0f / O5_N), t2 = (my2 + __uint_as_float(ov[1])) * (1.0f / O5_N);
    // ---- pass 2: dx = dres + rstd (gy - mean(gy) - xhat mean(gy xhat)), its dropout-masked copy
    const unsigned long long rng_off = p.offset + (p.step_ptr ? (unsigned long long)(*p.step_ptr) * MFP_RNG_STEP_STRIDE : 0ull);
    const float inv_keep = p.dropout_p > 0.f ? 1.f / (1.f - p.dropout_p) : 1.f;
    const unsigned int dkey = drop_key(p.seed, rng_off), dthr = drop_thr16(p.dropout_p);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = row0 + r0 + i;
      const float rs = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(rs_l), i));
      const float s1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(t1), i));
      const float s2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(t2), i));
      const f32x4 d = *reinterpret_cast<const f32x4*>(E + (r0 + i) * O5_EROW + lane * 16);
      const float xh[4] = {__uint_as_float(xv[i][0] << 16), __uint_as_float(xv[i][0] & 0xFFFF0000u), __uint_as_float(xv[i][1] << 16),
                           __uint_as_float(xv[i][1] & 0xFFFF0000u)};
      const float res[4] = {__uint_as_float(rv[i][0] << 16), __uint_as_float(rv[i][0] & 0xFFFF0000u), __uint_as_float(rv[i][1] << 16),
                            __uint_as_float(rv[i][1] & 0xFFFF0000u)};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rs * (d[e] * gam[e] - s1 - xh[e] * s2) + res[e];
      __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])}, rs_dx, (unsigned int)row * (O5_N * 2) + cb, 0, 0);
      if (p.ddrop != nullptr) {
        if (p.dropout_p > 0.f) {
          bool keep[4];
          drop_keep4(drop_row(dkey, (unsigned int)row), (unsigned int)(n0 + lane * 4), dthr, keep);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = keep[e] ? o[e] * inv_keep : 0.f;
        }
        __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])}, rs_dd, (unsigned int)row * (O5_N * 2) + cb, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) dc[e] += o[e];
      }
    }
    // ---- the tile's parameter-gradient sums: eight waves through LDS (the result image is dead), one [3][256] slice of the
    // tile's partial row for the batched reduction at the end of the backward pass
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    *reinterpret_cast<f32x4*>(red + (wave * 3 + 0) * O5_COLS + lane * 4) = (f32x4){dg[0], dg[1], dg[2], dg[3]};
    *reinterpret_cast<f32x4*>(red + (wave * 3 + 1) * O5_COLS + lane * 4) = (f32x4){db[0], db[1], db[2], db[3]};
    *reinterpret_cast<f32x4*>(red + (wave * 3 + 2) * O5_COLS + lane * 4) = (f32x4){dc[0], dc[1], dc[2], dc[3]};
    __syncthreads();
    for (int c = tid; c < 3 * O5_COLS; c += 512) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += red[w * 3 * O5_COLS + c];
      p.part[(size_t)tile * (3 * O5_N) + (c / O5_COLS) * O5_N + n0 + (c % O5_COLS)] = a;
    }
  } else {
    // bf16 rows: a lane packs 8 consecutive columns, a wave instruction covers two rows of this column half (2 x 512 B)
    const __amdgpu_buffer_rsrc_t rs_ob = __builtin_amdgcn_make_buffer_rsrc(p.outb, 0, (unsigned int)p.T * (O5_N * 2), 0x00020000);
    const int c8 = (lane & 31) * 8;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bv + c8), b1 = *reinterpret_cast<const f32x4*>(Bv + c8 + 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int lr = wave * 16 + 2 * i + (lane >> 5);
      const f32x4 a = *reinterpret_cast<const f32x4*>(E + lr * O5_EROW + c8 * 4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(E + lr * O5_EROW + c8 * 4 + 16);
      const u32x4 pk = {pack_bf16x2(a[0] + b0[0], a[1] + b0[1]), pack_bf16x2(a[2] + b0[2], a[3] + b0[3]),
                        pack_bf16x2(b[0] + b1[0], b[1] + b1[1]), pack_bf16x2(b[2] + b1[2], b[3] + b1[3])};
      __builtin_amdgcn_raw_buffer_store_b128(pk, rs_ob, (unsigned int)(row0 + lr) * (O5_N * 2) + (n0 + c8) * 2, 0, 0);
    }
  }
#if D5_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  D5_TR(41);
  D5_TR_END();
}

template <typename Kern>
int set_lds(Kern k, int bytes, const char* what) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    mfp_set_error("%s: cannot raise dynamic LDS to %d: %s", what, bytes, hipGetErrorString(e));
    return MFP_ELAUNCH;
  }
  return MFP_OK;
}

// column ranges per row tile: as few as keep a range <= 768 columns, doubled while the grid would leave CUs idle
int as512_nsplit(int T, int N) {
  const int tiles = (T + A5_ROWS - 1) / A5_ROWS, ncu = mfp_ncu_physical();
  int ns = (N + A5_MAXCOLS - 1) / A5_MAXCOLS;
  while (N % (128 * ns) != 0) ++ns;
  while (tiles * ns < ncu && N % (128 * ns * 2) == 0) ns *= 2;
  return ns;
}

template <int ASRC, int EPI, bool XH = false>
int launch_as512(As512Params& p, hipStream_t st) {
  static bool done[MFP_MAX_DEVICES] = {};
  bool& attr = done[mfp_device_slot()];
  if (!attr) {
    if (int rc = set_lds(as512_kernel<ASRC, EPI, XH>, A5_LDS, "as512")) return rc;
    attr = true;
  }
  const int ns = as512_nsplit(p.T, p.N);
  p.ncols = p.N / ns;
#if D5_TRACE
  p.trace = g_d5_trace;
#endif
  hipLaunchKernelGGL((as512_kernel<ASRC, EPI, XH>), dim3((p.T + A5_ROWS - 1) / A5_ROWS, ns), dim3(512), A5_LDS, st, p);
  return MFP_OK;
}

template <int EPI, bool DROPOUT>
int launch_os512(const Os512Params& p_, hipStream_t st) {
  Os512Params p = p_;
#if D5_TRACE
  p.trace = g_d5_trace;
#endif
  static bool done[MFP_MAX_DEVICES] = {};
  bool& attr = done[mfp_device_slot()];
  if (!attr) {
    if (int rc = set_lds(os512_kernel<EPI, DROPOUT>, O5_LDS, "os512")) return rc;
    attr = true;
  }
  hipLaunchKernelGGL((os512_kernel<EPI, DROPOUT>), dim3((p.T + O5_ROWS - 1) / O5_ROWS, O5_N / O5_COLS), dim3(512), O5_LDS, st, p);
  return MFP_OK;
}

bool al16(const void* q) { return ((uintptr_t)q % 16) == 0; }

}  // namespace

static int ln_dense_d512_impl(bool xhat, const float* x, const float* gamma, const float* beta, const void* W, const float* bias, void* y,
                              float* mean, float* rstd, void* out, int32_t T, int32_t N, int32_t relu, float eps, mfp_stream_t stream) {
  MFP_CHECK_ARG(x && gamma && beta && W && y && mean && rstd && out);
  MFP_CHECK_ARG(T > 0 && T <= (1 << 20) && N > 0 && N % 128 == 0 && N <= 8192 && eps > 0.f && (long long)T * N * 2 < 0x40000000LL);
  MFP_CHECK_ARG(al16(x) && al16(gamma) && al16(beta) && al16(W) && al16(y) && al16(out) && (bias == nullptr || al16(bias)));
  As512Params p = {};
  p.x = x; p.gamma = gamma; p.beta = beta;
  p.W = reinterpret_cast<const unsigned short*>(W); p.bias = bias;
  p.y = reinterpret_cast<unsigned short*>(y); p.mean = mean; p.rstd = rstd;
  p.out = reinterpret_cast<unsigned short*>(out); p.ldo = N;
  p.T = T; p.N = N; p.eps = eps;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rc = xhat ? (relu ? launch_as512<A5_SRC_LN, A5_EPI_RELU, true>(p, st) : launch_as512<A5_SRC_LN, A5_EPI_BIAS, true>(p, st))
                      : (relu ? launch_as512<A5_SRC_LN, A5_EPI_RELU>(p, st) : launch_as512<A5_SRC_LN, A5_EPI_BIAS>(p, st));
  if (rc) return rc;
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_ln_dense_d512(const float* x, const float* gamma, const float* beta, const void* W, const float* bias, void* y,
                                 float* mean, float* rstd, void* out, int32_t T, int32_t N, int32_t relu, float eps,
                                 mfp_stream_t stream) {
  return ln_dense_d512_impl(false, x, gamma, beta, W, bias, y, mean, rstd, out, T, N, relu, eps, stream);
}

// The same launch leaving x-hat = (x - mean) rstd (bf16) in the place of y = LN(x) (see mfp_block_fwd_xhat)
extern "C" int mfp_ln_dense_d512_xhat(const float* x, const float* gamma, const float* beta, const void* W, const float* bias, void* xhat,
                                      float* mean, float* rstd, void* out, int32_t T, int32_t N, int32_t relu, float eps,
                                      mfp_stream_t stream) {
  return ln_dense_d512_impl(true, x, gamma, beta, W, bias, xhat, mean, rstd, out, T, N, relu, eps, stream);
}

extern "C" int mfp_dense_relumask_d512(const void* A, const void* W, const void* aux, void* out, int32_t T, int32_t N,
                                       mfp_stream_t stream) {
  MFP_CHECK_ARG(A && W && aux && out);
  MFP_CHECK_ARG(T > 0 && T <= (1 << 20) && N > 0 && N % 128 == 0 && N <= 8192 && (long long)T * N * 2 < 0x40000000LL);
  MFP_CHECK_ARG(al16(A) && al16(W) && al16(aux) && al16(out));
  As512Params p = {};
  p.A = reinterpret_cast<const unsigned short*>(A); p.lda = A5_K;
  p.W = reinterpret_cast<const unsigned short*>(W);
  p.aux = reinterpret_cast<const unsigned short*>(aux); p.ldaux = N;
  p.out = reinterpret_cast<unsigned short*>(out); p.ldo = N;
  p.T = T; p.N = N; p.eps = 1.f;
  if (int rc = launch_as512<A5_SRC_BF16, A5_EPI_MASK>(p, reinterpret_cast<hipStream_t>(stream))) return rc;
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_dense_n512_res(const void* A, const void* W, const float* bias, const float* residual, float* out, void* out_bf16,
                                  int32_t T, int32_t K, float dropout_p, uint64_t seed, uint64_t offset, const int32_t* step_ptr,
                                  mfp_stream_t stream) {
  MFP_CHECK_ARG(A && W && residual && out);
  MFP_CHECK_ARG(T > 0 && T <= (1 << 19) && K >= 128 && K % 64 == 0 && K <= 8192 && (long long)T * K * 2 < 0x40000000LL);
  MFP_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f);
  MFP_CHECK_ARG(al16(A) && al16(W) && al16(residual) && al16(out) && al16(out_bf16) && (bias == nullptr || al16(bias)));
  Os512Params p = {};
  p.A = reinterpret_cast<const unsigned short*>(A); p.lda = K; p.K = K;
  p.W = reinterpret_cast<const unsigned short*>(W); p.bias = bias;
  p.res = residual; p.out = out; p.outc = reinterpret_cast<unsigned short*>(out_bf16);
  p.T = T; p.dropout_p = dropout_p; p.seed = seed; p.offset = offset; p.step_ptr = step_ptr;
  const int rc = dropout_p > 0.f ? launch_os512<O5_EPI_RES, true>(p, reinterpret_cast<hipStream_t>(stream))
                                 : launch_os512<O5_EPI_RES, false>(p, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_dense_n512(const void* A, const void* W, void* out, int32_t T, int32_t K, mfp_stream_t stream) {
  MFP_CHECK_ARG(A && W && out);
  MFP_CHECK_ARG(T > 0 && T <= (1 << 19) && K >= 128 && K % 64 == 0 && K <= 8192 && (long long)T * K * 2 < 0x40000000LL);
  MFP_CHECK_ARG(al16(A) && al16(W) && al16(out));
  Os512Params p = {};
  p.A = reinterpret_cast<const unsigned short*>(A); p.lda = K; p.K = K;
  p.W = reinterpret_cast<const unsigned short*>(W);
  p.outb = reinterpret_cast<unsigned short*>(out);
  p.T = T;
  if (int rc = launch_os512<O5_EPI_BF16, false>(p, reinterpret_cast<hipStream_t>(stream))) return rc;
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

// The same product from an A whose rows are `lda` (<= K) elements apart and whose columns lda .. K - 1 do not exist: the k-range
// is rounded up to the 64 the stages move, the weight's columns >= lda MUST be zero (what the stage reads past a row's end is
// the head of the next row -- finite values -- or, behind the last row, zeros from the bounds check).  The decoder heads' input
// gradient at d_model 512: A = d(logits) [T][U], W = the transposed heads [512][U rounded up to 128] (decoder.py:39-43).
extern "C" int mfp_dense_n512_lda(const void* A, int32_t lda, const void* W, void* out, int32_t T, int32_t K, mfp_stream_t stream) {
  MFP_CHECK_ARG(A && W && out);
  MFP_CHECK_ARG(T > 0 && T <= (1 << 19) && K >= 128 && K % 64 == 0 && K <= 8192 && (long long)T * K * 2 < 0x40000000LL);
  MFP_CHECK_ARG(lda > K - 128 && lda <= K && lda % 8 == 0);
  MFP_CHECK_ARG(al16(A) && al16(W) && al16(out));
  Os512Params p = {};
  p.A = reinterpret_cast<const unsigned short*>(A); p.lda = lda; p.K = K;
  p.W = reinterpret_cast<const unsigned short*>(W);
  p.outb = reinterpret_cast<unsigned short*>(out);
  p.T = T;
  if (int rc = launch_os512<O5_EPI_BF16, false>(p, reinterpret_cast<hipStream_t>(stream))) return rc;
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

// mfp_dense_n512 followed by the x-hat form of mfp_layernorm_bwd_xhat on its result, in one launch (O5_EPI_LNB): A bf16 [T][K]
// is the gradient of the Dense behind a LayerNorm (dh, dqkv), W bf16 [512][K] that Dense's transposed kernel, so A W^T = dy.
//   dx = dres + rstd (dy gamma - mean_c(dy gamma) - xhat mean_c(dy gamma xhat)),  ddrop = Dropout-mask(dx) / keep (or nullptr),
//   part[T / 128][3][512] = per-tile sums of dy xhat | dy | ddrop (dgamma, dbeta, the consuming Dense's bias gradient: the caller
//   reduces them, mfp_reduce_partials[_batch] with P = T / 128, pstride = 1536).
// exch: f32 [T / 128][2][128][2] scratch; flags: int32 [T / 128][2], ZERO on entry and zero again on exit.  T % 128 == 0.
extern "C" int mfp_dense_n512_lnb(const void* A, const void* W, const void* xhat, const float* gamma, const float* rstd, const void* dres,
                                  void* dx, void* ddrop, float* part, float* exch, int32_t* flags, int32_t T, int32_t K, float dropout_p,
                                  uint64_t seed, uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream) {
  MFP_CHECK_ARG(A && W && xhat && gamma && rstd && dres && dx && part && exch && flags);
  MFP_CHECK_ARG(T > 0 && T <= (1 << 19) && T % O5_ROWS == 0 && K >= 128 && K % 64 == 0 && K <= 8192 && (long long)T * K * 2 < 0x40000000LL);
  MFP_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f);
  MFP_CHECK_ARG(al16(A) && al16(W) && al16(xhat) && al16(gamma) && al16(dres) && al16(dx) && al16(ddrop) && al16(part) && al16(exch));
  Os512Params p = {};
  p.A = reinterpret_cast<const unsigned short*>(A); p.lda = K; p.K = K;
  p.W = reinterpret_cast<const unsigned short*>(W);
  p.xhat = reinterpret_cast<const unsigned short*>(xhat); p.gamma = gamma; p.rstd = rstd;
  p.dres = reinterpret_cast<const unsigned short*>(dres); p.dx = reinterpret_cast<unsigned short*>(dx);
  p.ddrop = reinterpret_cast<unsigned short*>(ddrop); p.part = part; p.exch = exch; p.flags = flags;
  p.T = T; p.dropout_p = dropout_p; p.seed = seed; p.offset = offset; p.step_ptr = step_ptr;
  if (int rc = launch_os512<O5_EPI_LNB, false>(p, reinterpret_cast<hipStream_t>(stream))) return rc;
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

#if D5_TRACE
extern "C" int mfp_debug_d512_trace(void* buf) { g_d5_trace = reinterpret_cast<unsigned long long*>(buf); return 0; }
#endif
