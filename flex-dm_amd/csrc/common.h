// Shared device/host helpers for libmfp_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mfp_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define MFP_LDS __attribute__((address_space(3)))

// ----------------------------------------------------------------------------- host errors
void mfp_set_error(const char* fmt, ...);

#define MFP_CHECK_ARG(cond)                                                        \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      mfp_set_error("%s:%d: argument check failed: %s", __FILE__, __LINE__, #cond); \
      return MFP_EINVAL;                                                           \
    }                                                                              \
  } while (0)

#define MFP_CHECK_LAUNCH()                                                            \
  do {                                                                                \
    hipError_t e_ = hipGetLastError();                                                \
    if (e_ != hipSuccess) {                                                           \
      mfp_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return MFP_ELAUNCH;                                                             \
    }                                                                                 \
  } while (0)

// ----------------------------------------------------------------------------- bf16 helpers
__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
  return __uint_as_float(((unsigned int)h) << 16);
}
// f32 -> bf16, round-to-nearest-even, through the native __bf16 type: on gfx950 this is ONE
// v_cvt_pk_bf16_f32 per pair (a software round costs ~5 VALU per value; the GEMM epilogues were
// VALU-bound on it).
typedef __attribute__((ext_vector_type(2))) __bf16 mfp_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float mfp_f32x2_t;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  const mfp_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, mfp_bf16x2_t));
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}

template <typename T> struct cdt_traits;
template <> struct cdt_traits<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct cdt_traits<unsigned short> {
  static __device__ __forceinline__ float load(const unsigned short* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void store(unsigned short* p, float v) { *p = f32_to_bf16(v); }
};

// ----------------------------------------------------------------------------- wave reductions
// All-reduce over the 64 lanes: four DPP steps inside each 16-lane row (quad xor 1, quad xor 2, half-row mirror, row
// mirror), then the four row results through scalar registers.  (`__shfl_xor` is a ds_bpermute round trip of ~100+ clocks:
// a chain of six per sum was what a LayerNorm-backward row spent most of its time in.)
// PRECONDITION (wave_sum, wave_max, lane_xor16, lane_xor32): all 64 lanes active and the call site convergent -- a DPP
// step with an inactive source lane keeps the lane's own value (it would be counted twice) and readlane of an inactive
// row returns stale data.  Every caller sits at wave-uniform control flow; row loops are bounded per wave, not per lane.
template <int CTRL>
__device__ __forceinline__ float mfp_dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += mfp_dpp_f<0xB1>(v); v += mfp_dpp_f<0x4E>(v); v += mfp_dpp_f<0x141>(v); v += mfp_dpp_f<0x140>(v);
  const int vi = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, mfp_dpp_f<0xB1>(v)); v = fmaxf(v, mfp_dpp_f<0x4E>(v)); v = fmaxf(v, mfp_dpp_f<0x141>(v)); v = fmaxf(v, mfp_dpp_f<0x140>(v));
  const int vi = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// The value of the lane 16 / 32 lanes away (lane ^ 16, lane ^ 32) by a row / half-wave swap on the VALU
// (v_permlane16_swap / v_permlane32_swap, gfx950) instead of a ds_bpermute round trip through the LDS crossbar.
__device__ __forceinline__ float lane_xor16(float v) {
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // r[0] = rows {0,0,2,2}, r[1] = rows {1,1,3,3}
  const unsigned int mine_odd = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) >> 4) & 1u;
  return __builtin_bit_cast(float, mine_odd ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor32(float v) {
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // r[0] = halves {lo, lo}, r[1] = halves {hi, hi}
  const unsigned int mine_hi = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) >> 5) & 1u;
  return __builtin_bit_cast(float, mine_hi ? r[0] : r[1]);
}

// ----------------------------------------------------------------------------- per-device caches
// hipFuncSetAttribute and the CU count are properties of (function, device): a process that drives
// several devices must not reuse what it learnt on the first one.  Host-side caches are indexed by
// the current device (slot 0 for ordinals beyond the table).
#define MFP_MAX_DEVICES 64
inline int mfp_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MFP_MAX_DEVICES) dev = 0;
  return dev;
}

// ----------------------------------------------------------------------------- Philox4x32-10
// Counter-based RNG for dropout: the keep mask of element (row, col) of an [M,N] activation is
// philox(key = seed, ctr = (row, col>>2, offset_lo, offset_hi))[col & 3] so that the GEMM epilogue
// (4 consecutive columns of one row per lane and quad) needs one call per 16-byte store, and the
// backward kernel regenerates the identical mask from (seed, offset).
__device__ __forceinline__ void philox_round(unsigned int (&c)[4], unsigned int k0, unsigned int k1) {
  const unsigned int M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  unsigned int hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  unsigned int hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  unsigned int n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32(unsigned long long seed, unsigned int c0, unsigned int c1,
                                           unsigned long long offset, unsigned int (&out)[4]) {
  unsigned int c[4] = {c0, c1, (unsigned int)offset, (unsigned int)(offset >> 32)};
  unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
// keep iff uniform >= p  ([TF-EXT] tf.nn.dropout keeps where random_uniform >= rate)
__device__ __forceinline__ bool philox_keep(unsigned int r, float p) {
  return (float)(r >> 8) * (1.0f / 16777216.0f) >= p;
}

// ----------------------------------------------------------------------------- dropout RNG
// Dropout masks use a cheaper counter-based generator than Philox: 32-bit integer multiplies are
// quarter rate on CDNA (16 clk per wave instruction), so one Philox4x32-10 call (40 multiplies)
// costs ~800 clk per 4 keep decisions -- the dropout epilogues and the fused LayerNorm-backward
// consumer were VALU-bound on it.  Here: a 32-bit avalanche hash ("lowbias32": 2 multiplies,
// measured bias 0.17 bits) of a per-row hash plus the column pair gives 2 x 16 uniform bits:
//   key  = hash(seed, site offset + step * MFP_RNG_STEP_STRIDE)       once per kernel
//   rowh = hash32(row * 0x9E3779B1 ^ key)                             once per row
//   u16(col), u16(col + 1) = halves of hash32(rowh + (col >> 1))      col even
//   keep iff u16 >= thr16 = round(p * 65536)   ([TF-EXT] tf.nn.dropout keeps where uniform >= rate)
// ~7x fewer VALU cycles per decision; forward epilogue and backward consumers regenerate the
// identical mask from (seed, offset, step, row, col).
__device__ __forceinline__ unsigned int hash32(unsigned int x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned int drop_key(unsigned long long seed, unsigned long long rng_off) {
  return hash32((unsigned int)seed ^ hash32((unsigned int)rng_off ^ 0x68E31DA4u) ^
                (unsigned int)(seed >> 32) * 0x9E3779B1u ^ hash32((unsigned int)(rng_off >> 32) + 0xB5297A4Du));
}
__device__ __forceinline__ unsigned int drop_thr16(float p) { return (unsigned int)(p * 65536.f + 0.5f); }
__device__ __forceinline__ unsigned int drop_row(unsigned int key, unsigned int row) {
  return hash32(row * 0x9E3779B1u ^ key);
}
// keep decisions of the 4 consecutive columns col .. col+3 (col % 4 == 0) of the row hashed as rowh
__device__ __forceinline__ void drop_keep4(unsigned int rowh, unsigned int col, unsigned int thr16, bool (&keep)[4]) {
  const unsigned int a = hash32(rowh + (col >> 1)), b = hash32(rowh + (col >> 1) + 1u);
  keep[0] = (a & 0xffffu) >= thr16; keep[1] = (a >> 16) >= thr16;
  keep[2] = (b & 0xffffu) >= thr16; keep[3] = (b >> 16) >= thr16;
}

// CU count of the current device (csrc/error.cpp): physical, and what a persistent launch sizes its grid for
// (physical - mfp_set_reserved_cus)
int mfp_ncu_physical();
int mfp_ncu_launch();

// ----------------------------------------------------------------------------- XCD-aware block order
// Hardware places block b on XCD b % 8 (speed only, never correctness).  This bijective remap gives
// every XCD a CONTIGUOUS range of logical ids, so neighbouring logical ids (tiles sharing an operand
// panel, or the 8 heads of a document whose 64-byte slices share 128-byte lines) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}
