// LayerNorm backward on a 128-row x 256-column tile whose dy rows sit in LDS: the epilogue of the launches that produce the
// gradient of a LayerNorm output at d_model 256 (block_fused.hip mlp_bwd_kernel<true>: LN2; block_attn_bwd.hip
// attn_block_bwd_kernel<.., true>: LN1), so that dy never crosses HBM.  The row arithmetic is ln_bwd_kernel's
// (csrc/layernorm.hip; reference: Keras autodiff of LayerNormalization, transformer.py:216-217 / 222-223):
//     xh = (x - mean) rstd,  gy = dy gamma,  dx = dres + rstd (gy - mean_c(gy) - xh mean_c(gy xh)),
//     ddrop = Dropout-mask(dx) / keep  (the stream of mfp_dropout_bwd for (seed, offset, *step_ptr); ddrop == nullptr: none)
//     part[tile][3][256] = sum over the tile's rows of  dy xh | dy | ddrop   (dgamma, dbeta, the consuming Dense's bias gradient)
// x-hat form (round 5, LnTileArgs::xhat != nullptr): the forward pass left xh = (x - mean) rstd in bf16 (in the place of
// y = LN(x): the launch that stashes it is mfp_block_fwd with xhat_stash = 1) and the epilogue reads 0.5 KB per element
// instead of x's 1 KB; mean is not needed then.
// bf16 residual-gradient stream (dres in, dx out: mfp_layernorm_bwd_res16's types).  Wave w owns rows 16 w .. + 15, a lane 4
// consecutive columns: x in whole 1 KB rows, dres / dx / ddrop in 512-byte rows.  Every load is issued before the first store:
// a load between two rows' stores would wait for them (vmcnt counts both).
#pragma once
#include "common.h"

struct LnTileArgs {
  const float* x; const float* gamma; const float* mean; const float* rstd;
  const unsigned short* xhat;      // bf16 [T][256] or nullptr: x-hat form (x, mean unused)
  const unsigned short* dres;      // bf16 [T][256]
  unsigned short* dx;              // bf16 [T][256]
  unsigned short* ddrop;           // bf16 [T][256] or nullptr
  float* part;                     // [T / 128][3][256]
  float drop_p; unsigned long long seed, offset; const int* step_ptr;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ln_tile_x_rsrc(const LnTileArgs& q, int T) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.x), 0, (unsigned int)T * 1024u, 0x00020000);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ln_tile_xh_rsrc(const LnTileArgs& q, int T) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(q.xhat), 0, (unsigned int)T * 512u, 0x00020000);
}
template <int NR>
__device__ __forceinline__ void ln_tile_load_xh(const __amdgpu_buffer_rsrc_t rs_xh, int row0, int wv, int lane, u32x2 (&xv)[NR]) {
#pragma unroll
  for (int i = 0; i < NR; ++i)
    xv[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_xh, (unsigned int)(row0 + wv * NR + i) * 512u + lane * 8, 0, 0));
}

// the x rows of wave `wv` (may be issued early by the caller: 64 registers in flight)
template <int NR>
__device__ __forceinline__ void ln_tile_load_x(const __amdgpu_buffer_rsrc_t rs_x, int row0, int wv, int lane, f32x4 (&xv)[NR]) {
#pragma unroll
  for (int i = 0; i < NR; ++i)
    xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned int)(row0 + wv * NR + i) * 1024u + lane * 16, 0, 0));
}

// dy_of(r) = the 4 bf16 values (u32x2) of tile row r at this lane's columns 4 lane .. + 3; red = 24 KB of free LDS; every wave
// of the 512-thread workgroup calls it (one __syncthreads inside)
// XT = f32x4 (x rows) or u32x2 (x-hat rows, bf16); NR = rows per wave: 16 (a 128-row tile) or 8 (a 64-row half tile: `tile`
// then counts half tiles and `part` holds T / 64 rows)
// NWV = waves of the workgroup (8; 4: dgrad_half_kernel's 64-row tiles, 16 rows per wave)
template <int NWV = 8, typename XT, typename DyOf, int NR>
__device__ __forceinline__ void ln_bwd_tile(const LnTileArgs& q, int T, int row0, int tile, int wv, int lane, int tid,
                                            const XT (&xv)[NR], DyOf dy_of, float* red) {
  constexpr bool XH = sizeof(XT) == 8;
  constexpr int D = 256;
  const unsigned int rbytes = (unsigned int)T * (D * 2);
  const __amdgpu_buffer_rsrc_t rs_dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(q.dres), 0, rbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dx = __builtin_amdgcn_make_buffer_rsrc(q.dx, 0, rbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dd = __builtin_amdgcn_make_buffer_rsrc(q.ddrop ? q.ddrop : q.dx, 0, q.ddrop ? rbytes : 0u, 0x00020000);
  const int r0 = wv * NR;
  u32x2 rv[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i)
    rv[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_dr, (unsigned int)(row0 + r0 + i) * (D * 2) + lane * 8, 0, 0));
  const f32x4 gam = *reinterpret_cast<const f32x4*>(q.gamma + lane * 4);
  // the 16 rows' statistics in lanes 0..15, broadcast per row by v_readlane
  float mu_l = 0.f, rs_l = 0.f;
  if (lane < NR && row0 + r0 + lane < T) { mu_l = XH ? 0.f : q.mean[row0 + r0 + lane]; rs_l = q.rstd[row0 + r0 + lane]; }
  const unsigned long long rng_off = q.offset + (q.step_ptr ? (unsigned long long)(*q.step_ptr) * MFP_RNG_STEP_STRIDE : 0ull);
  const float inv_keep = q.drop_p > 0.f ? 1.f / (1.f - q.drop_p) : 1.f;
  const unsigned int dkey = drop_key(q.seed, rng_off), dthr = drop_thr16(q.drop_p);
  float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f}, dc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = r0 + i, row = row0 + r;
    const bool live = row < T;
    const float mu = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(mu_l), i));
    const float rs = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(rs_l), i));
    const u32x2 dv = dy_of(r);
    const float d[4] = {__uint_as_float(dv[0] << 16), __uint_as_float(dv[0] & 0xFFFF0000u), __uint_as_float(dv[1] << 16),
                        __uint_as_float(dv[1] & 0xFFFF0000u)};
    float xh[4], gy[4];
    if constexpr (XH) {
      xh[0] = __uint_as_float(xv[i][0] << 16); xh[1] = __uint_as_float(xv[i][0] & 0xFFFF0000u);
      xh[2] = __uint_as_float(xv[i][1] << 16); xh[3] = __uint_as_float(xv[i][1] & 0xFFFF0000u);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (!XH) xh[e] = (xv[i][e] - mu) * rs;
      dg[e] += d[e] * xh[e];
      db[e] += d[e];
      gy[e] = d[e] * gam[e];
    }
    float s1 = gy[0] + gy[1] + gy[2] + gy[3];
    float s2 = gy[0] * xh[0] + gy[1] * xh[1] + gy[2] * xh[2] + gy[3] * xh[3];
    s1 = wave_sum(s1) * (1.0f / D);
    s2 = wave_sum(s2) * (1.0f / D);
    const float res[4] = {__uint_as_float(rv[i][0] << 16), __uint_as_float(rv[i][0] & 0xFFFF0000u), __uint_as_float(rv[i][1] << 16),
                          __uint_as_float(rv[i][1] & 0xFFFF0000u)};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = rs * (gy[e] - s1 - xh[e] * s2) + res[e];
    __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])}, rs_dx, (unsigned int)row * (D * 2) + lane * 8, 0, 0);
    if (q.ddrop != nullptr) {
      if (q.drop_p > 0.f) {
        bool keep[4];
        drop_keep4(drop_row(dkey, (unsigned int)row), (unsigned int)(lane * 4), dthr, keep);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = keep[e] ? o[e] * inv_keep : 0.f;
      }
      __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])}, rs_dd, (unsigned int)row * (D * 2) + lane * 8, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) dc[e] += live ? o[e] : 0.f;
    }
  }
  // the tile's partial sums: eight waves through LDS, one row of [3][256] per workgroup for the batched reduction at the end of
  // the backward pass
  *reinterpret_cast<f32x4*>(red + (wv * 3 + 0) * D + lane * 4) = (f32x4){dg[0], dg[1], dg[2], dg[3]};
  *reinterpret_cast<f32x4*>(red + (wv * 3 + 1) * D + lane * 4) = (f32x4){db[0], db[1], db[2], db[3]};
  *reinterpret_cast<f32x4*>(red + (wv * 3 + 2) * D + lane * 4) = (f32x4){dc[0], dc[1], dc[2], dc[3]};
  __syncthreads();
  for (int c = tid; c < 3 * D; c += NWV * 64) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) a += red[w * 3 * D + c];
    q.part[(size_t)tile * (3 * D) + c] = a;
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// ln_bwd_tile with 16-BYTE accesses (round 5, x-hat form only): a lane owns 8 consecutive columns, a wave instruction covers
// TWO rows (lanes 0-31: row 2 i, lanes 32-63: row 2 i + 1) -- x-hat / dres loads and dx / masked-copy stores of 16 bytes per
// lane instead of 8 (8-byte accesses run at 0.54-0.70 of the 16-byte rate: MI355X_MICROARCH.md; the epilogue is a pure memory
// phase).  NH = row PAIRS per wave (8: a 128-row tile on eight waves or a 64-row tile on four; 4: a 64-row tile on eight waves).
// dy_of8(r, s) = the 8 bf16 values (u32x4) of tile row r at columns 8 s .. + 7.  Row sums: four DPP steps inside each 16-lane
// row, then the two rows of a half-wave through scalar registers.
template <int NH>
__device__ __forceinline__ void ln_tile_load_xh16(const __amdgpu_buffer_rsrc_t rs_xh, int row0, int wv, int lane, u32x4 (&xv)[NH]) {
#pragma unroll
  for (int i = 0; i < NH; ++i)
    xv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_xh, (unsigned int)(row0 + wv * (2 * NH) + 2 * i + (lane >> 5)) * 512u + (lane & 31) * 16, 0, 0));
}

template <int NWV = 8, typename DyOf8, int NH>
__device__ __forceinline__ void ln_bwd_tile16(const LnTileArgs& q, int T, int row0, int tile, int wv, int lane, int tid,
                                              const u32x4 (&xv)[NH], DyOf8 dy_of8, float* red) {
  constexpr int D = 256;
  const unsigned int rbytes = (unsigned int)T * (D * 2);
  const __amdgpu_buffer_rsrc_t rs_dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(q.dres), 0, rbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dx = __builtin_amdgcn_make_buffer_rsrc(q.dx, 0, rbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dd = __builtin_amdgcn_make_buffer_rsrc(q.ddrop ? q.ddrop : q.dx, 0, q.ddrop ? rbytes : 0u, 0x00020000);
  const int r0 = wv * (2 * NH), half = lane >> 5, s5 = lane & 31;
  u32x4 rv[NH];
#pragma unroll
  for (int i = 0; i < NH; ++i)
    rv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dr, (unsigned int)(row0 + r0 + 2 * i + half) * (D * 2) + s5 * 16, 0, 0));
  const f32x4 gam0 = *reinterpret_cast<const f32x4*>(q.gamma + s5 * 8), gam1 = *reinterpret_cast<const f32x4*>(q.gamma + s5 * 8 + 4);
  const float gam[8] = {gam0[0], gam0[1], gam0[2], gam0[3], gam1[0], gam1[1], gam1[2], gam1[3]};
  // the 2 NH rows' rstd in lanes 0 .. 2 NH - 1, broadcast per row by v_readlane
  float rs_l = 0.f;
  if (lane < 2 * NH && row0 + r0 + lane < T) rs_l = q.rstd[row0 + r0 + lane];
  const unsigned long long rng_off = q.offset + (q.step_ptr ? (unsigned long long)(*q.step_ptr) * MFP_RNG_STEP_STRIDE : 0ull);
  const float inv_keep = q.drop_p > 0.f ? 1.f / (1.f - q.drop_p) : 1.f;
  const unsigned int dkey = drop_key(q.seed, rng_off), dthr = drop_thr16(q.drop_p);
  float dg[8], db[8], dc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dg[e] = db[e] = dc[e] = 0.f;
  auto unpack8 = [](const u32x4& v, float (&o)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[2 * j] = __uint_as_float(v[j] << 16); o[2 * j + 1] = __uint_as_float(v[j] & 0xFFFF0000u); }
  };
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int r = r0 + 2 * i + half, row = row0 + r;
    const bool live = row < T;
    const float rsa = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(rs_l), 2 * i));
    const float rsb = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(rs_l), 2 * i + 1));
    const float rs = half ? rsb : rsa;
    float d[8], xh[8], res[8], gy[8];
    unpack8(dy_of8(r, s5), d);
    unpack8(xv[i], xh);
    unpack8(rv[i], res);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dg[e] += d[e] * xh[e];
      db[e] += d[e];
      gy[e] = d[e] * gam[e];
      s1 += gy[e];
      s2 += gy[e] * xh[e];
    }
    // sums over the 32 lanes of this lane's row: 16-lane rows by DPP, the two rows of a half-wave through scalar registers
    s1 += mfp_dpp_f<0xB1>(s1); s1 += mfp_dpp_f<0x4E>(s1); s1 += mfp_dpp_f<0x141>(s1); s1 += mfp_dpp_f<0x140>(s1);
    s2 += mfp_dpp_f<0xB1>(s2); s2 += mfp_dpp_f<0x4E>(s2); s2 += mfp_dpp_f<0x141>(s2); s2 += mfp_dpp_f<0x140>(s2);
    const int v1 = __builtin_bit_cast(int, s1), v2 = __builtin_bit_cast(int, s2);
    const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v1, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(v1, 16));
    const float b1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v1, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(v1, 48));
    const float a2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v2, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(v2, 16));
    const float b2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v2, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(v2, 48));
    const float m1 = (half ? b1 : a1) * (1.0f / D), m2 = (half ? b2 : a2) * (1.0f / D);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = rs * (gy[e] - m1 - xh[e] * m2) + res[e];
    __builtin_amdgcn_raw_buffer_store_b128((u32x4){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])},
                                           rs_dx, (unsigned int)row * (D * 2) + s5 * 16, 0, 0);
    if (q.ddrop != nullptr) {
      if (q.drop_p > 0.f) {
        const unsigned int rowh = drop_row(dkey, (unsigned int)row);
        bool k0[4], k1[4];
        drop_keep4(rowh, (unsigned int)(s5 * 8), dthr, k0);
        drop_keep4(rowh, (unsigned int)(s5 * 8 + 4), dthr, k1);
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = k0[e] ? o[e] * inv_keep : 0.f; o[4 + e] = k1[e] ? o[4 + e] * inv_keep : 0.f; }
      }
      __builtin_amdgcn_raw_buffer_store_b128((u32x4){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])},
                                             rs_dd, (unsigned int)row * (D * 2) + s5 * 16, 0, 0);
#pragma unroll
      for (int e = 0; e < 8; ++e) dc[e] += live ? o[e] : 0.f;
    }
  }
  // the tile's partial sums: the two half-waves hold the same columns (even / odd rows) -> one exchange, then NWV waves through LDS
#pragma unroll
  for (int e = 0; e < 8; ++e) { dg[e] += lane_xor32(dg[e]); db[e] += lane_xor32(db[e]); dc[e] += lane_xor32(dc[e]); }
  if (half == 0) {
    *reinterpret_cast<f32x4*>(red + (wv * 3 + 0) * D + s5 * 8) = (f32x4){dg[0], dg[1], dg[2], dg[3]};
    *reinterpret_cast<f32x4*>(red + (wv * 3 + 0) * D + s5 * 8 + 4) = (f32x4){dg[4], dg[5], dg[6], dg[7]};
    *reinterpret_cast<f32x4*>(red + (wv * 3 + 1) * D + s5 * 8) = (f32x4){db[0], db[1], db[2], db[3]};
    *reinterpret_cast<f32x4*>(red + (wv * 3 + 1) * D + s5 * 8 + 4) = (f32x4){db[4], db[5], db[6], db[7]};
    *reinterpret_cast<f32x4*>(red + (wv * 3 + 2) * D + s5 * 8) = (f32x4){dc[0], dc[1], dc[2], dc[3]};
    *reinterpret_cast<f32x4*>(red + (wv * 3 + 2) * D + s5 * 8 + 4) = (f32x4){dc[4], dc[5], dc[6], dc[7]};
  }
  __syncthreads();
  for (int c = tid; c < 3 * D; c += NWV * 64) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) a += red[w * 3 * D + c];
    q.part[(size_t)tile * (3 * D) + c] = a;
  }
}
