// LayerNorm forward / backward (Keras LayerNormalization(), eps 1e-3; reference
// architecture/transformer.py:172-173,216,222).  HBM-bound: one wave per token row, float4
// loads, wavefront-wide shuffle reductions; mean/rstd saved (8 B/row) for the backward.
//   fwd bytes/row: D*4 read + D*e write (+8);  bwd: D*(4 + e + 4[dres]) read + D*4 write.
#include "common.h"
#include "reduce.h"

namespace {


template <typename TOUT, int NVEC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     TOUT* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int T, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const float* xr = x + (long long)row * D;
  float v[NVEC * 4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NVEC; ++i) {
    const int c = lane * 4 + i * 256;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < D) t = *reinterpret_cast<const float4*>(xr + c);
    v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    s += t.x + t.y + t.z + t.w;
  }
  s = wave_sum(s);
  const float mu = s / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NVEC; ++i) {
    if (lane * 4 + i * 256 < D) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { float d = v[4 * i + e] - mu; q += d * d; }
    }
  }
  q = wave_sum(q);
  const float rs = rsqrtf(q / (float)D + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  TOUT* yr = y + (long long)row * D;
#pragma unroll
  for (int i = 0; i < NVEC; ++i) {
    const int c = lane * 4 + i * 256, j = 4 * i;
    if (c >= D) continue;
    float4 g = *reinterpret_cast<const float4*>(gamma + c);
    float4 b = *reinterpret_cast<const float4*>(beta + c);
    float o0 = (v[j] - mu) * rs * g.x + b.x, o1 = (v[j + 1] - mu) * rs * g.y + b.y;
    float o2 = (v[j + 2] - mu) * rs * g.z + b.z, o3 = (v[j + 3] - mu) * rs * g.w + b.w;
    if constexpr (sizeof(TOUT) == 4) {
      *reinterpret_cast<float4*>(yr + c) = make_float4(o0, o1, o2, o3);
    } else {
      u32x2 pk = {pack_bf16x2(o0, o1), pack_bf16x2(o2, o3)};
      *reinterpret_cast<u32x2*>(yr + c) = pk;
    }
  }
}

#ifndef LN_BWD_DEPTH
#define LN_BWD_DEPTH 2
#endif
// rows per workgroup: 32 (4 waves x 8 rows; 64 and 16 measured 6-15 % slower at 32 768 rows), 16 when 32 would leave the
// chip under-occupied (round 5, BASELINE config 5: 16 384 rows = 512 workgroups = two waves per SIMD, each walking a chain
// load -> reduce -> store per row -- latency-bound at 3.4 us per row)
static int ln_bwd_rows(int T) {
  static const int force = getenv("MFP_LN_BWD_ROWS") ? atoi(getenv("MFP_LN_BWD_ROWS")) : 0;      // A/B switch: 16 | 32
  if (force == 16 || force == 32) return force;
  return (T + 31) / 32 < 3 * mfp_ncu_physical() ? 16 : 32;
}

// Optional fused consumer: the LN-backward output dx is, in the DeepSVG block, immediately fed to
// the backward of a Dropout + Dense pair (x1 = x + Dropout(Dense(.))): when `ddrop` is given the
// kernel also emits ddrop = cdt(keep ? dx/(1-p) : 0) with the dropout keying of the GEMM epilogue
// (a lane holds 4 consecutive columns = one drop_keep4 call) and the column sums of ddrop (the Dense
// bias gradient) as a third partial vector -- saving a full re-read of dx and two launches.
// TRES = type of the residual gradient stream (dres in, dx out): float, or bf16 (unsigned short) when the step carries
// the residual gradient in the compute dtype (mfp_layernorm_bwd_res16: 1 KB per element and layer less).
// XH: `x` holds x-hat = (x - mean) rstd in bf16 (the stash of mfp_ln_dense_d512_xhat / mfp_block_fwd_xhat) -- 2 instead of 4 bytes
// per element read, mean unused (mfp_layernorm_bwd_xhat).
template <typename TDY, int NVEC, typename TRES = float, int LN_BWD_ROWS = 32, bool XH = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDY* __restrict__ dy,
                                                     const float* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     const TRES* __restrict__ dres,
                                                     TRES* __restrict__ dx, float* __restrict__ part,
                                                     int T, int D, TDY* __restrict__ ddrop, float drop_p,
                                                     unsigned long long seed, unsigned long long offset0,
                                                     const int* __restrict__ step_ptr) {
  // part: [gridDim.x][3][D]  (dgamma, dbeta, colsum(ddrop) partials)
  __shared__ float red[4][3][NVEC * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long rng_off = offset0 + (step_ptr ? (unsigned long long)(*step_ptr) * MFP_RNG_STEP_STRIDE : 0ull);
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const unsigned int dkey = drop_key(seed, rng_off), dthr = drop_thr16(drop_p);
  float dg[NVEC * 4], db[NVEC * 4], dc[NVEC * 4];
#pragma unroll
  for (int j = 0; j < NVEC * 4; ++j) { dg[j] = 0.f; db[j] = 0.f; dc[j] = 0.f; }
  const int row0 = blockIdx.x * LN_BWD_ROWS;
  // A wave walks its rows with the NEXT row's operands already in flight: a row is a chain
  // load -> two wave reductions -> store, and a wave that waits out a full memory round trip per row
  // leaves the kernel latency-bound (measured 3.3 TB/s in the step).
  struct RowIn {
    float4 xv[NVEC], rv[NVEC];
    float4 dvf[NVEC];    // f32 dy ...
    u32x2 dvh[NVEC];     // ... or four packed bf16 (only the member matching TDY is ever touched; a single
                         // punned member was miscompiled by hipcc 7.2: upper half treated as dead)
    float mu, rs;
  };
  auto load_row = [&](int row, RowIn& in) {
    const float* xr = x + (long long)row * D;
    const unsigned short* xhr = reinterpret_cast<const unsigned short*>(x) + (long long)row * D;
    const TDY* dyr = dy + (long long)row * D;
    const TRES* drr = dres ? dres + (long long)row * D : nullptr;
    in.mu = XH ? 0.f : mean[row];
    in.rs = rstd[row];
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
      const int c = lane * 4 + i * 256;
      if (c >= D) continue;
      if constexpr (XH) {
        const u32x2 t = *reinterpret_cast<const u32x2*>(xhr + c);
        in.xv[i] = make_float4(__uint_as_float(t[0] << 16), __uint_as_float(t[0] & 0xFFFF0000u), __uint_as_float(t[1] << 16),
                               __uint_as_float(t[1] & 0xFFFF0000u));
      } else {
        in.xv[i] = *reinterpret_cast<const float4*>(xr + c);
      }
      if constexpr (sizeof(TDY) == 4) {
        in.dvf[i] = *reinterpret_cast<const float4*>(dyr + c);
      } else {
        in.dvh[i] = *reinterpret_cast<const u32x2*>(dyr + c);
      }
      if (drr) {
        if constexpr (sizeof(TRES) == 4) {
          in.rv[i] = *reinterpret_cast<const float4*>(drr + c);
        } else {
          const u32x2 t = *reinterpret_cast<const u32x2*>(drr + c);
          in.rv[i] = make_float4(bf16_to_f32((unsigned short)(t[0] & 0xffff)), bf16_to_f32((unsigned short)(t[0] >> 16)),
                                 bf16_to_f32((unsigned short)(t[1] & 0xffff)), bf16_to_f32((unsigned short)(t[1] >> 16)));
        }
      }
    }
  };
  float4 gam[NVEC];
#pragma unroll
  for (int i = 0; i < NVEC; ++i) {
    const int c = lane * 4 + i * 256;
    gam[i] = c < D ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto process = [&](const RowIn& cur, int row) {
    const float mu = cur.mu, rs = cur.rs;
    const unsigned int rowh = drop_row(dkey, (unsigned int)row);
    float xh[NVEC * 4], gy[NVEC * 4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
      const int c = lane * 4 + i * 256, j = 4 * i;
      if (c >= D) { xh[j] = xh[j + 1] = xh[j + 2] = xh[j + 3] = 0.f; gy[j] = gy[j + 1] = gy[j + 2] = gy[j + 3] = 0.f; continue; }
      const float4 xv = cur.xv[i], g = gam[i];
      float d0, d1, d2, d3;
      if constexpr (sizeof(TDY) == 4) {
        d0 = cur.dvf[i].x; d1 = cur.dvf[i].y; d2 = cur.dvf[i].z; d3 = cur.dvf[i].w;
      } else {
        const unsigned int t0 = cur.dvh[i][0], t1 = cur.dvh[i][1];
        d0 = bf16_to_f32((unsigned short)(t0 & 0xffff)); d1 = bf16_to_f32((unsigned short)(t0 >> 16));
        d2 = bf16_to_f32((unsigned short)(t1 & 0xffff)); d3 = bf16_to_f32((unsigned short)(t1 >> 16));
      }
      if constexpr (XH) {
        xh[j] = xv.x; xh[j + 1] = xv.y; xh[j + 2] = xv.z; xh[j + 3] = xv.w;
      } else {
        xh[j] = (xv.x - mu) * rs; xh[j + 1] = (xv.y - mu) * rs;
        xh[j + 2] = (xv.z - mu) * rs; xh[j + 3] = (xv.w - mu) * rs;
      }
      dg[j] += d0 * xh[j]; dg[j + 1] += d1 * xh[j + 1]; dg[j + 2] += d2 * xh[j + 2]; dg[j + 3] += d3 * xh[j + 3];
      db[j] += d0; db[j + 1] += d1; db[j + 2] += d2; db[j + 3] += d3;
      gy[j] = d0 * g.x; gy[j + 1] = d1 * g.y; gy[j + 2] = d2 * g.z; gy[j + 3] = d3 * g.w;
      s1 += gy[j] + gy[j + 1] + gy[j + 2] + gy[j + 3];
      s2 += gy[j] * xh[j] + gy[j + 1] * xh[j + 1] + gy[j + 2] * xh[j + 2] + gy[j + 3] * xh[j + 3];
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    TRES* dxr = dx + (long long)row * D;
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
      const int c = lane * 4 + i * 256, j = 4 * i;
      if (c >= D) continue;
      float4 o;
      o.x = rs * (gy[j] - s1 - xh[j] * s2);
      o.y = rs * (gy[j + 1] - s1 - xh[j + 1] * s2);
      o.z = rs * (gy[j + 2] - s1 - xh[j + 2] * s2);
      o.w = rs * (gy[j + 3] - s1 - xh[j + 3] * s2);
      if (dres) {
        const float4 r = cur.rv[i];
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      if constexpr (sizeof(TRES) == 4) {
        *reinterpret_cast<float4*>(dxr + c) = o;
      } else {
        *reinterpret_cast<u32x2*>(dxr + c) = (u32x2){pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w)};
      }
      if (ddrop != nullptr) {
        if (drop_p > 0.f) {
          bool keep[4];
          drop_keep4(rowh, (unsigned int)c, dthr, keep);
          o.x = keep[0] ? o.x * inv_keep : 0.f;
          o.y = keep[1] ? o.y * inv_keep : 0.f;
          o.z = keep[2] ? o.z * inv_keep : 0.f;
          o.w = keep[3] ? o.w * inv_keep : 0.f;
        }
        TDY* dr = ddrop + (long long)row * D + c;
        if constexpr (sizeof(TDY) == 4) {
          *reinterpret_cast<float4*>(dr) = o;
        } else {
          u32x2 pk = {pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w)};
          *reinterpret_cast<u32x2*>(dr) = pk;
        }
        dc[j] += o.x; dc[j + 1] += o.y; dc[j + 2] += o.z; dc[j + 3] += o.w;
      }
    }
  };
  // LN_BWD_DEPTH rows in flight per wave (statically indexed buffers: the 8-row walk of a wave is fully unrolled)
  constexpr int DEPTH = LN_BWD_DEPTH, PER_WAVE = LN_BWD_ROWS / 4;
  RowIn buf[DEPTH];
  const int rlast = min(LN_BWD_ROWS, T - row0);     // rows of this workgroup
#pragma unroll
  for (int k = 0; k < DEPTH; ++k)
    if (wave + 4 * k < rlast) load_row(row0 + wave + 4 * k, buf[k]);
#pragma unroll
  for (int k = 0; k < PER_WAVE; ++k) {
    const int rr = wave + 4 * k;
    if (rr < rlast) {
      process(buf[k % DEPTH], row0 + rr);
      if (k + DEPTH < PER_WAVE && rr + 4 * DEPTH < rlast) load_row(row0 + rr + 4 * DEPTH, buf[k % DEPTH]);
    }
  }
  // cross-wave reduction of the dgamma/dbeta(/colsum) partials
#pragma unroll
  for (int i = 0; i < NVEC; ++i) {
    const int c = lane * 4 + i * 256, j = 4 * i;
    if (c >= D) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[wave][0][c + e] = dg[j + e]; red[wave][1][c + e] = db[j + e]; red[wave][2][c + e] = dc[j + e]; }
  }
  __syncthreads();
  float* pout = part + (long long)blockIdx.x * 3 * D;
  for (int c = threadIdx.x; c < 3 * D; c += 256) {
    int which = c / D, col = c % D;
    pout[c] = red[0][which][col] + red[1][which][col] + red[2][which][col] + red[3][which][col];
  }
}

}  // namespace

extern "C" int mfp_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y,
                                 float* mean, float* rstd, int32_t T, int32_t D, float eps,
                                 int32_t out_dtype, mfp_stream_t stream) {
  MFP_CHECK_ARG(x && gamma && beta && y && mean && rstd);
  MFP_CHECK_ARG(T > 0 && D > 0 && D % 4 == 0 && D <= 1024);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((T + 3) / 4), block(256);
  MFP_CHECK_ARG(out_dtype == MFP_F32 || out_dtype == MFP_BF16);
  const int nvec = (D + 255) / 256;
#define LN_FWD(TT, NV) hipLaunchKernelGGL((ln_fwd_kernel<TT, NV>), grid, block, 0, st, x, gamma, beta, (TT*)y, mean, rstd, T, D, eps)
  if (out_dtype == MFP_F32) {
    if (nvec == 1) LN_FWD(float, 1); else if (nvec == 2) LN_FWD(float, 2); else LN_FWD(float, 4);
  } else {
    if (nvec == 1) LN_FWD(unsigned short, 1); else if (nvec == 2) LN_FWD(unsigned short, 2); else LN_FWD(unsigned short, 4);
  }
#undef LN_FWD
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" size_t mfp_layernorm_bwd_workspace_bytes(int32_t T, int32_t D) {
  const int rows = ln_bwd_rows(T);
  size_t nblk = (size_t)(T + rows - 1) / rows;
  return nblk * 3 * D * sizeof(float);
}

template <typename TRES, bool XH = false>
static int ln_bwd_impl(const char* who, const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                       const TRES* dres, TRES* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                       int32_t T, int32_t D, int32_t dy_dtype, void* ddrop, float* drop_colsum, float drop_p, uint64_t seed,
                       uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream) {
  MFP_CHECK_ARG(dy && x && gamma && (XH || mean) && rstd && dx && (dgamma != nullptr) == (dbeta != nullptr));
  MFP_CHECK_ARG(T > 0 && D > 0 && D % 4 == 0 && D <= 1024);
  MFP_CHECK_ARG((ddrop == nullptr) == (drop_colsum == nullptr) && drop_p >= 0.f && drop_p < 1.f);
  if (!workspace || workspace_bytes < mfp_layernorm_bwd_workspace_bytes(T, D)) {
    mfp_set_error("%s: workspace too small", who);
    return MFP_EWORKSPACE;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rows = ln_bwd_rows(T);
  int nblk = (T + rows - 1) / rows;
  float* part = reinterpret_cast<float*>(workspace);
  MFP_CHECK_ARG(dy_dtype == MFP_F32 || dy_dtype == MFP_BF16);
  const int nvec = (D + 255) / 256;
#define LN_BWD_(TT, NV, R) hipLaunchKernelGGL((ln_bwd_kernel<TT, NV, TRES, R, XH>), dim3(nblk), dim3(256), 0, st, (const TT*)dy, x, gamma, mean, rstd, dres, dx, part, T, D, (TT*)ddrop, drop_p, seed, offset, step_ptr)
#define LN_BWD(TT, NV) do { if (rows == 16) LN_BWD_(TT, NV, 16); else LN_BWD_(TT, NV, 32); } while (0)
  if (dy_dtype == MFP_F32) {
    if (nvec == 1) LN_BWD(float, 1); else if (nvec == 2) LN_BWD(float, 2); else LN_BWD(float, 4);
  } else {
    if (nvec == 1) LN_BWD(unsigned short, 1); else if (nvec == 2) LN_BWD(unsigned short, 2); else LN_BWD(unsigned short, 4);
  }
#undef LN_BWD
#undef LN_BWD_
  MFP_CHECK_LAUNCH();
  if (dgamma == nullptr) return MFP_OK;   // partials only: the caller reduces them (mfp_reduce_partials)
  // one launch: columns [0,D) -> dgamma, [D,2D) -> dbeta, [2D,3D) -> dropout bias gradient
  launch_reduce_rows3(part, dgamma, dbeta, drop_colsum, D, 2 * D, nblk, ddrop != nullptr ? 3 * D : 2 * D, 3 * D, st);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* mean,
                                 const float* rstd, const float* dres, float* dx, float* dgamma,
                                 float* dbeta, void* workspace, size_t workspace_bytes, int32_t T,
                                 int32_t D, int32_t dy_dtype, void* ddrop, float* drop_colsum, float drop_p,
                                 uint64_t seed, uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream) {
  return ln_bwd_impl<float>("mfp_layernorm_bwd", dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, workspace, workspace_bytes, T, D,
                            dy_dtype, ddrop, drop_colsum, drop_p, seed, offset, step_ptr, stream);
}

extern "C" int mfp_layernorm_bwd_res16(const void* dy, const float* x, const float* gamma, const float* mean,
                                       const float* rstd, const void* dres, void* dx, float* dgamma,
                                       float* dbeta, void* workspace, size_t workspace_bytes, int32_t T,
                                       int32_t D, int32_t dy_dtype, void* ddrop, float* drop_colsum, float drop_p,
                                       uint64_t seed, uint64_t offset, const int32_t* step_ptr, mfp_stream_t stream) {
  return ln_bwd_impl<unsigned short>("mfp_layernorm_bwd_res16", dy, x, gamma, mean, rstd, reinterpret_cast<const unsigned short*>(dres),
                                     reinterpret_cast<unsigned short*>(dx), dgamma, dbeta, workspace, workspace_bytes, T, D, dy_dtype,
                                     ddrop, drop_colsum, drop_p, seed, offset, step_ptr, stream);
}

extern "C" int mfp_layernorm_bwd_xhat(const void* dy, const void* xhat, const float* gamma, const float* rstd, const void* dres, void* dx,
                                      float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, int32_t T, int32_t D,
                                      int32_t dy_dtype, void* ddrop, float* drop_colsum, float drop_p, uint64_t seed, uint64_t offset,
                                      const int32_t* step_ptr, mfp_stream_t stream) {
  return ln_bwd_impl<unsigned short, true>("mfp_layernorm_bwd_xhat", dy, reinterpret_cast<const float*>(xhat), gamma, nullptr, rstd,
                                           reinterpret_cast<const unsigned short*>(dres), reinterpret_cast<unsigned short*>(dx), dgamma, dbeta,
                                           workspace, workspace_bytes, T, D, dy_dtype, ddrop, drop_colsum, drop_p, seed, offset, step_ptr, stream);
}

extern "C" int32_t mfp_layernorm_bwd_partial_rows(int32_t T) { const int rows = ln_bwd_rows(T); return (T + rows - 1) / rows; }

extern "C" int mfp_reduce_partials(const float* part, float* out0, float* out1, float* out2, int64_t split1,
                                   int64_t split2, int32_t P, int64_t N, int64_t pstride, mfp_stream_t stream) {
  MFP_CHECK_ARG(part && out0 && P > 0 && N > 0 && pstride >= N && split1 >= 0 && split2 >= split1);
  MFP_CHECK_ARG((split1 >= N || out1) && (split2 >= N || out2));
  launch_reduce_rows3(part, out0, out1, out2, split1, split2, P, N, pstride, reinterpret_cast<hipStream_t>(stream));
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

extern "C" int mfp_reduce_partials_batch(const mfp_reduce_job* jobs, int32_t njobs, mfp_stream_t stream) {
  MFP_CHECK_ARG(jobs && njobs > 0 && njobs <= MFP_MAX_REDUCE_JOBS);
  ReduceJobs rj;
  long long maxn = 0;
  for (int i = 0; i < MFP_MAX_REDUCE_JOBS; ++i) {
    const mfp_reduce_job& j = jobs[i < njobs ? i : 0];
    if (i < njobs) {
      MFP_CHECK_ARG(j.part && j.out0 && j.P > 0 && j.N > 0 && j.N <= 8192 && j.pstride >= j.N && j.split1 >= 0 && j.split2 >= j.split1);
      MFP_CHECK_ARG((j.split1 >= j.N || j.out1) && (j.split2 >= j.N || j.out2));
      if (j.N > maxn) maxn = j.N;
    }
    rj.part[i] = j.part; rj.out0[i] = j.out0; rj.out1[i] = j.out1; rj.out2[i] = j.out2;
    rj.split1[i] = j.split1; rj.split2[i] = j.split2; rj.N[i] = j.N; rj.pstride[i] = j.pstride; rj.P[i] = j.P;
  }
  hipLaunchKernelGGL(reduce_rows_multi_kernel<16>, dim3((unsigned)((maxn + 15) / 16), (unsigned)njobs), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), rj);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}
