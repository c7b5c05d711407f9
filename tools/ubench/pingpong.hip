// Microbenchmark (round 6): ONE weight chunk of the activation-stationary block kernels (block_attn.hip's q / k / v chunk) with
// everything a chunk carries -- LDS-DMA of a later chunk (4 x 1 KB pieces per wave, L2-resident weights), 16 fragment reads + 32
// MFMAs per wave, bias / pack epilogue into a swizzled [128][128 B] image, stash of the previous image to HBM (2 x 16-byte stores
// per thread), counted wait, barrier -- under three schedules:
//   MODE 0  in phase (the kernels of rounds 1-5): all eight waves run the same step of the same chunk at the same time
//   MODE 1  ping-pong: waves 0-3 (X, column half 0) and 4-7 (Y, column half 1) run half a chunk apart -- on every SIMD one wave
//           multiplies while its partner runs the epilogue / DMA issue / stash of ITS last chunk; a barrier per half chunk
//   MODE 2  MODE 1 + s_setprio 1 on the multiplying wave
//   MODE 3  in phase, the epilogue of chunk c - 1 issued inside the products of chunk c (same wave, accumulators double-buffered)
// Prints microseconds per chunk per CU (256 workgroups of 512 threads, one per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ unsigned int pk(float a, float b) { const f2 v = {a, b}; return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf2)); }
__device__ __forceinline__ int isw(int row) { return (row >> 1) & 7; }
constexpr int IMG = 16384, WSB = 32768, WS_OFF = 3 * IMG, VEC_OFF = WS_OFF + 3 * WSB, LDS = VEC_OFF + 1024;

#define WAITV(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(n) : "memory")

template <int MODE, bool STASH = true, bool DMA = true, int PF = 1>
__global__ __launch_bounds__(512) void k(const unsigned short* W, unsigned short* out, int nch, int nwch, unsigned int obytes, unsigned long long* tr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Im = smem;
  unsigned char* const Ws = smem + WS_OFF;
  float* const Bq = reinterpret_cast<float*>(smem + VEC_OFF);
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rp = wave & 3, nh = wave >> 2;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, nwch * WSB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(out, 0, obytes, 0x00020000);
  if (tid < 256) Bq[tid] = 0.01f * tid;
  bf16x8 xf[2][8];
  for (int rt = 0; rt < 2; ++rt) for (int ks = 0; ks < 8; ++ks) for (int e = 0; e < 8; ++e) xf[rt][ks][e] = (short)(0x3c00 + ((lane * 7 + ks * 3 + rt + e) & 63));
  int xs[4];
  for (int ks = 0; ks < 4; ++ks) xs[ks] = ((ks * 4 + g) ^ li) << 4;
  const unsigned int w1off = (unsigned int)((wave * 8 + (lane >> 5)) * 512 + (((lane & 31) ^ ((wave & 1) * 8 + (lane >> 5))) << 4));
  const int row0 = blockIdx.x * 128;

  auto wload = [&](int c) {
    if (!DMA) return;
    unsigned char* dst = Ws + (c % 3) * WSB + wave * 4096;
    const int base = (c % nwch) * WSB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_u8*)(dst + i * 1024), 16, w1off ^ (i << 5), base + i * 1024, 0, 0);
  };
  auto stash = [&](int c) {
    if (!STASH) return;
    const unsigned char* img = Im + (c % 3) * IMG;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i, r = idx >> 3, c16 = idx & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(img + r * 128 + ((c16 ^ isw(r)) << 4));
      __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, (unsigned int)(row0 + r) * 1536 + (c % 12) * 128 + c16 * 16, 0, 0);
    }
  };
  auto compute = [&](int c, f32x4 (&acc)[2][2]) {
    const unsigned char* wa = Ws + (c % 3) * WSB + ((nh * 2) * 16 + li) * 512;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[nt][0] = acc[nt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (PF >= 9) {
      // every fragment read written up front, the interleave pinned: DEPTH reads ahead, then one read per two products
      constexpr int DEPTH = PF - 5;      // PF 9 -> 4 fragments in flight, 10 -> 5, 11 -> 6
      bf16x8 wq[8][2];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) wq[ks][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[ks & 3] + (ks >> 2) * 256);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) acc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ks][nt], xf[rt][ks], acc[nt][rt], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, DEPTH, 0);
#pragma unroll
      for (int i = 0; i < 16 - DEPTH; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * DEPTH, 0);
      return;
    }
    constexpr int NB = PF + 1 > 8 ? 8 : PF + 1;      // fragment buffers: PF k-steps ahead
    bf16x8 wf[NB][2];
#pragma unroll
    for (int k0 = 0; k0 < PF && k0 < 8; ++k0)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) wf[k0 % NB][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[k0 & 3] + (k0 >> 2) * 256);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + PF < 8) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) wf[(ks + PF) % NB][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[(ks + PF) & 3] + ((ks + PF) >> 2) * 256);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) acc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks % NB][nt], xf[rt][ks], acc[nt][rt], 0, 0, 0);
    }
  };
  auto epilogue = [&](int c, const f32x4 (&acc)[2][2]) {
    unsigned char* img = Im + (c % 3) * IMG;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(Bq + (c & 3) * 64 + (nh * 2 + nt) * 16 + 4 * g);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int row = rp * 32 + rt * 16 + li;
        const u32x2 p2 = {pk(acc[nt][rt][0] + bb[0], acc[nt][rt][1] + bb[1]), pk(acc[nt][rt][2] + bb[2], acc[nt][rt][3] + bb[3])};
        *reinterpret_cast<u32x2*>(img + row * 128 + ((((nh * 2 + nt) * 2 + (g >> 1)) ^ isw(row)) << 4) + (g & 1) * 8) = p2;
      }
    }
  };

  wload(0);
  wload(1);
  if (MODE == 1 || MODE == 2) { if (nh) wload(2); }
  WAITV(0);
  __builtin_amdgcn_s_barrier();

  if (MODE == 4) {
    // in phase; chunk c + 2 is awaited at the END of chunk c (it was requested at its head), so chunk c + 1 is readable during
    // chunk c: its first fragments are requested before chunk c's epilogue
    bf16x8 wn[2];
    {
      const unsigned char* wa = Ws + ((nh * 2) * 16 + li) * 512;
      for (int nt = 0; nt < 2; ++nt) wn[nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[0]);
    }
    for (int c = 0; c < nch; ++c) {
      f32x4 acc[2][2];
      wload(c + 2);
      if (c >= 1) stash(c - 1);
      const unsigned char* wa = Ws + (c % 3) * WSB + ((nh * 2) * 16 + li) * 512;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[nt][0] = acc[nt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      bf16x8 wf[2][2];
      wf[0][0] = wn[0]; wf[0][1] = wn[1];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) wf[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[(ks + 1) & 3] + ((ks + 1) >> 2) * 256);
        } else {
          const unsigned char* wb = Ws + ((c + 1) % 3) * WSB + ((nh * 2) * 16 + li) * 512;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) wn[nt] = *reinterpret_cast<const bf16x8*>(wb + nt * 8192 + xs[0]);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) acc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][nt], xf[rt][ks], acc[nt][rt], 0, 0, 0);
      }
      epilogue(c, acc);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STASH ? 2 : 0) : "memory");
      asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");      // (the two prefetched fragments may stay in flight)
      __builtin_amdgcn_s_barrier();
    }
  } else if (MODE == 7 || MODE == 8) {
    // ONE barrier per chunk as MODE 0, but the two waves of a SIMD walk the chunk's steps in a different ORDER: X (waves 0-3)
    // multiplies first and issues its DMA pieces / stash stores last, Y (waves 4-7) issues them first and multiplies last
    for (int c = 0; c < nch; ++c) {
      f32x4 acc[2][2];
      if (nh) { wload(c + 2); if (c >= 1) stash(c - 1); }
      if (MODE == 8 && nh) __builtin_amdgcn_sched_barrier(0);
      compute(c, acc);
      epilogue(c, acc);
      if (MODE == 8 && !nh) __builtin_amdgcn_sched_barrier(0);
      if (!nh) { wload(c + 2); if (c >= 1) stash(c - 1); }
      if (c == 0) WAITV(4); else if (c == 1) WAITV(6); else WAITV(8);
      __builtin_amdgcn_s_barrier();
    }
  } else if (MODE == 9 || MODE == 10) {
    // MODE 0 / MODE 7 with shader-clock stamps between the steps of a chunk (sums over the chunks, per wave)
    unsigned long long acc_t[6] = {0, 0, 0, 0, 0, 0};
    auto now = [&]() { unsigned long long t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return t; };
    for (int c = 0; c < nch; ++c) {
      f32x4 acc[2][2];
      const bool first = MODE == 9 || nh;
      unsigned long long t0 = now();
      if (first) { wload(c + 2); if (c >= 1) stash(c - 1); }
      __builtin_amdgcn_sched_barrier(0);
      unsigned long long t1 = now();
      compute(c, acc);
      __builtin_amdgcn_sched_barrier(0);
      unsigned long long t2 = now();
      epilogue(c, acc);
      __builtin_amdgcn_sched_barrier(0);
      unsigned long long t3 = now();
      if (!first) { wload(c + 2); if (c >= 1) stash(c - 1); }
      __builtin_amdgcn_sched_barrier(0);
      unsigned long long t4 = now();
      if (c == 0) WAITV(4); else if (c == 1) WAITV(6); else WAITV(8);
      unsigned long long t5 = now();
      __builtin_amdgcn_s_barrier();
      unsigned long long t6 = now();
      acc_t[0] += t1 - t0; acc_t[1] += t2 - t1; acc_t[2] += t3 - t2; acc_t[3] += t4 - t3; acc_t[4] += t5 - t4; acc_t[5] += t6 - t5;
    }
    if (lane == 0 && blockIdx.x == 0) for (int i = 0; i < 6; ++i) tr[wave * 6 + i] = acc_t[i];
  } else if (MODE == 5 || MODE == 6) {
    for (int c = 0; c < nch; ++c) {
      f32x4 acc[2][2];
      wload(c + 2);
      if (c >= 1) stash(c - 1);
      compute(c, acc);
      if (MODE == 6) epilogue(c, acc);
      else asm volatile("" :: "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[1][0]), "v"(acc[1][1]));
      if (c == 0) WAITV(4); else if (c == 1) WAITV(6); else WAITV(8);
      if (MODE == 5) __builtin_amdgcn_s_barrier();
    }
  } else if (MODE == 11) {
    if (nh) __builtin_amdgcn_s_setprio(1);
    for (int c = 0; c < nch; ++c) {
      f32x4 acc[2][2];
      wload(c + 2);
      if (c >= 1) stash(c - 1);
      compute(c, acc);
      epilogue(c, acc);
      if (c == 0) WAITV(4); else if (c == 1) WAITV(6); else WAITV(8);
      __builtin_amdgcn_s_barrier();
    }
  } else if (MODE == 12 || MODE == 13) {
    // the older half (waves 0-3, which wins every arbitration and then idles at the barrier) takes the younger half's stash
    // stores (MODE 12) or its DMA pieces too (MODE 13): X walks rows 0..127 of the image alone / loads wave w's and w + 4's pieces
    for (int c = 0; c < nch; ++c) {
      f32x4 acc[2][2];
      if (MODE == 13) {
        if (!nh) {
          unsigned char* dst = Ws + ((c + 2) % 3) * WSB;
          const int base = ((c + 2) % nwch) * WSB;
#pragma unroll
          for (int hw = 0; hw < 2; ++hw) {
            const int wv = wave + 4 * hw;
            const unsigned int off = (unsigned int)((wv * 8 + (lane >> 5)) * 512 + (((lane & 31) ^ ((wv & 1) * 8 + (lane >> 5))) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_u8*)(dst + wv * 4096 + i * 1024), 16, off ^ (i << 5), base + i * 1024, 0, 0);
          }
        }
      } else wload(c + 2);
      if (c >= 1 && !nh && STASH) {
        const unsigned char* img = Im + ((c - 1) % 3) * IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int idx = tid + 256 * i, r = idx >> 3, c16 = idx & 7;
          const u32x4 v = *reinterpret_cast<const u32x4*>(img + r * 128 + ((c16 ^ isw(r)) << 4));
          __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, (unsigned int)(row0 + r) * 1536 + ((c - 1) % 12) * 128 + c16 * 16, 0, 0);
        }
      }
      compute(c, acc);
      epilogue(c, acc);
      // X: 4 or 8 loads + 4 stores per chunk; Y: 4 or 0 loads, no stores
      if (c < 2) WAITV(0);
      else if (!nh) { if (MODE == 13) WAITV(16); else WAITV(12); }
      else { if (MODE == 13) WAITV(0); else WAITV(4); }
      __builtin_amdgcn_s_barrier();
    }
  } else if (MODE == 0) {
    for (int c = 0; c < nch; ++c) {
      f32x4 acc[2][2];
      wload(c + 2);
      if (c >= 1) stash(c - 1);
      compute(c, acc);
      epilogue(c, acc);
      if (c == 0) WAITV(4); else if (c == 1) WAITV(6); else WAITV(8);
      __builtin_amdgcn_s_barrier();
    }
  } else if (MODE == 3) {
    f32x4 acc[2][2][2];
    for (int c2 = 0; c2 < nch; c2 += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = c2 + h;
        wload(c + 2);
        if (c >= 2) stash(c - 2);
        compute(c, acc[h]);
        if (c >= 1) epilogue(c - 1, acc[h ^ 1]);
        if (c < 3) WAITV(0); else WAITV(8);
        __builtin_amdgcn_s_barrier();
      }
    }
  } else {
    // X: compute(c) in slot 2 c, post(c) in slot 2 c + 1 (epilogue, wload(c + 2), stash(c - 1), wait for chunk c + 1)
    // Y: compute(c) in slot 2 c + 1 (then wait for chunk c + 1), post(c) in slot 2 c + 2 (epilogue, wload(c + 3), stash(c - 1))
    f32x4 acc[2][2];
    const int nslot = 2 * nch + 2;
    for (int s = 0; s < nslot; ++s) {
      const int sx = s - nh;            // this group's own slot clock
      const int c = sx >> 1;
      if (sx >= 0 && c < nch) {
        if (!(sx & 1)) {
          if (MODE == 2) __builtin_amdgcn_s_setprio(1);
          compute(c, acc);
          if (MODE == 2) __builtin_amdgcn_s_setprio(0);
          if (nh) { if (s < 8) WAITV(0); else WAITV(8); }
        } else {
          epilogue(c, acc);
          const int cl = c + 2 + nh;
          wload(cl);
          if (c >= 1) stash(c - 1);
          if (!nh) { if (s < 8) WAITV(0); else WAITV(8); }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  WAITV(0);
}

template <int MODE, bool STASH = true, bool DMA = true, int PF = 1>
void run(const unsigned short* W, unsigned short* out, unsigned int obytes, const char* what) {
  static unsigned long long* tr = nullptr; if (!tr) hipMalloc(&tr, 8 * 6 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nch = 480, nwch = 48;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, STASH, DMA, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, STASH, DMA, PF>), dim3(256), dim3(512), LDS, 0, W, out, nch, nwch, obytes, tr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("%-44s %.3f us per chunk (128 rows x 64 cols x K 256 per CU)\n", what, ms * 1e3 / nch);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("error: %s\n", hipGetErrorString(e));
  if (MODE == 9 || MODE == 10) {
    unsigned long long h[48]; hipMemcpy(h, tr, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[6] = {"dma+stash (first)", "fragment reads + products (issue)", "epilogue", "dma+stash (last)", "counted wait", "barrier"};
    for (int w = 0; w < 8; w += 4) for (int i = 0; i < 6; ++i) printf("    wave %d  %-36s %7.0f clocks per chunk\n", w, nm[i], (double)h[w * 6 + i] / nch);
  }
  // checksum of the stash: the schedules must leave the same bytes
  unsigned short* ho = (unsigned short*)malloc(obytes);
  hipMemcpy(ho, out, obytes, hipMemcpyDeviceToHost);
  unsigned long long cs = 0; for (unsigned int i = 0; i < obytes / 2; ++i) cs = cs * 1315423911ull + ho[i];
  printf("    stash checksum %llx\n", cs);
  free(ho);
  hipMemset(out, 0, obytes);
}
int main() {
  unsigned short* W; unsigned short* out;
  const unsigned int obytes = 32768u * 1536u;
  hipMalloc(&W, 48 * WSB); hipMalloc(&out, obytes);
  unsigned short* h = (unsigned short*)malloc(48 * WSB);
  for (int i = 0; i < 48 * WSB / 2; ++i) h[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 26));
  hipMemcpy(W, h, 48 * WSB, hipMemcpyHostToDevice);
  run<0>(W, out, obytes, "in phase (shipped schedule)");
  run<1>(W, out, obytes, "ping-pong X / Y half a chunk apart");
  run<2>(W, out, obytes, "ping-pong + setprio on the multiplying wave");
  run<3>(W, out, obytes, "in phase, epilogue(c-1) inside products(c)");
  run<4>(W, out, obytes, "in phase, next chunk's fragments early");
  run<4, false>(W, out, obytes, "in phase, next fragments early, no stash");
  run<4, false, false>(W, out, obytes, "... no stash, no DMA");
  run<5, false, false>(W, out, obytes, "no epilogue, barrier, no stash, no DMA");
  run<6, false, false>(W, out, obytes, "epilogue, NO barrier, no stash, no DMA");
  run<7>(W, out, obytes, "X: products first, Y: DMA + stash first");
  run<8>(W, out, obytes, "... with scheduling fences");
  run<7, false>(W, out, obytes, "X / Y order, no stash");
  run<7, false, false>(W, out, obytes, "X / Y order, no stash, no DMA");
  run<7, true, true, 3>(W, out, obytes, "X / Y order, fragments 3 k-steps ahead");
  run<7, true, true, 8>(W, out, obytes, "X / Y order, all fragments up front");
  run<0, true, true, 8>(W, out, obytes, "in phase, all fragments up front");
  run<0, true, true, 2>(W, out, obytes, "in phase, fragments 2 k-steps ahead");
  run<7, true, true, 2>(W, out, obytes, "X / Y order, fragments 2 k-steps ahead");
  run<10, true, true, 3>(W, out, obytes, "X / Y order, 3 ahead, stamped");
  run<11>(W, out, obytes, "in phase, waves 4-7 at priority 1");
  run<12>(W, out, obytes, "waves 0-3 carry all stash stores");
  run<13>(W, out, obytes, "waves 0-3 carry all stash stores + DMA");
  run<0, true, true, 9>(W, out, obytes, "in phase, pinned interleave 4 deep");
  run<0, true, true, 11>(W, out, obytes, "in phase, pinned interleave 6 deep");
  run<7, true, true, 9>(W, out, obytes, "X / Y order, pinned interleave 4 deep");
  run<7, true, true, 11>(W, out, obytes, "X / Y order, pinned interleave 6 deep");
  run<10, true, true, 11>(W, out, obytes, "X / Y order, pinned 6 deep, stamped");
  run<1, true, true, 11>(W, out, obytes, "ping-pong, pinned 6 deep");
  run<9>(W, out, obytes, "in phase, stamped");
  run<10>(W, out, obytes, "X / Y order, stamped");
  run<0, false>(W, out, obytes, "in phase, no stash stores");
  run<0, true, false>(W, out, obytes, "in phase, no DMA");
  run<0, false, false>(W, out, obytes, "in phase, no stash, no DMA");
  run<1, false>(W, out, obytes, "ping-pong, no stash stores");
  run<1, false, false>(W, out, obytes, "ping-pong, no stash, no DMA");
  run<1, true, true, 3>(W, out, obytes, "ping-pong, fragments 3 k-steps ahead");
  run<1, true, true, 8>(W, out, obytes, "ping-pong, all fragments up front");
  run<1, false, true, 8>(W, out, obytes, "ping-pong, all fragments up front, no stash");
  run<0, true, true, 3>(W, out, obytes, "in phase, fragments 3 k-steps ahead");
  run<3, false>(W, out, obytes, "in phase, epilogue inside, no stash");
  return 0;
}
