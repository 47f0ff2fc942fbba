// stem_bf16_kernel (csrc/stem_bf16.h) with the waves of a workgroup specialised -- measured in round 3 and NOT adopted:
// 0.646 ms per 64 HD frames against 0.652 for the kernel it was meant to replace (bit-identical output).  Ablations in
// experiments/harness/stem_bf16_bench.hip: without the K loop 0.45 ms, without K loop and pooling 0.29; the pooling and
// the window's trip from HBM are hidden behind the computing waves as intended (0.62 without either), but ONE computing
// wave per SIMD exposes its own LDS read latency inside the K loop, which two waves per SIMD of the adopted kernel
// cover for each other -- the chain acc init -> K loop -> tile write of a computing wave stays serial.
#pragma once
#include "../feature-point-cnn_amd/csrc/stem_bf16.h"

namespace fpc {

// ---------------------------------------------------------------------------------
// The same tile, the same LDS layouts, the same arithmetic -- with the waves of a workgroup SPECIALISED.  In
// stem_bf16_kernel every wave walks stage -> barrier -> K loop -> tile -> barrier -> pool, two workgroups per CU: by the
// counters the matrix cores are 35 % busy and the LDS 48 %, i.e. neither bounds it; what does is each workgroup's serial
// chain of latencies (the window's trip from HBM, the LDS round trips of the staging and of the pooling, two barriers),
// which two workgroups per CU only half hide.  Here a workgroup is EIGHT waves, one per CU: waves 0-3 (one per SIMD)
// only run K loops -- window buffer -> 44 MFMAs -> bf16 tile buffer -- back to back; waves 4-7 (their SIMD partners)
// feed and drain them: request the window of tile i + 2, convert + stage the window of tile i, pool + store tile i - 2.
// Window and tile are double-buffered in LDS (2 x 22 KB + 2 x 34 KB), ONE barrier per tile for all eight waves; a
// bf16 MFMA leaves the VALU and the LDS port to the partner wave (unlike the fp32 one).
// ---------------------------------------------------------------------------------
constexpr int SB3_THREADS = 512;
template <int CIN>
struct StemB3Cfg {
  using C2 = StemB2Cfg<CIN>;
  static constexpr int WIN_BYTES = (C2::WIN_BYTES + 15) / 16 * 16;
  static constexpr int OFF_TILE = 2 * WIN_BYTES, OFF_BIAS = OFF_TILE + 2 * SB2_TILE_BYTES, OFF_SPARE = OFF_BIAS + 256;
  static constexpr int LDS_BYTES = OFF_SPARE + 16;
};

template <int CIN, unsigned ABL = 0>
__global__ __launch_bounds__(SB3_THREADS, 2) void stem_bf16_ws_kernel(const StemX3Args a) {
  using C = StemB2Cfg<CIN>;
  using C3 = StemB3Cfg<CIN>;
  constexpr int ROWS = C::ROWS, STEPS = C::STEPS, IT = C::IT;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* bias_lds = reinterpret_cast<float*>(lds_raw + C3::OFF_BIAS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the role branch below is a scalar branch
  const bool helper = wave8 >= 4;
  const int wave = wave8 & 3, htid = tid & 255;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y;
  // XCD k (workgroups with blockIdx.x & 7 == k) walks the tiles [k T / 8, (k + 1) T / 8) of the launch
  const int T = tiles * a.frames, per = gridDim.x >> 3, slot0 = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  const int t_end = (int)(((long long)(xcd + 1) * T) >> 3);
  const int t0 = (int)(((long long)xcd * T) >> 3) + slot0;
  const int n = t0 < t_end ? (t_end - t0 + per - 1) / per : 0;   // tiles of this workgroup: t0 + i per, i < n

  if (tid < 64) bias_lds[tid] = a.bias[tid];
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in), 0, (int)((unsigned)a.frames * CIN * a.H * a.W * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.out, 0, (int)((unsigned)a.frames * a.Hp * a.Wp * 128u), 0x00020000);

  // ONE register array for both roles (the allocator cannot know that a wave is either the one or the other: with two
  // arrays it keeps both alive around the loop and spills): a computing wave's weight fragments of every step, or a
  // feeding wave's two windows in flight -- the tiles i (even) / i + 1 (odd), requested two tiles ahead
  u32x4 R[STEPS * 2];
  static_assert(STEPS * 2 >= 2 * IT, "two windows fit in the fragments' registers");
  u32x4* const va = R;
  u32x4* const vb = R + IT;
  const int srow = htid / SB2_NQ, sq4 = htid - srow * SB2_NQ;   // staging role: window row (+ 25 per pass), float4 of the row
  const bool sact = htid < C::RPP * SB2_NQ;
  const int pc = htid & 7, pg = (htid >> 3) & 3, pj = htid >> 5;   // pooling role: pooled row, column pair, channel group
  auto request = [&](u32x4* v, int i) {   // the input window of the workgroup's tile i (i >= n: nothing is in range, zeros)
    const bool live = i < n;
    const int tc = live ? t0 + i * per : 0;
    const int b = tc / tiles, t = tc - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int iy0 = ty * (4 * SB2_PH) - 5, ixa = tx * (4 * SB2_PW) - 8;
    const int hlim = live ? a.H : 0;
    const unsigned tbase = (unsigned)((b * CIN * a.H + iy0) * a.W + ixa) * 4u;
    const bool xok = sact & ((unsigned)(ixa + 4 * sq4) < (unsigned)a.W);
    unsigned toff = tbase + (unsigned)((srow * a.W + 4 * sq4) * 4);
    asm volatile("" : "+v"(toff));
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int row = srow + k * C::RPP;
      const int c = (row >= SB2_ROWS ? 1 : 0) + (row >= 2 * SB2_ROWS ? 1 : 0);
      const int iy = iy0 + row - c * SB2_ROWS;
      const bool ok = xok & (row < C::WROWS) & ((unsigned)iy < (unsigned)hlim);
      const unsigned off = toff + (unsigned)((k * C::RPP + c * (a.H - SB2_ROWS)) * a.W * 4);
      v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(ok ? off : 0xfffffff0u), 0, 0));
    }
  };
  auto stage = [&](u32x4* v, unsigned char* win) {   // fp32 -> bf16, both copies of the window
    const int pe0 = (srow * SB2_PITCH + 2 * sq4) * 4;
    // (pinned behind the step's barrier: the conversions are register arithmetic, and the compiler lifts those of the
    // NEXT step's window in front of the barrier -- waiting there for loads requested one step ago instead of two)
#pragma unroll
    for (int k = 0; k < IT; ++k) asm volatile("" : "+v"(v[k]));
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const f32x4 f = __builtin_bit_cast(f32x4, v[k]);
      const unsigned lo = pk_bf16(f.x, f.y), hi = pk_bf16(f.z, f.w);
      const bool in = sact & (srow + k * C::RPP < C::WROWS);
      unsigned char* pe = in ? win + pe0 + k * (C::RPP * SB2_PITCH * 4) : lds_raw + C3::OFF_SPARE;
      unsigned char* po = in ? win + pe0 + k * (C::RPP * SB2_PITCH * 4) + C::COPY1 : lds_raw + C3::OFF_SPARE + 8;
      *reinterpret_cast<uint2*>(pe) = make_uint2(lo, hi);
      unsigned* o = reinterpret_cast<unsigned*>(po);
      o[0] = lo;
      o[1] = hi;
    }
  };
  auto pool = [&](const unsigned char* tile, int i) {   // 3x3/2 max-pool + ReLU of the workgroup's tile i, 16-byte stores
    const bool live = (i >= 0) & (i < n);
    const int tc = live ? t0 + i * per : 0;
    const int b = tc / tiles, t = tc - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int gpy = ty * SB2_PH + pj;
    u32x4 cm[5];
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) {
      const int q = 4 * pg + cc < SB2_Q ? 4 * pg + cc : 4 * pg + cc == SB2_Q ? 14 : 11;   // (see stem_bf16_kernel)
      const unsigned char* p0 = tile + sb2_tile_addr(2 * pj, q, pc);
      const u32x4 r0 = *reinterpret_cast<const u32x4*>(p0),
                  r1 = *reinterpret_cast<const u32x4*>(tile + ((sb2_tile_addr(2 * pj, q, pc) + 16 * SB2_PIX) ^ 32)),
                  r2 = *reinterpret_cast<const u32x4*>(p0 + 32 * SB2_PIX);
      cm[cc].x = stemb_pk_max(stemb_pk_max(r0.x, r1.x), r2.x);
      cm[cc].y = stemb_pk_max(stemb_pk_max(r0.y, r1.y), r2.y);
      cm[cc].z = stemb_pk_max(stemb_pk_max(r0.z, r1.z), r2.z);
      cm[cc].w = stemb_pk_max(stemb_pk_max(r0.w, r1.w), r2.w);
    }
    const int gpx = tx * SB2_PW + 2 * pg;
    const unsigned ooff = (unsigned)(((b * a.Hp + gpy) * a.Wp + gpx) * 64 + pc * 8) * 2u;
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      u32x4 o;
      o.x = stemb_pk_max(stemb_pk_max(stemb_pk_max(cm[2 * px].x, cm[2 * px + 1].x), cm[2 * px + 2].x), 0u);
      o.y = stemb_pk_max(stemb_pk_max(stemb_pk_max(cm[2 * px].y, cm[2 * px + 1].y), cm[2 * px + 2].y), 0u);
      o.z = stemb_pk_max(stemb_pk_max(stemb_pk_max(cm[2 * px].z, cm[2 * px + 1].z), cm[2 * px + 2].z), 0u);
      o.w = stemb_pk_max(stemb_pk_max(stemb_pk_max(cm[2 * px].w, cm[2 * px + 1].w), cm[2 * px + 2].w), 0u);
      const bool on = live & (!(ABL & STEMB_ABL_STORE) || o.x == 0x12345678u) & (gpy < a.Hp) & (2 * pg + px < SB2_PW) & (gpx + px < a.Wp);
      __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, (int)(on ? ooff + px * 128 : 0xfffffff0u), 0, 0);
    }
  };

  // ---- the computing waves' state
  int prow[2], pcol[2], abase[2], abase8[2], tbase[2], tswz[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int i = l31 >> 3, j = l31 & 7;
    int r = 4 * wave + i, q = 2 * j + mb;
    bool nobody = false;
    if (mb == 1 && j == 7) {
      q = (int)((SB2_SPARE_COL >> (4 * (4 * wave + i))) & 15ull);
      r = SB2_R - 1;
      nobody = q == 15;
      q = nobody ? SB2_Q - 1 : q;
    }
    prow[mb] = r;
    pcol[mb] = q;
    abase[mb] = ((q & 1) ? 0 : C::COPY1) + ((2 * r) * SB2_PITCH + q + 1) * 4;
    abase8[mb] = abase[mb] + 8;
    asm volatile("" : "+v"(abase8[mb]));
    tbase[mb] = (nobody ? (SB2_R - 1) * 16 + 15 : sb2_slot(r, q)) * SB2_PIX + 8 * half;
    tswz[mb] = sb2_swz(r, q);
  }
  auto compute = [&](const unsigned char* win, unsigned char* tile, int i) {   // the workgroup's tile i: window -> bf16 tile
    const int tc = t0 + i * per;
    const int b = tc / tiles, t = tc - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int gy = ty * (2 * SB2_PH) - 1 + prow[mb], gx = tx * (2 * SB2_PW) - 1 + pcol[mb];
      const bool inside = ((unsigned)gy < (unsigned)a.Ho) & ((unsigned)gx < (unsigned)a.Wo);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bq = *reinterpret_cast<const float4*>(bias_lds + nb * 32 + 8 * g + 4 * half);
          acc[mb][nb][4 * g + 0] = inside ? bq.x : -3.0e38f;
          acc[mb][nb][4 * g + 1] = inside ? bq.y : -3.0e38f;
          acc[mb][nb][4 * g + 2] = inside ? bq.z : -3.0e38f;
          acc[mb][nb][4 * g + 3] = inside ? bq.w : -3.0e38f;
        }
    }
#pragma unroll
    for (int s = 0; s < ((ABL & STEMB_ABL_K) ? 0 : STEPS); ++s) {
      constexpr int RB = SB2_PITCH * 4;
      const int r0 = 2 * s < ROWS ? 2 * s : ROWS - 1, r1 = 2 * s + 1 < ROWS ? 2 * s + 1 : ROWS - 1;
      const int off0 = ((r0 / 7) * SB2_ROWS + (r0 % 7)) * RB, off1 = ((r1 / 7) * SB2_ROWS + (r1 % 7)) * RB;
      const int off = half ? off1 : off0;
      uint4 av[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const uint2 lo = *reinterpret_cast<const uint2*>(__builtin_assume_aligned(win + abase[mb] + off, 8));
        const uint2 hi = *reinterpret_cast<const uint2*>(__builtin_assume_aligned(win + abase8[mb] + off, 8));
        av[mb] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, R[s * 2 + nb]), __builtin_bit_cast(bf16x8, av[mb]), acc[mb][nb], 0, 0, 0);
    }
#pragma unroll
    for (int mb = 0; mb < ((ABL & STEMB_ABL_TILE) ? 0 : 2); ++mb) {
      unsigned char* row = tile + tbase[mb];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<uint2*>(row + 16 * ((nb * 4 + g) ^ tswz[mb])) =
              make_uint2(pk_bf16(acc[mb][nb][4 * g], acc[mb][nb][4 * g + 1]), pk_bf16(acc[mb][nb][4 * g + 2], acc[mb][nb][4 * g + 3]));
    }
  };

  unsigned char* const win0 = lds_raw;
  unsigned char* const win1 = lds_raw + C3::WIN_BYTES;
  unsigned char* const tile0 = lds_raw + C3::OFF_TILE;
  unsigned char* const tile1 = lds_raw + C3::OFF_TILE + SB2_TILE_BYTES;
  if (helper) {
    // the queue of outstanding requests the loop is entered with = the one its back edge has: window, two stores,
    // window, two stores (see stem_bf16_kernel: otherwise the compiler's wait for a window takes the stores along)
    const u32x4 z = {0u, 0u, 0u, 0u};
    request(va, 0);
    __builtin_amdgcn_raw_buffer_store_b128(z, orsrc, (int)0xfffffff0u, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(z, orsrc, (int)0xffffffe0u, 0, 0);
    request(vb, 1);
    __builtin_amdgcn_raw_buffer_store_b128(z, orsrc, (int)0xffffffd0u, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(z, orsrc, (int)0xffffffc0u, 0, 0);
  } else {
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) R[s * 2 + nb] = __builtin_bit_cast(u32x4, a.wfrag[(s * 2 + nb) * 64 + lane]);
  }
  FPC_LDS_BARRIER();   // the biases are in LDS

  // step i (two per trip, so that buffers and register sets have compile-time names): the feeding waves stage tile i,
  // request tile i + 2 and pool tile i - 2; the computing waves run tile i - 1; one barrier.  n + 2 steps in all.  The
  // two roles run their OWN loops -- the same number of barriers in each (n is the workgroup's) -- so that the
  // compiler's count of a feeding wave's outstanding requests is exact (in one loop with a role branch it merges the
  // two paths' states and waits for the window requested ONE step ago when it needs the one requested two steps ago).
  if (helper) {
    for (int i = 0; i < n + 2; i += 2) {
      stage(va, win0);
      if constexpr (!(ABL & STEMB_ABL_LOAD)) request(va, i + 2);
      if constexpr (!(ABL & STEMB_ABL_POOL)) pool(tile0, i - 2);
      FPC_LDS_BARRIER();
      stage(vb, win1);
      if constexpr (!(ABL & STEMB_ABL_LOAD)) request(vb, i + 3);
      if constexpr (!(ABL & STEMB_ABL_POOL)) pool(tile1, i - 1);
      FPC_LDS_BARRIER();
    }
  } else {
    for (int i = 0; i < n + 2; i += 2) {
      if (i - 1 >= 0 && i - 1 < n) compute(win1, tile1, i - 1);
      FPC_LDS_BARRIER();
      if (i < n) compute(win0, tile0, i);
      FPC_LDS_BARRIER();
    }
  }
}

}  // namespace fpc
