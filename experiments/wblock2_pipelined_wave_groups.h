// wblock2_mfma.h -- the Winograd ResNetBlock kernel of wblock_mfma.h with its first phase software-pipelined across
// the two waves that share a SIMD.
//
// In wblock_mfma_kernel all eight waves walk the same phases together: stage a channel chunk, transform it (VALU +
// LDS, matrix cores idle), multiply (MFMA, VALU idle), next chunk.  Here the waves form two groups -- waves 0..3 and
// 4..7, one of each per SIMD (wave w and w + 4 share one) -- that run half a period apart:
//
//     period c:   group 0:  transform(c + 1)  then  GEMM(c)
//                 group 1:  GEMM(c)           then  transform(c + 1)
//
// so while one wave of a SIMD feeds the matrix cores the other one does the input transform of the NEXT chunk (each
// group produces two of the four transform rows for every tile and channel), and one barrier per period is all the
// synchronisation there is.  That takes two V buffers and three halo buffers in LDS, which fit with 16-channel chunks:
// 3 x 14.4 KB + 2 x 40 KB = 122 KB (the second half of the tile needs 141 KB anyway).  Chunk k of a tile is written to
// halo buffer k % 3 at the start of period k - 2, transformed in period k - 1 into V[k & 1], multiplied in period k.
// Chunk k is requested from global memory at the end of period k - 4 (see `request`).
#pragma once
#include "wblock_mfma.h"

namespace fpc {

template <int KC, int NBT, int CMID_>
struct WBlock2Cfg {
  static constexpr int TH = 8, TW = 16, NT = 512;
  static constexpr int HW = TW + 2, HH = TH + 2;
  static constexpr int ROW4 = KC / 4 + 1;                       // float4 per halo pixel / per V row (odd: conflict-free)
  static constexpr int CMID = CMID_;
  static constexpr int ROWH4 = CMID / 4 + 1;
  static constexpr int HALO_BYTES = HH * HW * ROW4 * 16;
  static constexpr int V_BYTES = 16 * 32 * ROW4 * 16;
  static constexpr int M_BYTES = 16 * 32 * 36 * 4;
  static constexpr int H_BYTES = 128 * ROWH4 * 16;
  static constexpr int P1_BYTES = 3 * HALO_BYTES + 2 * V_BYTES;
  static constexpr int LDS_BYTES = P1_BYTES > (M_BYTES + H_BYTES) ? P1_BYTES : (M_BYTES + H_BYTES);
  static constexpr int GN = NBT >= 4 ? 2 : 1;
  static constexpr int GX = 8 / GN;
  static constexpr int PX = 16 / GX;
  static constexpr int NBW = NBT / GN;
  static_assert(KC == 16, "the item mapping of the split transform assumes 8 channel pairs per chunk");
};

template <int KC, int NBT, int CMID_>
__global__ __launch_bounds__(512, 2) void wblock2_mfma_kernel(const WBlockArgs a) {
  using C = WBlock2Cfg<KC, NBT, CMID_>;
  constexpr int NT = C::NT, TH = C::TH, TW = C::TW, HW = C::HW, HH = C::HH, ROW4 = C::ROW4, K8 = KC / 8, KC4 = KC / 4;
  constexpr int NV = HH * HW * KC4, ITER = (NV + NT - 1) / NT, ROWH4 = C::ROWH4, CMID = C::CMID;
  constexpr int GN = C::GN, PX = C::PX, NBW = C::NBW;
  constexpr int HALO4 = C::HALO_BYTES / 16, V4 = C::V_BYTES / 16;
  extern __shared__ float4 lds4[];
  float4* halo4 = lds4;               // three buffers
  float4* vv4 = lds4 + 3 * HALO4;     // two buffers

  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gx = wave / GN, gn = wave % GN;
  const int grp = wave >> 2;          // phase group; wave w and w + 4 run on the same SIMD
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y;
  FPC_STAMP(0)

  typedef float f4v __attribute__((ext_vector_type(4)));   // plain vectors: struct copies turn into memcpy + scratch
  f4v stage0[ITER], stage1[ITER];  // chunk k of a tile travels in set k & 1 (two named arrays: a runtime index
                                      // into one array would put it in scratch memory)
  auto load_chunk = [&](int set, int wg, int chunk) {
    const int bl = wg / tiles;
    const int bb = a.frame0 + bl;
    const int t = wg - bl * tiles;
    const int tyy = t / a.tiles_x, txx = t - tyy * a.tiles_x;
    const int iy0 = tyy * TH - 1, ix0 = txx * TW - 1;
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int e = tid + i * NT;
      const int pix = e / KC4, c4 = e - pix * KC4;
      const int hy = pix / HW, hx = pix - hy * HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = (NV % NT == 0 || e < NV) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const size_t off = ok ? ((size_t)(bb * a.H + iy) * a.W + ix) * a.csx + chunk * KC + c4 * 4 : 0;
      f4v v = *reinterpret_cast<const f4v*>(a.x + off);
      if (!ok) v = (f4v)(0.f);
      if (set == 0) stage0[i] = v; else stage1[i] = v;
    }
  };
  auto store_chunk = [&](int set, int buf) {
    float4* hb = halo4 + buf * HALO4;
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int e = tid + i * NT;
      const int pix = e / KC4, c4 = e - pix * KC4;
      const f4v v = set == 0 ? stage0[i] : stage1[i];
      if (NV % NT == 0 || e < NV) *reinterpret_cast<f4v*>(hb + pix * ROW4 + c4) = v;
    }
  };

  // B fragments of this wave: [chunk][xi][k8][nb][lane]; its positions are contiguous, so step ls = p * K8 + k8 of a
  // chunk is at (chunk * 16 * K8 + ls) steps from the wave's base.  A ring of four steps with static slots (STEPS is a
  // multiple of four); reading ahead of the last step ends in the zero chunk the host appends.
  constexpr int STEPS = PX * K8, RING = 4;
  static_assert(STEPS % RING == 0, "ring period divides the chunk");
  constexpr int stepstride = NBT * 64;
  constexpr bool PIN = NBT != 3;
  const float4* wbase = a.w1 + (size_t)(gx * PX * K8 * NBT + gn * NBW) * 64 + (PIN ? 0 : lane);
  auto ldw = [&](const float4* pw) { return PIN ? fpc_ldg_su(pw, lane16) : *pw; };
  auto wptr = [&](int s) {
    const int chunk = s / STEPS, ls = s - chunk * STEPS;
    return wbase + (size_t)(chunk * 16 * K8 + ls) * stepstride;
  };

  const bool xcd_order = (a.xcd_order & 1) && (gridDim.x & 7) == 0;
  const int wg_step = xcd_order ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int xchunk = (a.total + 7) >> 3;
  const int wg_first = xcd_order ? (int)(blockIdx.x & 7) * xchunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int wg_end = xcd_order ? min(a.total, ((int)(blockIdx.x & 7) + 1) * xchunk) : a.total;
  if (wg_first < wg_end) {
    load_chunk(0, wg_first, 0);
    load_chunk(1, wg_first, 1);
  }
  const int wg_stamp = wg_first + 2 * wg_step < wg_end ? wg_first + 2 * wg_step : wg_first;
  const int n = a.nchunk;             // even and >= 4 (checked when the plan is built)
  for (int wg = wg_first; wg < wg_end; wg += wg_step) {
  const int bl = wg / tiles;
  const int b = a.frame0 + bl;
  const int t = wg - bl * tiles;
  const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
  if (wg == wg_stamp && wg != wg_first) { FPC_STAMP(0) }
  int tid_t = tid;
  asm volatile("" : "+v"(tid_t));

  f32x16 acc[PX][NBW];
#pragma unroll
  for (int p = 0; p < PX; ++p)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][nb][r] = 0.f;

  // ---------------------------------------------------------------- phase 1: Winograd 3x3, pipelined
  float4 bw[RING][NBW];
#pragma unroll
  for (int s_ = 0; s_ < RING - 1; ++s_) {
    const float4* p0 = wptr(s_);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) bw[s_][nb] = ldw(p0 + nb * 64);
  }
  FPC_LDS_BARRIER();                   // the previous tile's epilogue is done with the LDS
  store_chunk(0, 0);
  store_chunk(1, 1);
  load_chunk(0, wg, 2);                // needed at the start of period 0; the transform of chunk 0 comes first
  FPC_LDS_BARRIER();
  if (wg == wg_stamp) { FPC_STAMP(6) }

  // this group's half of the input transform of chunk k: rows i = 2 grp, 2 grp + 1 of V = B^T d B for every (tile,
  // channel pair) -- one item per thread of the group; lanes run over channel pairs first
  auto transform = [&](int k) {
    const float2* halo2 = reinterpret_cast<const float2*>(halo4 + (k % 3) * HALO4);
    float2* v2 = reinterpret_cast<float2*>(vv4 + (k & 1) * V4);
    const int u = tid_t & 255;
    const int c2 = u & 7, wt = u >> 3;
    const int ty2 = wt >> 3, tx2 = wt & 7;
    const int base = ((2 * ty2 + grp) * HW + 2 * tx2) * (ROW4 * 2) + c2;   // patch rows grp .. grp + 2
    float2 x[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) x[i][j] = halo2[base + (i * HW + j) * (ROW4 * 2)];
    float2 r[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (grp == 0) {   // rows 0, 1:  d0 - d2,  d1 + d2
        r[0][j] = make_float2(x[0][j].x - x[2][j].x, x[0][j].y - x[2][j].y);
        r[1][j] = make_float2(x[1][j].x + x[2][j].x, x[1][j].y + x[2][j].y);
      } else {          // rows 2, 3:  d2 - d1,  d1 - d3   (x[0] = d1, x[1] = d2, x[2] = d3)
        r[0][j] = make_float2(x[1][j].x - x[0][j].x, x[1][j].y - x[0][j].y);
        r[1][j] = make_float2(x[0][j].x - x[2][j].x, x[0][j].y - x[2][j].y);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float2 q0 = make_float2(r[i][0].x - r[i][2].x, r[i][0].y - r[i][2].y);
      const float2 q1 = make_float2(r[i][1].x + r[i][2].x, r[i][1].y + r[i][2].y);
      const float2 q2 = make_float2(r[i][2].x - r[i][1].x, r[i][2].y - r[i][1].y);
      const float2 q3 = make_float2(r[i][1].x - r[i][3].x, r[i][1].y - r[i][3].y);
      const int xi = (2 * grp + i) * 4;
      v2[((xi + 0) * 32 + wt) * (ROW4 * 2) + c2] = q0;
      v2[((xi + 1) * 32 + wt) * (ROW4 * 2) + c2] = q1;
      v2[((xi + 2) * 32 + wt) * (ROW4 * 2) + c2] = q2;
      v2[((xi + 3) * 32 + wt) * (ROW4 * 2) + c2] = q3;
    }
  };
  int gs = 0;
  // the 16 GEMMs of chunk c: this wave's PX positions x NBW channel blocks; A one step ahead, B three
  auto gemm = [&](int c) {
    const float4* vb = vv4 + (c & 1) * V4;
    const int ab0 = (gx * PX * 32 + l31) * ROW4 + half;
    float4 av = vb[ab0];
#pragma unroll
    for (int ls = 0; ls < STEPS; ++ls) {
      const int p = ls / K8, k8 = ls - p * K8;
      const int lsn = ls + 1 < STEPS ? ls + 1 : ls;
      const int pn_ = lsn / K8, k8n = lsn - pn_ * K8;
      const float4* pn = wptr(gs + RING - 1);
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) bw[(ls + RING - 1) % RING][nb] = ldw(pn + nb * 64);
      ++gs;
#ifdef FPC_W2_NOAPF
      __builtin_amdgcn_sched_barrier(0);
      av = vb[ab0 + p * 32 * ROW4 + k8 * 2];
      const float4 an = av;
#else
      const float4 an = vb[ab0 + pn_ * 32 * ROW4 + k8n * 2];
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
          const float4& bq = bw[ls % RING][nb];
          const float af = j == 0 ? av.x : j == 1 ? av.y : j == 2 ? av.z : av.w;
          const float bf = j == 0 ? bq.x : j == 1 ? bq.y : j == 2 ? bq.z : bq.w;
          acc[p][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[p][nb], 0, 0, 0);
        }
      av = an;
    }
  };
  for (int c = -1; c < n; ++c) {
#ifdef FPC_DIAG
    if (!((a.xcd_order >> 1) & 8))
#endif
    if (c >= 0 && c + 2 < n) store_chunk((c + 2) & 1, (c + 2) % 3);   // chunk c + 2 -> its halo buffer
    // Chunk c + 4 (or the next tile's chunk 0 / 1) is requested AFTER this wave's GEMM: the loads complete in order,
    // so a halo request in front of the period's B fragments would make the GEMM wait for HBM; behind the GEMM it has
    // the transform, the barrier and three GEMM steps before anything issued after it is consumed.
    auto request = [&]() {
#ifdef FPC_DIAG
      if ((a.xcd_order >> 1) & 8) return;
#endif
      const int k = c + 4;
      if (k < n) load_chunk(k & 1, wg, k);
      else if (k - n < 2 && wg + wg_step < wg_end) load_chunk((k - n) & 1, wg + wg_step, k - n);
    };
#ifdef FPC_DIAG
    const int dbg = a.xcd_order >> 1;    // diagnostic build only: 1 skip transform, 2 skip GEMM, 4 no phase shift
    const bool first = (dbg & 4) ? true : grp == 0;
    if (first && c + 1 < n && !(dbg & 1)) transform(c + 1);
    if (c >= 0 && !(dbg & 2)) gemm(c);
    request();
    if (!first && c + 1 < n && !(dbg & 1)) transform(c + 1);
#else
    if (grp == 0 && c + 1 < n) transform(c + 1);   // (one copy of the GEMM code: the accumulators stay put)
    if (c >= 0) gemm(c);
    request();
    if (grp != 0 && c + 1 < n) transform(c + 1);
#endif
    FPC_LDS_BARRIER();
    if (c == -1 && wg == wg_stamp) { FPC_STAMP(1) }
    if (c == 0 && wg == wg_stamp) { FPC_STAMP(7) }
  }
  if (wg == wg_stamp) { FPC_STAMP(2) }

#include "wblock_tail.inc"
  }  // persistent tile loop
}

}  // namespace fpc
