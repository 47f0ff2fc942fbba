// wblock36p_mfma.h -- the 64-channel Winograd F(4x4,3x3) ResNetBlock kernel at TWO waves per SIMD (round 5).
//
// wblock36_kernel<1, ...> (wblock36_mfma.h) holds 144 accumulator registers + a 72-register fragment ring per wave: one wave
// per SIMD, and nothing runs on a SIMD while that wave sits in a gap between two MFMA blocks, at a barrier, in an LDS
// round trip or behind an LDS-DMA request.  Measured (round 5, branch-free in-kernel stamps): a 16-channel chunk of the
// 64-channel instance takes 7.4-9 k cycles for 4.6 k cycles of MFMAs, the whole tile 51 k for 26.6 k.  The 128-channel
// instance pays the same fixed costs per chunk for twice the MFMAs (0.76 in its chunk loop).
//
// Here a workgroup is EIGHT waves, two per SIMD, and the two waves of a SIMD share a channel group (16 output channels):
//     waves 0..3 ("T"): Winograd positions  0..17 + the input transform of the next chunk (halo -> V)
//     waves 4..7 ("G"): positions 18..35 + the halo requests (LDS-DMA) and the per-tile address arithmetic
// 72 accumulator registers each; the fp32 MFMA and the VALU exclude each other on a SIMD whichever wave issues them
// (wblock36_mfma.h), so what the second wave buys is every LATENCY: while one wave waits, the other's MFMAs issue.
//   * Fragments: the same packed streams ([chunk][position][lane]); a wave reads its 18 positions, two steps ahead (ring
//     of three).  Accumulators in ARCHITECTURAL registers ("+v"): a kernel that names an AGPR anywhere gets its 256
//     registers split 128 / 128 by the compiler (137 spills); one that names none may use all 256 as VGPRs.
//   * Output transform Y = A^T M A: rows 0..2 of M are with the T wave, rows 3..5 with the G wave; the transform is
//     linear, so each wave transforms its rows and the two partial Y are added: each wave hands the partial sums of the
//     two Winograd tiles its partner finishes over through a compact buffer (eight 16-byte LDS writes per lane, aliasing
//     the not yet written h tile), reads its partner's, adds its own, the bias, ReLU, and stores h.
//   * The 64-channel h of the WHOLE tile (256 pixels x 72 floats = 72 KB) fits beside the next tile's pipelined input, so
//     the tail is one pass (no halves): 1x1 over h (+ projection over x, staged 64 channels at a time), each wave eight
//     pixel blocks of its channel group; epilogue through LDS into 16-byte stores.
// Same WBlockArgs, same packed weights, same results up to the order of two additions as wblock36_kernel<1, TYT, TXT>.
// The detector's 65-channel blocks (the DUST instance) stay on the one-wave kernel.
#pragma once
#include "wblock36_mfma.h"

namespace fpc {

template <int TYT_, int TXT_, bool DUST_ = false>
struct W36PCfg {
  static constexpr int TYT = TYT_, TXT = TXT_;
  static constexpr bool DUST = DUST_;
  static_assert(TYT * TXT == 16 && (TYT == 4 || TYT == 2), "16 Winograd tiles per workgroup tile: 4 x 4 or 2 x 8");
  static constexpr int NT = 512, NTH = 256, NPOS = 36, KC = 16, N = 64;
  static constexpr int TH = 4 * TYT, TW = 4 * TXT, HH = TH + 2, HW = TW + 2, NHALO = HH * HW;
  static constexpr int HROW = 20;
  static constexpr int HIT = (NHALO * 5 + NTH - 1) / NTH;           // LDS-DMA requests per G thread and chunk (7)
  static constexpr int HALO_FLOATS = HIT * NTH * 4;
  static constexpr int V_FLOATS = NPOS * 16 * 16;
  static constexpr int OFF_V0 = 0, OFF_H1 = V_FLOATS, OFF_H0 = OFF_H1 + HALO_FLOATS, OFF_V1 = OFF_H0 + HALO_FLOATS;
  static constexpr int RH = N + 8;                                  // h / x staging / output row (floats)
  static constexpr int PX = TH * TW;                                // 256 pixels: the whole tile
  static constexpr int T_FLOATS = PX * RH;
  static constexpr int BASE_FLOATS = (OFF_V1 + V_FLOATS) > (OFF_H0 + T_FLOATS) ? (OFF_V1 + V_FLOATS) : (OFF_H0 + T_FLOATS);
  // DUST (the detector's 65th channel, as W36Cfg<1, .., true>): the dust block's first 448 floats as they are, UIN (2304), V of
  // input channel 64 [36][16], M of output channel 64 [36][16], the halo of input channel 64 (one 16-byte slot per pixel)
  static constexpr int HIT64 = (NHALO + NTH - 1) / NTH;
  static constexpr int D_HEAD = BASE_FLOATS, D_COL = D_HEAD + 64, D_UIN = D_HEAD + 448, D_V64 = D_UIN + 2304, D_M64 = D_V64 + 576, D_H64 = D_M64 + 576;
  static constexpr int LDS_FLOATS = DUST ? D_H64 + HIT64 * NTH * 4 : BASE_FLOATS;
  static constexpr int LDS_BYTES = LDS_FLOATS * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static constexpr int STEPS = 9;                                   // per wave: 18 positions, two per step
  static constexpr int WPAD = W36Cfg<1, TYT, TXT>::WPAD, WPAD2 = W36Cfg<1, TYT, TXT>::WPAD2;
  static_assert(WPAD >= 36, "a wave prefetches up to one chunk past its stream's end");
};

// RING: depth of the fragment ring in steps (3 or 9: slot indices must line up across chunks); a wave's fragments are
// requested RING - 1 steps ahead.  Measured: 9 (a whole chunk ahead, 72 registers, 16 of them spilled) is 5-7 % SLOWER
// than 3 (24 registers; two steps = 1 k cycles of cover with the partner's MFMAs in between)
template <int TYT, int TXT, int RING, bool DUST>
__device__ __forceinline__ void wblock36p_body(const WBlockArgs& a) {
  using C = W36PCfg<TYT, TXT, DUST>;
  constexpr int TH = C::TH, TW = C::TW, HW = C::HW, HH = C::HH, HROW = C::HROW, HIT = C::HIT, NTH = C::NTH;
  constexpr int N = C::N, RH = C::RH, STEPS = C::STEPS, PX = C::PX;
  static_assert(STEPS % RING == 0 && RING >= 2, "ring slots must line up across chunks");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x4* const lds4 = reinterpret_cast<f32x4*>(lds);
  constexpr int TL4 = C::OFF_H0 / 4;        // h, x staging, output tile (float4 units)

  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave & 3;                 // channel group (16 output channels); the SIMD, as waves are dealt round-robin
  const int hs = wave >> 2;                 // 0: T wave (positions 0..17, transform), 1: G wave (18..35, halo requests)
  const int tiles = a.tiles_x * a.tiles_y;
  const int nchunk = a.nchunk;              // Cin / 16, even, >= 4
  const float4* const w1h = a.w1 + (size_t)blockIdx.y * (a.ysplit_floats / 4);
  const float* const b1h = a.b1 + (size_t)blockIdx.y * a.ysplit_floats;

  // tile walk (persistent, XCD-aware): as wblock36_kernel
  const bool xcd_order = a.xcd_order && (gridDim.x & 7) == 0;
  const int wg_step = xcd_order ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int xchunk = (a.total + 7) >> 3;
  const int wg_first = xcd_order ? (int)(blockIdx.x & 7) * xchunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int wg_end = xcd_order ? min(a.total, ((int)(blockIdx.x & 7) + 1) * xchunk) : a.total;
  if (wg_first >= wg_end) return;

  // ---------------------------------------------------------------- input side
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  struct TilePos { int b, ty, tx, live; };
  auto tile_pos = [&](int wg) {
    TilePos q;
    q.live = wg < wg_end;
    const int wgc = q.live ? wg : wg_first;
    const int bl = wgc / tiles;
    const int t = wgc - bl * tiles;
    q.b = a.frame0 + bl;
    q.ty = t / a.tiles_x;
    q.tx = t - q.ty * a.tiles_x;
    return q;
  };
  auto halo_base = [&](const TilePos& q) {
    return __builtin_amdgcn_readfirstlane(((q.b * a.H + q.ty * TH) * a.W + q.tx * TW) * a.csx * 4);
  };
  // G waves.  Slot q = t + 256 i of G thread t (t = tid & 255): halo pixel q / 5 = (hy, hx), channel quad q % 5 (4 = the skew
  // slot).  What depends on the thread alone is computed ONCE per launch -- hyx[i] = hy << 16 | c4 << 8 | hx (hy = 0x7fff for a
  // skew / padding slot: it fails every bounds test) -- and a tile costs a dozen VALU instructions per slot (the one-wave
  // kernel recomputes the divisions per tile: ~60 per slot, 1.0-1.3 k cycles of a 64-channel tile's 51 k, measured).
  // (the DUST instance is short of registers -- its per-launch values were spilled and reloaded one scratch round trip at a
  // time inside halo_tile --: it derives them from the thread id per tile instead, six more VALU per slot)
  constexpr bool KEEP_HYX = !DUST;
  auto slot_hyx = [&](int tl, int i) {
    const int e = tl + i * NTH;
    const int pix = (e * 13108) >> 16, c4 = e - pix * 5;                    // e / 5 (e < 1792)
    const int hy = HW == 18 ? (pix * 3641) >> 16 : (pix * 1928) >> 16;     // pix / 18 (pix < 469), pix / 34 (pix < 441)
    const int hx = pix - hy * HW;
    const bool slot_ok = (hy < HH) & (c4 < 4);
    return slot_ok ? (hy << 16) | (c4 << 8) | hx : 0x7fff0000;
  };
  int okoff[HIT];
  [[maybe_unused]] int hyx[KEEP_HYX ? HIT : 1];
  if (hs == 1) {
    const int tl = tid & (NTH - 1);
#pragma unroll
    for (int i = 0; i < HIT; ++i) {
      if constexpr (KEEP_HYX) hyx[i] = slot_hyx(tl, i);
      okoff[i] = W36_MARKER;
    }
  }
  // DUST: the same for input channel 64 (detector.layer.1), one 16-byte slot per halo pixel: channels 64..67
  [[maybe_unused]] int okoff64[C::HIT64];
  auto slot_hyx64 = [&](int tl, int i) {
    const int pix = tl + i * NTH;
    const int hy = HW == 18 ? (pix * 3641) >> 16 : (pix * 1928) >> 16;
    const int hx = pix - hy * HW;
    return (hy < HH && a.dust_in) ? (hy << 16) | hx : 0x7fff0000;
  };
  if constexpr (DUST) {
    if (hs == 1) {
#pragma unroll
      for (int i = 0; i < C::HIT64; ++i) okoff64[i] = W36_MARKER;
    }
  }
  auto halo_tile = [&](const TilePos& q) {               // okoff[] for that tile (G waves)
    const int iy0 = q.ty * TH - 1, ix0 = q.tx * TW - 1;
    const int hlim = q.live ? a.H : 0;
    int tlq = tid & (NTH - 1);
    asm volatile("" : "+v"(tlq));       // (opaque per tile: computed from loop invariants, everything below is hoisted out of the tile loop -- and spilled)
#pragma unroll
    for (int i = 0; i < HIT; ++i) {
      int hv = KEEP_HYX ? hyx[KEEP_HYX ? i : 0] : slot_hyx(tlq, i);
      asm volatile("" : "+v"(hv));
      const int hy = hv >> 16, hx = hv & 0xff, c4 = (hv >> 8) & 0xff;
      const bool ok = ((unsigned)(iy0 + hy) < (unsigned)hlim) & ((unsigned)(ix0 + hx) < (unsigned)a.W);
      okoff[i] = ok ? (((hy - 1) * a.W + (hx - 1)) * a.csx + c4 * 4) * 4 : W36_MARKER;
    }
    if constexpr (DUST) {
#pragma unroll
      for (int i = 0; i < C::HIT64; ++i) {
        int hv = slot_hyx64(tlq, i);
        asm volatile("" : "+v"(hv));
        const int hy = hv >> 16, hx = hv & 0xffff;
        const bool ok = ((unsigned)(iy0 + hy) < (unsigned)hlim) & ((unsigned)(ix0 + hx) < (unsigned)a.W);
        okoff64[i] = ok ? (((hy - 1) * a.W + (hx - 1)) * a.csx + 64) * 4 : W36_MARKER;
      }
    }
  };
  auto load_halo64 = [&](int base) {                     // DUST, G waves: channels 64..67 of the tile's halo pixels -> D_H64 [pixel][4]
    if constexpr (DUST) {
#pragma unroll
      for (int i = 0; i < C::HIT64; ++i) {
        int voff;
        asm("v_add_i32 %0, %1, %2 clamp" : "=v"(voff) : "v"(okoff64[i]), "s"(base));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(lds + C::D_H64 + (i * NTH + grp * 64) * 4), 16, voff, 0, 0, 0);
      }
    }
  };
  auto load_halo = [&](int base, int hoff_f) {            // G waves: base = halo_base of the tile + 64 bytes per chunk; hoff_f: the halo buffer (floats)
#pragma unroll
    for (int i = 0; i < HIT; ++i) {
      int voff;
      asm("v_add_i32 %0, %1, %2 clamp" : "=v"(voff) : "v"(okoff[i]), "s"(base));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(lds + hoff_f + (i * NTH + grp * 64) * 4), 16, voff, 0, 0, 0);
    }
  };
  // T waves: the transform item of thread t = tid (< 256): Winograd tile wt (0..15), channel ch (0..15) of the chunk
  const int tq = tid & (NTH - 1);
  const int wt_t = tq >> 4, ch_t = tq & 15;
  const int m_t = ((wt_t & 7) >> 1) * 4 + (wt_t >> 3) * 2 + (wt_t & 1);
  const int trd = ((4 * (wt_t / TXT)) * HW + 4 * (wt_t % TXT)) * HROW + ch_t;
  const int twr = m_t * 16 + ((((ch_t >> 2) ^ (2 * ((m_t >> 3) & 1))) << 2) | (ch_t & 3));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 tp[6][3];
  auto t_read = [&](int rd, int i, int j) { tp[i][j >> 1][j & 1] = lds[rd + (i * HW + j) * HROW]; };
  auto bt6 = [&](f32x2& x0, f32x2& x1, f32x2& x2, f32x2& x3, f32x2& x4, f32x2& x5) {
    const f32x2 k4 = {4.f, 4.f}, km4 = {-4.f, -4.f}, k5 = {-5.f, -5.f}, k2 = {2.f, 2.f}, km2 = {-2.f, -2.f};
    const f32x2 p = __builtin_elementwise_fma(km4, x2, x4), q = __builtin_elementwise_fma(km4, x1, x3);
    const f32x2 r = x4 - x2, s = x3 - x1;
    const f32x2 y0 = __builtin_elementwise_fma(k4, x0, __builtin_elementwise_fma(k5, x2, x4));
    const f32x2 y5 = __builtin_elementwise_fma(k4, x1, __builtin_elementwise_fma(k5, x3, x5));
    x0 = y0;
    x1 = p + q;
    x2 = p - q;
    x3 = __builtin_elementwise_fma(k2, s, r);
    x4 = __builtin_elementwise_fma(km2, s, r);
    x5 = y5;
  };
  auto t_burst = [&]() {
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) bt6(tp[0][jj], tp[1][jj], tp[2][jj], tp[3][jj], tp[4][jj], tp[5][jj]);
#pragma unroll
    for (int ii = 0; ii < 3; ++ii)
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        const f32x2 ra = tp[2 * ii][jj], rb = tp[2 * ii + 1][jj];
        tp[2 * ii][jj] = __builtin_shufflevector(ra, rb, 0, 2);
        tp[2 * ii + 1][jj] = __builtin_shufflevector(ra, rb, 1, 3);
      }
#pragma unroll
    for (int ii = 0; ii < 3; ++ii) bt6(tp[2 * ii][0], tp[2 * ii + 1][0], tp[2 * ii][1], tp[2 * ii + 1][1], tp[2 * ii][2], tp[2 * ii + 1][2]);
  };
  auto t_write = [&](int wr, int i, int j) { lds[wr + (i * 6 + j) * 256] = tp[2 * (i >> 1) + (j & 1)][j >> 1][i & 1]; };

  // ---------------------------------------------------------------- operands of the GEMMs
  const int aoff4 = (lane & 15) * 4 + ((lane >> 4) ^ (2 * ((lane & 15) >> 3))) + hs * 18 * 64;      // float4 units; this wave's first position
  const unsigned gstride = ((unsigned)nchunk * 36u + (unsigned)C::WPAD) * 1024u;                      // bytes per channel group
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(w1h), 0, (int)(4u * gstride), 0x00020000);
  const unsigned wlane = (unsigned)grp * gstride + (unsigned)hs * (18u * 1024u) + lane16;
  struct BF { f32x4 v[2]; };
  auto ldb = [&](int cc, int s) {      // step s (0..8, compile time) of chunk cc: positions 18 hs + 2 s, + 1
    BF r;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      r.v[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)wlane, cc * (36 * 1024) + (2 * s + q) * 1024, 0));
    return r;
  };

  // ---------------------------------------------------------------- DUST: the 65th output (and input) channel beside the 64 on the MFMAs
  // As in wblock36_dust_kernel: output channel 64 of the 3x3 is a dot product per (position, Winograd tile) and chunk in
  // the Winograd domain, on the VALU.  Here the G waves do it (the T waves have the transform): G wave g takes positions
  // g, g + 4, ..., g + 32; its lane l reads float4 l of a position's V block and keeps one partial sum per position.
  [[maybe_unused]] int d_va = 0, d_ua = 0;
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t drsrc =
      DUST ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dust + W36Dust::UOUT), 0, nchunk * 36 * 16 * 4, 0x00020000) : xrsrc;
  [[maybe_unused]] f32x4 du[6];      // two batches of three filter quads (a third of a chunk's nine at a time)
  [[maybe_unused]] float dacc[9];
  if constexpr (DUST) {
    d_va = grp * 64 + lane;
    d_ua = grp * 64 + 16 * ((lane & 3) ^ (2 * (lane >> 5)));
    for (int i = tid; i < 448 / 4; i += C::NT) lds4[C::D_HEAD / 4 + i] = reinterpret_cast<const f32x4*>(a.dust)[i];
    for (int i = tid; i < 2304 / 4; i += C::NT) lds4[C::D_UIN / 4 + i] = reinterpret_cast<const f32x4*>(a.dust + W36Dust::UIN)[i];
  }

  // ---------------------------------------------------------------- pipeline fill for the first tile
  TilePos pos_cur = tile_pos(wg_first);
  int base_cur = halo_base(pos_cur);
  if (hs == 1) {
    halo_tile(pos_cur);
    load_halo64(base_cur);
    load_halo(base_cur, C::OFF_H0);
    load_halo(base_cur + 64, C::OFF_H1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HIT) : "memory");      // chunk 0 has landed (this wave's part of it)
  }
  FPC_LDS_BARRIER();
  if (hs == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) t_read(trd + C::OFF_H0, i, j);
    t_burst();
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) t_write(twr + C::OFF_V0, i, j);
  }
  FPC_LDS_BARRIER();

  // the tail's per-thread constants: output float4 e = tid + 512 i lies at pixel m0 + 32 i, channel quad c4t
  constexpr int C4 = N / 4, EIT = PX * C4 / C::NT, PPI = C::NT / C4;   // 16 float4 per pixel; 8 per thread; 32 pixels per i
  constexpr int RPI = PPI / TW > 0 ? PPI / TW : 1;                    // pixel rows per i (TW = 16: 2, TW = 32: 1)
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + blockIdx.y * N, 0, W36_MARKER, 0x00020000);
  const int g2s = __builtin_amdgcn_readfirstlane(((a.k8_h + a.k8_x) / 2 + C::WPAD2) * 64);     // float4 per channel group of the 1x1 streams
  const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.w2), 0, a.conv_only ? 0 : 4 * g2s * 16, 0x00020000);
  const unsigned w2lane = (unsigned)(grp * g2s) * 16u + lane16;

  const int wg_stamp = wg_first + 2 * wg_step < wg_end ? wg_first + 2 * wg_step : wg_first;
  for (int wg = wg_first; wg < wg_end; wg += wg_step) {
    const int b = pos_cur.b, ty = pos_cur.ty, tx = pos_cur.tx;
    const TilePos pos_next = tile_pos(wg + wg_step);
    const int base_next = halo_base(pos_next);
    if (wg == wg_stamp) { FPC_STAMP(0) FPC_RSTAMP(6) }

    f32x4 acc[18];
    {
      float zf = 0.f;
      asm volatile("" : "+v"(zf));
#pragma unroll
      for (int p = 0; p < 18; ++p) acc[p] = f32x4{zf, zf, zf, zf};
    }
    if constexpr (DUST) {
      float zf = 0.f;
      asm volatile("" : "+v"(zf));
#pragma unroll
      for (int i = 0; i < 9; ++i) dacc[i] = zf;
    }
    // (ring filled per tile; kept alive ACROSS tiles -- the last chunk requesting the next tile's first fragments -- it
    // measured the same time at 16 more registers: the fill's L2 round trip hides under the partner wave)
    BF bq[RING];
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) bq[i] = ldb(0, i);

    // ---------------------------------------------------------------- phase 1: the wave's 18 GEMMs per chunk
    // One basic block per role and chunk (a block join costs an `s_waitcnt vmcnt(0)`); the two roles are two loops, not
    // a branch inside one.
    auto chunk_body = [&](auto ROLE, auto PAR, const int c) {
      constexpr int role = decltype(ROLE)::value, par = decltype(PAR)::value;
      constexpr int VB_OFF = par ? C::OFF_V1 : C::OFF_V0, VN_OFF = par ? C::OFF_V0 : C::OFF_V1;
      constexpr int HS_OFF = par ? C::OFF_H1 : C::OFF_H0;   // halo[c & 1]: receives chunk c + 2
      constexpr int HN_OFF = par ? C::OFF_H0 : C::OFF_H1;   // halo[(c + 1) & 1]: chunk c + 1, transformed now
      const int c3 = c + 2;
      const bool nxt = c3 >= nchunk;
      const int l_base = (nxt ? base_next : base_cur) + (nxt ? c3 - nchunk : c3) * 64;
      int ao = aoff4 + VB_OFF / 4, trd_c = trd + HN_OFF, twr_c = twr + VN_OFF;
      asm volatile("" : "+v"(ao), "+v"(trd_c), "+v"(twr_c));
      [[maybe_unused]] int dva = d_va + VB_OFF / 4;      // DUST: this lane's float4 of position g, this chunk's buffer
      [[maybe_unused]] f32x4 dv[3];
      if constexpr (DUST) asm volatile("" : "+v"(dva));
      f32x4 ac[2][2];
#pragma unroll
      for (int q = 0; q < 2; ++q) ac[0][q] = lds4[ao + q * 64];
      fpc_static_for<STEPS>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        // ---- the step's gap
        {
          constexpr int t = s + RING - 1;      // the step requested now: t of this chunk, or t - STEPS of the next one
          if constexpr (t < STEPS) bq[t % RING] = ldb(c, t);
          else bq[t % RING] = ldb(c + 1, t - STEPS);      // (behind the last chunk: the stream's zero padding)
        }
        if (s + 1 < STEPS) {
#pragma unroll
          for (int q = 0; q < 2; ++q) ac[(s + 1) & 1][q] = lds4[ao + ((s + 1) * 2 + q) * 64];
        }
        constexpr int SPS = 4;   // 36 input-side slots per chunk over nine steps
#pragma unroll
        for (int u = 0; u < SPS; ++u) {
          const int slot = s * SPS + u;
          if constexpr (role == 1) {
            if (slot == 1) {
              if (c3 == nchunk) halo_tile(pos_next);      // (uniform; no memory operation inside)
              load_halo(l_base, HS_OFF);
            }
            if constexpr (DUST) {
              // output channel 64: the chunk's nine filter quads requested early, the V quads read three positions at a time
              // and multiplied two slots later
              auto dread = [&](int i0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) dv[k] = lds4[dva + (i0 + k) * 256];
              };
              auto dfma = [&](int i0) {
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                  for (int e = 0; e < 4; ++e) dacc[i0 + k] = __builtin_fmaf(dv[k][e], du[(i0 / 3 % 2) * 3 + k][e], dacc[i0 + k]);
              };
              auto dload = [&](int i0) {      // three filter quads, ten slots (2.5 steps) ahead of their use
#pragma unroll
                for (int k = 0; k < 3; ++k)
                  du[(i0 / 3 % 2) * 3 + k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drsrc, d_ua + (i0 + k) * 256, c * (36 * 16 * 4), 0));
              };
              if (slot == 3) dload(0);
              else if (slot == 11) dread(0);
              else if (slot == 13) { dfma(0); dload(3); }
              else if (slot == 21) dread(3);
              else if (slot == 23) { dfma(3); dload(6); }
              else if (slot == 31) dread(6);
              else if (slot == 33) dfma(6);
            }
          } else {
            if (slot >= 2 && slot < 14) {
#pragma unroll
              for (int k = 0; k < 3; ++k) { const int e = (slot - 2) * 3 + k; t_read(trd_c, e / 6, e % 6); }
            } else if (slot == 14) {
              t_burst();
            } else if (slot >= 15 && slot < 33) {
#pragma unroll
              for (int k = 0; k < 2; ++k) { const int e = (slot - 15) * 2 + k; t_write(twr_c, e / 6, e % 6); }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        fpc_mfma_step<false, false>(acc[2 * s], acc[2 * s + 1], ac[s & 1][0], ac[s & 1][1], bq[s % RING].v[0], bq[s % RING].v[1]);
        __builtin_amdgcn_sched_barrier(0);
      });
      // (G: this wave's halo requests of the chunk have landed in LDS; T: stated for symmetry -- everything but the ring's
      // newest fragments is complete)
      // sixteen fragment loads were issued behind the halo request, whatever the ring's depth: at most that many outstanding
      // means the request has landed (vmcnt retires in order)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (STEPS - 1)) : "memory");
      FPC_LDS_BARRIER();
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): once per tile
    if (hs == 0) {
      int c = 0;
      do {
        chunk_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, c);
        chunk_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, c + 1);
        c += 2;
      } while (c < nchunk);
    } else {
      int c = 0;
      do {
        chunk_body(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, c);
        chunk_body(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, c + 1);
        c += 2;
      } while (c < nchunk);
    }
    if (wg == wg_stamp) { FPC_STAMP(1) }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");     // (asm MFMA results read by the VALU below)

    int tid_t = tid;
    asm volatile("" : "+v"(tid_t));
    const int lane_t = tid_t & 63, n16 = lane_t & 15, kq = lane_t >> 4;
    const bool proj = a.k8_x > 0;
    if constexpr (DUST) {
      if (a.dust_in) {
        // INPUT channel 64 (detector.layer.1): its halo arrived at the tile's start (D_H64, 16 bytes per pixel); threads
        // 0..15 transform the 6x6 patch of one Winograd tile each into V64 [position][row m] ...
        if (tid_t < 16) {
          const int rd = C::D_H64 + ((4 * (tid_t / TXT)) * HW + 4 * (tid_t % TXT)) * 4;
#pragma unroll
          for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) tp[i][j >> 1][j & 1] = lds[rd + (i * HW + j) * 4];
          t_burst();
          const int mrow = ((tid_t & 7) >> 1) * 4 + (tid_t >> 3) * 2 + (tid_t & 1);
#pragma unroll
          for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) lds[C::D_V64 + (i * 6 + j) * 16 + mrow] = tp[2 * (i >> 1) + (j & 1)][j >> 1][i & 1];
        }
        FPC_LDS_BARRIER();
        // ... and every wave adds its channel group's share at its 18 positions: ONE MFMA per position, K = 4 of which
        // k = 0 is real (the A operand of lanes 16..63 is zero); all 36 operands first, then the MFMAs back to back
        {
          const int va = C::D_V64 + hs * 18 * 16 + (lane_t & 15), ub = C::D_UIN + grp * 576 + hs * 18 * 16 + (lane_t & 15);
          float av[18], bv[18];
#pragma unroll
          for (int p = 0; p < 18; ++p) {
            av[p] = lds[va + p * 16];
            bv[p] = lds[ub + p * 16];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int p = 0; p < 18; ++p) {
            const float x = lane_t < 16 ? av[p] : 0.f;
            acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, bv[p], acc[p], 0, 0, 0);
          }
        }
      }
      if (hs == 1) {
        // (G waves) the four lanes of a V row add their partial sums up; lane 0 of the row adds input channel 64's share
        // (in 64 -> out 64) and writes M64[position][m]
        const int m_d = lane_t >> 2;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          float v = dacc[i];
          v += __shfl_xor(v, 1);
          v += __shfl_xor(v, 2);
          const int pos = grp + 4 * i;
          if (a.dust_in) v = __builtin_fmaf(lds[C::D_V64 + pos * 16 + m_d], lds[C::D_HEAD + W36Dust::UIN64 + pos], v);
          if ((lane_t & 3) == 0) lds[C::D_M64 + pos * 16 + m_d] = v;
        }
      }
      FPC_LDS_BARRIER();
      if (hs == 1) load_halo64(base_next);      // D_H64 has been read: the next tile's input channel 64 (okoff64 is the next tile's by now)
    }

    // ---------------------------------------------------------------- output transform: each wave its three rows of M, joined in place in h
    // h position of (row r of this lane's accumulators -> Winograd tile T = 8 (r >> 1) + 2 kq + (r & 1), pixel (0, 0), this lane's channel)
    int hw_base[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int T = 8 * (r >> 1) + 2 * kq + (r & 1);
      hw_base[r] = C::OFF_H0 + ((4 * (T / TXT)) * TW + 4 * (T % TXT)) * RH + 16 * grp + n16;
    }
    // One straight-line body per ROLE (the role as a compile-time constant): with `if (hs == 0)` inside the transform, every
    // column was a block join -- and the first join waits for the fragment ring's last (unused) prefetches, an L2 round
    // trip -- 2.1 k cycles for one partial transform, measured.
    auto out_transform = [&](auto ROLE) __attribute__((always_inline)) {
      constexpr int role = decltype(ROLE)::value;      // == hs
      // the partial transform of rows r = 2 rp, 2 rp + 1 (as register pairs): yv[i][jj] for the 4 x 4 pixels of the two tiles
      auto partial = [&](int rp, f32x2 (&yv)[4][4]) {
        f32x2 tt[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          auto pr = [&](int pos) { return f32x2{acc[pos][2 * rp], acc[pos][2 * rp + 1]}; };
          const f32x2 ma = pr(0 * 6 + j), mb = pr(1 * 6 + j), mc = pr(2 * 6 + j);
          if constexpr (role == 0) {      // rows 0, 1, 2 of M:  A^T = [1 1 1 ..; 0 1 -1 ..; 0 1 1 ..; 0 1 -1 ..]
            const f32x2 s12 = mb + mc, d12 = mb - mc;
            tt[0][j] = ma + s12;
            tt[1][j] = d12;
            tt[2][j] = s12;
            tt[3][j] = d12;
          } else {                        // rows 3, 4, 5:  [.. 1 1 0; .. 2 -2 0; .. 4 4 0; .. 8 -8 1]
            const f32x2 s34 = ma + mb, d34 = ma - mb;
            tt[0][j] = s34;
            tt[1][j] = f32x2{2.f, 2.f} * d34;
            tt[2][j] = f32x2{4.f, 4.f} * s34;
            tt[3][j] = __builtin_elementwise_fma(f32x2{8.f, 8.f}, d34, mc);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x2 m0 = tt[i][0], m1 = tt[i][1], m2 = tt[i][2], m3 = tt[i][3], m4 = tt[i][4], m5 = tt[i][5];
          const f32x2 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          yv[i][0] = m0 + s12 + s34;
          yv[i][1] = __builtin_elementwise_fma(f32x2{2.f, 2.f}, d34, d12);
          yv[i][2] = __builtin_elementwise_fma(f32x2{4.f, 4.f}, s34, s12);
          yv[i][3] = __builtin_elementwise_fma(f32x2{8.f, 8.f}, d34, d12) + m5;
        }
      };
      {
        // (1) the partner's two tiles (rows 2 (1 - role), + 1): 32 partial sums per lane, handed over through a compact
        // buffer -- float4 k of thread t at [k][t], eight conflict-free 16-byte writes -- that aliases the (still unwritten)
        // h tile
        f32x2 yv[4][4];
        partial(1 - role, yv);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int i = 0; i < 4; ++i) lds4[TL4 + (e * 4 + i) * C::NT + tid_t] = f32x4{yv[i][0][e], yv[i][1][e], yv[i][2][e], yv[i][3][e]};
      }
      FPC_LDS_BARRIER();
      {
        // (2) the partner's partial sums of this wave's own two tiles (requested first), + its own
        f32x4 other[2][4];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int i = 0; i < 4; ++i) other[e][i] = lds4[TL4 + (e * 4 + i) * C::NT + (tid_t ^ NTH)];
        f32x2 yv[4][4];
        partial(role, yv);
        const float bias1 = b1h[16 * grp + n16];
        float hv[2][16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const float v = (yv[i][jj][e] + other[e][i][jj]) + bias1;
              hv[e][i * 4 + jj] = v > 0.f ? v : 0.f;
            }
        FPC_LDS_BARRIER();      // every wave has read the hand-over buffer: (3) h may overwrite it
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) lds[hw_base[2 * role + e] + (i * TW + jj) * RH] = hv[e][i * 4 + jj];
        if constexpr (DUST && role == 0) {
          if (tid_t < 16) {
            // Y = A^T M A + bias, ReLU for output channel 64 of Winograd tile T(m), m = tid -> h[..][64] (here, with the
            // other channels' h: the hand-over buffer this region held has been read)
            float mm[36];
#pragma unroll
            for (int p = 0; p < 36; ++p) mm[p] = lds[C::D_M64 + p * 16 + tid_t];
            const float bias64 = lds[C::D_HEAD + W36Dust::B1];
            float tt[4][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
              const float m0 = mm[0 * 6 + j], m1 = mm[1 * 6 + j], m2 = mm[2 * 6 + j], m3 = mm[3 * 6 + j], m4 = mm[4 * 6 + j], m5 = mm[5 * 6 + j];
              const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
              tt[0][j] = m0 + s12 + s34;
              tt[1][j] = __builtin_fmaf(2.f, d34, d12);
              tt[2][j] = __builtin_fmaf(4.f, s34, s12);
              tt[3][j] = __builtin_fmaf(8.f, d34, d12) + m5;
            }
            const int T = 8 * ((tid_t >> 1) & 1) + 2 * (tid_t >> 2) + (tid_t & 1);
            const int hb = C::OFF_H0 + ((4 * (T / TXT)) * TW + 4 * (T % TXT)) * RH + 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float m0 = tt[i][0], m1 = tt[i][1], m2 = tt[i][2], m3 = tt[i][3], m4 = tt[i][4], m5 = tt[i][5];
              const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
              float yv64[4];
              yv64[0] = m0 + s12 + s34 + bias64;
              yv64[1] = __builtin_fmaf(2.f, d34, d12) + bias64;
              yv64[2] = __builtin_fmaf(4.f, s34, s12) + bias64;
              yv64[3] = __builtin_fmaf(8.f, d34, d12) + m5 + bias64;
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) lds[hb + (i * TW + jj) * RH] = yv64[jj] > 0.f ? yv64[jj] : 0.f;
            }
          }
        }
      }
      FPC_LDS_BARRIER();
    };
    if (hs == 0) out_transform(std::integral_constant<int, 0>{});
    else out_transform(std::integral_constant<int, 1>{});
    if (wg == wg_stamp) { FPC_STAMP(2) }

    // ---------------------------------------------------------------- the tail: one pass over the tile's 256 pixels
    const int y0 = ty * TH, x0 = tx * TW;
    const int rows_valid = min(TH, a.H - y0);                     // (uniform) pixel rows of the tile inside the frame
    const int xbase = __builtin_amdgcn_readfirstlane(((b * a.H + y0) * a.W + x0) * a.csx * 4);
    const int obase = __builtin_amdgcn_readfirstlane(((b * a.H + y0) * a.W + x0) * a.cso * 4);
    // output float4 i of this thread: LDS [m0 + 32 i][c4t]; pixel row i RPI + rsub (rsub uniform per wave), column pcol
    const int m0 = tid_t / C4, c4t = tid_t - m0 * C4;
    const int rsub = __builtin_amdgcn_readfirstlane(m0 / TW), pcol = m0 % TW;
    const int ocol = (x0 + pcol < a.W) ? (pcol * a.cso + c4t * 4) * 4 : W36_MARKER;
    int erd = TL4 + m0 * (RH / 4) + c4t;
    asm volatile("" : "+v"(erd));
    auto store_out = [&]() {
#pragma unroll
      for (int i = 0; i < EIT; ++i) {
        const int prow = i * RPI + rsub;                              // (uniform per wave)
        if (prow < rows_valid) {
          f32x4 v = lds4[erd + i * PPI * (RH / 4)];
          if (!a.conv_only) {
            v.x = v.x > 0.f ? v.x : 0.f;
            v.y = v.y > 0.f ? v.y : 0.f;
            v.z = v.z > 0.f ? v.z : 0.f;
            v.w = v.w > 0.f ? v.w : 0.f;
          }
          int voff;
          const int so = obase + prow * a.W * a.cso * 4;             // (uniform per wave)
          asm("v_add_i32 %0, %1, %2 clamp" : "=v"(voff) : "v"(ocol), "s"(so));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), orsrc, voff, 0, 0);
        }
      }
    };

    if (a.conv_only) {  // h is the result: [256 px][64] in LDS -> 16-byte stores (ReLU already applied)
      store_out();
      FPC_LDS_BARRIER();   // the region is reused by the next tile's pipeline
    } else {
      // ------------------------------------------------ phase 2: 1x1 over h (+ projection over x): eight pixel blocks of this wave's channel group
      f32x4 acc2[8];
      const int pb = hs;                          // pixel blocks 8 pb .. 8 pb + 7 (pixels 128 pb .. + 127)
      // DUST: output channel 64 of the tile's 256 pixels.  Lane (pxl = l >> 4, q = l & 15) of wave w takes float4 q of the rows
      // of pixels 32 w + 4 it + pxl, it = 0..7: a partial dot product per pixel over h here and over the projection's x in
      // its passes below, added up across the 16 lanes in the epilogue.  And h[64]'s share of outputs 0..63 is the INITIAL
      // VALUE of their accumulators.
      [[maybe_unused]] float dpart[8];
      [[maybe_unused]] float dw2r = 0.f;
      [[maybe_unused]] float dhv[8][4];
      [[maybe_unused]] const int dq = lane_t & 15, dpxl = lane_t >> 4;
      [[maybe_unused]] float dx64 = 0.f;      // identity shortcut: x[px][64] of the pixel this lane stores
      if constexpr (DUST) {
        if (!proj) {
          const int px = 32 * wave + 4 * (dq >> 1) + dpxl;
          dx64 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, (((px / TW) * a.W + px % TW) * a.csx + 64) * 4, xbase, 0));
        }
        const f32x4 wq = lds4[C::D_COL / 4 + dq];
        const float w64 = dq == 0 ? lds[C::D_COL + 64] : 0.f;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int px = 32 * wave + 4 * it + dpxl;
          const f32x4 hq = lds4[TL4 + px * (RH / 4) + dq];
          const float h64 = lds[C::OFF_H0 + px * RH + 64];
          float v = h64 * w64;
#pragma unroll
          for (int e = 0; e < 4; ++e) v = __builtin_fmaf(hq[e], wq[e], v);
          dpart[it] = v;
        }
        dw2r = lds[C::D_HEAD + W36Dust::W2ROW + 16 * grp + n16];
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) dhv[mb][r] = lds[C::OFF_H0 + (128 * pb + 16 * mb + 4 * kq + r) * RH + 64];
      }
      auto ldb2 = [&](int s) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, (int)w2lane, s * 1024, 0));
      };
      constexpr int KH = N / 16;
      auto gemm_over = [&](int ksteps, int s_first) {
        int ar = TL4 + (128 * pb + n16) * (RH / 4) + kq;
        asm volatile("" : "+v"(ar));
        f32x4 cb[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) cb[i] = ldb2(s_first + i);
        f32x4 af[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) af[0][q] = lds4[ar + q * 16 * (RH / 4)];
        for (int g = 0; g < ksteps; g += 4) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            cb[(u + 3) & 3] = ldb2(s_first + g + u + 3);
            fpc_static_for<4>([&](auto B_) __attribute__((always_inline)) {
              constexpr int blk = decltype(B_)::value;
              if (blk + 1 < 4) {
#pragma unroll
                for (int q = 0; q < 2; ++q) af[(blk + 1) & 1][q] = lds4[ar + (g + u) * 4 + ((blk + 1) * 2 + q) * 16 * (RH / 4)];
              } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) af[(blk + 1) & 1][q] = lds4[ar + (g + u + 1) * 4 + q * 16 * (RH / 4)];   // (next step; past the last one: read and dropped)
              }
              __builtin_amdgcn_sched_barrier(0);
              fpc_mfma_step<false, false>(acc2[2 * blk], acc2[2 * blk + 1], af[blk & 1][0], af[blk & 1][1], cb[u], cb[u]);
              __builtin_amdgcn_sched_barrier(0);
            });
          }
        }
      };
      if (proj) {
        // projection: more K for the same accumulators -- 64 channels of the tile's 256 pixels per pass, [px][16 float4];
        // thread (pixel (tid >> 4) + 32 i, quad tid & 15)
        constexpr int XIT = PX * 16 / C::NT;                          // 8 staging float4 per thread and pass
        f32x4 xst[XIT];
        auto load_x = [&](int pass) {
          const int kx4 = min(16, a.k8_x * 2 - pass * 16);
          const int px0 = tid_t >> 4;                                 // 0..31
          const int xp = (((px0 / TW) * a.W + px0 % TW) * a.csx + ((tid_t & 15) < kx4 ? pass * 64 + (tid_t & 15) * 4 : 0)) * 4;
#pragma unroll
          for (int i = 0; i < XIT; ++i) {
            const int so = xbase + i * RPI * a.W * a.csx * 4;        // (uniform) 32 pixels = RPI pixel rows further on
            xst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xp + so, 0, 0));
          }
        };
        load_x(0);
        {
          float zf = 0.f;
          asm volatile("" : "+v"(zf));
#pragma unroll
          for (int mb = 0; mb < 8; ++mb) acc2[mb] = f32x4{zf, zf, zf, zf};
          if constexpr (DUST) {
#pragma unroll
            for (int mb = 0; mb < 8; ++mb)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc2[mb][r] = dhv[mb][r] * dw2r;
          }
        }
        gemm_over(KH, 0);
        if (wg == wg_stamp) { FPC_STAMP(3) }
        const int npass = (a.k8_x + 7) >> 3;
        for (int pass = 0; pass < npass; ++pass) {
          FPC_LDS_BARRIER();   // h (or the previous pass's x) has been read by every wave
          {
            int xw = TL4 + (tid_t >> 4) * (RH / 4) + (tid_t & 15);
            asm volatile("" : "+v"(xw));
#pragma unroll
            for (int i = 0; i < XIT; ++i) lds4[xw + i * 32 * (RH / 4)] = xst[i];
          }
          FPC_LDS_BARRIER();
          if (pass + 1 < npass) load_x(pass + 1);
          if constexpr (DUST) {      // this pass's 64 channels of x against the projection's column for output 64
            const f32x4 wa = lds4[C::D_COL / 4 + 20 + pass * 16 + dq];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
              const int px = 32 * wave + 4 * it + dpxl;
              const f32x4 xa = lds4[TL4 + px * (RH / 4) + dq];
              float v = dpart[it];
#pragma unroll
              for (int e = 0; e < 4; ++e) v = __builtin_fmaf(xa[e], wa[e], v);
              dpart[it] = v;
            }
          }
          const int steps = min(4, a.k8_x / 2 - pass * 4);   // 16-channel steps of this pass: 4 (Cin is a multiple of 64 here)
          gemm_over(steps, KH + pass * 4);
        }
      } else {
        // identity: x is the INITIAL VALUE of the accumulators, read in their layout
        int xl = (4 * kq * a.csx + 16 * grp + n16) * 4;
        asm volatile("" : "+v"(xl));
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            constexpr int BPR = TW / 16;                                // pixel blocks per pixel row (1 or 2)
            const int mbg = 8 * pb + mb;                                // (uniform)
            const int so = xbase + ((mbg / BPR) * a.W + (mbg % BPR) * 16 + r) * a.csx * 4;
            acc2[mb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, xl, so, 0));
          }
        if constexpr (DUST) {
#pragma unroll
          for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2[mb][r] = __builtin_fmaf(dhv[mb][r], dw2r, acc2[mb][r]);
        }
        gemm_over(KH, 0);
        if (wg == wg_stamp) { FPC_STAMP(3) }
      }

      // ------------------------------------------------ epilogue: output tile through LDS -> 16-byte stores
      if (wg == wg_stamp) { FPC_STAMP(4) }
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
      FPC_LDS_BARRIER();
      {
        int ew = C::OFF_H0 + (128 * pb + 4 * kq) * RH + 16 * grp + n16;
        asm volatile("" : "+v"(ew));
        const float bias = a.b2[16 * grp + n16];
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[ew + (16 * mb + r) * RH] = acc2[mb][r] + bias;
      }
      FPC_LDS_BARRIER();
      store_out();
      if constexpr (DUST) {
        // channels 64..71 of the pixel: (out[64], 0, 0, 0) from the pair's first thread, zeros from the second (pad channels
        // are exact zeros: the next layer's 16-byte slot of input channel 64 relies on it).  The 16 lanes of a pixel add up
        // in a halving butterfly (4 + 2 + 1 + 1 exchanges) that ends with the total of pixel it = q >> 1 in both lanes q & 1.
        const bool b3 = (dq & 8) != 0, b2 = (dq & 4) != 0, b1 = (dq & 2) != 0;
        float r1[4], r2[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) r1[k] = (b3 ? dpart[4 + k] : dpart[k]) + __shfl_xor(b3 ? dpart[k] : dpart[4 + k], 8);
#pragma unroll
        for (int k = 0; k < 2; ++k) r2[k] = (b2 ? r1[2 + k] : r1[k]) + __shfl_xor(b2 ? r1[k] : r1[2 + k], 4);
        float tot = (b1 ? r2[1] : r2[0]) + __shfl_xor(b1 ? r2[0] : r2[1], 2);
        tot += __shfl_xor(tot, 1);
        const int dpx = 32 * wave + 4 * (dq >> 1) + dpxl, dhf = dq & 1, drow = dpx / TW, dcol = dpx % TW;
        tot += dx64 + lds[C::D_HEAD + W36Dust::B2];
        tot = tot > 0.f ? tot : 0.f;
        const bool ok = (drow < rows_valid) & (x0 + dcol < a.W);
        const int doff = ok ? ((drow * a.W + dcol) * a.cso + 64 + 4 * dhf) * 4 : W36_MARKER;
        int voff;
        asm("v_add_i32 %0, %1, %2 clamp" : "=v"(voff) : "v"(doff), "s"(obase));
        const f32x4 v = {dhf ? 0.f : tot, 0.f, 0.f, 0.f};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), orsrc, voff, 0, 0);
      }
      if (wg == wg_stamp) { FPC_STAMP(5) FPC_RSTAMP(7) }
      FPC_LDS_BARRIER();   // the region is reused by the next tile's pipeline
    }
    pos_cur = pos_next;
    base_cur = base_next;
  }  // persistent tile loop
}

template <int TYT, int TXT, int RING = 3>
__global__ __launch_bounds__(512, 1) void wblock36p_kernel(const WBlockArgs a) {
  wblock36p_body<TYT, TXT, RING, false>(a);
}

// The detector's blocks at two waves per SIMD: 64 channels as wblock36p_kernel + the 65th ("dustbin") channel beside them
// (W36Dust, wblock36_mfma.h), as wblock36_dust_kernel does for the one-wave instance.
template <int TYT, int TXT, int RING = 3>
__global__ __launch_bounds__(512, 1) void wblock36p_dust_kernel(const WBlockArgs a) {
  wblock36p_body<TYT, TXT, RING, true>(a);
}

}  // namespace fpc
