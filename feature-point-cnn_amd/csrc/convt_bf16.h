// convt_bf16.h -- ConvTranspose2d(k3, s2, p1, op1) + BN + ReLU in bf16 as ONE launch (round 5; `dtype = FPC_BF16`).
// Reference: python/src/superpoint.py:45-47,56-58 (`up_sample`, `bn`, `relu` of the descriptor head).
//
// The transposed convolution is four ordinary ones over the INPUT grid, one per output parity (py, px): output pixel
// (2y + py, 2x + px) sums 1 / 2 / 2 / 4 taps of input pixels (y + dy, x + dx), dy <= py, dx <= px.  Rounds 2-4 ran them as
// four launches of block_bf16_kernel's conv-only path, each staging the whole 256-channel input tile through LDS for one
// to four taps: 64 KB of staging for 2 k cycles of MFMAs in the one-tap phase -- 0.23 of the MFMA peak, 0.45 ms per 64 HD
// frames.  Here a workgroup stages a tile's halo chunk ONCE and issues all nine taps on it: the four parities'
// accumulators live side by side (4 x MB x 16 registers per lane, MB = 2 pixel blocks x one 32-channel block per wave),
// so a step of 16 input channels is eight pixel reads (four halo offsets x MB) and nine weight fragments for eighteen
// MFMAs -- 0.94 operand fetches per MFMA where the phase launches had 1.25, and a ninth of their staging.
//
// Structure as block_bf16.h: persistent grid with an XCD-aware tile walk, halo chunk (KC = 64 channels) global ->
// registers -> LDS with the next chunk's request in flight during this chunk's MFMAs, bank-conflict-free 4 x 8 pixel
// blocks, weights as the MFMA's A operand (accumulators hold the tile transposed: a lane owns one pixel and groups of four
// consecutive channels), fragments straight from L2 through a register ring one step (nine fragments) deep and read
// circularly, epilogue through LDS into 16-byte stores.  gridDim.y = the output-channel parts (CMIDP / (32 WN)): a
// workgroup of WM x WN waves owns TH x TW input pixels x 32 WN output channels x all four parities.
#pragma once
#include <type_traits>
#include <utility>

#include "block_bf16.h"

namespace fpc {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): ring slots, accumulators and taps are named at compile time
template <int... I, class F>
__device__ __forceinline__ void ct_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void ct_static_for(F&& f) {
  ct_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB, int CMIDP>
struct ConvTBfCfg {
  static_assert(S == 1 && EXT == 2, "halo = the tile + one row below and one column to the right");
  static_assert(NB == 1 && MB == 2, "a wave holds two pixel blocks x one channel block x four parities (128 accumulators)");
  static_assert(WM * MB * 32 == TH * TW && bf_blocked(TH, TW, 1), "whole 4 x 8 pixel blocks");
  static_assert(KC % 16 == 0 && CMIDP % (WN * 32) == 0, "channel blocks");
  static constexpr int NT = WM * WN * 64;
  static constexpr int HW = TW + 1, HH = TH + 1, HP = bf_halo_pitch(TH, TW, 1, 2);
  static constexpr int ROW16 = KC / 8 + 1;                // 16-byte units per halo pixel (+1 skew)
  static constexpr int M = TH * TW, NW = WN * 32;         // pixels / output channels of a workgroup
  static constexpr int NPARTS = CMIDP / NW;               // gridDim.y
  static constexpr int NBLK = CMIDP / 32;                 // 32-channel blocks of the fragment stream
  static constexpr int ROWO16 = NW / 8 + 1;               // output tile row, 16-byte units (+1 skew)
  static constexpr int HALO_BYTES = HH * HP * ROW16 * 16;
  static constexpr int OUT_BYTES = 4 * M * ROWO16 * 16;   // the four parities of the tile, bf16
  static constexpr int TILE_BYTES = HALO_BYTES > OUT_BYTES ? HALO_BYTES : OUT_BYTES;
  static constexpr int LDS_BYTES = TILE_BYTES + NW * 4;   // + bias
  static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
};

// Tap t of a step, in issue order; halo offset group o = (dy, dx): 0 = (1, 1), 1 = (0, 0), 2 = (0, 1), 3 = (1, 0); parity
// ph = 2 py + px.  Kernel element (ky, kx) of tap (o, ph): ky = py ? (dy ? 0 : 2) : 1, likewise kx -- out(2y + py) =
// sum over iy, ky with 2 iy - 1 + ky = 2y + py.  The order keeps two MFMAs on one accumulator at least two MFMAs apart.
//   t : 0        1        2        3        4        5        6        7        8
//   o : (1,1)    (0,0)    (0,0)    (0,0)    (0,0)    (0,1)    (0,1)    (1,0)    (1,0)
//   ph: 3        0        1        2        3        1        3        2        3
struct ConvTTaps {
  static constexpr int group[9] = {0, 1, 1, 1, 1, 2, 2, 3, 3};
  static constexpr int phase[9] = {3, 0, 1, 2, 3, 1, 3, 2, 3};
  static constexpr int gdy[4] = {1, 0, 0, 1}, gdx[4] = {1, 0, 1, 0};
  static constexpr int first[5] = {0, 1, 5, 7, 9};      // taps of group g: first[g] .. first[g + 1] - 1
  static constexpr int ky(int t) { return (phase[t] >> 1) ? (gdy[group[t]] ? 0 : 2) : 1; }
  static constexpr int kx(int t) { return (phase[t] & 1) ? (gdx[group[t]] ? 0 : 2) : 1; }
};

// Uses of BlockBfArgs: x, csx, x_bytes, H, W (the INPUT grid), nchunk, w1 ([16 nchunk K16 .. steps][9 taps][NBLK][64] uint4,
// + 2 steps of padding), b1 ([CMIDP] folded bias), out, cso, OH, OW (= 2H, 2W), tiles_x, tiles_y, frame0, total_tiles.
#ifdef FPC_DIAG
// wave 0's shader clock at a tile's phase boundaries (the workgroup's third tile): 16 slots per workgroup
#define CT_STAMP(i)                                                                                          \
  if (a.stamps && tcur == t_stamp && threadIdx.x == 0) {                                                     \
    unsigned long long t_;                                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                               \
    a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = t_;                                 \
  }
#else
#define CT_STAMP(i)
#endif

template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB, int CMIDP>
__global__ __launch_bounds__(WM* WN * 64, 2) void convt_bf16_kernel(const BlockBfArgs a) {
  using C = ConvTBfCfg<TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP>;
  using TP = ConvTTaps;
  constexpr int NT = C::NT, HW = C::HW, HH = C::HH, HP = C::HP, ROW16 = C::ROW16, K16 = KC / 16, KC8 = KC / 8;
  constexpr int NV = HH * HW * KC8, ITER = (NV + NT - 1) / NT, ROWO16 = C::ROWO16, M = C::M, NW = C::NW;
  extern __shared__ uint4 lds16[];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  const int part = blockIdx.y;
  const int tiles = a.tiles_x * a.tiles_y;
  const int T = a.total_tiles, per = gridDim.x >> 3, xcd = blockIdx.x & 7;
  const int t_first = (int)(((long long)xcd * T) >> 3), t_end = (int)(((long long)(xcd + 1) * T) >> 3);
  struct TileP { int b, ty, tx, iy0, ix0, live; unsigned xbase; };
  auto tile_params = [&](int tt) {
    TileP p;
    p.live = tt < t_end;
    const int tc = p.live ? tt : t_first;
    const int bl = tc / tiles, t = tc - bl * tiles;
    p.b = a.frame0 + bl;
    p.ty = t / a.tiles_x;
    p.tx = t - p.ty * a.tiles_x;
    p.iy0 = p.ty * TH;
    p.ix0 = p.tx * TW;
    p.xbase = (unsigned)((p.b * a.H + p.iy0) * a.W + p.ix0) * (unsigned)(a.csx * 2);
    return p;
  };

  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = (wm * MB + mb) * 32 + l31;
    int py, px;
    bf_pixel<TH, TW, 1>(m, py, px);
    abase[mb] = (py * HP + px) * ROW16 + half;
  }
  float* bias_lds = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(lds16) + C::TILE_BYTES);   // [NW]
  for (int i = tid; i < NW; i += NT) bias_lds[i] = a.b1[part * NW + i];

  // ---- the halo chunk: global -> registers (load_chunk) -> LDS (store_chunk), as block_bf16.h's carried form
  u32x4 stage[ITER];
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  static_assert(NT % KC8 == 0, "a thread keeps its channel group across the elements it stages");
  constexpr int PS = NT / KC8, PQ = PS / HW, PR = PS % HW;
  auto load_chunk = [&](const TileP& p, int chunk) {
    const int wlim = (chunk < a.nchunk && p.live) ? a.W : 0;   // nothing is in range past the last chunk / tile
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int pix0 = tl / KC8, c8 = tl - pix0 * KC8;
    int hy = pix0 / HW, hx = pix0 - hy * HW;
    unsigned off = p.xbase + (unsigned)(((hy * a.W + hx) * a.csx + chunk * KC + c8 * 8) * 2);
    const unsigned dstep = (unsigned)(((PQ * a.W + PR) * a.csx) * 2), dwrap = (unsigned)(((a.W - HW) * a.csx) * 2);
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int iy = p.iy0 + hy, ix = p.ix0 + hx;
      bool ok = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)wlim);
      if ((i + 1) * NT > NV) ok = ok & (hy < HH);
      stage[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(ok ? off : 0xfffffff0u), 0, 0));
      if (i + 1 < ITER) {
        hx += PR;
        const bool wrap = hx >= HW;
        hx -= wrap ? HW : 0;
        hy += PQ + (wrap ? 1 : 0);
        off += dstep + (wrap ? dwrap : 0u);
      }
    }
  };
  auto store_chunk = [&]() {
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int pix0 = tl / KC8, c8 = tl - pix0 * KC8;
    int hy = pix0 / HW, hx = pix0 - hy * HW;
    int slot = (hy * HP + hx) * ROW16 + c8;
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      int sl = slot;
      if ((i + 1) * NT > NV) sl = hy < HH ? sl : (HH * HW - 1 + (HH - 1) * (HP - HW)) * ROW16 + KC8;   // past the halo: the skew slot of its last pixel (never read)
      *reinterpret_cast<u32x4*>(&lds16[sl]) = stage[i];
      if (i + 1 < ITER) {
        hx += PR;
        const bool wrap = hx >= HW;
        hx -= wrap ? HW : 0;
        hy += PQ + (wrap ? 1 : 0);
        slot += (PQ * HP + PR) * ROW16 + (wrap ? (HP - HW) * ROW16 : 0);
      }
    }
  };

  // ---- weight fragments: ring of nine (one per tap), refilled for the NEXT step as soon as a tap's MFMAs are issued; the
  // stream is read circularly (a tile's last step requests step 0 again: the next tile's first)
  constexpr int FRAG = C::NBLK * 64 * 16;                      // bytes between two fragments of the stream
  const int nsteps = a.nchunk * K16;
  const int wtotal = nsteps * 9 * FRAG;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.w1), 0, wtotal, 0x00020000);
  const unsigned wlane = (unsigned)(((part * WN + wn) * 64 + lane) * 16);
  u32x4 ring[9];
  int wreq = 0;   // (scalar) byte offset of the next fragment to request
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    ring[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)wlane, wreq, 0));
    wreq += FRAG;
  }

  TileP cur = tile_params(t_first + (int)(blockIdx.x >> 3));
  load_chunk(cur, 0);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)0xffffff00u, 0x00020000);

  [[maybe_unused]] const int t_stamp = t_first + (int)(blockIdx.x >> 3) + 2 * per;
  for (int tcur = t_first + (int)(blockIdx.x >> 3); tcur < t_end; tcur += per) {
    const int b = cur.b, ty = cur.ty, tx = cur.tx;
    CT_STAMP(0)
    f32x16 acc[4][MB];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ph][mb][r] = 0.f;

    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      FPC_LDS_BARRIER();   // the previous chunk's pixels (or the previous tile's output tile) have been read
      store_chunk();
      FPC_LDS_BARRIER();
      CT_STAMP(1 + 2 * chunk)
      load_chunk(cur, chunk + 1);   // (all zeros past the last chunk: no branch between a request and its use)
      // pixel operands: a ring of three groups, read two groups ahead of their MFMAs
      constexpr int NG = 4 * K16;
      u32x4 av[3][MB];
      auto read_group = [&](auto GI) {
        constexpr int gi = decltype(GI)::value, k16 = gi >> 2, g = gi & 3;
        constexpr int off = (TP::gdy[g] * HP + TP::gdx[g]) * ROW16 + k16 * 2;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[gi % 3][mb] = *reinterpret_cast<const u32x4*>(&lds16[abase[mb] + off]);
      };
      read_group(std::integral_constant<int, 0>{});
      read_group(std::integral_constant<int, 1>{});
      ct_static_for<NG>([&](auto GI) __attribute__((always_inline)) {
        constexpr int gi = decltype(GI)::value, g = gi & 3;
        if constexpr (gi + 2 < NG) read_group(std::integral_constant<int, gi + 2>{});
        __builtin_amdgcn_sched_barrier(0);
        ct_static_for<TP::first[g + 1] - TP::first[g]>([&](auto J) __attribute__((always_inline)) {
          constexpr int t = TP::first[g] + decltype(J)::value, ph = TP::phase[t];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            acc[ph][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[t]), __builtin_bit_cast(bf16x8, av[gi % 3][mb]),
                                                                  acc[ph][mb], 0, 0, 0);
          ring[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)wlane, wreq, 0));
          wreq += FRAG;
          if constexpr (t == 8) wreq = wreq == wtotal ? 0 : wreq;   // (whole steps: the request runs exactly one step ahead)
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      CT_STAMP(2 + 2 * chunk)
    }

    // ---------------------------------------------------------------- epilogue: + bias, ReLU -> the four parities as a bf16 tile in LDS -> 16-byte stores
    const TileP nxt = tile_params(tcur + per);
    load_chunk(nxt, 0);   // lands behind the epilogue (all zeros past the last tile)
    FPC_LDS_BARRIER();    // the last chunk's pixels have been read
    CT_STAMP(9)
    {
      int nl = wn * 32 + 4 * half, ml = (wm * MB) * 32 + l31;
      asm volatile("" : "+v"(nl), "+v"(ml));
      unsigned char* ol = reinterpret_cast<unsigned char*>(lds16);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n0 = nl + 8 * g;
        const float4 bias = *reinterpret_cast<const float4*>(bias_lds + n0);
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const int m = ml + mb * 32;
            const float v0 = fmaxf(acc[ph][mb][4 * g + 0] + bias.x, 0.f), v1 = fmaxf(acc[ph][mb][4 * g + 1] + bias.y, 0.f);
            const float v2 = fmaxf(acc[ph][mb][4 * g + 2] + bias.z, 0.f), v3 = fmaxf(acc[ph][mb][4 * g + 3] + bias.w, 0.f);
            *reinterpret_cast<uint2*>(ol + (ph * M + m) * (ROWO16 * 16) + n0 * 2) = make_uint2(pk_bf16(v0, v1), pk_bf16(v2, v3));
          }
      }
    }
    FPC_LDS_BARRIER();
    CT_STAMP(10)
    {
      // element e = tid + i NT: 16 bytes = channels 8 c8 ..+7 of (parity, tile pixel).  NT / C8 = 32 pixels per step = one 4 x 8
      // block: c8 and the pixel's place inside its block are the thread's own, block and parity are the step's
      constexpr int C8 = NW / 8, MS = NT / C8, BX = TW / 8, EIT = 4 * M * C8 / NT;
      static_assert(NT % C8 == 0 && MS == 32 && M % 32 == 0, "one pixel block per store step");
      int tl = tid;
      asm volatile("" : "+v"(tl));
      const int l = tl / C8, c8 = tl - l * C8;
      const int yb = ty * TH + (l >> 3), xb = tx * TW + (l & 7);
      const unsigned obase = (unsigned)((b * a.OH + 2 * yb) * a.OW + 2 * xb) * (unsigned)(a.cso * 2) + (unsigned)((part * NW + c8 * 8) * 2);
      const int lbase = l * ROWO16 + c8;
#pragma unroll
      for (int i = 0; i < EIT; ++i) {
        const int ph = (i * MS) / M, blk = ((i * MS) % M) >> 5;
        const int cy = (blk / BX) * 4, cx = (blk % BX) * 8;
        const bool on = (yb + cy < a.H) & (xb + cx < a.W);
        const unsigned off = obase + (unsigned)((2 * cy + (ph >> 1)) * a.OW + 2 * cx + (ph & 1)) * (unsigned)(a.cso * 2);
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&lds16[lbase + (ph * M + blk * 32) * ROWO16]), orsrc,
                                               (int)(on ? off : 0xfffffff0u), 0, 0);
      }
    }
    CT_STAMP(11)
    cur = nxt;
  }
}
#undef CT_STAMP

}  // namespace fpc
