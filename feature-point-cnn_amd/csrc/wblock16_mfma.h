// wblock16_mfma.h -- second generation of the Winograd ResNetBlock kernel (wblock_mfma.h has the algorithm).
//
// What round 1's kernel left on the table (its in-kernel stamps): the GEMMs of a tile saturate
// the matrix cores, but 45 % of a tile is spent in phases that use none of them and that every wave of the workgroup
// walks through in lockstep -- halo staging, input transform, the accumulators' round trip through LDS for the output
// transform, barriers in between.  Two changes of structure remove most of that:
//
//  (1) A wave owns 16 OUTPUT CHANNELS for ALL 16 Winograd positions (v_mfma_f32_16x16x4_f32 blocks: 16 tiles x 16
//      channels, 4 accumulator registers) instead of 4 positions x 64 channels.  A lane then holds, for its channel and
//      its four tiles, all 16 positions M_xi: the output transform Y = A^T M A is register arithmetic -- no LDS round
//      trip of the accumulators, no barrier.  Weight traffic from L2 is unchanged (every fragment is still read once
//      per workgroup and tile); the A operand is read 4x as often from LDS (13 % of its bandwidth).
//  (2) The input side is SOFTWARE-PIPELINED into the GEMM: channels are walked in chunks of 16 with two V buffers and
//      two halo buffers; while the MFMAs of chunk c run, the same instruction stream transforms chunk c+1 (halo -> V),
//      stores chunk c+2 (registers -> halo) and requests chunk c+3 (global -> registers) -- a few dozen VALU / LDS
//      instructions spread over 128 MFMAs.  One LDS barrier per chunk, nothing but MFMA-paced code between them, and
//      the pipeline runs on across tile boundaries (the grid is persistent): during a tile's last chunk the NEXT
//      tile's first chunk is transformed.
//
// LDS (131 200 B): V0 | halo0 | halo1 | V1, and the region from V1 on is reused at the end of a tile -- when only V0
// and the halo buffers hold live data of the next tile -- for h, for the shortcut's x tile and for the output tile.
// V rows are 16 floats with an XOR swizzle of their four 16-byte slots (slot ^ 2*(tile>>3 & 1)), h / x / output rows
// N + 8 floats: ds_read_b128 of the A operand (lane = (tile or pixel) l & 15, k quarter l >> 4) is conflict-free.
//
// Fragment order (host: pack_w16 in fpc_api.hip): for 16 input channels c0..c0+15 and 16 output channels, lane
// (n = l & 15, kq = l >> 4) holds the float4 { B[c0 + 4 kq + j][n] }, j = 0..3; MFMA j of a step takes component j of
// that and component j of the A float4 { A[m][c0 + 4 kq .. + 3] } -- a permutation of K inside the step, the same on
// both operands.
#pragma once
#include <type_traits>

#include "wblock_mfma.h"

namespace fpc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// TH_ = 8: 8 x 16 pixel tiles (throughput).  TH_ = 4 (N = 128 only): 4 x 16 pixel tiles for calls of a few frames --
// twice as many workgroups, each with half the MFMAs: the latency of a layer is the time of ONE tile when the frame has
// fewer tiles than the chip has CUs (60 x 80 features: 40 tiles of 8 x 16).  The input side still stages and transforms
// the 10-row window (rows 6..9 feed tiles nobody multiplies): wasteful, but off the critical path of a single frame.
template <int NCG, int TH_ = 8>
struct W16Cfg {
  static constexpr int NT = 512, TH = TH_, TW = 16, HW = 18, HH = 10, KC = 16;
  static constexpr int TG = 8 / NCG;        // wave groups over tiles / pixels: 1 (N = 128) or 2 (N = 64)
  static constexpr int MBW = TH_ == 4 ? 1 : 2 / TG;   // 16-tile blocks per wave in the Winograd GEMMs
  static constexpr int MB2 = TH_ == 4 ? 4 : 8 / TG;   // 16-pixel blocks per wave in the 1x1
  static constexpr int PX = TH_ * TW;       // output pixels of a tile
  static_assert(TH_ == 8 || (TH_ == 4 && NCG == 8), "half-height tiles exist for the 128-channel instance");
  static constexpr int N = NCG * 16;
  static constexpr int V_FLOATS = 16 * 32 * 16;
  static constexpr int HROW = 20;           // halo row: 16 channels + 4 floats of skew
  static constexpr int NHALO = HH * HW;     // 180 pixels
  static constexpr int HALO_FLOATS = 2 * NT / 4 * HROW;   // 256 pixel slots: every thread stores both of its float4 (180 are real)
  static constexpr int OFF_V0 = 0, OFF_H0 = V_FLOATS, OFF_H1 = OFF_H0 + HALO_FLOATS, OFF_V1 = OFF_H1 + HALO_FLOATS;
  static constexpr int RH = N + 8;          // h / output row (floats): (N + 8) / 4 = 2 (mod 16)
  static constexpr int RX = 128 + 8;        // x staging row (up to 128 channels per pass)
  static constexpr int T_FLOATS = 128 * (RH > RX ? RH : RX);
  static constexpr int LDS_BYTES = (OFF_V1 + (T_FLOATS > V_FLOATS ? T_FLOATS : V_FLOATS)) * 4;
  static constexpr int WPAD = 16;           // zero steps behind every wave's fragment stream (the ring reads ahead)
  // Depth of the weight-fragment ring.  vmcnt retires IN ORDER: once a chunk's halo request (HBM / Infinity Cache, a few
  // thousand cycles under load) is in the queue, every younger fragment load waits behind it, so the ring must already
  // hold the fragments of that whole time: RING - 1 positions of 8 (4) MFMAs per wave pair.  With 4 slots every chunk
  // stalled ~3 k cycles (in-kernel stamps: 10.1 k per chunk against 7.2 k of MFMA time at N = 128, 6.9 k / 3.6 k at 64).
  static constexpr int RING = NCG == 8 ? 8 : 16;
  static_assert(NCG == 8 || NCG == 4, "8 waves = NCG channel groups x TG tile groups");
};

template <int NCG, int TH_ = 8>
__global__ __launch_bounds__(512, 2) void wblock16_kernel(const WBlockArgs a) {
  using C = W16Cfg<NCG, TH_>;
  constexpr int NT = C::NT, TH = C::TH, TW = C::TW, HW = C::HW, HROW = C::HROW;
  constexpr int MBW = C::MBW, MB2 = C::MB2, N = C::N, RH = C::RH, RX = C::RX, RING = C::RING, PX = C::PX;
  extern __shared__ float lds[];
  float* const V0 = lds + C::OFF_V0;
  float* const HB0 = lds + C::OFF_H0;
  float* const HB1 = lds + C::OFF_H1;
  float* const TL = lds + C::OFF_V1;        // h, x staging, output tile

  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % NCG, tg = wave / NCG;
  const int tiles = a.tiles_x * a.tiles_y;
  const int nchunk = a.nchunk;              // Cin / 16, even, >= 4
  // second half of a 256-wide conv-only layer (gridDim.y == 2)
  const float4* const w1h = a.w1 + (size_t)blockIdx.y * (a.ysplit_floats / 4);
  const float* const b1h = a.b1 + (size_t)blockIdx.y * a.ysplit_floats;
  float* const outh = a.out + blockIdx.y * N;

  // tile walk (persistent, XCD-aware): as wblock_mfma_kernel
  const bool xcd_order = a.xcd_order && (gridDim.x & 7) == 0;
  const int wg_step = xcd_order ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int xchunk = (a.total + 7) >> 3;
  const int wg_first = xcd_order ? (int)(blockIdx.x & 7) * xchunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int wg_end = xcd_order ? min(a.total, ((int)(blockIdx.x & 7) + 1) * xchunk) : a.total;
  if (wg_first >= wg_end) return;
  // (The second-dispatched half of the waves loses the issue arbitration against its SIMD partner on every
  // instruction -- MI355X_MICROARCH.md, two waves per SIMD.  Measured in round 2 and not kept: static s_setprio 1 for
  // waves 4-7 just swaps the roles (+-1 %); raising it for the first positions of every chunk only balanced the two
  // (+1.4 % on the chunk loop) at the cost of two branches in the loop body.)

  // ---------------------------------------------------------------- input side: L (global -> registers), S (-> halo), T (halo -> V)
  // The halo request is a BUFFER load (buffer_load_dwordx4 ... offen): out-of-frame pixels, the pixel slots past the
  // 180 real ones and the tiles behind a workgroup's last one get an offset outside the descriptor's range and come
  // back as zeros from the hardware's bounds check -- nothing touches the loaded registers before they are stored a
  // chunk later (a select right after the load made every chunk wait for its own request), and the per-call address
  // arithmetic is two adds and two compares per element on values precomputed once per thread (as 64-bit pointer
  // arithmetic it was ~60 VALU instructions in the middle of the MFMA stream, and the younger half of the waves --
  // which lose the issue arbitration -- fell 1.4 k cycles behind per chunk: in-kernel stamps).
  f32x4 stage[2];   // (vector values, not float4 structs: a struct copy becomes a memcpy into a private array that is never promoted)
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  auto load_halo = [&](int wg, int chunk) {            // chunk `chunk` of tile wg
    int tl = tid;
    asm volatile("" : "+v"(tl));                       // (recomputed per call: held across the loop the four index registers spilled)
    const bool live = wg < wg_end;
    const int wgc = live ? wg : wg_first;
    const int bl = wgc / tiles;
    const int bb = a.frame0 + bl;
    const int t = wgc - bl * tiles;
    const int tyy = t / a.tiles_x, txx = t - tyy * a.tiles_x;
    const int iy0 = tyy * TH - 1, ix0 = txx * TW - 1;
    const int hlim = live ? a.H : 0;                   // nothing is in range for a tile past the end
    const unsigned base = (unsigned)((bb * a.H + iy0) * a.W + ix0) * (unsigned)(a.csx * 4) + (unsigned)(chunk * 64);   // unsigned: mod 2^32, exact for in-frame pixels
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tl + i * NT;
      const int pix = e >> 2, c4 = e & 3;
      const int hy = (pix * 3641) >> 16, hx = pix - hy * HW;     // pix / 18 for pix < 256
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = ((unsigned)iy < (unsigned)hlim) & ((unsigned)ix < (unsigned)a.W) & (hy < C::HH);   // (& not &&: no branch in the loop body)
      unsigned in_off = base + (unsigned)((hy * a.W + hx) * a.csx * 4 + c4 * 16);
      asm volatile("" : "+v"(in_off));   // computed for every lane: as a conditional the compiler branches around it (a block split in the loop)
      const unsigned voff = ok ? in_off : 0xfffffff0u;
      stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)voff, 0, 0));
    }
  };
  auto store_halo = [&](float* hb) {
    int tl = tid;
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tl + i * NT;
      *reinterpret_cast<f32x4*>(hb + (e >> 2) * HROW + (e & 3) * 4) = stage[i];   // (slots >= 180 are padding: no branch in the loop)
    }
  };
  // the transform item of this thread: Winograd tile wt (0..31), channel ch (0..15) of the chunk
  const int wt_t = tid >> 4, ch_t = tid & 15;
  const int trd = ((2 * (wt_t >> 3)) * HW + 2 * (wt_t & 7)) * HROW + ch_t;                       // halo read base
  const int twr = wt_t * 16 + ((((ch_t >> 2) ^ (2 * ((wt_t >> 3) & 1))) << 2) | (ch_t & 3));     // V write base (swizzled)
  float td[4][4];   // d, then B^T d in place
  auto t_read_row = [&](const float* hb, int rd, int i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) td[i][j] = hb[rd + (i * HW + j) * HROW];
  };
  auto t_col = [&](int j) {  // B^T d, column j, in place
    const float d0 = td[0][j], d1 = td[1][j], d2 = td[2][j], d3 = td[3][j];
    td[0][j] = d0 - d2;
    td[1][j] = d1 + d2;
    td[2][j] = d2 - d1;
    td[3][j] = d1 - d3;
  };
  auto t_write_row = [&](float* vb, int wr, int i) {  // (B^T d) B, row i -> positions 4 i .. 4 i + 3
    vb[(i * 4 + 0) * 512 + wr] = td[i][0] - td[i][2];
    vb[(i * 4 + 1) * 512 + wr] = td[i][1] + td[i][2];
    vb[(i * 4 + 2) * 512 + wr] = td[i][2] - td[i][1];
    vb[(i * 4 + 3) * 512 + wr] = td[i][1] - td[i][3];
  };
  auto transform_all = [&](const float* hb, float* vb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) t_read_row(hb, trd, i);
#pragma unroll
    for (int j = 0; j < 4; ++j) t_col(j);
#pragma unroll
    for (int i = 0; i < 4; ++i) t_write_row(vb, twr, i);
  };

  // ---------------------------------------------------------------- operands of the GEMMs
  // A: V[pos][tile][16 ch] -- lane (tile t16 of block mb, k quarter kq) reads one float4
  int aoff[MBW];
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) aoff[mb] = (16 * (tg * MBW + mb) + (lane & 15)) * 16 + (((lane >> 4) ^ (2 * ((lane & 15) >> 3))) << 2);
  // B: this wave's fragment stream [chunk][pos][64 lanes] float4, contiguous per channel group
  // (buffer_load ... offen with the lane offset in a VGPR that never changes and the step in the scalar offset: one
  // s_add per load instead of a 64-bit scalar address + a 64-bit vector add)
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float4*>(w1h), 0, (int)((unsigned)NCG * ((unsigned)nchunk * 16u + (unsigned)C::WPAD) * 1024u), 0x00020000);
  const unsigned wlane = (unsigned)cg * ((unsigned)nchunk * 16u + (unsigned)C::WPAD) * 1024u + lane16;
  auto ldb = [&](int s) {
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)wlane, s * 1024, 0));
    return make_float4(v.x, v.y, v.z, v.w);
  };

  // ---------------------------------------------------------------- pipeline fill for the first tile
  load_halo(wg_first, 0);
  store_halo(HB0);
  load_halo(wg_first, 1);
  FPC_LDS_BARRIER();
  transform_all(HB0, V0);
  store_halo(HB1);
  load_halo(wg_first, 2);
  FPC_LDS_BARRIER();
  // state at the top of iteration c of a tile: V[c & 1] = chunk c transformed; halo[(c + 1) & 1] = chunk c + 1 stored;
  // stage = chunk c + 2 requested

  // diagnostic stamps (make diag) describe a workgroup's THIRD tile: steady state
  const int wg_stamp = wg_first + 2 * wg_step < wg_end ? wg_first + 2 * wg_step : wg_first;
  for (int wg = wg_first; wg < wg_end; wg += wg_step) {
    const int bl = wg / tiles;
    const int b = a.frame0 + bl;
    const int t = wg - bl * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    if (wg == wg_stamp) { FPC_STAMP(0) }

    f32x4 acc[16][MBW];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb) acc[p][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 bq[RING];
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) bq[i] = ldb(i);

    // ---------------------------------------------------------------- phase 1: 16 GEMMs per chunk, input side of the next chunks in between
    // (two chunks per trip so that the buffer of every LDS access is known at compile time: with `c & 1` selects the
    // addresses of the 32 operand reads and of the transform cost a vector add each -- and every instruction that is
    // not an MFMA costs the SIMD ~2.6 cycles of a chunk, in-kernel stamps of both instances)
    auto chunk_body = [&](auto PAR, const int c) {
      constexpr int par = decltype(PAR)::value;
      float* hs = par ? HB1 : HB0;         // halo[c & 1]
      // (LDS addresses as ONE opaque register per stream -- buffer base included, V1 and halo1 lie beyond the 64 KiB
      // an instruction's immediate offset reaches -- plus immediates; with the base added per access the loop carried
      // 60 vector adds per two chunks, and every non-MFMA instruction costs the SIMD ~2.6 cycles)
      constexpr int VB_OFF = par ? C::OFF_V1 : C::OFF_V0, VN_OFF = par ? C::OFF_V0 : C::OFF_V1;
      constexpr int HN_OFF = par ? C::OFF_H0 : C::OFF_H1;   // halo[(c + 1) & 1]
      const float* vb = lds;
      float* vn = lds;
      const float* hn = lds;
      // chunk c + 3 of this tile, or the next tile's chunk c + 3 - nchunk
      const int c3 = c + 3;
      const int l_wg = c3 < nchunk ? wg : wg + wg_step;
      const int l_c = c3 < nchunk ? c3 : c3 - nchunk;
      if (wg == wg_stamp && c == 2) { FPC_STAMP(6) }
#ifdef FPC_DIAG
      unsigned long long tq[5] = {0, 0, 0, 0, 0};
#define FPC_TQ(i) if (a.stamps) asm volatile("s_memtime %0" : "=s"(tq[i]));
#else
#define FPC_TQ(i)
#endif
      // A operand: ONE register set per 16-tile block, refilled for position p + 1 as soon as the block's last MFMA of
      // position p has been issued (an MFMA reads its operands when it issues); the other block's four MFMAs cover the
      // LDS latency.  (A second set, loaded a whole position ahead, cost the 8 registers the deeper ring needs.)
      float4 ac[MBW];
      // (opaque per-chunk copies of the lane's LDS offsets: left visible, every `buffer + offset + position` sum is
      // loop-invariant, gets its own register outside the tile loop -- 40 of them -- and is spilled; opaque, the
      // position becomes the instruction's immediate offset)
      int ao[MBW], trd_c = trd + HN_OFF, twr_c = twr + VN_OFF;
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb) {
        ao[mb] = aoff[mb] + VB_OFF;
        asm volatile("" : "+v"(ao[mb]));
      }
      asm volatile("" : "+v"(trd_c), "+v"(twr_c));
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb) ac[mb] = *reinterpret_cast<const float4*>(vb + ao[mb]);
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        bq[(p + RING - 1) % RING] = ldb(c * 16 + p + RING - 1);
        const float4 bv = bq[p % RING];
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float af = j == 0 ? ac[mb].x : j == 1 ? ac[mb].y : j == 2 ? ac[mb].z : ac[mb].w;
            const float bf = j == 0 ? bv.x : j == 1 ? bv.y : j == 2 ? bv.z : bv.w;
            acc[p][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[p][mb], 0, 0, 0);
          }
          if (p < 15) ac[mb] = *reinterpret_cast<const float4*>(vb + ao[mb] + (p + 1) * 512);
        }
        // Store, then request, at the START of the iteration: hipcc's wait-count pass merges the loop's entry and
        // back-edge states conservatively and puts s_waitcnt vmcnt(1) in front of the iteration's first MFMA whatever
        // the ring depth; with the halo request 15 positions old by then, that wait finds only L2-resident fragment
        // loads outstanding.  (Requested at p = 13 the halo was 3 positions old: a memory latency per chunk.)
        if (p == 0) store_halo(hs);
        else if (p == 1) load_halo(l_wg, l_c);
        else if (p < 6) t_read_row(hn, trd_c, p - 2);
        else if (p < 10) t_col(p - 6);
        else if (p < 14) t_write_row(vn, twr_c, p - 10);
        // (Tried and measured, in-kernel stamps per wave: the transform on waves 0-3 only, two items each, so that the
        // second-dispatched waves -- which lose the issue arbitration and end every chunk ~2 k cycles behind -- carry
        // less: the two waves of a SIMD then stop alternating on the matrix pipe altogether -- the light wave runs its
        // dependent MFMA chains back to back, 40 cycles apart, and the 8-cycle gaps are no use to its partner -- and a
        // chunk takes 10.2 k instead of 8.7 + 2.1 k.  Static or per-half-chunk s_setprio for waves 4-7: +-1 %.)
        // One position = one scheduling region: MFMAs first come, the other instructions dealt out BETWEEN them, a few
        // per MFMA (the matrix pipe runs an MFMA for 32 cycles; the wave may issue independent vector / LDS work
        // meanwhile).  As a clump behind the position's 8 MFMAs they left the pipe idle whenever the SIMD's other wave
        // was parked on an operand at that moment -- 23 % of a chunk, in-kernel stamps.
#pragma unroll
        for (int q = 0; q < 4 * MBW; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x300, 1, 0);   // one LDS read or write
          __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);   // up to four VALU / SALU
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one global read
        }
        __builtin_amdgcn_sched_barrier(0);
        if (p == 0) { FPC_TQ(0) } else if (p == 3) { FPC_TQ(1) } else if (p == 7) { FPC_TQ(2) } else if (p == 11) { FPC_TQ(3) } else if (p == 15) { FPC_TQ(4) }
      }
#ifdef FPC_DIAG
      if (a.stamps && wg == wg_stamp && c == 2 && (threadIdx.x & 63) == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        unsigned long long* q = a.stamps + 65536 * 4 + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8;
        for (int i = 0; i < 5; ++i) q[i] = tq[i];
      }
#endif
      if (wg == wg_stamp && c == 2) { FPC_STAMP(7) }
      FPC_LDS_BARRIER();
    };
    // (all of this wave's requests -- the ring's first entries among them -- are waited for HERE, once per tile: at the
    // loop header the compiler merges the counters of the two ways in, and with the ring just requested on one of them
    // it waited for every outstanding request at the top of EVERY trip, i.e. it drained the ring every two chunks)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    for (int c = 0; c < nchunk; c += 2) {
      chunk_body(std::integral_constant<int, 0>{}, c);
      chunk_body(std::integral_constant<int, 1>{}, c + 1);
    }
    if (wg == wg_stamp) { FPC_STAMP(1) }

    // Everything from here to the end of the tile depends on a thread id laundered HERE: per-thread index arithmetic
    // that only depends on threadIdx would otherwise be hoisted above the chunk loop (even out of the tile loop), live
    // through it, and push the halo staging registers into scratch -- with an s_waitcnt vmcnt(0) per chunk.
    int tid_t = tid;
    asm volatile("" : "+v"(tid_t));
    const int lane_t = tid_t & 63, t16 = lane_t & 15, kq = lane_t >> 4;
    // ---------------------------------------------------------------- shortcut operands requested now, used after the output transform
    f32x4 acc2[MB2];
    const bool proj = a.k8_x > 0;
    if (!a.conv_only) {
      if (!proj) {   // identity: the accumulators of the 1x1 start from x, read in the accumulator layout
#pragma unroll
        for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int px = 16 * (tg * MB2 + mb) + 4 * kq + r;
            int y = ty * TH + (px >> 4), x = tx * TW + (px & 15);
            y = y < a.H ? y : a.H - 1;
            x = x < a.W ? x : a.W - 1;
            acc2[mb][r] = a.x[((size_t)(b * a.H + y) * a.W + x) * a.csx + 16 * cg + t16];
          }
      } else {
#pragma unroll
        for (int mb = 0; mb < MB2; ++mb) acc2[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }

    // ---------------------------------------------------------------- output transform in registers -> h (LDS)
    {
      const float bias = b1h[16 * cg + t16];
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s0[4], s1[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            s0[j] = acc[0 * 4 + j][mb][r] + acc[1 * 4 + j][mb][r] + acc[2 * 4 + j][mb][r];
            s1[j] = acc[1 * 4 + j][mb][r] - acc[2 * 4 + j][mb][r] - acc[3 * 4 + j][mb][r];
          }
          const float y00 = s0[0] + s0[1] + s0[2] + bias, y01 = s0[1] - s0[2] - s0[3] + bias;
          const float y10 = s1[0] + s1[1] + s1[2] + bias, y11 = s1[1] - s1[2] - s1[3] + bias;
          const int wt = 16 * (tg * MBW + mb) + 4 * kq + r;          // C/D row 4 kq + r of the block = tile
          const int pm = (2 * (wt >> 3)) * TW + 2 * (wt & 7);
          float* hp = TL + pm * RH + 16 * cg + t16;
          hp[0] = y00 > 0.f ? y00 : 0.f;
          hp[RH] = y01 > 0.f ? y01 : 0.f;
          hp[TW * RH] = y10 > 0.f ? y10 : 0.f;
          hp[(TW + 1) * RH] = y11 > 0.f ? y11 : 0.f;
        }
    }
    FPC_LDS_BARRIER();
    if (wg == wg_stamp) { FPC_STAMP(2) }

    if (a.conv_only) {  // h is the result: [128 px][N] in LDS -> 16-byte stores
      constexpr int C4 = N / 4, EIT = PX * C4 / NT;
#pragma unroll
      for (int i = 0; i < EIT; ++i) {
        const int e = tid_t + i * NT;
        const int m = e / C4, c4 = e - m * C4;
        const int y = ty * TH + (m >> 4), x = tx * TW + (m & 15);
        if (y < a.H && x < a.W)
          *reinterpret_cast<float4*>(outh + ((size_t)(b * a.H + y) * a.W + x) * a.cso + c4 * 4) =
              *reinterpret_cast<const float4*>(TL + m * RH + c4 * 4);
      }
      FPC_LDS_BARRIER();   // the next tile's second chunk is transformed into V1 = this region
      continue;
    }

    // ---------------------------------------------------------------- phase 2: 1x1 over h (+ projection over x)
    const float4* const w2s = a.w2 + (size_t)cg * ((size_t)(a.k8_h + a.k8_x) / 2 + C::WPAD) * 64;   // steps of 16 channels
    auto ldb2 = [&](int s) { return fpc_ldg_su(w2s + (size_t)s * 64, lane16); };
    constexpr int KH = N / 16;
    int abase2[MB2];
#pragma unroll
    for (int mb = 0; mb < MB2; ++mb) abase2[mb] = (16 * (tg * MB2 + mb) + t16);
    auto gemm_step = [&](const float* rows, int rstride, int kcol, const float4& bv) {
#pragma unroll
      for (int mb = 0; mb < MB2; ++mb) {
        const float4 av = *reinterpret_cast<const float4*>(rows + abase2[mb] * rstride + kcol + 4 * kq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float af = j == 0 ? av.x : j == 1 ? av.y : j == 2 ? av.z : av.w;
          const float bf = j == 0 ? bv.x : j == 1 ? bv.y : j == 2 ? bv.z : bv.w;
          acc2[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc2[mb], 0, 0, 0);
        }
      }
    };
    // projection shortcut: the tile's centre pixels of x, up to 128 channels per pass, requested before the GEMM over h
    constexpr int XIT = PX * 32 / NT;
    float4 xst[XIT];
    auto load_x = [&](int pass) {
      const int kx4 = min(32, a.k8_x * 2 - pass * 32);   // float4 per pixel in this pass
#pragma unroll
      for (int i = 0; i < XIT; ++i) {
        const int e = tid_t + i * NT;
        const int m = e >> 5, c4 = e & 31;
        int y = ty * TH + (m >> 4), x = tx * TW + (m & 15);
        y = y < a.H ? y : a.H - 1;
        x = x < a.W ? x : a.W - 1;
        const bool ok = c4 < kx4;
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v v = *reinterpret_cast<const f4v*>(a.x + ((size_t)(b * a.H + y) * a.W + x) * a.csx + (ok ? pass * 128 + c4 * 4 : 0));
        xst[i] = make_float4(v.x, v.y, v.z, v.w);
      }
    };
    if (proj) load_x(0);
    {
      float4 cb[4];
      cb[0] = ldb2(0);
      cb[1] = ldb2(1);
      cb[2] = ldb2(2);
#pragma unroll
      for (int g = 0; g < KH; ++g) {
        cb[(g + 3) & 3] = ldb2(g + 3);
        __builtin_amdgcn_sched_barrier(0);
        gemm_step(TL, RH, 16 * g, cb[g & 3]);
      }
    }
    if (wg == wg_stamp) { FPC_STAMP(3) }
    if (proj) {
      const int npass = (a.k8_x + 15) >> 4;
      for (int pass = 0; pass < npass; ++pass) {
        FPC_LDS_BARRIER();   // h (or the previous pass's x) has been read by every wave
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
          const int e = tid_t + i * NT;
          *reinterpret_cast<float4*>(TL + (e >> 5) * RX + (e & 31) * 4) = xst[i];
        }
        FPC_LDS_BARRIER();
        if (pass + 1 < npass) load_x(pass + 1);
        const int steps = min(8, a.k8_x / 2 - pass * 8);   // 16-channel steps of this pass: 4 or 8
        const int s0 = KH + pass * 8;
        float4 cb[4];
        cb[0] = ldb2(s0);
        cb[1] = ldb2(s0 + 1);
        cb[2] = ldb2(s0 + 2);
        for (int g = 0; g < steps; g += 4) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            cb[(u + 3) & 3] = ldb2(s0 + g + u + 3);
            __builtin_amdgcn_sched_barrier(0);
            gemm_step(TL, RX, 16 * (g + u), cb[u & 3]);
          }
        }
      }
    }

    // ---------------------------------------------------------------- epilogue: output tile through LDS -> 16-byte stores
    if (wg == wg_stamp) { FPC_STAMP(4) }
    FPC_LDS_BARRIER();
    {
      const float bias = a.b2[16 * cg + t16];
#pragma unroll
      for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int px = 16 * (tg * MB2 + mb) + 4 * kq + r;
          TL[px * RH + 16 * cg + t16] = acc2[mb][r] + bias;
        }
    }
    FPC_LDS_BARRIER();
    {
      constexpr int C4 = N / 4, EIT = PX * C4 / NT;
#pragma unroll
      for (int i = 0; i < EIT; ++i) {
        const int e = tid_t + i * NT;
        const int m = e / C4, c4 = e - m * C4;
        const int y = ty * TH + (m >> 4), x = tx * TW + (m & 15);
        if (y < a.H && x < a.W) {
          float4 v = *reinterpret_cast<const float4*>(TL + m * RH + c4 * 4);
          v.x = v.x > 0.f ? v.x : 0.f;
          v.y = v.y > 0.f ? v.y : 0.f;
          v.z = v.z > 0.f ? v.z : 0.f;
          v.w = v.w > 0.f ? v.w : 0.f;
          *reinterpret_cast<float4*>(a.out + ((size_t)(b * a.H + y) * a.W + x) * a.cso + c4 * 4) = v;
        }
      }
    }
    if (wg == wg_stamp) { FPC_STAMP(5) }
    FPC_LDS_BARRIER();   // the next tile's second chunk is transformed into V1 = this region
  }  // persistent tile loop
}

}  // namespace fpc
