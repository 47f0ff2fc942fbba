// wblock36_mfma.h -- third generation of the Winograd ResNetBlock kernel: F(4x4, 3x3).
//
//     V = B^T d B   (6x6 input patch of every 4x4 output tile, per channel)        VALU, LDS -> LDS
//     M_xi = V_xi . U_xi   for the 36 patch positions xi, U = G g G^T (host, double) 36 GEMMs on MFMA
//     Y = A^T M A   (+ folded-BN bias, ReLU)  -> h                                  VALU, registers -> LDS
//     out = relu(conv1x1(h) + shortcut(x))                                         as wblock16_mfma.h
//
// 36 multiplications per 16 outputs and channel pair instead of 144: the 3x3 costs 4x fewer MFMAs than the direct
// convolution (F(2x2,3x3): 2.25x).  Interpolation points 0, +-1, +-2, inf; measured on He-scaled 128-channel layers
// against fp64: 6e-6 of the layer's scale (F(2x2,3x3): 4e-7; direct fp32: 2e-7) -- inside the 1e-4 bar with a factor
// of ten to spare per layer (experiments/harness/wino43_numerics.py).
//
// What shapes the kernel is a measurement (experiments/harness/mfma_f32_coissue.hip): on gfx950 the fp32 MFMA and the
// VALU exclude each other on a SIMD.  v_mfma_f32_16x16x4_f32 issues back to back at 32 cycles; one independent
// v_fma_f32 between two of them costs +15.5 cycles, every further one +4, with one wave or with two waves per SIMD
// alike -- a partner wave's VALU work does NOT hide under this wave's MFMAs.  LDS instructions do overlap (3
// ds_read_b32 or 2 ds_write_b32 per MFMA and wave are free).  So the time of a chunk is
//         32 x MFMAs  +  4 x VALU instructions  +  ~11 per switch from the matrix pipe to the VALU and back
// whatever the wave arrangement, and the levers are: fewer MFMAs (this transform), fewer VALU instructions, and VALU
// work in a few long bursts instead of dealt out between the MFMAs (generation 2 dealt it out: a switch per MFMA).
//
// Structure: ONE wave per SIMD (256 threads, up to 512 registers per lane).  A workgroup owns a tile of 16 Winograd
// tiles (TYT x TXT of 4x4 pixels) x ALL output channels; wave w owns 16 NB output channels for all 36 positions:
// 36 NB accumulator blocks of v_mfma_f32_16x16x4_f32 (NB = 2: 288 registers, N = 128; NB = 1: 144, N = 64).  As in
// generation 2 the output transform is then register arithmetic, and the input side is software-pipelined into the
// GEMM over chunks of 16 input channels (two V buffers, two halo buffers, the next tile's first chunks during this
// tile's last ones), with the transform of chunk c + 1 as ONE burst of 144 VALU instructions in the middle of chunk
// c's MFMAs, its 36 LDS reads dealt out before it and its 36 LDS writes after it.
//
// The MFMA's row index m (0..15) carries Winograd tile T(m) = 8 ((m >> 1) & 1) + 2 (m >> 2) + (m & 1): a lane's
// four accumulator rows 4 kq + r then hold two tiles of the tile's first half (r = 0, 1) and two of its second half
// (r = 2, 3), so the 1x1 phase can run per half (128 pixels: h for all 256 would not fit beside the next tile's
// pipelined input) with every lane at work in both halves.
#pragma once
#include <type_traits>
#include <utility>

#include "wblock16_mfma.h"

namespace fpc {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant in
// every trip whatever the optimiser's unrolling thresholds say (left to `#pragma unroll`, the 36-step chunk loop stayed
// a loop: ring slots and accumulators indexed at run time, i.e. in scratch memory)
template <int... I, class F>
__device__ __forceinline__ void fpc_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void fpc_static_for(F&& f) {
  fpc_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// Layout of the dust block (floats) a DUST instance reads through WBlockArgs::dust -- the 65th ("dustbin") channel of the
// detector's blocks, packed by fpc_api.hip (pack_w36_dust):
//   W2ROW[64]   conv2 weight of h[64] into outputs 0..63            (x folded-BN scale of the output)
//   W2COL[80]   conv2 weights of h[0..64] into output 64 (65 real)
//   WPCOL[256]  projection weights of x[0..cin) into output 64
//   B1, B2      folded biases of h[64] / out[64] (projection's bias included)
//   UIN64[36]   Winograd-domain filter  in 64 -> out 64
//   UIN[4][36][16]   ... in 64 -> out n (channel group, position, n & 15)
//   UOUT[nchunk][36][16]  ... in 16 c + k -> out 64
struct W36Dust {
  static constexpr int W2ROW = 0, W2COL = 64, WPCOL = 144, B1 = 400, B2 = 401, UIN64 = 404, UIN = 448, UOUT = 448 + 4 * 36 * 16;
  static constexpr int floats(int nchunk) { return UOUT + nchunk * 36 * 16; }
};

template <int NB_, int TYT_, int TXT_, bool DUST_ = false>
struct W36Cfg {
  static constexpr int NB = NB_, TYT = TYT_, TXT = TXT_;
  static constexpr bool DUST = DUST_;
  static_assert(!DUST || NB == 1, "the dust channel rides on the 64-channel instance");
  static_assert(TYT * TXT == 16 && (TYT == 4 || TYT == 2), "16 Winograd tiles per workgroup tile: 4 x 4 or 2 x 8");
  static_assert(NB == 1 || NB == 2, "a wave owns 16 or 32 output channels");
  static constexpr int NT = 256, NPOS = 36, KC = 16;
  static constexpr int TH = 4 * TYT, TW = 4 * TXT, HH = TH + 2, HW = TW + 2, NHALO = HH * HW;
  static constexpr int N = 64 * NB;
  static constexpr int HROW = 20;                                   // halo row: 16 channels + 4 floats of skew
  // The halo arrives by LDS-DMA (buffer_load_dwordx4 ... lds): a wave instruction writes 64 consecutive 16-byte slots,
  // so the buffer is addressed in slots -- five per pixel (four channel quads + the skew slot, which receives zeros) --
  // and thread t requests slots t, t + 256, ... : HIT per thread and chunk (7), no staging registers, no ds_write.
  static constexpr int HIT = (NHALO * 5 + NT - 1) / NT;
  static constexpr int HALO_FLOATS = HIT * NT * 4;
  static constexpr int V_FLOATS = NPOS * 16 * 16;
  // V0 | halo1 | halo0 | V1: at a tile's end V0 and halo1 hold the next tile's first chunks; halo0, V1 and the rest
  // of the LDS are free for h, the shortcut's x tile and the output tile of one half
  static constexpr int OFF_V0 = 0, OFF_H1 = V_FLOATS, OFF_H0 = OFF_H1 + HALO_FLOATS, OFF_V1 = OFF_H0 + HALO_FLOATS;
  static constexpr int RH = N + 8;                                  // h / output row (floats)
  static constexpr int RX = 128 + 8;                                // x staging row (up to 128 channels per pass)
  static constexpr int HPX = 128;                                   // pixels of one half
  static constexpr int T_FLOATS = HPX * (RH > RX ? RH : RX);
  static constexpr int BASE_FLOATS = (OFF_V1 + V_FLOATS) > (OFF_H0 + T_FLOATS) ? (OFF_V1 + V_FLOATS) : (OFF_H0 + T_FLOATS);
  // DUST: the dust block's first 448 floats as they are (W2ROW | W2COL | WPCOL | B1, B2 | UIN64: every constant the tile's
  // tail reads -- round 5: as scalar / vector GLOBAL loads in the tail each one was an L2 round trip in a serial chain),
  // UIN (2304), V of input channel 64 [36][16], M of output channel 64 [36][16], the halo of input channel 64 (one 16-byte
  // slot per pixel by LDS-DMA: HIT64 requests per thread)
  static constexpr int HIT64 = (NHALO + NT - 1) / NT;
  static constexpr int D_HEAD = BASE_FLOATS, D_COL = D_HEAD + 64, D_UIN = D_HEAD + 448, D_V64 = D_UIN + 2304, D_M64 = D_V64 + 576, D_H64 = D_M64 + 576;
  static constexpr int LDS_FLOATS = DUST ? D_H64 + HIT64 * NT * 4 : BASE_FLOATS;
  static constexpr int LDS_BYTES = LDS_FLOATS * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  // one ring step = TWO positions: 16 MFMAs (NB = 2: two positions x two channel blocks, four accumulators in turn) or 8
  // (NB = 1: two accumulators in turn -- a single dependent chain would run at 40 instead of 32 cycles per MFMA)
  static constexpr int STEPS = 18;
  // Ring depth in steps (8 NB registers each; must divide STEPS so that slot indices are compile-time constants across
  // chunks).  vmcnt retires in order: a fragment requested after the chunk's halo request waits for it, so the ring has
  // to hold the fragments of that whole latency (RING - 1 steps of 256 NB cycles).
  // (NB = 1: 9 steps = 2 k cycles of cover, like the 128-channel instance's.  Round 3 ran 18: 144 of the 256 ARCHITECTURAL
  // VGPRs, so the compiler parked ring slots in AGPRs -- 50 v_accvgpr moves per chunk in the MFMA stream; 9 measured 3 %
  // faster on layer1, and 6 the same as 9)
  static constexpr int RING = NB == 2 ? 6 : 9;
  static_assert(STEPS % RING == 0, "ring slots must line up across chunks");
  static constexpr int PAG = NB == 2 ? 30 : 36;                     // positions whose accumulators live in AGPRs (240 of 256; the other 48 registers in VGPRs)
  static constexpr int WPAD = 36;                                   // zero POSITIONS behind every channel group's stream (>= 2 RING)
  static constexpr int WPAD2 = 16;                                  // ... and steps behind the 1x1 streams (as generation 2)
};

// Eight v_mfma_f32_16x16x4_f32 as ONE asm statement: two accumulators alternate (a single dependent chain would run at
// 40 instead of 32 cycles per MFMA), each pinned to the AGPR half (AG) or the VGPR half of the register file.
//  * 288 accumulators do not fit the 256 AGPRs; left to the compiler (the builtin), the ones that do not fit travel
//    between the two halves around every MFMA (220 v_accvgpr moves per chunk, each a VALU instruction, and the VALU
//    excludes the MFMA, see above) and the chains of one accumulator are issued back to back.
//  * With one wave per SIMD ANY instruction between two MFMAs delays the second by ~6 cycles (co-issue harness: one
//    ds_read per MFMA 38.5 cycles instead of 32, three 39.6): the non-MFMA instructions of a step belong in ONE gap, in
//    front of its eight MFMAs.  As separate asm statements the compiler put its own s_nop / s_waitcnt / address
//    arithmetic between them: a step took 295 cycles instead of 256 + one gap.
//  * The compiler does not know what is inside the asm, so it cannot place the wait states an MFMA result needs before
//    a VALU instruction reads it (11 after this 8-pass MFMA) -- and it DOES read accumulators behind the asm: copies
//    between the register files where its allocator splits a live range (seen right behind a block: v_accvgpr_read of
//    the two accumulators just written -- stale values, wrong results).  Eleven s_nop wait states at the end of every
//    block fix that at 5 % of the kernel's time.  Instead the block's LAST MFMA is the compiler's own builtin: it then
//    knows the hazard of the one result that can still be in flight when the block ends and pays for it only where it
//    really puts a reader there (every other MFMA of the block was issued at least 32 cycles -- one MFMA -- earlier and
//    has all but written its result when the last one issues: 40 cycles from issue to result, 32 to the next issue
//    plus the builtin's own issue slots -- and two s_nop cycles at the end of the asm for margin).
// LAST_IN_ASM (round 5): the block's last MFMA inside the asm as well, no builtin.  For the 128-channel instance this is
// 3 % in the stand-alone harness (layer2.1-like 0.2317 -> 0.2253 ms, layer_out.0-like 0.469 -> 0.452) and 1.3 % in the
// library: with the builtin the compiler gives some results a different register than their source accumulator and
// carries them home across the unrolled chunk pair -- 26 v_accvgpr moves and 36 s_nop per chunk, VALU slots in the MFMA
// stream.  Safe ONLY where nothing can read an accumulator right behind a block: tests/test_static_isa.py scans the
// shipped code object for exactly that (every instance, straight line and back edges) and is the gate for this switch.
// NOT for wblock36p_kernel (accumulators in VGPRs: the compiler moves them around between blocks -- two of the harness's
// seven cases computed stale values with it) nor worth it for the 64-channel one-wave instances (all 144 accumulators in
// AGPRs, no moves to save: same time).
template <bool AG0, bool AG1, bool LAST_IN_ASM = false>
__device__ __forceinline__ void fpc_mfma_step(f32x4& c0, f32x4& c1, const f32x4& a0, const f32x4& a1, const f32x4& b0, const f32x4& b1) {
#define FPC_MFMA7                                                                                                      \
  "v_mfma_f32_16x16x4_f32 %0, %2, %10, %0\n\tv_mfma_f32_16x16x4_f32 %1, %6, %14, %1\n\t"                                \
  "v_mfma_f32_16x16x4_f32 %0, %3, %11, %0\n\tv_mfma_f32_16x16x4_f32 %1, %7, %15, %1\n\t"                                \
  "v_mfma_f32_16x16x4_f32 %0, %4, %12, %0\n\tv_mfma_f32_16x16x4_f32 %1, %8, %16, %1\n\t"                                \
  "v_mfma_f32_16x16x4_f32 %0, %5, %13, %0\n\t"
#define FPC_MFMA8 FPC_MFMA7 "v_mfma_f32_16x16x4_f32 %1, %9, %17, %1"
#define FPC_MFMA8_IN                                                                                                   \
  "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]), "v"(b0[0]), "v"(b0[1]), \
      "v"(b0[2]), "v"(b0[3]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3])
  if constexpr (LAST_IN_ASM) {
    if constexpr (AG0 && AG1) asm volatile(FPC_MFMA8 : "+a"(c0), "+a"(c1) : FPC_MFMA8_IN);
    else if constexpr (AG0) asm volatile(FPC_MFMA8 : "+a"(c0), "+v"(c1) : FPC_MFMA8_IN);
    else if constexpr (AG1) asm volatile(FPC_MFMA8 : "+v"(c0), "+a"(c1) : FPC_MFMA8_IN);
    else asm volatile(FPC_MFMA8 : "+v"(c0), "+v"(c1) : FPC_MFMA8_IN);
  } else {
    if constexpr (AG0 && AG1) asm volatile(FPC_MFMA7 "s_nop 1" : "+a"(c0), "+a"(c1) : FPC_MFMA8_IN);
    else if constexpr (AG0) asm volatile(FPC_MFMA7 "s_nop 1" : "+a"(c0), "+v"(c1) : FPC_MFMA8_IN);
    else if constexpr (AG1) asm volatile(FPC_MFMA7 "s_nop 1" : "+v"(c0), "+a"(c1) : FPC_MFMA8_IN);
    else asm volatile(FPC_MFMA7 "s_nop 1" : "+v"(c0), "+v"(c1) : FPC_MFMA8_IN);
    // the LAST MFMA of the block is the compiler's own builtin (see the header comment: it then knows the wait states)
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[3], b1[3], c1, 0, 0, 0);
  }
#undef FPC_MFMA7
#undef FPC_MFMA8
#undef FPC_MFMA8_IN
}

// Byte offsets of this kernel's buffer accesses are SIGNED 32-bit sums `per-lane part + uniform part`, added with
// saturation; a lane that must not touch memory carries this marker as its per-lane part: marker + anything >= 0 stays
// >= marker, which is the (or above the) range of every descriptor here.  Tensors must be smaller than this (the host
// checks: fpc_api.hip, w36_fits).
#define W36_MARKER 0x7fffff00

// Sixteen MFMAs as one statement -- two positions x two channel blocks, four accumulators in turn -- for the 128-channel
// instance: half as many gaps per MFMA as blocks of eight.  All sixteen in the asm (LAST_IN_ASM above: round 5).
template <bool AG>
__device__ __forceinline__ void fpc_mfma_step16(f32x4& c00, f32x4& c01, f32x4& c10, f32x4& c11, const f32x4& a0, const f32x4& a1,
                                                const f32x4& b00, const f32x4& b01, const f32x4& b10, const f32x4& b11) {
  // operands: 0-3 accumulators; 4-7 a0; 8-11 a1; 12-15 b00; 16-19 b01; 20-23 b10; 24-27 b11
#define FPC_MFMA16                                                                                                      \
  "v_mfma_f32_16x16x4_f32 %0, %4, %12, %0\n\tv_mfma_f32_16x16x4_f32 %1, %4, %16, %1\n\t"                               \
  "v_mfma_f32_16x16x4_f32 %2, %8, %20, %2\n\tv_mfma_f32_16x16x4_f32 %3, %8, %24, %3\n\t"                               \
  "v_mfma_f32_16x16x4_f32 %0, %5, %13, %0\n\tv_mfma_f32_16x16x4_f32 %1, %5, %17, %1\n\t"                               \
  "v_mfma_f32_16x16x4_f32 %2, %9, %21, %2\n\tv_mfma_f32_16x16x4_f32 %3, %9, %25, %3\n\t"                               \
  "v_mfma_f32_16x16x4_f32 %0, %6, %14, %0\n\tv_mfma_f32_16x16x4_f32 %1, %6, %18, %1\n\t"                               \
  "v_mfma_f32_16x16x4_f32 %2, %10, %22, %2\n\tv_mfma_f32_16x16x4_f32 %3, %10, %26, %3\n\t"                             \
  "v_mfma_f32_16x16x4_f32 %0, %7, %15, %0\n\tv_mfma_f32_16x16x4_f32 %1, %7, %19, %1\n\t"                               \
  "v_mfma_f32_16x16x4_f32 %2, %11, %23, %2\n\tv_mfma_f32_16x16x4_f32 %3, %11, %27, %3"
#define FPC_MFMA16_IN                                                                                                   \
  "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]), "v"(b00[0]), "v"(b00[1]), \
      "v"(b00[2]), "v"(b00[3]), "v"(b01[0]), "v"(b01[1]), "v"(b01[2]), "v"(b01[3]), "v"(b10[0]), "v"(b10[1]), "v"(b10[2]),  \
      "v"(b10[3]), "v"(b11[0]), "v"(b11[1]), "v"(b11[2]), "v"(b11[3])
  if constexpr (AG) asm volatile(FPC_MFMA16 : "+a"(c00), "+a"(c01), "+a"(c10), "+a"(c11) : FPC_MFMA16_IN);
  else asm volatile(FPC_MFMA16 : "+v"(c00), "+v"(c01), "+v"(c10), "+v"(c11) : FPC_MFMA16_IN);
#undef FPC_MFMA16
#undef FPC_MFMA16_IN
}

template <int NB, int TYT, int TXT, bool DUST>
__device__ __forceinline__ void wblock36_body(const WBlockArgs& a) {
  using C = W36Cfg<NB, TYT, TXT, DUST>;
  constexpr int NT = C::NT, TH = C::TH, TW = C::TW, HW = C::HW, HH = C::HH, HROW = C::HROW, HIT = C::HIT;
  constexpr int N = C::N, RH = C::RH, RX = C::RX, RING = C::RING, STEPS = C::STEPS, HPX = C::HPX;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x4* const lds4 = reinterpret_cast<f32x4*>(lds);
  constexpr int TL4 = C::OFF_H0 / 4;        // h, x staging, output tile of one half (float4 units)

  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = a.tiles_x * a.tiles_y;
  const int nchunk = a.nchunk;              // Cin / 16, even, >= 4
  // second half of a 256-wide conv-only layer (gridDim.y == 2)
  const float4* const w1h = a.w1 + (size_t)blockIdx.y * (a.ysplit_floats / 4);
  const float* const b1h = a.b1 + (size_t)blockIdx.y * a.ysplit_floats;

  // tile walk (persistent, XCD-aware): as wblock16_kernel
  const bool xcd_order = a.xcd_order && (gridDim.x & 7) == 0;
  const int wg_step = xcd_order ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int xchunk = (a.total + 7) >> 3;
  const int wg_first = xcd_order ? (int)(blockIdx.x & 7) * xchunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int wg_end = xcd_order ? min(a.total, ((int)(blockIdx.x & 7) + 1) * xchunk) : a.total;
  if (wg_first >= wg_end) return;

  // ---------------------------------------------------------------- input side: L (global -> registers), S (-> halo), T (halo -> V)
  // As generation 2: buffer loads with the hardware's bounds check (out-of-frame pixels, padding slots and tiles behind
  // the workgroup's last one get an offset outside the descriptor's range and come back as zeros).
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  // Slot q = t + 256 i of this thread: halo pixel q / 5 = (hy, hx), channel quad q % 5 (4 = the skew slot).  Its byte
  // offset from the tile's FIRST PIXEL is a constant of the thread; whether it lies inside the frame depends on the tile.
  // okoff[i] = that offset, or a huge positive number where the slot must read as zero (outside the frame, skew or
  // padding slot, a tile behind the workgroup's last); the request adds the tile's (uniform, non-negative) base with
  // SIGNED SATURATION, which leaves the huge ones outside the descriptor's range (zeros from the hardware's bounds
  // check): one VALU instruction per slot and chunk (the first version recomputed rows, columns and flags per chunk: 13
  // instructions per element in the MFMA stream, and the VALU excludes the MFMA).
  int okoff[HIT];
  int okoff64[DUST ? C::HIT64 : 1];            // DUST: the same for input channel 64, one slot per halo pixel
  struct TilePos { int b, ty, tx, live; };
  auto tile_pos = [&](int wg) {                          // uniform (SALU) arithmetic: once per tile, not per chunk
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
  auto halo_base = [&](const TilePos& q) {               // byte offset of the tile's first pixel, channel 0
    return __builtin_amdgcn_readfirstlane(((q.b * a.H + q.ty * TH) * a.W + q.tx * TW) * a.csx * 4);
  };
  auto halo_tile = [&](const TilePos& q) {               // okoff[] for that tile
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int iy0 = q.ty * TH - 1, ix0 = q.tx * TW - 1;
    const int hlim = q.live ? a.H : 0;
#pragma unroll
    for (int i = 0; i < HIT; ++i) {
      const int e = tl + i * NT;
      const int pix = (e * 13108) >> 16, c4 = e - pix * 5;                    // e / 5 (e < 1792: 13108 / 65536 = 0.2 + 1.2e-5)
      const int hy = HW == 18 ? (pix * 3641) >> 16 : (pix * 1928) >> 16;     // pix / 18 (pix < 469), pix / 34 (pix < 441)
      const int hx = pix - hy * HW;
      const bool ok = ((unsigned)(iy0 + hy) < (unsigned)hlim) & ((unsigned)(ix0 + hx) < (unsigned)a.W) & (hy < HH) & (c4 < 4);
      const int off = (((hy - 1) * a.W + (hx - 1)) * a.csx + c4 * 4) * 4;
      okoff[i] = ok ? off : W36_MARKER;
    }
    if constexpr (DUST) {
#pragma unroll
      for (int i = 0; i < C::HIT64; ++i) {
        const int pix = tl + i * NT;
        const int hy = HW == 18 ? (pix * 3641) >> 16 : (pix * 1928) >> 16;
        const int hx = pix - hy * HW;
        const bool ok = ((unsigned)(iy0 + hy) < (unsigned)hlim) & ((unsigned)(ix0 + hx) < (unsigned)a.W) & (hy < HH) & (a.dust_in != 0);
        okoff64[i] = ok ? (((hy - 1) * a.W + (hx - 1)) * a.csx + 64) * 4 : W36_MARKER;
      }
    }
  };
  auto load_halo = [&](int base, int hoff) {             // base: halo_base of the tile + 64 bytes per chunk; hoff: the halo buffer (floats)
#pragma unroll
    for (int i = 0; i < HIT; ++i) {
      int voff;
      asm("v_add_i32 %0, %1, %2 clamp" : "=v"(voff) : "v"(okoff[i]), "s"(base));
      // (the instruction's LDS address is M0 + 16 lane: M0 = this wave's 64 slots of request i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(lds + hoff + (i * NT + wave * 64) * 4), 16, voff, 0, 0, 0);
    }
  };
  auto load_halo64 = [&](int base) {                     // DUST: channels 64..67 of the tile's halo pixels -> D_H64 [pixel][4]
    if constexpr (DUST) {
#pragma unroll
      for (int i = 0; i < C::HIT64; ++i) {
        int voff;
        asm("v_add_i32 %0, %1, %2 clamp" : "=v"(voff) : "v"(okoff64[i]), "s"(base));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(lds + C::D_H64 + (i * NT + wave * 64) * 4), 16, voff, 0, 0, 0);
      }
    }
  };
  // the transform item of this thread: Winograd tile wt (0..15), channel ch (0..15) of the chunk; V row m = Tinv(wt)
  const int wt_t = tid >> 4, ch_t = tid & 15;
  const int m_t = ((wt_t & 7) >> 1) * 4 + (wt_t >> 3) * 2 + (wt_t & 1);
  const int trd = ((4 * (wt_t / TXT)) * HW + 4 * (wt_t % TXT)) * HROW + ch_t;                          // halo read base
  const int twr = m_t * 16 + ((((ch_t >> 2) ^ (2 * ((m_t >> 3) & 1))) << 2) | (ch_t & 3));            // V write base (swizzled)
  // The 6x6 patch lives in register PAIRS (v_pk_fma_f32 / v_pk_add_f32 do two fp32 operations per VALU slot, and on this GPU
  // every VALU slot is taken from the MFMAs): pairs along j for the column pass, re-paired along i (a 2x2 transpose of
  // pairs is two v_pk_mov_b32) for the row pass -- 36 + 18 + 36 instructions where scalar code had 144.  Same operations
  // in the same order on every element: bit-identical results.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 tp[6][3];      // (d[i][2 jj], d[i][2 jj + 1]) as read from the halo; after the burst tp[2 ii + (j & 1)][j >> 1] = (V[2 ii][j], V[2 ii + 1][j])
  auto t_read = [&](int rd, int i, int j) { tp[i][j >> 1][j & 1] = lds[rd + (i * HW + j) * HROW]; };
  // B^T x for the six values x0..x5 (in place): the F(4x4,3x3) input transform along one axis, 12 operations
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
        const f32x2 ra = tp[2 * ii][jj], rb = tp[2 * ii + 1][jj];      // the 2x2 block in place: no second array alive
        tp[2 * ii][jj] = __builtin_shufflevector(ra, rb, 0, 2);
        tp[2 * ii + 1][jj] = __builtin_shufflevector(ra, rb, 1, 3);
      }
#pragma unroll
    for (int ii = 0; ii < 3; ++ii) bt6(tp[2 * ii][0], tp[2 * ii + 1][0], tp[2 * ii][1], tp[2 * ii + 1][1], tp[2 * ii][2], tp[2 * ii + 1][2]);
  };
  auto t_write = [&](int wr, int i, int j) { lds[wr + (i * 6 + j) * 256] = tp[2 * (i >> 1) + (j & 1)][j >> 1][i & 1]; };
  auto transform_all = [&](int hoff, int voff) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) t_read(trd + hoff, i, j);
    t_burst();
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) t_write(twr + voff, i, j);
  };

  // ---------------------------------------------------------------- operands of the GEMMs
  // A: V[pos][m][16 ch] -- lane (row m = l & 15, k quarter kq = l >> 4) reads one float4
  const int aoff4 = (lane & 15) * 4 + ((lane >> 4) ^ (2 * ((lane & 15) >> 3)));      // in float4 units
  // B: the fragment streams of this wave's NB channel groups, [chunk][pos][64 lanes] float4 each
  const unsigned gstride = ((unsigned)nchunk * 36u + (unsigned)C::WPAD) * 1024u;        // bytes per channel group
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(w1h), 0, (int)(4u * (unsigned)NB * gstride), 0x00020000);
  const unsigned wlane = (unsigned)(wave * NB) * gstride + lane16;
  const int gs_s = __builtin_amdgcn_readfirstlane((int)gstride);
  struct BF { f32x4 v[2][NB]; };      // [position of the step][channel block]
  auto ldb = [&](int s) {      // ring step s (counted from the tile's first chunk): positions 2 s, 2 s + 1
    BF r;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        r.v[q][nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)wlane, (2 * s + q) * 1024 + nb * gs_s, 0));
    return r;
  };

  // ---------------------------------------------------------------- DUST: the 65th output (and input) channel beside the 64 on the MFMAs
  // Output channel 64 of the 3x3 is a dot product per (position, Winograd tile) and chunk IN THE WINOGRAD DOMAIN -- the V the
  // MFMAs read is in LDS anyway (v_fma at the fp32 MFMA's rate, where a fifth 16-channel group would issue 36 more MFMAs
  // per wave and chunk for one real channel in sixteen).  Wave w takes positions w, w + 4, ..., w + 32; its lane l reads
  // float4 l of a position's V block (row m = l >> 2, quad slot l & 3: the wave reads the block's 1 KB in order, no bank
  // conflict -- the first mapping, a thread per (position, two rows), hit two bank groups with 64 lanes and cost 28 k cycles
  // a tile) and keeps one partial sum per position over ITS four channels; the four lanes of a row are added up at the
  // tile's end.  A row's quads lie XOR-swizzled by 2 (m >> 3): the filter quad is LOADED accordingly.
  [[maybe_unused]] int d_va = 0, d_ua = 0;
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t drsrc =
      DUST ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dust + W36Dust::UOUT), 0, nchunk * 36 * 16 * 4, 0x00020000) : xrsrc;
  [[maybe_unused]] f32x4 du[9];
  [[maybe_unused]] float dacc[9];
  if constexpr (DUST) {
    d_va = wave * 64 + lane;                                                    // float4 index of (position w, this lane) in a V buffer
    d_ua = wave * 64 + 16 * ((lane & 3) ^ (2 * (lane >> 5)));                   // byte offset of its filter quad in a chunk's [36][16]
    // once per workgroup: W2COL | WPCOL and UIN into LDS (read back as broadcasts / MFMA operands at every tile's end)
    for (int i = tid; i < 448 / 4; i += NT) lds4[C::D_HEAD / 4 + i] = reinterpret_cast<const f32x4*>(a.dust)[i];
    for (int i = tid; i < 2304 / 4; i += NT) lds4[C::D_UIN / 4 + i] = reinterpret_cast<const f32x4*>(a.dust + W36Dust::UIN)[i];
  }

  // ---------------------------------------------------------------- pipeline fill for the first tile
  TilePos pos_cur = tile_pos(wg_first);
  int base_cur = halo_base(pos_cur);
  halo_tile(pos_cur);
  load_halo64(base_cur);
  load_halo(base_cur, C::OFF_H0);
  load_halo(base_cur + 64, C::OFF_H1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HIT) : "memory");      // chunk 0 has landed (this wave's part of it)
  FPC_LDS_BARRIER();
  transform_all(C::OFF_H0, C::OFF_V0);
  FPC_LDS_BARRIER();
  // state at the top of iteration c of a tile: V[c & 1] = chunk c transformed; halo[(c + 1) & 1] = chunk c + 1 requested
  // (landed, as far as this wave's own requests go, by the barrier that ended iteration c - 1)

  // the tail's per-thread constants: output float4 e = tid + 256 i of a half lies at pixel m0 + PPI i, channel quad c4
  constexpr int C4 = N / 4, EIT = HPX * C4 / NT, PPI = NT / C4;       // float4 per pixel; float4 per thread (16 / 8); pixels per i (8 / 16)
  constexpr int CPR = TW / PPI;                                      // column steps per pixel row (1, 2 or 4)
  // (range = the marker below: `marker + uniform offset`, saturated, must be out of range for EVERY uniform offset >= 0.
  // With a range above the marker, the masked columns of the first tile of the first frame -- uniform offset 0 -- were
  // in range and wrote 2 GB behind the tensor.)
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + blockIdx.y * N, 0, W36_MARKER, 0x00020000);

  const int g2s = __builtin_amdgcn_readfirstlane(((a.k8_h + a.k8_x) / 2 + C::WPAD2) * 64);     // float4 per channel group of the 1x1 streams
  const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.w2), 0, a.conv_only ? 0 : 4 * NB * g2s * 16, 0x00020000);
  const unsigned w2lane = (unsigned)(wave * NB * g2s) * 16u + lane16;

  const int wg_stamp = wg_first + 2 * wg_step < wg_end ? wg_first + 2 * wg_step : wg_first;
  for (int wg = wg_first; wg < wg_end; wg += wg_step) {
    const int b = pos_cur.b, ty = pos_cur.ty, tx = pos_cur.tx;
    const TilePos pos_next = tile_pos(wg + wg_step);
    const int base_next = halo_base(pos_next);
    if (wg == wg_stamp) { FPC_STAMP(0) FPC_RSTAMP(6) }

    f32x4 acc[36][NB];
    {
      // (a zero the optimiser cannot see through: as the constant 0.f, every accumulator's initial tuple was a loop
      // invariant of the persistent tile loop -- hoisted in front of it, 48 + 64 registers, and spilled)
      float zf = 0.f;
      asm volatile("" : "+v"(zf));
#pragma unroll
      for (int p = 0; p < 36; ++p)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[p][nb] = f32x4{zf, zf, zf, zf};
    }

    BF bq[RING];
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) bq[i] = ldb(i);
    if constexpr (DUST) {
      float zf = 0.f;
      asm volatile("" : "+v"(zf));
#pragma unroll
      for (int i = 0; i < 9; ++i) dacc[i] = zf;
    }

    // ---------------------------------------------------------------- phase 1: 36 GEMMs per chunk, input side of the next chunks in between
    auto chunk_body = [&](auto PAR, const int c) {
      constexpr int par = decltype(PAR)::value;
      constexpr int VB_OFF = par ? C::OFF_V1 : C::OFF_V0, VN_OFF = par ? C::OFF_V0 : C::OFF_V1;
      constexpr int HS_OFF = par ? C::OFF_H1 : C::OFF_H0;   // halo[c & 1]: receives chunk c + 2
      constexpr int HN_OFF = par ? C::OFF_H0 : C::OFF_H1;   // halo[(c + 1) & 1]: chunk c + 1, transformed now
      const int c3 = c + 2;                                 // the chunk requested now: c + 2 of this tile or c + 2 - nchunk of the next one
      const bool nxt = c3 >= nchunk;
      const int l_base = (nxt ? base_next : base_cur) + (nxt ? c3 - nchunk : c3) * 64;
      int ao = aoff4 + VB_OFF / 4, trd_c = trd + HN_OFF, twr_c = twr + VN_OFF;
      asm volatile("" : "+v"(ao), "+v"(trd_c), "+v"(twr_c));
      [[maybe_unused]] int dva = d_va + VB_OFF / 4;      // DUST: this lane's float4 of position w, this chunk's buffer
      [[maybe_unused]] f32x4 dv[3];
      if constexpr (DUST) asm volatile("" : "+v"(dva));
      // A operand: two register sets, the next step's read while this step's MFMAs run
      constexpr int AQ = 2;                     // positions per step
      f32x4 ac[2][AQ];
#pragma unroll
      for (int q = 0; q < AQ; ++q) ac[0][q] = lds4[ao + q * 64];
#ifdef FPC_DIAG_STEPS
      // (per-step stamps: their own switch of the diagnostic build, and BRANCH-FREE -- as `if (a.stamps) s_memtime` every
      // stamp split the chunk loop's basic block, and every block join costs an `s_waitcnt vmcnt(0)`: round 5 found the
      // diagnostic build's chunk 60 % slower than the product's, and its first two steps "taking 6 k cycles")
      unsigned long long tq[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define FPC_TQ(i) asm volatile("s_memtime %0" : "=s"(tq[i]));
#else
#define FPC_TQ(i)
#endif
      fpc_static_for<STEPS>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        if constexpr (s % 2 == 0) { FPC_TQ(s / 2) }
        // ---- the step's gap: everything that is not an MFMA, in front of the eight MFMAs
        bq[(s + RING - 1) % RING] = ldb(c * STEPS + s + RING - 1);
        if (s + 1 < STEPS) {
#pragma unroll
          for (int q = 0; q < AQ; ++q) ac[(s + 1) & 1][q] = lds4[ao + ((s + 1) * AQ + q) * 64];
        }
        // input side, dealt over the steps in `slots` (36 per chunk): LDS instructions ride in the gaps, VALU work comes
        // in two bursts (the halo request, the transform)
        constexpr int SPS = 36 / STEPS;   // slots per step
#pragma unroll
        for (int u = 0; u < SPS; ++u) {
          const int slot = s * SPS + u;
          if (slot == 1) {
            if (c3 == nchunk) halo_tile(pos_next);      // (uniform; no memory operation inside, so the join merges identical counter states)
            load_halo(l_base, HS_OFF);
          } else if (slot >= 2 && slot < 14) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { const int e = (slot - 2) * 3 + k; t_read(trd_c, e / 6, e % 6); }
          } else if (slot == 14) {
            t_burst();
          } else if (slot >= 15 && slot < 33) {
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int e = (slot - 15) * 2 + k; t_write(twr_c, e / 6, e % 6); }
          }
          if constexpr (DUST) {
            // output channel 64: the chunk's nine filter quads are requested early (L2, most of a chunk of cover); the V
            // quads are read three positions at a time and multiplied two slots later
            auto dread = [&](int i0) {
#pragma unroll
              for (int k = 0; k < 3; ++k) dv[k] = lds4[dva + (i0 + k) * 256];
            };
            auto dfma = [&](int i0) {
#pragma unroll
              for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) dacc[i0 + k] = __builtin_fmaf(dv[k][e], du[i0 + k][e], dacc[i0 + k]);
            };
            if (slot == 3) {
#pragma unroll
              for (int i = 0; i < 9; ++i)
                du[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drsrc, d_ua + i * 256, c * (36 * 16 * 4), 0));
            } else if (slot == 23) {
              dread(0);
            } else if (slot == 25) {
              dfma(0);
              dread(3);
            } else if (slot == 27) {
              dfma(3);
              dread(6);
            } else if (slot == 29) {
              dfma(6);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- the step's MFMAs
        {
          const BF& bv = bq[s % RING];
          constexpr int p0 = 2 * s, p1 = 2 * s + 1;
          if constexpr (NB == 2)
            fpc_mfma_step16<(p0 < C::PAG)>(acc[p0][0], acc[p0][1], acc[p1][0], acc[p1][1], ac[s & 1][0], ac[s & 1][1], bv.v[0][0], bv.v[0][1],
                                           bv.v[1][0], bv.v[1][NB - 1]);
          else
            fpc_mfma_step<true, true>(acc[p0][0], acc[p1][0], ac[s & 1][0], ac[s & 1][1], bv.v[0][0], bv.v[1][0]);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      FPC_TQ(9)
      // this wave's halo requests of the chunk have landed in LDS: everything but the ring's newest fragments is complete
      // (vmcnt counts in order; the ring waits above already imply it -- stated for the hardware, free)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB * (RING - 1)) : "memory");
      FPC_LDS_BARRIER();
      FPC_TQ(10)
#ifdef FPC_DIAG_STEPS
      if (a.stamps && wg == wg_stamp && c == 2 && (threadIdx.x & 63) == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        unsigned long long* q = a.stamps + 1024 * 8 + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
        for (int i = 0; i < 11; ++i) q[i] = tq[i];
      }
#endif
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): once per tile (see wblock16_kernel)
    {   // (do-while: nchunk >= 4.  With a loop that may run zero times the ring's first fragments were also live on the
        // path around it -- the compiler spilled all of them behind a vmcnt(0) and reloaded them after the loop.)
      int c = 0;
      do {
        chunk_body(std::integral_constant<int, 0>{}, c);
        chunk_body(std::integral_constant<int, 1>{}, c + 1);
        c += 2;
      } while (c < nchunk);
    }
    if (wg == wg_stamp) { FPC_STAMP(1) }
    // (the last MFMAs' results are read by VALU instructions below: a wait the compiler manages for its own MFMAs and
    // cannot see through the asm -- 8 passes need at most 18 wait states)
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");

    if constexpr (DUST) {
      // (every per-thread index below comes from a copy of the thread id the optimiser cannot see through: computed from
      // threadIdx.x they are invariants of the persistent tile loop, hoisted in front of it -- a hundred registers -- and
      // spilled: the first build of this instance had 101 scratch stores in its prologue)
      int tid_d = tid;
      asm volatile("" : "+v"(tid_d));
      const int lane_d = tid_d & 63;
      if (a.dust_in) {
        // INPUT channel 64 (detector.layer.1): its halo arrived at the tile's start (D_H64, 16 bytes per pixel); threads
        // 0..15 transform the 6x6 patch of one Winograd tile each into V64 [position][row m] ...
        if (tid_d < 16) {
          const int rd = C::D_H64 + ((4 * (tid_d / TXT)) * HW + 4 * (tid_d % TXT)) * 4;
#pragma unroll
          for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) tp[i][j >> 1][j & 1] = lds[rd + (i * HW + j) * 4];
          t_burst();
          const int mrow = ((tid_d & 7) >> 1) * 4 + (tid_d >> 3) * 2 + (tid_d & 1);
#pragma unroll
          for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) lds[C::D_V64 + (i * 6 + j) * 16 + mrow] = tp[2 * (i >> 1) + (j & 1)][j >> 1][i & 1];
        }
        FPC_LDS_BARRIER();
        // ... and every wave adds its 16 output channels' share: ONE MFMA per position, K = 4 of which k = 0 is real (the
        // A operand of lanes 16..63 is zero, so their B operand only has to be finite)
        {
          // (all 72 operands first -- the fragment ring's registers are free here --, then the 36 MFMAs back to back: as
          // read, wait, multiply per position the loop was 36 LDS round trips in a row)
          const int va = C::D_V64 + (lane_d & 15), ub = C::D_UIN + wave * 576 + (lane_d & 15);
          float av[36], bv[36];
#pragma unroll
          for (int p = 0; p < 36; ++p) {
            av[p] = lds[va + p * 16];
            bv[p] = lds[ub + p * 16];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int p = 0; p < 36; ++p) {
            const float x = lane_d < 16 ? av[p] : 0.f;
            acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, bv[p], acc[p][0], 0, 0, 0);
          }
        }
      }
      {
        // the four lanes of a V row add their partial sums up; lane 0 of the row adds input channel 64's share (in 64 -> out 64)
        // and writes M64[position][m]
        const int m_d = lane_d >> 2;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          float v = dacc[i];
          v += __shfl_xor(v, 1);
          v += __shfl_xor(v, 2);
          const int pos = wave + 4 * i;
          if (a.dust_in) v = __builtin_fmaf(lds[C::D_V64 + pos * 16 + m_d], lds[C::D_HEAD + W36Dust::UIN64 + pos], v);
          if ((lane_d & 3) == 0) lds[C::D_M64 + pos * 16 + m_d] = v;
        }
      }
      FPC_LDS_BARRIER();
      load_halo64(base_next);            // D_H64 has been read: the next tile's input channel 64 (okoff64 is the next tile's by now)
      if (tid_d < 16) {
        // Y = A^T M A + bias, ReLU for output channel 64 of Winograd tile T(m), m = tid
        float mm[36];
#pragma unroll
        for (int p = 0; p < 36; ++p) mm[p] = lds[C::D_M64 + p * 16 + tid_d];
        const float bias1 = lds[C::D_HEAD + W36Dust::B1];
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
        const int T = 8 * ((tid_d >> 1) & 1) + 2 * (tid_d >> 2) + (tid_d & 1), th = T & 7;
        const int hb = C::OFF_H0 + ((4 * (th / TXT)) * TW + 4 * (th % TXT)) * RH + 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float m0 = tt[i][0], m1 = tt[i][1], m2 = tt[i][2], m3 = tt[i][3], m4 = tt[i][4], m5 = tt[i][5];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          float yv[4];
          yv[0] = m0 + s12 + s34 + bias1;
          yv[1] = __builtin_fmaf(2.f, d34, d12) + bias1;
          yv[2] = __builtin_fmaf(4.f, s34, s12) + bias1;
          yv[3] = __builtin_fmaf(8.f, d34, d12) + m5 + bias1;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            // a first-half tile goes to h at once; a second-half tile's values wait in D_V64 (read, by now, for this tile;
            // registers held them at first -- and were spilled to scratch across the first half's GEMMs)
            const float v = yv[jj] > 0.f ? yv[jj] : 0.f;
            lds[T < 8 ? hb + (i * TW + jj) * RH : C::D_V64 + tid_d * 16 + i * 4 + jj] = v;
          }
        }
      }
    }

    // ---------------------------------------------------------------- output transform in registers; then the two halves of the tile
    int tid_t = tid;
    asm volatile("" : "+v"(tid_t));
    const int lane_t = tid_t & 63, n16 = lane_t & 15, kq = lane_t >> 4;
    const bool proj = a.k8_x > 0;
    // Y = A^T M A + bias, ReLU: the first half's two tiles of every lane go to LDS (h) at once, the second half's wait in
    // registers (64) while the first half runs its 1x1 -- so that the 288 accumulators are dead before phase 2 starts
    float yh1[NB][2][16];
    // h position of (tile-in-half th, pixel (0, 0), this lane's channel of block nb = 0): + (i TW + j) RH per pixel, + 16 per block
    int hw_base[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int th = 2 * kq + rr;
      hw_base[rr] = C::OFF_H0 + ((4 * (th / TXT)) * TW + 4 * (th % TXT)) * RH + 16 * (wave * NB) + n16;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const float bias1 = b1h[16 * (wave * NB + nb) + n16];
      if constexpr (NB == 1) {
        // (measured: the pair form is 3 % faster on the 64-channel layers and 5 % SLOWER on the 128-channel ones, whose
        // register file is full -- those keep the scalar form)
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        __builtin_amdgcn_sched_barrier(0);          // one (block, tile pair) at a time: reading all 288 accumulators ahead spills
        // rows r = 2 rp, 2 rp + 1 of the accumulators as register PAIRS (they are neighbours in every f32x4): the transform is
        // element-wise in r, so every operation is one v_pk_* for both tiles
        f32x2 tt[4][6];     // A^T M: rows 0..3, columns 0..5
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          auto pr = [&](int pos) { return f32x2{acc[pos][nb][2 * rp], acc[pos][nb][2 * rp + 1]}; };
          const f32x2 m0 = pr(0 * 6 + j), m1 = pr(1 * 6 + j), m2 = pr(2 * 6 + j), m3 = pr(3 * 6 + j), m4 = pr(4 * 6 + j), m5 = pr(5 * 6 + j);
          const f32x2 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          tt[0][j] = m0 + s12 + s34;
          tt[1][j] = __builtin_elementwise_fma(f32x2{2.f, 2.f}, d34, d12);
          tt[2][j] = __builtin_elementwise_fma(f32x2{4.f, 4.f}, s34, s12);
          tt[3][j] = __builtin_elementwise_fma(f32x2{8.f, 8.f}, d34, d12) + m5;
        }
        const f32x2 bias2 = {bias1, bias1}, zero2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x2 m0 = tt[i][0], m1 = tt[i][1], m2 = tt[i][2], m3 = tt[i][3], m4 = tt[i][4], m5 = tt[i][5];
          const f32x2 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          f32x2 yv[4];
          yv[0] = m0 + s12 + s34 + bias2;
          yv[1] = __builtin_elementwise_fma(f32x2{2.f, 2.f}, d34, d12) + bias2;
          yv[2] = __builtin_elementwise_fma(f32x2{4.f, 4.f}, s34, s12) + bias2;
          yv[3] = __builtin_elementwise_fma(f32x2{8.f, 8.f}, d34, d12) + m5 + bias2;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const f32x2 v = __builtin_elementwise_max(yv[jj], zero2);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              if (rp == 0) lds[hw_base[e] + 16 * nb + (i * TW + jj) * RH] = v[e];
              else yh1[nb][e][i * 4 + jj] = v[e];
            }
          }
        }
      }
      } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        __builtin_amdgcn_sched_barrier(0);          // one (block, tile) at a time: reading all 288 accumulators ahead spills
        float tt[4][6];     // A^T M: rows 0..3, columns 0..5
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const float m0 = acc[0 * 6 + j][nb][r], m1 = acc[1 * 6 + j][nb][r], m2 = acc[2 * 6 + j][nb][r];
          const float m3 = acc[3 * 6 + j][nb][r], m4 = acc[4 * 6 + j][nb][r], m5 = acc[5 * 6 + j][nb][r];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          tt[0][j] = m0 + s12 + s34;
          tt[1][j] = __builtin_fmaf(2.f, d34, d12);
          tt[2][j] = __builtin_fmaf(4.f, s34, s12);
          tt[3][j] = __builtin_fmaf(8.f, d34, d12) + m5;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float m0 = tt[i][0], m1 = tt[i][1], m2 = tt[i][2], m3 = tt[i][3], m4 = tt[i][4], m5 = tt[i][5];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          float yv[4];
          yv[0] = m0 + s12 + s34 + bias1;
          yv[1] = __builtin_fmaf(2.f, d34, d12) + bias1;
          yv[2] = __builtin_fmaf(4.f, s34, s12) + bias1;
          yv[3] = __builtin_fmaf(8.f, d34, d12) + m5 + bias1;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const float v = yv[jj] > 0.f ? yv[jj] : 0.f;
            if (r < 2) lds[hw_base[r & 1] + 16 * nb + (i * TW + jj) * RH] = v;
            else yh1[nb][r - 2][i * 4 + jj] = v;
          }
        }
      }
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // per-thread parts of the tail's global addresses (bytes), for this tile
    const int m0 = tid_t / C4, c4t = tid_t - m0 * C4;
    const int opart = (m0 * a.cso + c4t * 4) * 4;
    int ocol[CPR];                              // opart + column step k, or a huge positive number where that column is outside the frame
#pragma unroll
    for (int k = 0; k < CPR; ++k) ocol[k] = (tx * TW + m0 + k * PPI < a.W) ? opart + k * PPI * a.cso * 4 : W36_MARKER;

#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int y0 = ty * TH + half * (TH / 2), x0 = tx * TW;       // first pixel of the half: HPX = (TH / 2) x TW pixels, row-major
      const int rows_valid = min(TH / 2, a.H - y0);                 // (uniform) pixel rows of the half inside the frame
      const int xbase = __builtin_amdgcn_readfirstlane(((b * a.H + y0) * a.W + x0) * a.csx * 4);
      const int obase = __builtin_amdgcn_readfirstlane(((b * a.H + y0) * a.W + x0) * a.cso * 4);
      if (half == 1) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) lds[hw_base[rr] + 16 * nb + (i * TW + jj) * RH] = yh1[nb][rr][i * 4 + jj];
        if constexpr (DUST) {
          const int T = 8 * ((tid_t >> 1) & 1) + 2 * (tid_t >> 2) + (tid_t & 1), th = T & 7;
          if (tid_t < 16 && T >= 8) {
            const int hb = C::OFF_H0 + ((4 * (th / TXT)) * TW + 4 * (th % TXT)) * RH + 64;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) lds[hb + (i * TW + jj) * RH] = lds[C::D_V64 + tid_t * 16 + i * 4 + jj];
          }
        }
      }
      FPC_LDS_BARRIER();
      if (wg == wg_stamp && half == 0) { FPC_STAMP(2) }
      // DUST: output channel 64 of the half's 128 pixels.  Lane (pxl = l >> 4, q = l & 15) of wave w takes float4 q of the
      // rows of pixels 32 w + 4 it + pxl, it = 0..7 (a wave instruction reads four rows' 256 contiguous bytes: no bank
      // conflict; two threads per pixel with a K half each were 8-way conflicts on every read): a partial dot product per
      // pixel over h here and over the projection's x in its passes below, added up across the 16 lanes in the epilogue.
      // And h[64]'s share of outputs 0..63 is the INITIAL VALUE of their accumulators.
      [[maybe_unused]] float dpart[8];
      [[maybe_unused]] float dw2r = 0.f;
      [[maybe_unused]] float dhv[8][4];
      [[maybe_unused]] const int dq = lane_t & 15, dpxl = lane_t >> 4;
      [[maybe_unused]] float dx64 = 0.f;      // identity shortcut: x[px][64] of the pixel this lane stores (requested here, used in the epilogue)
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
        dw2r = lds[C::D_HEAD + W36Dust::W2ROW + 16 * wave + (lane_t & 15)];
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) dhv[mb][r] = lds[C::OFF_H0 + (16 * mb + 4 * (lane_t >> 4) + r) * RH + 64];
      }
      // output float4 i of this thread: LDS [m0 + PPI i][c4t], global row i / CPR, column step i % CPR
      int erd = TL4 + m0 * (RH / 4) + c4t;
      asm volatile("" : "+v"(erd));
      auto store_out = [&]() {
#pragma unroll
        for (int i = 0; i < EIT; ++i) {
          if (i / CPR < rows_valid) {                                     // (uniform)
            f32x4 v = lds4[erd + i * PPI * (RH / 4)];
            if (!a.conv_only) {
              v.x = v.x > 0.f ? v.x : 0.f;
              v.y = v.y > 0.f ? v.y : 0.f;
              v.z = v.z > 0.f ? v.z : 0.f;
              v.w = v.w > 0.f ? v.w : 0.f;
            }
            int voff;
            const int so = obase + (i / CPR) * a.W * a.cso * 4;           // (uniform)
            asm("v_add_i32 %0, %1, %2 clamp" : "=v"(voff) : "v"(ocol[i % CPR]), "s"(so));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), orsrc, voff, 0, 0);
          }
        }
      };

      if (a.conv_only) {  // h is the result: [128 px][N] in LDS -> 16-byte stores (ReLU already applied)
        store_out();
        FPC_LDS_BARRIER();   // the region is reused by the second half / the next tile's pipeline
        continue;
      }

      // ------------------------------------------------ phase 2: 1x1 over h (+ projection over x), 8 pixel blocks x NB channel blocks per wave
      f32x4 acc2[8][NB];
      // the 1x1 fragments through a buffer descriptor, as the 3x3's: step and channel block in the SCALAR offset (a
      // 64-bit address per load is two VALU instructions in an MFMA gap)
      auto ldb2 = [&](int s, int nb) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, (int)w2lane, (nb * g2s + s * 64) * 16, 0));
      };
      constexpr int KH = N / 16;
      // one 16-channel step of the GEMM: blocks of eight MFMAs (one pixel block x 2 channel blocks, or two pixel blocks x 1),
      // the next block's A fragment(s) read in the gap in front of this block's MFMAs
      constexpr int NBLK = NB == 2 ? 8 : 4, APB = NB == 2 ? 1 : 2;      // blocks per step, A fragments per block
      auto gemm_over = [&](int row_pitch4, int ksteps, int s_first) {
        int ar = TL4 + n16 * row_pitch4 + kq;
        asm volatile("" : "+v"(ar));
        f32x4 cb[4][NB];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) cb[i][nb] = ldb2(s_first + i, nb);
        f32x4 af[2][APB];
#pragma unroll
        for (int q = 0; q < APB; ++q) af[0][q] = lds4[ar + q * 16 * row_pitch4];
        for (int g = 0; g < ksteps; g += 4) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) cb[(u + 3) & 3][nb] = ldb2(s_first + g + u + 3, nb);
            // (same body as gemm_step, with this tile's row pitch)
            fpc_static_for<NBLK>([&](auto B_) __attribute__((always_inline)) {
              constexpr int blk = decltype(B_)::value;
              if (blk + 1 < NBLK) {
#pragma unroll
                for (int q = 0; q < APB; ++q) af[(blk + 1) & 1][q] = lds4[ar + (g + u) * 4 + ((blk + 1) * APB + q) * 16 * row_pitch4];
              } else {
#pragma unroll
                for (int q = 0; q < APB; ++q) af[(blk + 1) & 1][q] = lds4[ar + (g + u + 1) * 4 + q * 16 * row_pitch4];   // (next step; past the last one: read and dropped)
              }
              __builtin_amdgcn_sched_barrier(0);
              if constexpr (NB == 2) fpc_mfma_step<true, true, true>(acc2[blk][0], acc2[blk][1], af[blk & 1][0], af[blk & 1][0], cb[u][0], cb[u][NB - 1]);
              else fpc_mfma_step<true, true>(acc2[2 * blk][0], acc2[2 * blk + 1][0], af[blk & 1][0], af[blk & 1][APB - 1], cb[u][0], cb[u][0]);
              __builtin_amdgcn_sched_barrier(0);
            });
          }
        }
      };
      // Shortcut.  Every register of it is DEFINED AND USED inside one branch: declared in front of the branches and
      // used behind them, the staging registers were undefined values live around the whole tile loop -- and spilled.
      if (proj) {
        // projection: more K for the same accumulators -- up to 128 channels of the half's pixels per pass, [px][32 float4];
        // pixel m = (tid >> 5) + 8 i
        constexpr int XIT = HPX * 32 / NT;                            // staging float4 per thread and pass: 16
        f32x4 xst[XIT];
        auto load_x = [&](int pass) {
          const int kx4 = min(32, a.k8_x * 2 - pass * 32);
          const int xp = ((tid_t >> 5) * a.csx + ((tid_t & 31) < kx4 ? pass * 128 + (tid_t & 31) * 4 : 0)) * 4;
#pragma unroll
          for (int i = 0; i < XIT; ++i) {
            constexpr int XCPR = TW / 8;
            const int so = xbase + ((i / XCPR) * a.W + (i % XCPR) * 8) * a.csx * 4;      // (uniform)
            xst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xp + so, 0, 0));
          }
        };
        load_x(0);
        {
          float zf = 0.f;
          asm volatile("" : "+v"(zf));
#pragma unroll
          for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc2[mb][nb] = f32x4{zf, zf, zf, zf};
          if constexpr (DUST) {
#pragma unroll
            for (int mb = 0; mb < 8; ++mb)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc2[mb][0][r] = dhv[mb][r] * dw2r;
          }
        }
        gemm_over(RH / 4, KH, 0);
        if (wg == wg_stamp && half == 0) { FPC_STAMP(3) }
        const int npass = (a.k8_x + 15) >> 4;
        for (int pass = 0; pass < npass; ++pass) {
          FPC_LDS_BARRIER();   // h (or the previous pass's x) has been read by every wave
          {
            int xw = TL4 + (tid_t >> 5) * (RX / 4) + (tid_t & 31);
            asm volatile("" : "+v"(xw));
#pragma unroll
            for (int i = 0; i < XIT; ++i) lds4[xw + i * 8 * (RX / 4)] = xst[i];
          }
          FPC_LDS_BARRIER();
          if (pass + 1 < npass) load_x(pass + 1);
          if constexpr (DUST) {      // this pass's 128 channels of x against the projection's column for output 64
            const f32x4 wa = lds4[C::D_COL / 4 + 20 + pass * 32 + dq], wb = lds4[C::D_COL / 4 + 20 + pass * 32 + 16 + dq];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
              const int px = 32 * wave + 4 * it + dpxl;
              const f32x4 xa = lds4[TL4 + px * (RX / 4) + dq], xb = lds4[TL4 + px * (RX / 4) + 16 + dq];
              float v = dpart[it];
#pragma unroll
              for (int e = 0; e < 4; ++e) v = __builtin_fmaf(xb[e], wb[e], __builtin_fmaf(xa[e], wa[e], v));
              dpart[it] = v;
            }
          }
          const int steps = min(8, a.k8_x / 2 - pass * 8);   // 16-channel steps of this pass: 4 or 8
          gemm_over(RX / 4, steps, KH + pass * 8);
        }
      } else {
        // identity: x is the INITIAL VALUE of the accumulators, read in their layout (row 4 kq + r of pixel block mb, this
        // lane's channel): 64 four-byte loads per lane and half, no staging registers, no additions in the epilogue
        int xl = (4 * kq * a.csx + 16 * (wave * NB) + n16) * 4;
        asm volatile("" : "+v"(xl));
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            constexpr int BPR = TW / 16;                                // pixel blocks per pixel row (1 or 2)
            const int so = xbase + ((mb / BPR) * a.W + (mb % BPR) * 16 + r) * a.csx * 4;      // (uniform)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc2[mb][nb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, xl + so + 64 * nb, 0, 0));
          }
        if constexpr (DUST) {
#pragma unroll
          for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2[mb][0][r] = __builtin_fmaf(dhv[mb][r], dw2r, acc2[mb][0][r]);
        }
        gemm_over(RH / 4, KH, 0);
        if (wg == wg_stamp && half == 0) { FPC_STAMP(3) }
      }

      // ------------------------------------------------ epilogue: output tile through LDS -> 16-byte stores
      if (wg == wg_stamp && half == 0) { FPC_STAMP(4) }
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // (asm MFMA results read by the VALU, as above)
      FPC_LDS_BARRIER();
      {
        int ew = C::OFF_H0 + (4 * kq) * RH + 16 * (wave * NB) + n16;
        asm volatile("" : "+v"(ew));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float bias = a.b2[16 * (wave * NB + nb) + n16];
#pragma unroll
          for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[ew + 16 * nb + (16 * mb + r) * RH] = acc2[mb][nb][r] + bias;
        }
      }
      FPC_LDS_BARRIER();
      store_out();
      if constexpr (DUST) {
        // channels 64..71 of the pixel: (out[64], 0, 0, 0) from the pair's first thread, zeros from the second (pad
        // channels are exact zeros: the next layer's 16-byte slot of input channel 64 relies on it)
        // The 16 lanes of a pixel add up in a halving butterfly: each step a lane keeps the half of its values its q bit
        // selects and receives the partner's -- 4 + 2 + 1 + 1 exchanges instead of 8 x 4 -- and ends with the total of
        // pixel it = q >> 1 in both lanes q & 1: exactly the pair that stores that pixel's channels 64..67 and 68..71.
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
      if (wg == wg_stamp && half == 0) { FPC_STAMP(5) FPC_RSTAMP(7) }
      FPC_LDS_BARRIER();   // the region is reused by the second half / the next tile's pipeline
    }  // halves
    pos_cur = pos_next;
    base_cur = base_next;
  }  // persistent tile loop
}

template <int NB, int TYT, int TXT>
__global__ __launch_bounds__(256, 1) void wblock36_kernel(const WBlockArgs a) {
  wblock36_body<NB, TYT, TXT, false>(a);
}

// The detector's blocks: 64 channels as wblock36_kernel<1, ...> + the 65th ("dustbin") channel beside them (W36Dust).
template <int TYT, int TXT>
__global__ __launch_bounds__(256, 1) void wblock36_dust_kernel(const WBlockArgs a) {
  wblock36_body<1, TYT, TXT, true>(a);
}

}  // namespace fpc
