// block_bf16.h -- the ResNetBlock / convolution kernels of block_mfma.h and conv_mfma.h with bf16
// activations and weights and fp32 accumulation (`dtype = FPC_BF16`; BASELINE.json configs[4]:
// 1280x960 frames, the large-activation regime).  Same structure -- halo tile in LDS, weights
// pre-packed in MFMA lane order and read straight from global memory, h never leaves LDS, fused
// shortcut, vectorised epilogue -- on v_mfma_f32_32x32x16_bf16 (16 channels per instruction: lane
// (row = l & 31, half = l >> 5) supplies channels 8*half .. 8*half+7 as one 16-byte fragment).
// Tensors are bf16 NHWC with the channel count padded to a multiple of 16 where they feed a GEMM's K.
// The last detector / descriptor blocks write fp32 so that the post-processing kernels are the same as in the
// fp32 path.
#pragma once
#include "conv_mfma.h"
#include "nms_word.h"

namespace fpc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4b __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;  // storage type

__device__ __forceinline__ bf16_t f2bf(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float((unsigned)b << 16); }
// two floats -> one dword of two bf16 (round to nearest even, low half = first) as ONE v_cvt_pk_bf16_f32.  Written as
// f2bf(a) | f2bf(b) << 16 the compiler pairs the conversions ACROSS the two dwords of a store and spends four more
// instructions per dword on putting the halves where they belong.  (A vector conversion, not inline asm: the compiler
// must see a VALU instruction that reads MFMA results, or it leaves out the wait states between the two.)
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_));
}
__device__ __forceinline__ uint4 pack8(const float4& a, const float4& b) {
  return make_uint4(pk_bf16(a.x, a.y), pk_bf16(a.z, a.w), pk_bf16(b.x, b.y), pk_bf16(b.z, b.w));
}

struct BlockBfArgs {
  const void* x;         // input NHWC: bf16 (fp32 when in_f32: the split-operand kernels of block_x3.h only), offset to its first channel
  int csx, nchunk;       // pixel stride in ELEMENTS, Cin_pad / KC
  unsigned x_bytes;      // block_bf16_kernel: bytes of the input tensor from x on (buffer descriptor range)
  int H, W;
  int in_f32;
  const uint4* w1;       // 3x3 fragments: step = (chunk*ntaps + tap)*K16 + k16, [step][nb][64] uint4 (+2 steps)
  const float* b1;
  int ntaps;             // 9 for a block; 1..4 for a ConvTranspose phase (conv_only)
  int tapoff16[9];       // halo offset of each tap in 16-byte units
  const uint4* w2;       // 1x1 fragments: k16_h steps over h, then k16_x over x
  const float* b2;
  int k16_h, k16_x;      // k16_x == 0: identity shortcut
  int conv_only;         // 1: stop after phase 1 (bias, ReLU, store): ConvTranspose phases
  void* out;             // bf16 (or fp32 when out_f32), offset to its first channel
  int cso, out_f32;
  int Ho, Wo, tiles_x, tiles_y, frame0;
  int total_tiles;       // block_bf16_kernel (persistent grid): tiles_x * tiles_y * frames of the launch
  // fused exp-softmax + depth-to-space + threshold (the detector's last block in fpc_detect; instances whose waves hold
  // all channels of their pixels): instead of the logits, the NMS state map and the candidate lists are written
  int softmax;
  float thresh;
  uint32_t* nmsmap;      // [frames of the launch][8 Ho][8 Wo], offset to the launch's first frame
  uint32_t* cand;        // same shape
  int32_t* ncand;        // [frames of the launch], zeroed by the host
  int OH, OW, oys, oxs, oy0, ox0;   // output pixel = (y*oys + oy0, x*oxs + ox0) in an OH x OW buffer
  int pad;               // halo origin = tile origin * S - pad
  int norelu;            // conv_only: 1 = no ReLU (the last 1x1 of a head of the C++ network)
  int* range_flag;       // block_h2_kernel: set to 1 when a value that a later layer will split leaves fp16's range
#ifdef FPC_DIAG
  unsigned long long* stamps;
#endif
};

// Pixel order inside a tile, and the halo's row pitch (round 3).  Phase 1 reads a pixel's 16 bytes of a K16 step with
// ds_read_b128: the LDS serves that in groups of sixteen lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (banks =
// 16-byte unit mod 16), and a halo pixel is ROW16 = KC / 8 + 1 units (odd), so a group is conflict-free iff its sixteen
// halo pixel indices differ mod 16.  Row-major pixels (lane = pixel, 20 or 16 to a tile row, halo rows 22 or 18 wide)
// put lanes 20-27 on the next tile row and two of them on the units of lanes 12-15: PMC showed 43-50 % of these
// kernels' LDS cycles as bank conflicts at 0.5-0.6 LDS busy.  With 32-pixel blocks of FOUR tile rows x EIGHT pixels
// (lane = 8 row + column) and a halo pitch = 8 mod 16 pixels, a group's lanes are columns 0-3 of rows 0 and 3 and 4-7 of
// rows 1 and 2 (or the complement): indices {0-3}, {8 + 4..7}, {16 + 4..7}, {24 + 0..3} mod 16 -- all sixteen.
// Stride-1 tiles whose sides are multiples of 4 x 8 use it; the others stay row-major.
__host__ __device__ constexpr bool bf_blocked(int TH, int TW, int S) { return S == 1 && TH % 4 == 0 && TW % 8 == 0; }
__host__ __device__ constexpr int bf_halo_pitch(int TH, int TW, int S, int EXT) {
  const int hw = (TW - 1) * S + EXT;
  return bf_blocked(TH, TW, S) ? (hw <= 8 ? 8 : (hw - 8 + 15) / 16 * 16 + 8) : hw;
}
// pixel m of a tile (block m >> 5, lane m & 31) -> tile row / column
template <int TH, int TW, int S>
__device__ __forceinline__ void bf_pixel(int m, int& py, int& px) {
  if constexpr (bf_blocked(TH, TW, S)) {
    constexpr int BX = TW / 8;
    const int blk = m >> 5, l = m & 31;
    py = (blk / BX) * 4 + (l >> 3);
    px = (blk % BX) * 8 + (l & 7);
  } else {
    py = m / TW;
    px = m - py * TW;
  }
}

template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB, int CMIDP>
struct BlockBfCfg {
  static constexpr int NT = WM * WN * 64;
  static constexpr int HW = (TW - 1) * S + EXT, HH = (TH - 1) * S + EXT;
  static constexpr int HP = bf_halo_pitch(TH, TW, S, EXT);   // halo row pitch in LDS, pixels
  static constexpr int ROW16 = KC / 8 + 1;            // 16-byte units per halo pixel (+1 skew)
  static constexpr int ROWH16 = CMIDP / 8 + 1;        // per h row
  static constexpr int M = WM * MB * 32, N = WN * NB * 32;
  static constexpr int HALO_BYTES = HH * HP * ROW16 * 16;
  static constexpr int H_BYTES = M * ROWH16 * 16;
  static constexpr int TILE_BYTES = HALO_BYTES > H_BYTES ? HALO_BYTES : H_BYTES;   // the halo chunk, then h
  static constexpr int LDS_BYTES = TILE_BYTES + 2 * N * 4;                         // + b1, b2
  static_assert(KC % 16 == 0 && CMIDP % 16 == 0 && CMIDP <= N, "bf16 MFMA consumes 16 channels per step");
};

// ONE: the block's input is ONE chunk (Cin_pad == KC; the host checks it) -- known at compile time, so that the second
// accumulator set (the shortcut's, which conv2 continues) is born at the chunk's tenth "tap" instead of being carried
// around the chunk loop, and no request for a "next chunk" holds the staging registers through the steps: 64 + 48 registers
// fewer inside the unrolled steps, which is what lets a wave keep FOUR pixel blocks per weight fragment (MB = 4).
// NCH: the number of chunks as a compile-time fact (1: ONE above; 2: a 256-channel input in two 128-channel chunks -- the
// second accumulator set is born at the FIRST chunk's tenth tap and the last chunk requests no successor); 0: a.nchunk.
template <class T> struct bf_chunk_index { static constexpr int value = T::value; };
template <> struct bf_chunk_index<int> { static constexpr int value = -1; };
template <int V> struct bf_ic { static constexpr int value = V; constexpr operator int() const { return V; } };

template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB, int CMIDP, int NCH>
__device__ __forceinline__ void block_bf16_body(const BlockBfArgs& a) {
  constexpr bool ONE = NCH > 0;   // (the name of round 4's first form: "the trip count is known")
  using C = BlockBfCfg<TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP>;
  constexpr int NT = C::NT, HW = C::HW, HH = C::HH, HP = C::HP, ROW16 = C::ROW16, K16 = KC / 16, KC8 = KC / 8;
  constexpr int NV = HH * HW * KC8, ITER = (NV + NT - 1) / NT, ROWH16 = C::ROWH16, NBT = WN * NB;
  extern __shared__ uint4 lds16[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y;
  // Persistent grid (a multiple of 8 workgroups, a few per CU): XCD k -- the workgroups with blockIdx.x & 7 == k -- walks
  // the tiles [k T / 8, (k + 1) T / 8) of the launch, so neighbouring tiles (which share halo rows) meet in one L2, and a
  // workgroup requests its NEXT tile's first halo chunk before the current tile's epilogue: that request's trip to HBM
  // was 11 % of a 128-channel tile and 28 % of a 64-channel one (in-kernel stamps).
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
    p.iy0 = p.ty * TH * S - a.pad;
    p.ix0 = p.tx * TW * S - a.pad;
    p.xbase = (unsigned)((p.b * a.H + p.iy0) * a.W + p.ix0) * (unsigned)(a.csx * 2);   // unsigned: mod 2^32, exact for in-frame pixels
    return p;
  };

  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    int m = (wm * MB + mb) * 32 + l31;
    m = m < TH * TW ? m : TH * TW - 1;
    int py, px;
    bf_pixel<TH, TW, S>(m, py, px);
    abase[mb] = ((py * S) * HP + px * S) * ROW16 + half;
  }
  constexpr int stepstride = NBT * 64;
  // The MFMAs below take the weight fragment as the A operand and the pixels as B: the accumulators then hold the tile
  // TRANSPOSED -- a lane owns ONE pixel (l31 of its 32-pixel block) and, per 32-channel block, four groups of four
  // consecutive channels (group g: channels 8 g + 4 half ..+3 in registers 4 g ..+3) -- so both epilogues work on whole
  // 8- / 16-byte channel groups straight from registers: h goes to LDS in 16 ds_write_b64 per lane (it was 64 two-byte
  // writes), the output leaves as 8-byte (bf16) or 16-byte (fp32) stores with no fp32 tile in LDS in between (67 KB of
  // the workgroup's LDS, a barrier and ~250 VALU / LDS instructions per lane).  Biases sit in LDS, read 16 bytes at a time.
  float* bias_lds = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(lds16) + C::TILE_BYTES);   // [2][N]
  for (int i = tid; i < 2 * C::N; i += NT) {
    const int which = i / C::N, n = i - which * C::N;
    bias_lds[i] = which == 0 ? a.b1[n] : (a.b2 ? a.b2[n] : 0.f);
  }

  // The halo chunk through a buffer descriptor: a position outside the frame (or a chunk past the last one, or a tile
  // past the workgroup's last) gets an offset beyond the descriptor's range and comes back as zeros from the hardware's
  // bounds check.  No branch anywhere between a request and its use: around a block boundary the compiler waits for
  // EVERY outstanding load (it did, once per chunk, for the next chunk's halo right after requesting it and for the
  // fragments in flight: PMC showed the waves of this kernel parked 61 % of the time).
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 stage[ITER];   // (vector values: an array of uint4 structs is not promoted to registers)
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  // Element e = tid + i NT of a halo chunk is 16 bytes: channels 8 c8 ..+7 of halo pixel e / KC8.  Where NT is a multiple
  // of KC8 (every instance but the 80-channel one) c8 is the thread's own for every i and the pixel advances by
  // PS = NT / KC8 per step, so (hy, hx), the global byte offset and the LDS slot are carried from element to element -- a
  // compare and two selects -- instead of two divisions by constants, the frame's pitch and the pixel stride per element:
  // round 4 counted 40 VALU instructions per requested element (18 per staged one) in the ISA, 1 230 per wave and
  // 256-pixel tile of layer1 beside 176 MFMAs, and the VALU slots of the workgroup in its epilogue are the other
  // workgroup's MFMA slots.
  constexpr bool CARRY = NT % KC8 == 0;
  constexpr int PS = CARRY ? NT / KC8 : 1, PQ = PS / HW, PR = PS % HW;
  auto load_chunk = [&](const TileP& p, int chunk) {
    const int wlim = (chunk < a.nchunk && p.live) ? a.W : 0;   // nothing is in range past the last chunk / tile
    int tl = tid;
    asm volatile("" : "+v"(tl));   // (recomputed per call: hoisted out of the tile loop, the per-element offsets spill)
    if constexpr (CARRY) {
      const int pix0 = tl / KC8, c8 = tl - pix0 * KC8;
      int hy = pix0 / HW, hx = pix0 - hy * HW;
      unsigned off = p.xbase + (unsigned)(((hy * a.W + hx) * a.csx + chunk * KC + c8 * 8) * 2);
      const unsigned dstep = (unsigned)(((PQ * a.W + PR) * a.csx) * 2), dwrap = (unsigned)(((a.W - HW) * a.csx) * 2);
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
        const int iy = p.iy0 + hy, ix = p.ix0 + hx;
        bool ok = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)wlim);
        if ((i + 1) * NT > NV) ok = ok & (hy < HH);   // (only the last step can run past the halo)
        stage[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(ok ? off : 0xfffffff0u), 0, 0));
        if (i + 1 < ITER) {
          hx += PR;
          const bool wrap = hx >= HW;
          hx -= wrap ? HW : 0;
          hy += PQ + (wrap ? 1 : 0);
          off += dstep + (wrap ? dwrap : 0u);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
        const int e = tl + i * NT;
        const int pix = e / KC8, c8 = e - pix * KC8;
        const int hy = pix / HW, hx = pix - hy * HW;
        const int iy = p.iy0 + hy, ix = p.ix0 + hx;
        const bool ok = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)wlim) & (hy < HH);   // (& not &&: no branch)
        unsigned off = p.xbase + (unsigned)(((hy * a.W + hx) * a.csx + chunk * KC + c8 * 8) * 2);
        asm volatile("" : "+v"(off));   // computed for every lane: as a conditional the compiler branches around it
        stage[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(ok ? off : 0xfffffff0u), 0, 0));
      }
    }
  };
  auto store_chunk = [&]() {
    int tl = tid;
    asm volatile("" : "+v"(tl));
    if constexpr (CARRY) {
      const int pix0 = tl / KC8, c8 = tl - pix0 * KC8;
      int hy = pix0 / HW, hx = pix0 - hy * HW;
      int slot = (hy * HP + hx) * ROW16 + c8;
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
        int sl = slot;
        if ((i + 1) * NT > NV) sl = hy < HH ? sl : (HH * HW - 1 + (HH - 1) * (HP - HW)) * ROW16 + KC8;   // past the halo: the skew column of its last pixel (never read)
        *reinterpret_cast<u32x4*>(&lds16[sl]) = stage[i];
        if (i + 1 < ITER) {
          hx += PR;
          const bool wrap = hx >= HW;
          hx -= wrap ? HW : 0;
          hy += PQ + (wrap ? 1 : 0);
          slot += (PQ * HP + PR) * ROW16 + (wrap ? (HP - HW) * ROW16 : 0);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
        const int e = tl + i * NT;
        int pix = e / KC8, c8 = e - pix * KC8;
        if (NV % NT != 0) {   // elements past the halo go to the skew column of its last pixel (never read)
          const bool in = e < NV;
          pix = in ? pix : HH * HW - 1;
          c8 = in ? c8 : KC8;
        }
        if constexpr (HP != HW) pix += (pix / HW) * (HP - HW);   // halo row pitch in LDS
        *reinterpret_cast<u32x4*>(&lds16[pix * ROW16 + c8]) = stage[i];
      }
    }
  };

  TileP cur = tile_params(t_first + (int)(blockIdx.x >> 3));
  load_chunk(cur, 0);
  // conv1's (+ the shortcut's) fragment ring, see phase 1.  The stream is read CIRCULARLY: the last D steps of a tile
  // request steps 0 .. D - 1 again, which are the next tile's first -- no L2 round trip at the head of a tile.
  constexpr int NS = 10 * K16;          // steps of a chunk: 9 taps of conv1 and the shortcut, K16 each
  // (2 for detector.layer.1's 50 steps of three fragments: five steps ahead were 60 registers of a kernel that then spilled
  // 22 -- every spill reload waits for ALL outstanding requests -- 0.40 -> 0.33 ms per 64 HD frames; for the others 4
  // measured better than 2)
  constexpr int D = NS % 4 == 0 ? 4 : 2;   // fragment ring: D steps ahead
  static_assert(NS % D == 0, "the ring position of a step must not depend on the chunk");
  u32x4 ring[D][NB];
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint4*>(a.w1), 0, (int)((unsigned)(a.nchunk * NS + 2) * (unsigned)(stepstride * 16)), 0x00020000);
  const unsigned wlane = (unsigned)((wn * NB) * 64 + lane) * 16u;
  const int wtotal = a.nchunk * NS * stepstride * 16;
  int wstep = 0;   // (scalar) byte offset of the next step to request
  if (ONE || a.ntaps == 9) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        ring[d][nb] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)(wlane + nb * 1024), wstep, 0));
      wstep += stepstride * 16;
    }
  }
#ifdef FPC_DIAG
  // second record of a workgroup (+ 32768): shader clock / 100 MHz clock at its start and at the end of every tile, tile count
  if (a.stamps && threadIdx.x == 0) {
    unsigned long long t0_, r0_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_), "=s"(r0_)::"memory");
    a.stamps[((size_t)blockIdx.x + 32768) * 8 + 0] = t0_;
    a.stamps[((size_t)blockIdx.x + 32768) * 8 + 1] = r0_;
    a.stamps[((size_t)blockIdx.x + 32768) * 8 + 4] = 0;
  }
#endif
  for (int tcur = t_first + (int)(blockIdx.x >> 3); tcur < t_end; tcur += per) {
  const int b = cur.b, ty = cur.ty, tx = cur.tx;
  const uint4* wp = a.w1 + (size_t)(wn * NB) * 64 + lane;   // (the ConvTranspose phases' path below)
  // acc: conv1 (then dead once h is written); acc2: the shortcut + conv2 = the block's output before bias / ReLU
  f32x16 acc[MB][NB], acc2[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[mb][nb][r] = 0.f;
        if (!ONE) acc2[mb][nb][r] = 0.f;
      }

  // ---------------------------------------------------------------- phase 1: KxK conv (+ the shortcut)
  FPC_STAMP(0)
  FPC_RSTAMP(6)
  uint4 b0[NB], b1[NB];
  // A chunk of a ResNetBlock is 9 x K16 steps of conv1 and K16 steps of the SHORTCUT on the same chunk of x: the 1x1
  // projection (or the identity, as a unit matrix: exact in bf16 x fp32) reads exactly the pixels the 3x3's centre tap
  // reads, which are in LDS right now -- as phase 2b it fetched them again from global memory, 16 bytes per lane at a
  // pixel stride (32 cache lines per request): 23 k of layer_out.0's 85 k cycles per tile for 3 k cycles of MFMAs, and the
  // identity's loads stood at the head of the epilogue (in-kernel stamps).  Its fragments follow the chunk's conv1
  // fragments in the w1 stream ("tap 9").
  if (ONE || a.ntaps == 9) {
    // Two steps of fragments ahead -- 256 MFMA cycles of this wave -- do not cover an L2 round trip, and the LDS read of a
    // step's pixels sat right in front of its MFMAs.  Here the chunk's steps are unrolled, fragments run D steps ahead
    // in a register ring whose slots are compile-time names (through a buffer descriptor too: the lane's offset in a
    // VGPR that never changes, the step in the scalar offset, and a request past the last step returns zeros), and a
    // step's pixels are read while the previous step's MFMAs run.
    auto run_chunk = [&](auto chunk_c) {
      const int chunk = chunk_c;
      constexpr int CI = bf_chunk_index<decltype(chunk_c)>::value;   // the chunk's index where it is a compile-time fact, else -1
      FPC_LDS_BARRIER();   // the previous chunk's pixels have been read
      store_chunk();
      FPC_LDS_BARRIER();
      if (chunk == 0) { FPC_STAMP(1) }
      if (!ONE || CI + 1 < NCH) load_chunk(cur, chunk + 1);
      u32x4 av[MB], an[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) av[mb] = *reinterpret_cast<const u32x4*>(&lds16[abase[mb] + a.tapoff16[0]]);
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        if (st + 1 < NS) {
          const int tapn = (st + 1) / K16;
          const int toff = a.tapoff16[tapn < 9 ? tapn : 4];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) an[mb] = *reinterpret_cast<const u32x4*>(&lds16[abase[mb] + toff + ((st + 1) % K16) * 2]);
        }
        // (the NEXT step's pixels are requested here, in front of this step's MFMAs: without the fence the scheduler
        // sinks the reads to the end of the step -- shorter live ranges -- and every MFMA of the next step then waits
        // out the LDS latency of a read issued two instructions earlier: the ISA showed ds_read, ds_read,
        // s_waitcnt lgkmcnt(1), v_mfma all through the chunk loop)
        __builtin_amdgcn_sched_barrier(0);
        if (st / K16 < 9) {
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[st % D][nb]),
                                                                    __builtin_bit_cast(bf16x8, av[mb]), acc[mb][nb], 0, 0, 0);
        } else {
          if (ONE && CI == 0 && st == 9 * K16) {   // the shortcut's accumulators start here
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[mb][nb][r] = 0.f;
          }
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc2[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[st % D][nb]),
                                                                     __builtin_bit_cast(bf16x8, av[mb]), acc2[mb][nb], 0, 0, 0);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          ring[st % D][nb] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)(wlane + nb * 1024), wstep, 0));
        wstep += stepstride * 16;
        wstep = wstep == wtotal ? 0 : wstep;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[mb] = an[mb];
        __builtin_amdgcn_sched_barrier(0);   // (left alone, the scheduler sinks the ring's requests to just before their use)
      }
    };
    if constexpr (NCH == 1) {
      run_chunk(bf_ic<0>{});
    } else if constexpr (NCH == 2) {
      run_chunk(bf_ic<0>{});
      run_chunk(bf_ic<1>{});
    } else {
      for (int chunk = 0; chunk < a.nchunk; ++chunk) run_chunk(chunk);
    }
  } else {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) b0[nb] = wp[nb * 64];
  wp += stepstride;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) b1[nb] = wp[nb * 64];
  wp += stepstride;
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    FPC_LDS_BARRIER();
    store_chunk();
    FPC_LDS_BARRIER();
    load_chunk(cur, chunk + 1);   // (all zeros past the last chunk: no branch between a request and its use)
    for (int tap = 0; tap < a.ntaps; ++tap) {
      const int toff = a.tapoff16[tap];
#pragma unroll
      for (int k = 0; k < K16; ++k) {
        uint4 b2[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b2[nb] = wp[nb * 64];
        wp += stepstride;
        __builtin_amdgcn_sched_barrier(0);
        uint4 av[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[mb] = lds16[abase[mb] + toff + k * 2];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b0[nb]),
                                                                  __builtin_bit_cast(bf16x8, av[mb]), acc[mb][nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          b0[nb] = b1[nb];
          b1[nb] = b2[nb];
        }
      }
    }
  }
  }

  FPC_STAMP(2)
  if (ONE || !a.conv_only) {
    // -------------------------------------------------------------- h = relu(acc + b1) -> LDS (bf16)
    // phase 2's first D2 steps of fragments are requested before h is written and land behind that.
    constexpr int KH = CMIDP / 16, D2 = KH < 4 ? KH : 4;
    u32x4 ring2[D2][NB];
    const __amdgpu_buffer_rsrc_t wrsrc2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(a.w2), 0, (int)((unsigned)(KH + 2) * (unsigned)(stepstride * 16)), 0x00020000);
    const unsigned wlane2 = (unsigned)((wn * NB) * 64 + lane) * 16u;
#pragma unroll
    for (int d = 0; d < D2; ++d)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        ring2[d][nb] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc2, (int)(wlane2 + nb * 1024), d * stepstride * 16, 0));
    FPC_LDS_BARRIER();
    {
      unsigned char* hl = reinterpret_cast<unsigned char*>(lds16);
      int nl = (wn * NB) * 32 + 4 * half, ml = (wm * MB) * 32 + l31;
      asm volatile("" : "+v"(nl), "+v"(ml));   // (recomputed per tile: hoisted out of the tile loop, the 24 addresses below spill)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n0 = nl + nb * 32 + 8 * g;   // this lane's four channels of the group
          const float4 bias = *reinterpret_cast<const float4*>(bias_lds + n0);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const int m = ml + mb * 32;
            float v0 = acc[mb][nb][4 * g + 0] + bias.x, v1 = acc[mb][nb][4 * g + 1] + bias.y;
            float v2 = acc[mb][nb][4 * g + 2] + bias.z, v3 = acc[mb][nb][4 * g + 3] + bias.w;
            v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; v2 = v2 > 0.f ? v2 : 0.f; v3 = v3 > 0.f ? v3 : 0.f;
            if (CMIDP == C::N || n0 < CMIDP)
              *reinterpret_cast<uint2*>(hl + m * (ROWH16 * 16) + n0 * 2) =
                  make_uint2(pk_bf16(v0, v1), pk_bf16(v2, v3));
          }
        }
    }
    FPC_LDS_BARRIER();
    FPC_STAMP(3)
    // -------------------------------------------------------------- phase 2: K over h (LDS), on top of the shortcut
    int hbase[MB];
    {
      int hb = ((wm * MB) * 32 + l31) * ROWH16 + half;
      asm volatile("" : "+v"(hb));
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) hbase[mb] = hb + mb * 32 * ROWH16;
    }
    {
      u32x4 av[MB], an[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) av[mb] = *reinterpret_cast<const u32x4*>(&lds16[hbase[mb]]);
#pragma unroll
      for (int k = 0; k < KH; ++k) {
        if (k + 1 < KH) {
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) an[mb] = *reinterpret_cast<const u32x4*>(&lds16[hbase[mb] + (k + 1) * 2]);
        }
        __builtin_amdgcn_sched_barrier(0);   // (as in phase 1: the next step's reads in front of this step's MFMAs)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc2[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring2[k % D2][nb]),
                                                                   __builtin_bit_cast(bf16x8, av[mb]), acc2[mb][nb], 0, 0, 0);
        if (k + D2 < KH) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            ring2[k % D2][nb] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc2, (int)(wlane2 + nb * 1024), (k + D2) * stepstride * 16, 0));
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[mb] = an[mb];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = acc2[mb][nb];
  }

  // ---------------------------------------------------------------- epilogue: + bias, ReLU -> bf16 tile in LDS -> 16-byte stores
  // (fp32 outputs -- the logits and the descriptor map -- leave straight from the registers, 16 bytes per lane)
  FPC_STAMP(4)
  const TileP nxt = tile_params(tcur + per);
  bool fused = false;
  if constexpr (WN == 1 && MB == 1 && NB == 3) fused = a.softmax != 0;
  if (!fused) load_chunk(nxt, 0);   // lands behind the epilogue (all zeros past the last tile)
  {
    // (stores through a buffer descriptor as well: a dead pixel's offset is out of range and the hardware drops the
    // store -- an `if` around a store is a block boundary, and the compiler drains every outstanding request there,
    // the next tile's halo included)
    const float* bl = bias_lds + (a.conv_only ? 0 : C::N);
    const int oyb = ty * TH, oxb = tx * TW;
    const float floor_ = a.norelu ? -3.0e38f : 0.f;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)0xffffff00u, 0x00020000);
    int nl = (wn * NB) * 32 + 4 * half, ml = (wm * MB) * 32 + l31;
    asm volatile("" : "+v"(nl), "+v"(ml));
    if (fused) {
      if constexpr (WN == 1 && MB == 1 && NB == 3) {
        // exp-softmax over the 65 channels of a cell, dustbin dropped, depth-to-space, threshold (superpoint.py:111-114,
        // netutils.py:56-75) -- softmax_d2s_kernel<true>'s arithmetic on the logits this epilogue would have stored, BIT FOR
        // BIT: the same exponential (sm_exp<true>: kernels_misc.h), the sum of a cell's 64 exps in the same tree (that kernel: lane q of sixteen holds channels
        // 4 q ..+3, (e0 + e1) + (e2 + e3), then a butterfly over q ^ 8, 4, 2, 1; here q = 8 nb + 2 g + half, so the first
        // three levels are sums of this lane's own registers and the last one is the partner lane of the other half),
        // the same (s + ed) + 1e-5 and reciprocal (sm_scale / sm_prob).  The 295 MB of fp32 logits per 64 HD frames are neither written nor
        // read back, and a launch disappears.
        __shared__ int s_cnt, s_base;
        unsigned short* s_list = reinterpret_cast<unsigned short*>(lds16);   // [TH * TW * 64] local ids (m << 6 | c)
        if (tid == 0) s_cnt = 0;
        FPC_LDS_BARRIER();   // phase 2 has read h; the counter is visible
        const int m = ml;
        int py, px;
        bf_pixel<TH, TW, S>(m, py, px);
        const int y = oyb + py, x = oxb + px;
        const bool live = (m < TH * TW) & (y < a.Ho) & (x < a.Wo);
        const int W8 = a.Wo * 8;
        const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(a.nmsmap, 0, (int)0xffffff00u, 0x00020000);
        float e[2][16], P[2][4];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 bias = *reinterpret_cast<const float4*>(bl + nl + nb * 32 + 8 * g);
            e[nb][4 * g + 0] = sm_exp<true>(fmaxf(acc[0][nb][4 * g + 0] + bias.x, floor_));
            e[nb][4 * g + 1] = sm_exp<true>(fmaxf(acc[0][nb][4 * g + 1] + bias.y, floor_));
            e[nb][4 * g + 2] = sm_exp<true>(fmaxf(acc[0][nb][4 * g + 2] + bias.z, floor_));
            e[nb][4 * g + 3] = sm_exp<true>(fmaxf(acc[0][nb][4 * g + 3] + bias.w, floor_));
            P[nb][g] = (e[nb][4 * g] + e[nb][4 * g + 1]) + (e[nb][4 * g + 2] + e[nb][4 * g + 3]);
          }
        float ed = sm_exp<true>(fmaxf(acc[0][2][0] + bl[64], floor_));   // the dustbin: channel 64 = block 2, register 0 of the half-0 lanes
        ed = __shfl(ed, l31);
        const float a0 = P[0][0] + P[1][0], a1 = P[0][1] + P[1][1], a2 = P[0][2] + P[1][2], a3 = P[0][3] + P[1][3];
        const float b0_ = a0 + a2, b1_ = a1 + a3;
        float ssum = b0_ + b1_;
        ssum += __shfl_xor(ssum, 32);
        const float den = sm_scale<true>((ssum + ed) + .00001f);
        const unsigned mapbase = (unsigned)(((b - a.frame0) * a.Ho * 8 + 8 * y) * W8 + 8 * x + 4 * half);
        unsigned cmask = 0;   // this lane's candidates among its 32 probabilities: bit 16 nb + 4 g + j
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            u32x4 wv;
            unsigned* wp_ = reinterpret_cast<unsigned*>(&wv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float p = sm_prob<true>(e[nb][4 * g + j], den);
              const bool c = live & (p >= a.thresh);
              wp_[j] = c ? nms_state_word(p) : 0u;
              cmask |= c ? 1u << (16 * nb + 4 * g + j) : 0u;
            }
            const unsigned off = (mapbase + (unsigned)((4 * nb + g) * W8)) * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(wv, mrsrc, (int)(live ? off : 0xfffffff0u), 0, 0);
          }
        // The tile's candidates -> s_list: ONE LDS atomic per wave (round 3).  Round 2 compacted every one of the 32
        // probabilities of a lane on its own -- ballot, and wherever a wave had a candidate (four times in five at
        // real densities) an LDS atomic whose result the wave waits for: 32 serial LDS round trips per wave and tile.
        // Now: a lane counts its own, an inclusive scan over the wave, one atomicAdd of the wave's total.  (The list's
        // order inside a tile changes; nothing depends on it.)
        {
          const int cnt = __popc(cmask);
          int pre = cnt;
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(pre, d);
            pre += lane >= d ? t : 0;
          }
          const int total = __shfl(pre, 63);
          if (total) {   // (wave-uniform)
            int off = 0;
            if (lane == 63) off = atomicAdd(&s_cnt, total);
            off = __shfl(off, 63) + pre - cnt;
            for (unsigned mk = cmask; mk; mk &= mk - 1) {
              const int i = __ffs((int)mk) - 1;
              s_list[off++] = (unsigned short)((m << 6) | ((i >> 4) * 32 + ((i >> 2) & 3) * 8 + 4 * half + (i & 3)));
            }
          }
        }
        FPC_LDS_BARRIER();
        const int ncl = s_cnt;
        if (tid == 0 && ncl) s_base = atomicAdd(&a.ncand[b - a.frame0], ncl);   // ONE global atomic per tile
        FPC_LDS_BARRIER();
        if (ncl) {
          uint32_t* dst = a.cand + (size_t)(b - a.frame0) * a.Ho * 8 * W8 + s_base;
          for (int i = tid; i < ncl; i += NT) {
            const int id = s_list[i], mm = id >> 6, cc = id & 63;
            int my, mx;
            bf_pixel<TH, TW, S>(mm, my, mx);
            const int yy = oyb + my, xx = oxb + mx;
            dst[i] = (uint32_t)((8 * yy + (cc >> 3)) * W8 + 8 * xx + (cc & 7));
          }
        }
        // (the branches above are block boundaries, where the compiler waits for every outstanding request: the next
        // tile's first halo chunk is requested here, after them)
        load_chunk(nxt, 0);
      }
    } else if (a.out_f32) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int m = ml + mb * 32;
        int py, px;
        bf_pixel<TH, TW, S>(m, py, px);
        const int y = oyb + py, x = oxb + px;
        const bool live = (m < TH * TW) & (y < a.Ho) & (x < a.Wo);
        const unsigned obase = (unsigned)((b * a.OH + y * a.oys + a.oy0) * a.OW + x * a.oxs + a.ox0) * (unsigned)(a.cso * 4);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n0 = nl + nb * 32 + 8 * g;
            const float4 bias = *reinterpret_cast<const float4*>(bl + n0);
            f32x4b v;
            v.x = fmaxf(acc[mb][nb][4 * g + 0] + bias.x, floor_);
            v.y = fmaxf(acc[mb][nb][4 * g + 1] + bias.y, floor_);
            v.z = fmaxf(acc[mb][nb][4 * g + 2] + bias.z, floor_);
            v.w = fmaxf(acc[mb][nb][4 * g + 3] + bias.w, floor_);
            const bool on = live & (CMIDP == C::N || n0 < CMIDP);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, (int)(on ? obase + n0 * 4 : 0xfffffff0u), 0, 0);
          }
      }
    } else {
      FPC_LDS_BARRIER();   // phase 2 (or the last chunk) has read the region
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int m = ml + mb * 32;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n0 = nl + nb * 32 + 8 * g;
            const float4 bias = *reinterpret_cast<const float4*>(bl + n0);
            const float v0 = fmaxf(acc[mb][nb][4 * g + 0] + bias.x, floor_), v1 = fmaxf(acc[mb][nb][4 * g + 1] + bias.y, floor_);
            const float v2 = fmaxf(acc[mb][nb][4 * g + 2] + bias.z, floor_), v3 = fmaxf(acc[mb][nb][4 * g + 3] + bias.w, floor_);
            if (CMIDP == C::N || n0 < CMIDP)
              *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(lds16) + m * (ROWH16 * 16) + n0 * 2) =
                  make_uint2(pk_bf16(v0, v1), pk_bf16(v2, v3));
          }
      }
      FPC_LDS_BARRIER();
      constexpr int C8 = CMIDP / 8;
      constexpr int NE = TH * TW * C8, EIT = (NE + NT - 1) / NT;
      // (element e = tid + i NT: 16 bytes = channels 8 c8 ..+7 of tile pixel m = e / C8.  With the blocked pixel order and
      // NT / C8 = 8, 16 or 32 pixels per step, c8 and the pixel's place inside its 4 x 8 block are the thread's own and the
      // step only adds compile-time rows / columns: one offset per thread + a scalar per step, as in load_chunk)
      constexpr int MS = NT % C8 == 0 ? NT / C8 : 0;
      constexpr bool OCARRY = bf_blocked(TH, TW, S) && NE % NT == 0 && (MS == 8 || MS == 16 || MS == 32);
      int tl = tid;
      asm volatile("" : "+v"(tl));
      if constexpr (OCARRY) {
        constexpr int BX = TW / 8;
        const int m0 = tl / C8, c8 = tl - m0 * C8;
        const int yb = oyb + (m0 >> 3), xb = oxb + (m0 & 7);
        const unsigned obase = (unsigned)((b * a.OH + yb * a.oys + a.oy0) * a.OW + xb * a.oxs + a.ox0) * (unsigned)(a.cso * 2) + (unsigned)(c8 * 16);
        const int lbase = m0 * ROWH16 + c8;
#pragma unroll
        for (int i = 0; i < EIT; ++i) {
          const int blk = (i * MS) >> 5, ci = (i * MS) & 31;
          const int cy = (blk / BX) * 4 + (ci >> 3), cx = (blk % BX) * 8;
          const bool on = (yb + cy < a.Ho) & (xb + cx < a.Wo);
          const unsigned off = obase + (unsigned)(cy * a.oys * a.OW + cx * a.oxs) * (unsigned)(a.cso * 2);
          __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&lds16[lbase + i * MS * ROWH16]), orsrc, (int)(on ? off : 0xfffffff0u), 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < EIT; ++i) {
          const int e = tl + i * NT;
          const int m = e / C8, c8 = e - m * C8;
          int py, px;
          bf_pixel<TH, TW, S>(m, py, px);
          const int y = oyb + py, x = oxb + px;
          const bool on = (NE % NT == 0 || e < NE) & (y < a.Ho) & (x < a.Wo);
          const unsigned off = (unsigned)((b * a.OH + y * a.oys + a.oy0) * a.OW + x * a.oxs + a.ox0) * (unsigned)(a.cso * 2) + (unsigned)(c8 * 16);
          const int mm = (NE % NT == 0 || e < NE) ? m : 0;
          __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&lds16[mm * ROWH16 + c8]), orsrc, (int)(on ? off : 0xfffffff0u), 0, 0);
        }
      }
    }
  }
  FPC_STAMP(5)
  FPC_RSTAMP(7)
#ifdef FPC_DIAG
  if (a.stamps && threadIdx.x == 0) {
    unsigned long long t0_, r0_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_), "=s"(r0_)::"memory");
    a.stamps[((size_t)blockIdx.x + 32768) * 8 + 2] = t0_;
    a.stamps[((size_t)blockIdx.x + 32768) * 8 + 3] = r0_;
    a.stamps[((size_t)blockIdx.x + 32768) * 8 + 4] += 1;
  }
#endif
  cur = nxt;
  }
}

template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB, int CMIDP>
__global__ __launch_bounds__(WM* WN * 64, 2) void block_bf16_kernel(const BlockBfArgs a) {
  block_bf16_body<TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP, 0>(a);
}
// the one-chunk form (3x3 blocks whose Cin_pad == KC: layer1, detector.layer.1)
template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB, int CMIDP>
__global__ __launch_bounds__(WM* WN * 64, 2) void block_bf16_one_kernel(const BlockBfArgs a) {
  block_bf16_body<TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP, 1>(a);
}
// the two-chunk form (Cin_pad == 2 KC: layer_out.0, layer_in.1 on 128-channel chunks)
template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB, int CMIDP>
__global__ __launch_bounds__(WM* WN * 64, 2) void block_bf16_two_kernel(const BlockBfArgs a) {
  block_bf16_body<TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP, 2>(a);
}

}  // namespace fpc
