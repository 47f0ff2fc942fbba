// FPC_BF16's stem: encoder.conv1 (7x7/2, BN folded) + ReLU + MaxPool2d(3, 2, 1) in one launch, bf16 out.
// Reference: python/src/superpoint.py:11-14,20-23 (Encoder.conv1 / bn1 / relu / max_pool).
//
// Round 3 rebuild of round 2's stem_pool_bf16_kernel.  Its ablations put half of its time in
// the K loop for a quarter of it in MFMAs, and the model below says why -- per wave and step of four MFMAs (128 cycles
// of its SIMD's matrix core) the old loop asked the LDS for
//     8 x ds_read_b32   pixel fragments: 128 B/clk, banks mod 32, and with 17 pixels per tile row at a pitch of 44
//                       dwords lanes 12-16 and 17-21 of a half-wave share banks: 4 cycles each      = 32 cycles
//     2 x ds_read_b128  weight fragments                                                            =  8 cycles
// i.e. 40 LDS cycles per 128 MFMA cycles and SIMD, 160 % of the LDS for the CU's four SIMDs.  Here:
//   * the weights live in REGISTERS for the whole persistent workgroup (11 steps x 2 channel blocks x 16 bytes = 88
//     VGPRs; two waves per SIMD leave 256);
//   * a pixel's eight K values (seven taps of one filter row + a zero-weight pad in front) start at an even image
//     column = a 4-byte boundary that is 8-byte aligned for every OTHER pixel of a row, so the window is staged twice,
//     the second copy one dword further on, and every lane reads two aligned ds_read_b64 (256 B/clk, banks mod 64)
//     from the copy its pixel's column parity picks;
//   * lane -> pixel is chosen for the banks, not row-major: a block of 32 pixels = four tile rows x eight pixels of one
//     column parity; consecutive lanes are 8 bytes apart, the next tile row (two window rows on) starts 48 dwords
//     later = bank 48, 32, 16: the 32 lanes of a half-wave tile the 64 banks exactly.  (The odd-parity blocks have
//     seven pixels per row; their eighth lanes take the tile's 17th row, two-way conflicts there.)
//   => 2 x 2 x 2 = 8 LDS cycles per step and wave instead of 40.
//   * the tile: 8 x 7 pooled pixels over 17 x 15 convolution outputs = 255 of the 256 pixel slots of four waves x two
//     blocks (the old 17 x 17 = 289 on five waves x two blocks used 289 of 320, and ten waves on four SIMDs are 3 + 3
//     + 2 + 2); 240 x 320 pooled = 30 x 46 tiles with 0.6 % overhang.
//   * window and tile in separate LDS regions: two barriers per tile instead of four.
// Arithmetic, fragments and K order are stem_pool_bf16_kernel's (the same packed blob; results bit-identical).
#pragma once
#include "block_x3.h"

namespace fpc {

constexpr int SB2_PH = 8, SB2_PW = 7;           // pooled rows x columns of a tile
constexpr int SB2_R = 2 * SB2_PH + 1;           // 17 convolution rows under them
constexpr int SB2_Q = 2 * SB2_PW + 1;           // 15 convolution columns
constexpr int SB2_ROWS = 2 * (SB2_R - 1) + 7;   // 39 window rows
constexpr int SB2_NQ = 10;                      // float4 per window row: image columns 28 tx - 8 .. 28 tx + 31
constexpr int SB2_PITCH = 24;                   // dwords per window row in LDS (2 * PITCH = 48 = -16 mod 64: see above)
constexpr int SB2_THREADS = 256;
constexpr int SB2_PIX = 128;                    // bytes per pixel slot of the bf16 tile: 64 channels, no padding (see sb2_tile_addr)
constexpr int SB2_TILE_BYTES = SB2_R * 16 * SB2_PIX;
constexpr int SB2_OPAD = 40;                    // bytes between the window's two copies (chosen with the spare-lane table below)

// max of two packed pairs of bf16 as signed 16-bit integers: among non-negative floats the integer order is the float
// order, every negative float is below every non-negative one, and a window of negatives only has to come out negative
// (the ReLU that follows is one more packed max with 0)
__device__ __forceinline__ unsigned stemb_pk_max(unsigned x, unsigned y) {
  unsigned r;
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// ABL: phases removed for the ablation runs of DESIGN.md section 3.7 -- a TEMPLATE argument that only
// experiments/harness/stem_bf16_bench.hip ever sets (the library instantiates <CIN> alone, so no -D can make the
// product skip work).
enum : unsigned { STEMB_ABL_LOAD = 1, STEMB_ABL_K = 2, STEMB_ABL_TILE = 4, STEMB_ABL_POOL = 8, STEMB_ABL_STORE = 16 };

template <int CIN>
struct StemB2Cfg {
  static constexpr int ROWS = CIN * 7, STEPS = (ROWS + 1) / 2;   // filter rows (c, ky); two per MFMA step
  static constexpr int COPY_BYTES = CIN * SB2_ROWS * SB2_PITCH * 4;
  static constexpr int COPY1 = COPY_BYTES + SB2_OPAD + 4;        // copy 1 holds dword d of a row at d + 1
  static constexpr int WIN_BYTES = 2 * COPY_BYTES + SB2_OPAD + 8;
  static constexpr int LDS_BYTES = WIN_BYTES + SB2_TILE_BYTES + 256 + 16;   // window x 2 | tile | 64 biases | a spare slot
  // staging: thread t < 250 moves float4 (t % 10) of window rows t / 10 + 25 i, i < IT
  static constexpr int WROWS = CIN * SB2_ROWS, RPP = 25, IT = (WROWS + RPP - 1) / RPP;
};

// The odd-parity blocks' eighth lanes (16 of them) take the 15 pixels of the tile's 17th row: spare position s = 4 wave
// + row-in-block -> column (15: nobody's; a second copy of column 14 into a slot nothing reads).  Found by a random
// search over the LDS bank model (scratch: K-loop reads + tile writes).
constexpr unsigned long long SB2_SPARE_COL = 0x23f47d0b6ca5198eull;   // 4 bits per spare position

// The bf16 tile in LDS.  Pixel (r, q) sits in slot n = 16 r + perm(q) whose PARITY is bit 2 of q, 128 bytes per slot,
// and its eight 16-byte channel groups are XOR-swizzled with s(r, q), bit 2 of s = bit 2 of q again.  The pooling's
// ds_read_b128 is served in groups of 16 lanes {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS): with 8 lanes per
// pixel (channel group = lane & 7) and the four pixels of a half-wave at columns q, q + 4, q + 8, q + 12, a group takes
// channel groups 0-3 of two pixels of opposite slot parity and 4-7 of the two others -- exactly the 64 banks, whatever
// the low bits of s are.  Those spread the K-loop lanes' 8-byte tile writes (16 lanes, banks mod 32) two-way, the
// least a half-block that writes ONE 8-byte half of every 16-byte group can have.
__device__ __forceinline__ int sb2_swz(int r, int q) { return (((q >> 2) & 1) << 2) | ((r & 1) << 1) | ((q >> 1) & 1); }
__device__ __forceinline__ int sb2_slot(int r, int q) { return r * 16 + ((q >> 3) * 4 + (q & 3)) * 2 + ((q >> 2) & 1); }
// byte offset of channel group cg (8 channels) of pixel (r, q)
__device__ __forceinline__ int sb2_tile_addr(int r, int q, int cg) { return sb2_slot(r, q) * SB2_PIX + 16 * (cg ^ sb2_swz(r, q)); }

// (diagnostic build only: wave 0 of every workgroup records the shader clock at the phase boundaries of its TENTH tile --
// a tile in the steady state, both workgroups of the CU at work -- experiments/harness/stem_bf16_bench.hip -DFPC_DIAG)
#ifdef FPC_DIAG
#define SB2_STAMP(i)                                                                     \
  if (a.stamps && threadIdx.x == 0 && ntile == 10) {                                     \
    unsigned long long t_;                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
    a.stamps[(size_t)blockIdx.x * 8 + (i)] = t_;                                         \
  }
#else
#define SB2_STAMP(i)
#endif

template <int CIN, unsigned ABL = 0>
__global__ __launch_bounds__(SB2_THREADS, 2) void stem_bf16_kernel(const StemX3Args a) {
  using C = StemB2Cfg<CIN>;
  constexpr int ROWS = C::ROWS, STEPS = C::STEPS, IT = C::IT;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* tile = lds_raw + C::WIN_BYTES;
  float* bias_lds = reinterpret_cast<float*>(lds_raw + C::WIN_BYTES + SB2_TILE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y;
  // XCD k (workgroups with blockIdx.x & 7 == k) walks the tiles [k T / 8, (k + 1) T / 8) of the launch
  const int T = tiles * a.frames, per = gridDim.x >> 3, slot0 = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  const int t_end = (int)(((long long)(xcd + 1) * T) >> 3);
  int tcur = (int)(((long long)xcd * T) >> 3) + slot0;

  if (tid < 64) bias_lds[tid] = a.bias[tid];
  uint4 wreg[STEPS][2];   // this lane's weight fragments of every step: registers for the workgroup's whole life
#pragma unroll
  for (int s = 0; s < STEPS; ++s)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) wreg[s][nb] = a.wfrag[(s * 2 + nb) * 64 + lane];

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in), 0, (int)((unsigned)a.frames * CIN * a.H * a.W * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.out, 0, (int)((unsigned)a.frames * a.Hp * a.Wp * 128u), 0x00020000);
  f32x4 v[IT];
  const int srow = tid / SB2_NQ, sq4 = tid - srow * SB2_NQ;   // staging role: window row (+ 25 per pass), float4 of the row
  const bool sact = tid < C::RPP * SB2_NQ;
  auto request = [&](int tt) {  // the input window of tile tt as aligned float4 row segments (W is a multiple of 8: a
                                // float4 is entirely inside or outside the frame); outside -> zeros from the bounds check
    const bool live = tt < t_end;
    const int tc = live ? tt : 0;
    const int b = tc / tiles, t = tc - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int iy0 = ty * (4 * SB2_PH) - 5, ixa = tx * (4 * SB2_PW) - 8;  // image row / column of window row / column 0
    const int hlim = live ? a.H : 0;
    const unsigned tbase = (unsigned)((b * CIN * a.H + iy0) * a.W + ixa) * 4u;   // unsigned: mod 2^32, exact for in-frame pixels
    const bool xok = sact & ((unsigned)(ixa + 4 * sq4) < (unsigned)a.W);
    unsigned toff = tbase + (unsigned)((srow * a.W + 4 * sq4) * 4);
    asm volatile("" : "+v"(toff));
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int row = srow + i * C::RPP;                                        // window row c * 39 + hy
      const int c = (row >= SB2_ROWS ? 1 : 0) + (row >= 2 * SB2_ROWS ? 1 : 0);
      const int iy = iy0 + row - c * SB2_ROWS;
      const bool ok = xok & (row < C::WROWS) & ((unsigned)iy < (unsigned)hlim);
      const unsigned off = toff + (unsigned)((i * C::RPP + c * (a.H - SB2_ROWS)) * a.W * 4);
      v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(ok ? off : 0xfffffff0u), 0, 0));
    }
  };
  request(tcur);
  // (two stores that the bounds check drops: the loop is then entered with the SAME queue of outstanding requests as
  // its back edge has -- five loads, two stores behind them -- and the compiler's wait for the loads is vmcnt(6..2)
  // on both paths; with five outstanding on one and seven on the other it takes vmcnt(4..0), i.e. waits for the
  // previous tile's stores to be acknowledged at the head of every tile)
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
    __builtin_amdgcn_raw_buffer_store_b128(z, orsrc, (int)0xfffffff0u, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(z, orsrc, (int)0xffffffe0u, 0, 0);   // (another offset: two equal stores are merged)
  }

  // this lane's two pixels (block mb = column parity): tile row / column, byte address of its K values in the window
  // (copy 0 holds dword d of a row at d, copy 1 at d + 1; a pixel's values start at dword q + 1), slot in the tile
  int prow[2], pcol[2], abase[2], abase8[2], tbase[2], tswz[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int i = l31 >> 3, j = l31 & 7;
    int r = 4 * wave + i, q = 2 * j + mb;
    bool nobody = false;
    if (mb == 1 && j == 7) {   // the spare lanes of the odd-parity blocks: the tile's 17th row
      q = (int)((SB2_SPARE_COL >> (4 * (4 * wave + i))) & 15ull);
      r = SB2_R - 1;
      nobody = q == 15;
      q = nobody ? SB2_Q - 1 : q;
    }
    prow[mb] = r;
    pcol[mb] = q;
    abase[mb] = ((q & 1) ? 0 : C::COPY1) + ((2 * r) * SB2_PITCH + q + 1) * 4;
    // (the second 8 bytes through an address the compiler cannot see next to the first: it would merge the two reads
    // into ds_read2_b64 -- 8 cycles, banks mod 32 in groups of 16 lanes -- where two ds_read_b64 take 2 x 2, banks mod 64)
    abase8[mb] = abase[mb] + 8;
    asm volatile("" : "+v"(abase8[mb]));
    tbase[mb] = (nobody ? (SB2_R - 1) * 16 + 15 : sb2_slot(r, q)) * SB2_PIX + 8 * half;
    tswz[mb] = sb2_swz(r, q);
  }
  // pooling role of this thread: pooled row pj, channels 8 pc .. 8 pc + 7, pooled columns 2 pg and 2 pg + 1
  const int pc = tid & 7, pg = (tid >> 3) & 3, pj = tid >> 5;

  int ntile = 0;
  for (; tcur < t_end; tcur += per, ++ntile) {
    const int b = tcur / tiles, t = tcur - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    SB2_STAMP(0)
    // (every wave is past the previous tile's K loop -- the barrier in front of its pooling -- so the window is free)
    {
      // (no branch anywhere in the tile loop: behind one the compiler waits for EVERY outstanding request, the
      // previous tile's stores among them; an element past the window goes to a spare slot)
      constexpr int SPARE = C::WIN_BYTES + SB2_TILE_BYTES + 256;
      const int pe0 = (srow * SB2_PITCH + 2 * sq4) * 4;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const unsigned lo = pk_bf16(v[i].x, v[i].y), hi = pk_bf16(v[i].z, v[i].w);
        const bool in = sact & (srow + i * C::RPP < C::WROWS);
        const int pe = in ? pe0 + i * (C::RPP * SB2_PITCH * 4) : SPARE, po = in ? pe0 + i * (C::RPP * SB2_PITCH * 4) + C::COPY1 : SPARE + 8;
        *reinterpret_cast<uint2*>(lds_raw + pe) = make_uint2(lo, hi);
        unsigned* o = reinterpret_cast<unsigned*>(lds_raw + po);
        o[0] = lo;
        o[1] = hi;
      }
    }
    SB2_STAMP(1)
    __syncthreads();   // window complete; the previous tile's pooling reads of the tile region are done as well
    SB2_STAMP(2)
    if constexpr (!(ABL & STEMB_ABL_LOAD)) request(tcur + per);  // lands behind the K loop and the epilogue

    // accumulators start at the bias; a pixel of the 17 x 15 outside the convolution's output starts (and stays) hugely
    // negative, so the max-pool ignores it as it ignores MaxPool2d's padding
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
      // this lane's filter row: (c, ky); a padded row (weights zero) re-reads the last real one
      constexpr int RB = SB2_PITCH * 4;
      const int r0 = 2 * s < ROWS ? 2 * s : ROWS - 1, r1 = 2 * s + 1 < ROWS ? 2 * s + 1 : ROWS - 1;
      const int off0 = ((r0 / 7) * SB2_ROWS + (r0 % 7)) * RB, off1 = ((r1 / 7) * SB2_ROWS + (r1 % 7)) * RB;
      const int off = half ? off1 : off0;
      uint4 av[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const uint2 lo = *reinterpret_cast<const uint2*>(__builtin_assume_aligned(lds_raw + abase[mb] + off, 8));
        const uint2 hi = *reinterpret_cast<const uint2*>(__builtin_assume_aligned(lds_raw + abase8[mb] + off, 8));
        av[mb] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) mfma_split<1>(acc[mb][nb], &wreg[s][nb], &av[mb]);   // weights as A: the tile comes out transposed
    }

    SB2_STAMP(3)
    // the tile -> LDS as bf16: [slot][64 channels], 8 bytes per write (a lane holds one pixel, four groups of four channels per block)
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
    SB2_STAMP(4)
    __syncthreads();   // tile complete (and every wave has finished reading the window)
    SB2_STAMP(5)
    // 3x3/2 max-pool + ReLU on packed bf16 pairs as signed 16-bit integers (stem_pool_bf16_kernel's argument)
    const int gpy = ty * SB2_PH + pj;
    if constexpr (!(ABL & STEMB_ABL_POOL)) {
      u32x4 cm[5];
#pragma unroll
      for (int cc = 0; cc < 5; ++cc) {
        // (pg = 3: its second pooled column does not exist; columns "15" and "16" read pixels of the slot parity the
        // bank argument above needs, nobody uses the result)
        const int q = 4 * pg + cc < SB2_Q ? 4 * pg + cc : 4 * pg + cc == SB2_Q ? 14 : 11;
        // rows 2 pj, + 1, + 2 of column q: 16 slots on, and bit 1 of the swizzle follows the row's parity
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
        const bool on = (!(ABL & STEMB_ABL_STORE) || o.x == 0x12345678u) & (gpy < a.Hp) & (2 * pg + px < SB2_PW) & (gpx + px < a.Wp);
        __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, (int)(on ? ooff + px * 128 : 0xfffffff0u), 0, 0);   // a dead pixel's store: dropped by the bounds check
      }
    }
    SB2_STAMP(6)
  }
}

}  // namespace fpc
