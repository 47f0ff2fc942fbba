// Round 2's bf16 stem (stem_pool_bf16_kernel), replaced in the library by stem_bf16_kernel (csrc/stem_bf16.h, round 3).
// Kept here, built by nothing but experiments/harness/stem_bf16_bench.hip, which runs the two side by side and compares
// the pooled maps bit for bit.
#pragma once
#include "../feature-point-cnn_amd/csrc/stem_bf16.h"

namespace fpc {

// ---------------------------------------------------------------------------------
// FPC_BF16's stem: 7x7/2 convolution + bias + ReLU + 3x3/2 max-pool, bf16 in, bf16 out, every pooled value written
// ONCE.  stem_pool_x3_kernel completes the pooling windows that straddle two tiles with atomicMax on an fp32 buffer:
// at 64 HD frames that is a 0.94 GB memset, a read-modify-write of the same bytes, and the next layer reading 4-byte
// values (2.5 GB of HBM traffic per launch by the PMC counters, for 0.7 GB of frames in and 0.47 GB of bf16 out).
// Here a tile is 8 x 8 POOLED pixels and the 17 x 17 convolution outputs under them (one row above and one column
// left of the 16 x 16 the old tile had: 13 % more MFMA work on a kernel whose matrix cores were 18 % busy), as ten
// 32-pixel blocks on five waves.
//
// What bounds this kernel is the number of instructions per output, not memory and not the matrix cores (ablations
// in a stand-alone harness: the pooling alone was 36 % of its time).  So: the MFMAs take the weights as the A operand,
// which leaves a lane with ONE pixel and four groups of four consecutive channels per 32-channel block; the bias is
// the accumulators' initial value (minus a huge number for the pixels of the 17 x 17 that lie outside the convolution's
// output, so that MaxPool2d's padding never wins); the tile goes to LDS as bf16, 8 bytes per write (rounding to bf16
// is monotone, so it commutes with the max); the max-pool runs on packed pairs of bf16 AS SIGNED 16-BIT INTEGERS
// (v_pk_max_i16: among non-negative floats the integer order is the float order, every negative float is below every
// non-negative one, and a window of negatives only has to come out negative) and ReLU is one more packed max with 0 on
// the pooled value; a thread pools 8 channels of two pooled pixels from 15 sixteen-byte LDS reads and stores 16 bytes
// per pixel.  The grid is persistent (two workgroups per CU, each XCD walking a contiguous range of tiles so that
// neighbours share their input halos in one L2): the weight fragments are staged in LDS once per workgroup, and the
// next tile's input window is requested before the current tile's K loop.  K layout and fragments: stem_pool_x3_kernel's.
// ---------------------------------------------------------------------------------
constexpr int STEMB_C = 17;                        // convolution rows / columns per tile
constexpr int STEMB_ROWS = 2 * (STEMB_C - 1) + 7;  // 39 input rows
constexpr int STEMB_LW = 44;                       // bf16 per LDS row: image columns 32 tx - 8 .. 32 tx + 35
constexpr int STEMB_THREADS = 320;
constexpr int STEMB_M = 320;                       // 289 real pixels in ten blocks of 32
constexpr int STEMB_PIX = 144;                     // bytes per pixel of the bf16 tile (64 channels + 16 of skew)
constexpr int STEMB_TILE_BYTES = STEMB_M * STEMB_PIX;  // 46 KB; the input window (10 KB) uses the same region first

template <int CIN>
struct StemBCfg {
  static constexpr int ROWS = CIN * 7, STEPS = (ROWS + 1) / 2;  // 21 -> 11 steps; 7 -> 4
  static constexpr int W_BYTES = STEPS * 2 * 64 * 16;
  static constexpr int LDS_BYTES = STEMB_TILE_BYTES + W_BYTES + 256;   // tile | weight fragments | 64 biases
  static constexpr int NQ = STEMB_LW / 4, NE = CIN * STEMB_ROWS * NQ, IT = (NE + STEMB_THREADS - 1) / STEMB_THREADS;
  static_assert(CIN * STEMB_ROWS * STEMB_LW * 2 <= STEMB_TILE_BYTES, "input window fits in the tile's LDS");
};

template <int CIN, unsigned ABL = 0>
__global__ __launch_bounds__(STEMB_THREADS) void stem_pool_bf16_kernel(const StemX3Args a) {
  using C = StemBCfg<CIN>;
  constexpr int ROWS = C::ROWS, STEPS = C::STEPS, IT = C::IT, NQ = C::NQ, NE = C::NE;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const unsigned* lds32 = reinterpret_cast<const unsigned*>(lds_raw);
  uint4* wl = reinterpret_cast<uint4*>(lds_raw + STEMB_TILE_BYTES);  // [STEPS][2 nb][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y;
  // XCD k (workgroups with blockIdx.x & 7 == k) walks the tiles [k T / 8, (k + 1) T / 8) of the launch
  const int T = tiles * a.frames, per = gridDim.x >> 3, slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  const int t_end = (int)(((long long)(xcd + 1) * T) >> 3);
  int tcur = (int)(((long long)xcd * T) >> 3) + slot;

  for (int i = tid; i < STEPS * 2 * 64; i += STEMB_THREADS) wl[i] = a.wfrag[i];
  float* bias_lds = reinterpret_cast<float*>(lds_raw + STEMB_TILE_BYTES + C::W_BYTES);
  if (tid < 64) bias_lds[tid] = a.bias[tid];

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in), 0, (int)((unsigned)a.frames * CIN * a.H * a.W * 4u), 0x00020000);
  f32x4 v[IT];
  auto request = [&](int tt) {  // the input window of tile tt as aligned float4 row segments (W is a multiple of 8: a
                                // float4 is entirely inside or outside the frame); outside -> zeros from the bounds check
    const bool live = tt < t_end;
    const int tc = live ? tt : 0;
    const int b = tc / tiles, t = tc - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int iy0 = ty * 32 - 5, ixa = tx * 32 - 8;  // image row of LDS row 0, image column of LDS column 0
    const int hlim = live ? a.H : 0;
    const unsigned tbase = (unsigned)((b * CIN * a.H + iy0) * a.W + ixa) * 4u;   // unsigned: mod 2^32, exact for in-frame pixels
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      // (per-thread constants of element i -- plane, row, column -- are loop invariants the compiler keeps in
      // registers: computed per tile they were ~40 VALU instructions per element, two divisions by constants)
      const int e = tid + i * STEMB_THREADS;
      const int row = e / NQ, q = e - row * NQ;
      const int c = row / STEMB_ROWS, hy = row - c * STEMB_ROWS;
      const int iy = iy0 + hy, ix = ixa + 4 * q;
      const bool ok = (e < NE) & ((unsigned)iy < (unsigned)hlim) & ((unsigned)ix < (unsigned)a.W);
      unsigned off = tbase + (unsigned)(((c * a.H + hy) * a.W + 4 * q) * 4);
      asm volatile("" : "+v"(off));
      v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(ok ? off : 0xfffffff0u), 0, 0));
    }
  };
  request(tcur);

  // pixel m of the tile = convolution output (row 16 ty - 1 + m / 17, column 16 tx - 1 + m % 17); its 8 K-values of a
  // filter row start at LDS column 2 (m % 17) + 2 (the zero-weight pad in front), LDS row 2 (m / 17) + ky
  int abase[2], prow[2], pcol[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    int m = (wave * 2 + mb) * 32 + l31;
    m = m < STEMB_C * STEMB_C ? m : STEMB_C * STEMB_C - 1;
    prow[mb] = m / STEMB_C;
    pcol[mb] = m % STEMB_C;
    abase[mb] = (2 * prow[mb]) * (STEMB_LW / 2) + pcol[mb] + 1;  // dwords
  }
  // pooling role of this thread: pooled row pj, channels 8 pc .. 8 pc + 7, pooled columns 2 pg and 2 pg + 1
  const int pc = tid & 7, pg = (tid >> 3) & 3, pj = tid >> 5;

  for (; tcur < t_end; tcur += per) {
    const int b = tcur / tiles, t = tcur - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    __syncthreads();  // the previous tile's pooling has read the region (first tile: the weights are in LDS)
    {
      uint2* lds64 = reinterpret_cast<uint2*>(lds_raw);
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int e = tid + i * STEMB_THREADS;
        const int slot_ = e < NE ? e : NE;  // (one spare 8-byte slot behind the window: no branch)
        lds64[slot_] = make_uint2(f2bf(v[i].x) | ((unsigned)f2bf(v[i].y) << 16), f2bf(v[i].z) | ((unsigned)f2bf(v[i].w) << 16));
      }
    }
    __syncthreads();
    if constexpr (!(ABL & STEMB_ABL_LOAD)) request(tcur + per);  // lands behind the K loop and the epilogue

    // accumulators start at the bias; a pixel of the 17 x 17 outside the convolution's output starts (and stays) hugely
    // negative, so the max-pool ignores it as it ignores MaxPool2d's padding
    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int gy = ty * 16 - 1 + prow[mb], gx = tx * 16 - 1 + pcol[mb];
      const bool inside = ((unsigned)gy < (unsigned)a.Ho) & ((unsigned)gx < (unsigned)a.Wo);
      // (biases from LDS, 16 bytes at a time: channels nb * 32 + 8 g + 4 half + j  <->  register 4 g + j of block nb; held in
      // registers across tiles they cost the second resident workgroup of the CU)
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
      const int r0 = 2 * s < ROWS ? 2 * s : ROWS - 1, r1 = 2 * s + 1 < ROWS ? 2 * s + 1 : ROWS - 1;
      const int off0 = ((r0 / 7) * STEMB_ROWS + (r0 % 7)) * (STEMB_LW / 2), off1 = ((r1 / 7) * STEMB_ROWS + (r1 % 7)) * (STEMB_LW / 2);
      const int off = half ? off1 : off0;
      uint4 av[2], bw[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const unsigned* q = lds32 + abase[mb] + off;
        av[mb] = make_uint4(q[0], q[1], q[2], q[3]);
      }
      bw[0] = wl[(s * 2 + 0) * 64 + lane];
      bw[1] = wl[(s * 2 + 1) * 64 + lane];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) mfma_split<1>(acc[mb][nb], &bw[nb], &av[mb]);   // weights as A: the tile comes out transposed
    }

    // the tile -> LDS as bf16: [pixel m][64 channels], 8 bytes per write
    __syncthreads();  // every wave has read its pixels of the input window
#pragma unroll
    for (int mb = 0; mb < ((ABL & STEMB_ABL_TILE) ? 0 : 2); ++mb) {
      unsigned char* row = lds_raw + ((wave * 2 + mb) * 32 + l31) * STEMB_PIX + 8 * half;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<uint2*>(row + nb * 64 + g * 16) =
              make_uint2(f2bf(acc[mb][nb][4 * g]) | ((unsigned)f2bf(acc[mb][nb][4 * g + 1]) << 16),
                         f2bf(acc[mb][nb][4 * g + 2]) | ((unsigned)f2bf(acc[mb][nb][4 * g + 3]) << 16));
    }
    __syncthreads();
    // 3x3/2 max-pool + ReLU on packed bf16 pairs (as signed 16-bit integers, see above)
    const int gpy = ty * 8 + pj;
    if (!(ABL & STEMB_ABL_POOL) && pj < 8 && gpy < a.Hp) {
      u32x4 cm[5];
#pragma unroll
      for (int cc = 0; cc < 5; ++cc) {
        const unsigned char* q = lds_raw + ((2 * pj) * STEMB_C + 4 * pg + cc) * STEMB_PIX + pc * 16;
        const u32x4 r0 = *reinterpret_cast<const u32x4*>(q), r1 = *reinterpret_cast<const u32x4*>(q + STEMB_C * STEMB_PIX),
                    r2 = *reinterpret_cast<const u32x4*>(q + 2 * STEMB_C * STEMB_PIX);
        cm[cc].x = stemb_pk_max(stemb_pk_max(r0.x, r1.x), r2.x);
        cm[cc].y = stemb_pk_max(stemb_pk_max(r0.y, r1.y), r2.y);
        cm[cc].z = stemb_pk_max(stemb_pk_max(r0.z, r1.z), r2.z);
        cm[cc].w = stemb_pk_max(stemb_pk_max(r0.w, r1.w), r2.w);
      }
      unsigned short* orow = reinterpret_cast<unsigned short*>(a.out) + ((size_t)(b * a.Hp + gpy) * a.Wp + tx * 8 + 2 * pg) * 64 + pc * 8;
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        u32x4 o;
        o.x = stemb_pk_max(stemb_pk_max(stemb_pk_max(cm[2 * px].x, cm[2 * px + 1].x), cm[2 * px + 2].x), 0u);
        o.y = stemb_pk_max(stemb_pk_max(stemb_pk_max(cm[2 * px].y, cm[2 * px + 1].y), cm[2 * px + 2].y), 0u);
        o.z = stemb_pk_max(stemb_pk_max(stemb_pk_max(cm[2 * px].z, cm[2 * px + 1].z), cm[2 * px + 2].z), 0u);
        o.w = stemb_pk_max(stemb_pk_max(stemb_pk_max(cm[2 * px].w, cm[2 * px + 1].w), cm[2 * px + 2].w), 0u);
        if (!(ABL & STEMB_ABL_STORE) || o.x == 0x12345678u)
        if (tx * 8 + 2 * pg + px < a.Wp) *reinterpret_cast<u32x4*>(orow + px * 64) = o;
      }
    }
  }
}

}  // namespace fpc
