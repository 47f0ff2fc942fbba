// kernels_misc.h -- everything on the path that is not the halo-tile convolution:
// the 7x7/2 stem (MFMA, K = 147), max-pool, exp-softmax + depth-to-space +
// threshold, greedy NMS + sort + border crop, descriptor sampling, layout helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_mfma.h"
#include "nms_word.h"

namespace fpc {

// ---------------------------------------------------------------------------------
// Stem: Conv2d(3, 64, 7, stride 2, padding 3, bias=False) + BN + ReLU
// (python/src/superpoint.py:12-14,20-22).  Input NCHW planar (the layout the
// reference's forward takes), output NHWC.  Implicit GEMM with K = 3*7*7 = 147
// (padded to 152 = 19 groups of 8): one workgroup = 16x16 output pixels x 64
// channels; the 37x37x3 input window sits in LDS and every A operand is one
// ds_read_b32 at a compile-time offset.
// ---------------------------------------------------------------------------------
struct StemArgs {
  const float* in;      // [B,3,H,W]
  const float4* wfrag;  // [19][2][64] float4
  const float* bias;    // [64]
  float* out;           // [B,H/2,W/2,64]
  int H, W, Ho, Wo, tiles_x, tiles_y;
};

constexpr int STEM_T = 16;                       // tile edge (output pixels)
constexpr int STEM_HALO = (STEM_T - 1) * 2 + 7;  // 37
constexpr int STEM_LW = 40;                      // LDS row stride (floats)
constexpr int STEM_KG = 19;                      // groups of 8 k-values
constexpr int STEM_LDS_FLOATS = 3 * STEM_HALO * STEM_LW;

__device__ __forceinline__ constexpr int stem_koff(int k) {
  k = k < 147 ? k : 146;  // padded k: weight is zero, any valid address will do
  return (k / 49) * (STEM_HALO * STEM_LW) + ((k % 49) / 7) * STEM_LW + (k % 7);
}

// K order of the stem's GEMM (round 3).  One v_mfma_f32_32x32x2_f32 takes K = 2: lanes 0-31 supply one filter tap, lanes
// 32-63 another.  With the taps in flat (c, ky, kx) order the two LDS addresses of a step differ by an amount that
// changes from step to step (1 inside a filter row, 34 across its end, ...), i.e. a per-lane select + add in front of
// every read -- two VALU instructions between MFMAs, and on gfx950 a VALU instruction between two fp32 MFMAs costs 15
// cycles (generation 3).  Pairing taps so that the second lies either ONE COLUMN to the right of
// the first or ONE ROW below it leaves two per-lane bases (a + half, a + half * STEM_LW) and a compile-time offset per
// step: no VALU instruction in the K loop.  Per channel: 21 column pairs (kx = 0|1, 2|3, 4|5 of the 7 rows), 3 row pairs
// (kx = 6 of rows 0|1, 2|3, 4|5) and the last tap (6, 6) as the SECOND of a column pair whose first, (6, 5) again, has
// weight zero: 25 steps per channel (the flat order: 24.5).  The host packs the fragments from the same function.
struct StemPair {
  int addr_k;   // tap whose LDS address lanes 0-31 read (flat index c * 49 + ky * 7 + kx)
  int wa, wb;   // taps whose WEIGHTS lanes 0-31 / 32-63 multiply (-1: zero)
  int row;      // 1: lanes 32-63 read one LDS row below lanes 0-31; 0: one column to the right
};
__host__ __device__ constexpr StemPair stem_pair(int cin, int t) {
  const int c = t / 25, i = t % 25;
  if (c >= cin) return StemPair{0, -1, -1, 0};                                        // padding steps of the last group
  if (i < 21) return StemPair{c * 49 + (i / 3) * 7 + 2 * (i % 3), c * 49 + (i / 3) * 7 + 2 * (i % 3), c * 49 + (i / 3) * 7 + 2 * (i % 3) + 1, 0};
  if (i < 24) return StemPair{c * 49 + 2 * (i - 21) * 7 + 6, c * 49 + 2 * (i - 21) * 7 + 6, c * 49 + (2 * (i - 21) + 1) * 7 + 6, 1};
  return StemPair{c * 49 + 6 * 7 + 5, -1, c * 49 + 6 * 7 + 6, 0};
}

__global__ __launch_bounds__(256) void stem_kernel(const StemArgs a) {
  __shared__ float lds[STEM_LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y;
  const int b = blockIdx.x / tiles;
  const int t = blockIdx.x - b * tiles;
  const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
  const int iy0 = ty * STEM_T * 2 - 3, ix0 = tx * STEM_T * 2 - 3;

  for (int e = tid; e < 3 * STEM_HALO * STEM_HALO; e += 256) {
    const int c = e / (STEM_HALO * STEM_HALO);
    const int r = e - c * (STEM_HALO * STEM_HALO);
    const int hy = r / STEM_HALO, hx = r - hy * STEM_HALO;
    const int iy = iy0 + hy, ix = ix0 + hx;
    float v = 0.f;
    if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = a.in[((size_t)(b * 3 + c) * a.H + iy) * a.W + ix];
    lds[(c * STEM_HALO + hy) * STEM_LW + hx] = v;
  }
  __syncthreads();

  int acol[2], arow[2];   // A operand bases: lanes 32-63 one column to the right / one row below (stem_pair)
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int m = (wave * 2 + mb) * 32 + l31;
    const int ab = (2 * (m / STEM_T)) * STEM_LW + 2 * (m % STEM_T);
    acol[mb] = ab + half;
    arow[mb] = ab + half * STEM_LW;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float4* wp = a.wfrag + lane;
#pragma unroll
  for (int g = 0; g < STEM_KG; ++g) {
    const float4 bq0 = wp[(g * 2 + 0) * 64], bq1 = wp[(g * 2 + 1) * 64];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      constexpr int CIN3 = 3;
      const StemPair P = stem_pair(CIN3, g * 4 + j);
      if (P.wa < 0 && P.wb < 0) continue;   // an empty step; the first one carries the bias for stem_pool2_kernel (pack layout revision 4)
      const int off = stem_koff(P.addr_k);
      const float bf0 = j == 0 ? bq0.x : j == 1 ? bq0.y : j == 2 ? bq0.z : bq0.w;
      const float bf1 = j == 0 ? bq1.x : j == 1 ? bq1.y : j == 2 ? bq1.z : bq1.w;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float af = lds[(P.row ? arow[mb] : acol[mb]) + off];
        acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf0, acc[mb][0], 0, 0, 0);
        acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf1, acc[mb][1], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = nb * 32 + l31;
    const float bias = a.bias[n];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (wave * 2 + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int y = ty * STEM_T + m / STEM_T, x = tx * STEM_T + m % STEM_T;
        if (y < a.Ho && x < a.Wo) {
          const float v = acc[mb][nb][r] + bias;
          a.out[((size_t)(b * a.Ho + y) * a.Wo + x) * 64 + n] = v > 0.f ? v : 0.f;
        }
      }
  }
}

// ---------------------------------------------------------------------------------
// Stem + max-pool in one launch (superpoint.py:20-23): the 19.7 MB/frame conv1 output
// never reaches HBM.  Same GEMM as stem_kernel; the epilogue turns each 32-channel half
// of the 16x16 conv tile around through LDS and emits MaxPool2d(3, stride 2, padding 1)
// outputs.  A pooling window that straddles two tiles is completed with atomicMax on the
// float bits -- exact and order-independent because post-ReLU values are >= +0 -- into a
// buffer the host zeroes per call; windows inside one tile are stored directly.
// ---------------------------------------------------------------------------------
struct StemPoolArgs {
  const float* in;      // [B,3,H,W]
  const float4* wfrag;  // [KG + 2][2][64] float4 (two zero groups of prefetch padding); KG = 19 (RGB) / 7 (gray)
  const float* bias;    // [64]
  float* out;           // [B,Hp,Wp,64], zero-filled
  int H, W, Ho, Wo, Hp, Wp, tiles_x, tiles_y;
  uint32_t* range;      // [B][FPC_RANGE_WORDS] per-frame range words (nullable): word RANGE_BAD_INPUT is set to 1 when a pixel of the
                        // frame is NaN or +-Inf (include/fpc.h: FPC_E_NONFINITE)
};

// Per-frame range words the path keeps for its caller (fpc_get_counts, fpc_output_range).
constexpr int FPC_RANGE_WORDS = 4;
enum { RANGE_MAX_LOGIT = 0, RANGE_BAD_INPUT = 1, RANGE_MAX_DESC = 2, RANGE_SPARE = 3 };

// Non-finite pixels, found while the stem stages its window: 0 * (x + y + z + w) is NaN iff one of the four is NaN or
// +-Inf (or their sum overflows: |pixel| > 8e37, no image).  Three VALU instructions per 16-byte load, outside the K loop.
__device__ __forceinline__ float nonfinite_probe(float chk, const float4& x) {
  return fmaf((x.x + x.y) + (x.z + x.w), 0.f, chk);
}

constexpr int STEM_TROW = 33;  // floats per pixel of the epilogue tile (32 channels + 1 skew)
constexpr int STEM_POOL_LDS_FLOATS =
    STEM_LDS_FLOATS > 256 * STEM_TROW ? STEM_LDS_FLOATS : 256 * STEM_TROW;

template <int KREAL>
__device__ __forceinline__ constexpr int stem_koff_c(int k) {
  k = k < KREAL ? k : KREAL - 1;  // padded k: weight is zero, any valid address will do
  return (k / 49) * (STEM_HALO * STEM_LW) + ((k % 49) / 7) * STEM_LW + (k % 7);
}

// MaxPool2d(3, stride 2, padding 1) of one 32-channel half of the 16x16 conv tile held in LDS
// (lds[(row * 16 + col) * STEM_TROW + channel], post-ReLU so every value is >= 0).  Thread (c = tid & 31,
// j = tid >> 5) owns pooled row j of the tile for its channel: it reads the three conv rows once (48 LDS reads),
// reduces them to 16 column maxima and emits the 9 window maxima of the row from registers -- straight-line code
// with compile-time column indices.  Pooled row 8 (conv row 15 only; the rest of its windows belongs to the tile
// below) is done by group j == 0.  Windows that straddle tiles are completed with atomicMax on the float bits.
__device__ __forceinline__ void stem_pool_emit(const float* lds, float* out, int b, int ty, int tx, int nb, int Ho, int Wo,
                                               int Hp, int Wp, int tid, int* range_flag) {
  const int c = tid & 31, j = tid >> 5;
  const int gy0 = ty * STEM_T, gx0 = tx * STEM_T;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int py = pass == 0 ? j : 8;
    if (pass == 1 && j != 0) break;
    const int gpy = ty * 8 + py;
    if (gpy >= Hp) continue;
    // (values may be negative: the fp32 stem hands over conv + bias and the ReLU comes after the max; the sentinel for
    // "no pixel of this window in this tile" is far below anything a convolution produces)
    constexpr float NONE = -3.0e38f;
    float cm[STEM_T];
#pragma unroll
    for (int cc = 0; cc < STEM_T; ++cc) cm[cc] = NONE;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int rr = 2 * py - 1 + dy;
      if (rr < 0 || rr > STEM_T - 1 || gy0 + rr >= Ho) continue;
#pragma unroll
      for (int cc = 0; cc < STEM_T; ++cc) cm[cc] = fmaxf(cm[cc], lds[(rr * STEM_T + cc) * STEM_TROW + c]);
    }
#pragma unroll
    for (int cc = 0; cc < STEM_T; ++cc)
      if (gx0 + cc >= Wo) cm[cc] = NONE;  // conv columns beyond the image
    float* row = out + ((size_t)(b * Hp + gpy) * Wp + tx * 8) * 64 + nb * 32 + c;
#pragma unroll
    for (int px = 0; px < 9; ++px) {
      if (tx * 8 + px >= Wp) continue;
      float mx = px < 8 ? cm[2 * px] : NONE;
      if (px > 0) mx = fmaxf(mx, cm[2 * px - 1]);
      if (px < 8) mx = fmaxf(mx, cm[2 * px + 1]);
      if (mx < -1.0e38f) continue;  // no pixel of this window lies in this tile
      mx = fmaxf(mx, 0.f);         // ReLU (>= +0 from here on: the atomicMax on the float bits below relies on it)
      if (range_flag && mx > 65504.f) atomicOr(range_flag, 1);
      if (py >= 1 && py <= 7 && px >= 1 && px <= 7)
        row[px * 64] = mx;  // whole window inside this tile
      else
        atomicMax(reinterpret_cast<unsigned int*>(row + px * 64), __float_as_uint(mx));
    }
  }
}

// The pooled map's cells that stem_pool_emit completes ACROSS tiles with atomicMax -- pooled rows / columns that are
// multiples of 8 (window row / column 0 and 8 of a 16 x 16 conv tile) -- must hold +0 before the stem runs; every other
// cell is written by a plain store of the one tile that holds its whole window.  Zeroing just those cells (23 % of the
// map: whole rows y % 8 == 0, and every eighth pixel of the other rows) replaces the hipMemsetAsync of the whole map
// (round 3: a 29 us fill launch per sub-batch at VGA x 32).  out = [n, Hp, Wp, 64] floats.
__global__ __launch_bounds__(256) void stem_border_clear_kernel(float4* out, int Hp, int Wp, int rows) {
  // one workgroup per 8 pooled rows of a frame (row y0 whole, every eighth pixel of the other seven): `rows` = frames x Hp
  const int y0 = blockIdx.x * 8;              // frame * Hp + y, Hp a multiple of 8 or not: rows are taken modulo Hp below
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nsparse = ((Wp + 7) / 8) * 16;    // pixels 0, 8, 16, ...: 16 float4 each
  for (int r = 0; r < 8 && y0 + r < rows; ++r) {
    float4* p = out + (size_t)(y0 + r) * Wp * 16;
    if ((((y0 + r) % Hp) & 7) == 0) {
      for (int i = threadIdx.x; i < Wp * 16; i += 256) p[i] = z;
    } else {
      for (int i = threadIdx.x; i < nsparse; i += 256) p[(i >> 4) * 128 + (i & 15)] = z;
    }
  }
}

template <int CIN>
__global__ __launch_bounds__(256) void stem_pool_kernel(const StemPoolArgs a) {
  constexpr int KREAL = CIN * 49, KG = (KREAL + 7) / 8;  // 147 -> 19 groups of 8; 49 -> 7
  __shared__ float lds[STEM_POOL_LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y;
  const int bidx = fpc_xcd_tile_index();   // neighbouring tiles complete each other's border windows with atomics: keep them in one L2
  const int b = bidx / tiles;
  const int t = bidx - b * tiles;
  const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
  const int iy0 = ty * STEM_T * 2 - 3, ix0 = tx * STEM_T * 2 - 3;

  // B fragments run two groups ahead, through a buffer descriptor: the lane's offset in a VGPR that never changes, the
  // group in the instruction's constant offset.  As wp[(g * 2 + nb) * 64] the loads past the 4 KB immediate range cost
  // two 64-bit pointer additions each -- VALU instructions in the K loop, and on this GPU a VALU instruction between two
  // fp32 MFMAs holds the matrix core up (generation 3).
  typedef float f32x4w __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.wfrag), 0, (KG + 2) * 2 * 64 * 16, 0x00020000);
  const int wlane = lane * 16;
  auto wfrag = [&](int i) {   // fragment i = group * 2 + nb
    const f32x4w r = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, i * 1024, 0));
    return make_float4(r.x, r.y, r.z, r.w);
  };
  float4 q0[2], q1[2];
  q0[0] = wfrag(0);
  q0[1] = wfrag(1);
  q1[0] = wfrag(2);
  q1[1] = wfrag(3);

  {  // input window -> LDS as aligned float4 row segments: LDS column 0 is image column ix0 - 1 (= 32 tx - 4, a multiple
     // of 4, and W is a multiple of 8, so every float4 lies entirely inside or entirely outside the frame); 40 columns
     // per row cover the 37 the tile needs.  All loads of a thread are issued before its first LDS write.
    constexpr int NQ = STEM_LW / 4, NE = CIN * STEM_HALO * NQ, IT = (NE + 255) / 256;
    float4 v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int e = tid + i * 256;
      const int row = e / NQ, q = e - row * NQ;
      const int c = row / STEM_HALO, hy = row - c * STEM_HALO;
      const int iy = iy0 + hy, ix = ix0 - 1 + 4 * q;
      const bool ok = e < NE && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const float4 x = *reinterpret_cast<const float4*>(a.in + (ok ? ((size_t)(b * CIN + c) * a.H + iy) * a.W + ix : 0));
      v[i] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float chk = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int e = tid + i * 256;
      if (e < NE) *reinterpret_cast<float4*>(lds + e * 4) = v[i];  // e * 4 == row * STEM_LW + 4 * q
      chk = nonfinite_probe(chk, v[i]);
    }
    if (chk != chk && a.range) a.range[b * FPC_RANGE_WORDS + RANGE_BAD_INPUT] = 1u;   // (rare; any number of writers, one value)
  }
  __syncthreads();

  int acol[2], arow[2];   // A operand bases: lanes 32-63 one column to the right / one row below (stem_pair)
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int m = (wave * 2 + mb) * 32 + l31;
    const int ab = (2 * (m / STEM_T)) * STEM_LW + 2 * (m % STEM_T) + 1;  // + 1: LDS column 0 is image column ix0 - 1
    acol[mb] = ab + half;
    arow[mb] = ab + half * STEM_LW;
  }
  // the accumulators START at the folded-BN bias (every register of a lane belongs to channel nb * 32 + l31): no
  // addition in the epilogue; the ReLU moves behind the max-pool (max and ReLU commute: one per pooled value, not per pixel)
  f32x16 acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float bias = a.bias[j * 32 + l31];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = bias;
  }

#pragma unroll
  for (int g = 0; g < KG; ++g) {
    float4 q2[2];
    q2[0] = wfrag((g + 2) * 2 + 0);
    q2[1] = wfrag((g + 2) * 2 + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const StemPair P = stem_pair(CIN, g * 4 + j);
      if (P.wa < 0 && P.wb < 0) continue;   // an empty step; the first one carries the bias for stem_pool2_kernel (pack layout revision 4)
      const int off = stem_koff_c<KREAL>(P.addr_k);
      const float bf0 = j == 0 ? q0[0].x : j == 1 ? q0[0].y : j == 2 ? q0[0].z : q0[0].w;
      const float bf1 = j == 0 ? q0[1].x : j == 1 ? q0[1].y : j == 2 ? q0[1].z : q0[1].w;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float af = lds[(P.row ? arow[mb] : acol[mb]) + off];
        acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf0, acc[mb][0], 0, 0, 0);
        acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf1, acc[mb][1], 0, 0, 0);
      }
    }
    q0[0] = q1[0];
    q0[1] = q1[1];
    q1[0] = q2[0];
    q1[1] = q2[1];
  }

  // epilogue: per 32-channel half, tile -> LDS -> 3x3/2 max-pool
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    __syncthreads();  // nb == 0: halo reads done; nb == 1: previous half's pooling reads done
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (wave * 2 + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        lds[m * STEM_TROW + l31] = acc[mb][nb][r];      // conv + bias, before the ReLU (stem_pool_emit applies it to the pooled value)
      }
    __syncthreads();
    stem_pool_emit(lds, a.out, b, ty, tx, nb, a.Ho, a.Wo, a.Hp, a.Wp, tid, nullptr);
  }
}

// ---------------------------------------------------------------------------------
// Round 5: stem_pool_kernel on a VALU diet, for maps whose WIDTH is whole tiles (Wo a multiple of 16: VGA, HD, QVGA; a
// last tile row of fewer than 16 conv rows takes a masked tile write; other widths keep stem_pool_kernel).  On gfx950 the fp32 MFMA and the VALU exclude each other on a SIMD (wblock36_mfma.h), and the
// shipped stem_pool_kernel<3> issues ~900 VALU instructions per wave and tile around its 304 MFMAs (static count of the
// code object: 251 in the prologue, 52 in the K loop, 669 in the epilogue, 408 of them v_max_f32) -- a fifth of its
// matrix time.  Here:
//   * the folded-BN bias is a K step of the GEMM (the step the K order leaves empty behind the last channel: A = 1.0,
//     B = bias in lanes 0-31, stem_pair2 below) and the accumulators start at the MFMA's inline constant 0: no 64 moves per
//     tile, no addition in the epilogue;
//   * at most 128 registers (__launch_bounds__(256, 4): the kernel names no AGPR, so the accumulators are architectural
//     VGPRs): the tile write is 32 ds_write_b32 per half straight from them, where the AGPR form went through 94
//     v_accvgpr moves per tile;
//   * the window is staged through a buffer descriptor of the frame: 32-bit offsets, out-of-frame float4s get an offset
//     outside its range and come back as zeros (no 64-bit address arithmetic, no select on the data);
//   * pooling: a pooled row is v_max3_f32 over its three conv rows (16 instructions; row -1 of the tile's first pooled
//     row reads row 0 again -- a duplicate does not change a maximum) and v_max3_f32 over three columns + the ReLU per
//     output, no masks (whole tiles), one branch per half (the tile's first pooled row goes out by atomics, the others by
//     seven plain stores and two atomics).
// Same K order, same fragments (the bias step is in them for every kernel since pack layout revision 4; stem_pool_kernel
// and stem_kernel skip it), same cells completed by atomicMax as stem_pool_kernel.
// ---------------------------------------------------------------------------------
constexpr int STEM_BIAS_TAP = -2;   // StemPair::wa of the step that carries the bias
__host__ __device__ constexpr StemPair stem_pair2(int cin, int t) {
  if (t == cin * 25) return StemPair{0, STEM_BIAS_TAP, -1, 0};
  return stem_pair(cin, t);
}

// (as asm, like stem_max3: behind an asm result the compiler's own fmaxf first canonicalises its operand -- v_max_f32 x, x)
__device__ __forceinline__ float stem_relu(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float stem_max3(float x, float y, float z) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  return r;
}

template <int CIN>
__global__ __launch_bounds__(256, 4) void stem_pool2_kernel(const StemPoolArgs a) {
  constexpr int KREAL = CIN * 49, KG = (KREAL + 7) / 8;
  static_assert(CIN * 25 < KG * 4, "an empty step behind the last channel carries the bias");
  __shared__ float lds[STEM_POOL_LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y;
  const int bidx = fpc_xcd_tile_index();
  const int b = bidx / tiles;
  const int t = bidx - b * tiles;
  const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
  const int iy0 = ty * STEM_T * 2 - 3, ix0 = tx * STEM_T * 2 - 3;

  typedef float f32x4w __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.wfrag), 0, (KG + 2) * 2 * 64 * 16, 0x00020000);
  const int wlane = lane * 16;
  auto wfrag = [&](int i) {   // fragment i = group * 2 + nb
    const f32x4w r = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, i * 1024, 0));
    return make_float4(r.x, r.y, r.z, r.w);
  };
  float4 q0[2], q1[2];
  q0[0] = wfrag(0);
  q0[1] = wfrag(1);
  q1[0] = wfrag(2);
  q1[1] = wfrag(3);

  {  // input window -> LDS as aligned float4 row segments (stem_pool_kernel), through a descriptor of THIS frame
    constexpr int NQ = STEM_LW / 4, NE = CIN * STEM_HALO * NQ, IT = (NE + 255) / 256;
    const int plane = a.H * a.W;
    const __amdgpu_buffer_rsrc_t irsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in) + (size_t)b * CIN * plane, 0, CIN * plane * 4, 0x00020000);
    f32x4w v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int e = tid + i * 256;
      const int row = e / NQ, q = e - row * NQ;
      const int c = row / STEM_HALO, hy = row - c * STEM_HALO;
      const int iy = iy0 + hy, ix = ix0 - 1 + 4 * q;
      bool ok = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
      if ((i + 1) * 256 > NE) ok = ok & (e < NE);
      const int off = ((c * a.H + iy) * a.W + ix) * 4;
      v[i] = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(irsrc, ok ? off : (int)0x7ffffff0, 0, 0));
    }
    float chk = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int e = tid + i * 256;
      if ((i + 1) * 256 <= NE || e < NE) *reinterpret_cast<f32x4w*>(lds + e * 4) = v[i];
      chk = fmaf((v[i].x + v[i].y) + (v[i].z + v[i].w), 0.f, chk);   // nonfinite_probe
    }
    if (chk != chk && a.range) a.range[b * FPC_RANGE_WORDS + RANGE_BAD_INPUT] = 1u;
  }
  __syncthreads();

  // A operand bases PER CHANNEL, as registers of their own: a tap's offset inside its channel's window is at most 6 rows + 6
  // columns = 246 dwords, inside ds_read2_b32's 8-bit offsets -- from one base per (column | row) form the compiler made new
  // bases on the way (v_add_u32 in the K loop: 30 of them, and each VALU instruction between two fp32 MFMAs costs 15 cycles)
  int acol[CIN][2], arow[CIN][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int m = (wave * 2 + mb) * 32 + l31;
    const int ab = (2 * (m / STEM_T)) * STEM_LW + 2 * (m % STEM_T) + 1;
#pragma unroll
    for (int ch = 0; ch < CIN; ++ch) {
      acol[ch][mb] = (ab + half + ch * (STEM_HALO * STEM_LW)) * 4;            // BYTE offsets (a dword index is shifted at every use)
      arow[ch][mb] = (ab + half * STEM_LW + ch * (STEM_HALO * STEM_LW)) * 4;
      asm volatile("" : "+v"(acol[ch][mb]), "+v"(arow[ch][mb]));
    }
  }
  float one = 1.f;
  asm volatile("" : "+v"(one));   // (a register the MFMA can take as its A operand)
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int g = 0; g < KG; ++g) {
    float4 q2[2];
    q2[0] = wfrag((g + 2) * 2 + 0);
    q2[1] = wfrag((g + 2) * 2 + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const StemPair P = stem_pair2(CIN, g * 4 + j);
      if (P.wa == -1 && P.wb == -1) continue;   // an empty step: no MFMA
      const int kk = P.addr_k < KREAL ? P.addr_k : KREAL - 1, ch = kk / 49;
      const int off = ((kk % 49) / 7) * STEM_LW + (kk % 7);   // inside the channel's window
      const float bf0 = j == 0 ? q0[0].x : j == 1 ? q0[0].y : j == 2 ? q0[0].z : q0[0].w;
      const float bf1 = j == 0 ? q0[1].x : j == 1 ? q0[1].y : j == 2 ? q0[1].z : q0[1].w;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float af = P.wa == STEM_BIAS_TAP
                             ? one
                             : *reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + (P.row ? arow[ch][mb] : acol[ch][mb]) + off * 4);
        acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf0, acc[mb][0], 0, 0, 0);
        acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf1, acc[mb][1], 0, 0, 0);
      }
    }
    q0[0] = q1[0];
    q0[1] = q1[1];
    q1[0] = q2[0];
    q1[1] = q2[1];
  }
  // (no fence here: with one, the epilogue's address arithmetic waits behind the last MFMA and the kernel measures 1.7 % slower
  // than with those dozen instructions under the last steps' MFMAs, where the compiler puts them)
  // epilogue: per 32-channel half, tile -> LDS -> 3x3/2 max-pool
  const int c = tid & 31, j = tid >> 5;
  const int wbase = (wave * 64 + 4 * half) * STEM_TROW + l31;
  const int r0 = j == 0 ? 0 : 2 * j - 1;
  const int rd0 = r0 * STEM_T * STEM_TROW + c, rd1 = 2 * j * STEM_T * STEM_TROW + c;
  const bool has_col8 = tx * 8 + 8 < a.Wp, has_row8 = ty * 8 + 8 < a.Hp;
  const int rows_valid = min(STEM_T, a.Ho - ty * STEM_T);   // (uniform) conv rows of the tile inside the map
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    FPC_LDS_BARRIER();   // (LDS hand-offs only: __syncthreads() would also wait for the first half's stores and atomics)
    if (rows_valid == STEM_T) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[wbase + (mb * 32 + (r & 3) + 8 * (r >> 2)) * STEM_TROW] = acc[mb][nb][r];
    } else {   // the map's last tile row (Ho not a multiple of 16): conv rows below the map leave every maximum alone
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wave * 2 + mb) * 2 + (r >> 3);   // (wave-uniform)
          lds[wbase + (mb * 32 + (r & 3) + 8 * (r >> 2)) * STEM_TROW] = row < rows_valid ? acc[mb][nb][r] : -3.0e38f;
        }
    }
    FPC_LDS_BARRIER();
    float cm[STEM_T];
#pragma unroll
    for (int cc = 0; cc < STEM_T; ++cc)
      cm[cc] = stem_max3(lds[rd0 + cc * STEM_TROW], lds[rd1 + cc * STEM_TROW], lds[rd1 + (STEM_T + cc) * STEM_TROW]);
    float mx[9];
    mx[0] = stem_max3(cm[0], cm[1], 0.f);
#pragma unroll
    for (int px = 1; px < 8; ++px) mx[px] = stem_relu(stem_max3(cm[2 * px - 1], cm[2 * px], cm[2 * px + 1]));
    mx[8] = stem_relu(cm[15]);
    float* row = a.out + ((size_t)(b * a.Hp + ty * 8 + j) * a.Wp + tx * 8) * 64 + nb * 32 + c;
    if (ty * 8 + j >= a.Hp) {
      // (a pooled row below the map: only in the last tile row of a map whose height is not a multiple of 16)
    } else if (j == 0) {
#pragma unroll
      for (int px = 0; px < 8; ++px) atomicMax(reinterpret_cast<unsigned int*>(row + px * 64), __float_as_uint(mx[px]));
    } else {
      atomicMax(reinterpret_cast<unsigned int*>(row), __float_as_uint(mx[0]));
#pragma unroll
      for (int px = 1; px < 8; ++px) row[px * 64] = mx[px];
    }
    if (has_col8 && ty * 8 + j < a.Hp) atomicMax(reinterpret_cast<unsigned int*>(row + 8 * 64), __float_as_uint(mx[8]));
    if (has_row8 && j == 0) {   // pooled row 8: conv row 15 only, the rest of its windows belongs to the tile below
      float* row8 = row + (size_t)8 * a.Wp * 64;
      const int rd = 15 * STEM_T * STEM_TROW + c;
#pragma unroll
      for (int cc = 0; cc < STEM_T; ++cc) cm[cc] = lds[rd + cc * STEM_TROW];
      atomicMax(reinterpret_cast<unsigned int*>(row8), __float_as_uint(stem_max3(cm[0], cm[1], 0.f)));
#pragma unroll
      for (int px = 1; px < 8; ++px)
        atomicMax(reinterpret_cast<unsigned int*>(row8 + px * 64), __float_as_uint(stem_relu(stem_max3(cm[2 * px - 1], cm[2 * px], cm[2 * px + 1]))));
      if (has_col8) atomicMax(reinterpret_cast<unsigned int*>(row8 + 8 * 64), __float_as_uint(stem_relu(cm[15])));
    }
  }
}

// MaxPool2d(kernel_size=3, stride=2, padding=1) on NHWC, C = 64 (superpoint.py:15,23).
__global__ __launch_bounds__(256) void maxpool_kernel(const float4* in, float4* out, int B, int H, int W,
                                                      int Ho, int Wo) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // over B*Ho*Wo*16 float4
  const size_t total = (size_t)B * Ho * Wo * 16;
  if (i >= total) return;
  const int c4 = i & 15;
  size_t p = i >> 4;
  const int x = p % Wo;
  p /= Wo;
  const int y = p % Ho;
  const int b = p / Ho;
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = 2 * y + ky - 1;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = 2 * x + kx - 1;
      if (ix < 0 || ix >= W) continue;
      const float4 v = in[((size_t)(b * H + iy) * W + ix) * 16 + c4];
      m.x = fmaxf(m.x, v.x);
      m.y = fmaxf(m.y, v.y);
      m.z = fmaxf(m.z, v.z);
      m.w = fmaxf(m.w, v.w);
    }
  }
  out[i] = m;
}

// ---------------------------------------------------------------------------------
// Detector post-processing, part 1 (python/src/superpoint.py:111-114,
// python/src/netutils.py:56-75): p = exp(l) / (sum_c exp(l) + 1e-5), drop the
// dustbin, depth-to-space, threshold.  One wave per 8x8 cell: lane c owns
// sub-pixel (c/8, c%8).  Writes the dense probability map, the NMS state map
// (float bits of p where p >= thresh, else 0) and appends candidates to the
// frame's list (order irrelevant: NMS below is order-free, the sort is total).
// ---------------------------------------------------------------------------------
// (nms_state_word / nms_state_conf: nms_word.h)

// (sm_exp / sm_scale / sm_prob, the arithmetic of FAST = true: nms_word.h)
template <bool FAST = false>
__global__ __launch_bounds__(256) void softmax_d2s_kernel(const float* logits, int cs, int B, int Hc, int Wc,
                                                          float thresh, float* prob, uint32_t* nmsmap,
                                                          uint32_t* cand, int32_t* ncand, float* rowmax) {
  // one workgroup per ROW of cells: the 8 x W strip of probabilities is assembled in LDS so
  // that the dense maps are written as whole rows, and the strip's candidates are appended
  // with ONE global atomic (per-wave atomics on one counter serialise at ~11 ns each).
  extern __shared__ float strip[];                 // [8][W] floats, then [8*W] 16-bit candidate slots (8 W < 65536)
  __shared__ int s_cnt, s_base;
  __shared__ float s_max[4];
  const int W = Wc * 8, H = Hc * 8;
  unsigned short* s_list = reinterpret_cast<unsigned short*>(strip + 8 * W);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / Hc, i = blockIdx.x - b * Hc;
  if (tid == 0) s_cnt = 0;
  // Sixteen lanes per cell, four channels each (one 16-byte load; the dustbin a broadcast word): a wave's load covers
  // four cells, the sum over a cell's channels is a 4-step butterfly inside its sixteen lanes, and a lane's four
  // probabilities are four neighbouring pixels of the strip (one 16-byte LDS write).  As one wave per cell this phase was
  // 6 cross-lane steps, 2 exps and 4 one-word LDS writes per lane and CELL.  Four such loads per wave are in flight.
  const float* lrow = logits + (size_t)(b * Hc + i) * Wc * cs;
  const int q = lane & 15, gidx = lane >> 4;
  float lmax = 0.f;       // the largest logit of this row of cells (logits are post-ReLU: >= 0; NaN never wins a max)
  for (int j0 = wave * 16; j0 < Wc; j0 += 64) {
    float4 lv[4];
    float ld[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + 4 * u + gidx < Wc ? j0 + 4 * u + gidx : Wc - 1;
      lv[u] = *reinterpret_cast<const float4*>(lrow + (size_t)j * cs + 4 * q);
      ld[u] = lrow[(size_t)j * cs + 64];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + 4 * u + gidx;
      const float e0 = sm_exp<FAST>(lv[u].x), e1 = sm_exp<FAST>(lv[u].y), e2 = sm_exp<FAST>(lv[u].z), e3 = sm_exp<FAST>(lv[u].w);
      const float ed = sm_exp<FAST>(ld[u]);
      lmax = fmaxf(fmaxf(fmaxf(lmax, fabsf(ld[u])), fmaxf(fabsf(lv[u].x), fabsf(lv[u].y))), fmaxf(fabsf(lv[u].z), fabsf(lv[u].w)));
      float s = (e0 + e1) + (e2 + e3);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
      const float den = sm_scale<FAST>((s + ed) + .00001f);
      if (j < Wc)
        *reinterpret_cast<float4*>(strip + (q >> 1) * W + j * 8 + 4 * (q & 1)) =
            make_float4(sm_prob<FAST>(e0, den), sm_prob<FAST>(e1, den), sm_prob<FAST>(e2, den), sm_prob<FAST>(e3, den));
    }
  }
  // the largest logit of this row of cells -> rowmax[frame][row] (include/fpc.h: fpc_output_range reduces the rows): a wave's
  // maximum to LDS in front of the barrier that is there anyway, one plain store per workgroup behind it.  (Round 5's first
  // form -- an atomicMax per wave on the frame's word -- doubled this kernel's time: 240 atomics per word.)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
  if (lane == 0) s_max[wave] = lmax;
  __syncthreads();
  // write-out, four pixels per thread and trip (16-byte LDS reads and stores: as single floats this loop was 80 store
  // instructions per thread at HD); `prob` is NULL in fpc_detect, whose callers never see the dense map (a third of the
  // kernel's bytes)
  if (rowmax && tid == 0) rowmax[b * Hc + i] = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
  const size_t fbase = (size_t)b * H * W + (size_t)i * 8 * W;
  for (int k4 = tid; k4 < 2 * W; k4 += 256) {      // 8 W / 4 groups; 2 W is a multiple of 16, not always of 64
    const float4 p4 = *reinterpret_cast<const float4*>(strip + 4 * k4);
    const float pv[4] = {p4.x, p4.y, p4.z, p4.w};
    uint4 st;
    unsigned* sw = &st.x;
    if (prob) *reinterpret_cast<float4*>(prob + fbase + 4 * k4) = p4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool c = pv[u] >= thresh;
      sw[u] = c ? nms_state_word(pv[u]) : 0u;
      const unsigned long long mask = __ballot(c);
      if (mask) {
        int off = 0;
        if (lane == __ffsll((long long)mask) - 1) off = atomicAdd(&s_cnt, __popcll(mask));
        off = __shfl(off, __ffsll((long long)mask) - 1);
        if (c) s_list[off + __popcll(mask & ((1ull << lane) - 1))] = (unsigned short)(4 * k4 + u);
      }
    }
    *reinterpret_cast<uint4*>(nmsmap + fbase + 4 * k4) = st;
  }
  __syncthreads();
  const int n = s_cnt;
  if (n == 0) return;
  if (tid == 0) s_base = atomicAdd(&ncand[b], n);
  __syncthreads();
  uint32_t* dst = cand + (size_t)b * H * W + s_base;
  for (int k = tid; k < n; k += 256) dst[k] = (uint32_t)(i * 8 * W) + s_list[k];
}

// Same outputs from a caller-provided dense probability map (fpc_get_points).  The map is a probability map: entries
// must be >= 0 (fpc_create rejects conf_thresh < 0; with conf_thresh == 0 every non-negative pixel is a candidate).
__global__ __launch_bounds__(256) void threshold_kernel(const float* prob, int B, int HW, float thresh,
                                                        uint32_t* nmsmap, uint32_t* cand, int32_t* ncand) {
  const int lane = threadIdx.x & 63;
  const int per = (HW + 255) / 256 * 256;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int b = g / per;
  const int idx = g - (size_t)b * per;
  if (b >= B) return;
  const bool in = idx < HW;
  const float p = in ? prob[(size_t)b * HW + idx] : 0.f;
  const bool c = in && p >= thresh && p >= 0.f;   // probabilities: negative entries (and NaN) never become candidates
  if (in) nmsmap[(size_t)b * HW + idx] = c ? nms_state_word(p) : 0u;
  const unsigned long long mask = __ballot(c);
  if (mask) {
    int base = 0;
    if (lane == 0) base = atomicAdd(&ncand[b], __popcll(mask));
    base = __shfl(base, 0);
    if (c) cand[(size_t)b * HW + base + __popcll(mask & ((1ull << lane) - 1))] = idx;
  }
}

// ---------------------------------------------------------------------------------
// Detector post-processing, part 2: greedy NMS + sort + border crop
// (python/src/nms.py:4-53, python/src/netutils.py:90-99), one 1024-thread
// workgroup per frame.
//
// The reference walks candidates in descending confidence and keeps a point iff
// no already-kept point lies within infinity-distance nms_dist.  Equivalent
// order-free rule, iterated to a fixed point: an undecided candidate is KEPT once
// every higher-priority candidate in its window is suppressed, and SUPPRESSED as
// soon as any candidate in its window is kept (a kept neighbour always has higher
// priority, otherwise it could not have been decided before this one).  Decisions
// are final and only depend on final decisions, so reading neighbours while other
// threads update them is safe: a stale read only delays a decision by a round.
// Priority = (confidence, then smaller row-major index) -- the tie order this
// build defines (the reference's is unspecified: numpy's unstable argsort).
//
// State map word: 0 = empty / suppressed, float bits + 1 (> 0, see nms_state_word) = undecided
// candidate, that | 0x80000000 = kept.
// ---------------------------------------------------------------------------------
struct NmsArgs {
  uint32_t* nmsmap;        // [B][H*W]
  uint32_t* cand;          // [B][H*W]
  const int32_t* ncand;    // [B]
  unsigned long long* sort_scratch;  // [B][sort_cap] used when the kept set exceeds LDS
  int sort_cap;            // power of two
  int H, W, r, border, cap;
  int32_t* count;          // [B]
  int32_t* xy;             // [B][cap][2]
  float* conf;             // [B][cap]
  int32_t* status;         // [1] reserved (the round structure cannot fail to converge)
  int max_rounds;
  int32_t* aux;            // [B][NMS_AUX_INTS]: [0] 1 = this frame takes nms_sort_kernel's one-workgroup path,
                           // [2 + 2c], [3 + 2c] = offset and length of chunk c's sorted keys in sort_scratch
};

constexpr int NMS_MAX_CHUNKS = 8;    // slices of a frame's candidate list (nms_chunk_sort_kernel)
constexpr int NMS_AUX_INTS = 2 + 2 * NMS_MAX_CHUNKS;

constexpr int NMS_LDS_KEYS = 16384;  // 128 KiB of 64-bit keys

__device__ __forceinline__ uint32_t ld_relaxed(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_relaxed(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Rounds.  grid = (G, B): the frame's candidates are dealt over G workgroups; a workgroup
// leaves as soon as all of ITS candidates are decided.  Neighbour words are read and
// written with relaxed agent-scope atomics only (they may belong to a workgroup on another
// CU / XCD); a workgroup never blocks on another one, it just runs another round, so the
// kernel needs no co-residency guarantee.
template <int R>
__device__ __forceinline__ void nms_scan(const uint32_t* map, int H, int W, int r, uint32_t ci, uint32_t v,
                                         bool* kept_nb, bool* wait) {
  const int y = ci / W, x = ci - y * W;
  bool k = false, w = false;
  if (R > 0) {  // compile-time radius: three window rows (3*(2R+1) loads) in flight at a time --
                // the whole window at once would not fit in registers (it went to scratch)
    constexpr int D = 2 * R + 1;
#pragma unroll
    for (int g = -R; g <= R; g += 3) {
      uint32_t u[3 * D];
#pragma unroll
      for (int dy = g; dy < g + 3 && dy <= R; ++dy)
#pragma unroll
        for (int dx = -R; dx <= R; ++dx) {
          const int yy = y + dy, xx = x + dx;
          const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
          u[(dy - g) * D + dx + R] = ld_relaxed(map + (ok ? yy * W + xx : (int)ci));
        }
#pragma unroll
      for (int dy = g; dy < g + 3 && dy <= R; ++dy)
#pragma unroll
        for (int dx = -R; dx <= R; ++dx) {
          if (dy == 0 && dx == 0) continue;
          const int yy = y + dy, xx = x + dx;
          const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
          const uint32_t q = yy * W + xx, uu = u[(dy - g) * D + dx + R];
          k |= ok && (uu & 0x80000000u);
          w |= ok && !(uu & 0x80000000u) && (uu > v || (uu == v && q < ci));
        }
    }
  } else {
    const int y0 = max(y - r, 0), y1 = min(y + r, H - 1), x0 = max(x - r, 0), x1 = min(x + r, W - 1);
    for (int yy = y0; yy <= y1 && !k; ++yy)
      for (int xx = x0; xx <= x1; ++xx) {
        const uint32_t q = yy * W + xx;
        if (q == ci) continue;
        const uint32_t uu = ld_relaxed(map + q);
        if (uu & 0x80000000u) k = true;
        else if (uu > v || (uu == v && q < ci)) w = true;
      }
  }
  *kept_nb = k;
  *wait = w;
}

// One pass of rounds over the candidates [first, n) in steps of `stride` owned by this
// workgroup.  Runs rounds while (a) something of ours is still undecided and (b) the last
// round decided at least one of ours (`until_done` = false), or until everything is decided
// (`until_done` = true, legal only when this workgroup owns ALL candidates of the frame).
template <int R>
__device__ __forceinline__ void nms_run_rounds(uint32_t* map, uint32_t* cand, int n, int first, int stride, int H,
                                               int W, int r, bool until_done) {
  while (true) {
    int pending = 0, progress = 0;
    for (int i = first; i < n; i += stride) {
      const uint32_t ci = cand[i];
      if (ci & 0x80000000u) continue;  // decided earlier (only the owning thread writes cand[i])
      const uint32_t v = ld_relaxed(map + ci);
      bool kept_nb, wait;
      nms_scan<R>(map, H, W, r, ci, v, &kept_nb, &wait);
      if (kept_nb) {
        st_relaxed(map + ci, 0u);
        cand[i] = ci | 0x80000000u;
        progress = 1;
      } else if (!wait) {
        st_relaxed(map + ci, v | 0x80000000u);
        cand[i] = ci | 0x80000000u;
        progress = 1;
      } else {
        pending = 1;
      }
    }
    const int f = __syncthreads_or(pending | (progress << 1));
    if (!(f & 1)) break;                  // nothing of ours left
    if (!until_done && !(f & 2)) break;   // stuck on another workgroup's candidates: let the launch end
  }
}

// Parallel rounds: grid = (G, B), the frame's candidates are dealt over G workgroups.  A
// workgroup NEVER waits for another one without bound: a wave whose candidates make no progress for
// NMS_IDLE_ROUNDS rounds leaves, and whatever is still undecided is picked up by the next launch (the host
// issues a few of these back to back) and finally by nms_finish_kernel, whose single workgroup per frame owns
// every candidate and therefore always terminates.  No co-residency assumption anywhere.
//
// What a round costs decides everything here: the words of other workgroups' candidates are read past this XCD's
// L2 (agent scope), 81 of them per window at r = 4, and the longest dependency chain of a frame is some tens of
// rounds.  But only the undecided neighbours of HIGHER priority matter to a candidate (a kept neighbour always has
// higher priority; lower ones and suppressed ones never influence it), at the densities of real maps one or two of
// the 80: the first scan of a candidate records them -- up to eight window positions packed in 64 bits, in an LDS
// slot of its own -- and every later round re-reads just those words.  A candidate with more than eight (dense maps)
// or beyond the LDS slots (NMS_LIST_SLOTS per thread) is re-scanned every round, as round 1 of this build did for all.
constexpr int NMS_ROUNDS_THREADS = 512;  // 8 waves: the unrolled window scan needs > 128 VGPRs
constexpr int NMS_LIST_SLOTS = 16;       // LDS: 16 x 512 x 8 B = 64 KiB
constexpr int NMS_IDLE_ROUNDS = 16;

// First look at a candidate: the scan of nms_scan plus the list of what it has to wait for (0: too many to list).
// FIRST: the scan of a launch's first round.  81 one-word requests per candidate is what bounds that round (the
// vector memory pipeline, not latency), so there a window row is two 16-byte loads and a word -- plain loads, through
// the caches: a word older than the truth only shows a decided neighbour as still undecided; it then goes on the list
// and is re-read coherently in the next round.  Candidates within R of the frame's edge take the word-by-word scan.
typedef uint32_t nms_u4 __attribute__((ext_vector_type(4), aligned(4)));

template <int R, bool FIRST>
__device__ __forceinline__ void nms_scan_list(const uint32_t* map, int H, int W, uint32_t ci, uint32_t v, bool* kept_nb,
                                              bool* wait, unsigned long long* list) {
  constexpr int D = 2 * R + 1;
  static_assert(D * D < 255, "window positions are stored in a byte");
  const int y = ci / W, x = ci - y * W;
  bool k = false;
  int cnt = 0;
  unsigned long long l = 0ull;
  const bool inner = FIRST && R == 4 && y >= R && y < H - R && x >= R && x < W - R;
#pragma unroll
  for (int g = -R; g <= R; g += 3) {
    uint32_t u[3 * D];
    if (inner) {
      if constexpr (R == 4) {
#pragma unroll
        for (int dy = g; dy < g + 3 && dy <= R; ++dy) {
          const uint32_t* q = map + (y + dy) * W + x - R;
          const nms_u4 a = *reinterpret_cast<const nms_u4*>(q), c = *reinterpret_cast<const nms_u4*>(q + 4);
          uint32_t* o = u + (dy - g) * D;
          o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = c.x; o[5] = c.y; o[6] = c.z; o[7] = c.w; o[8] = q[8];
        }
      }
    } else {
#pragma unroll
      for (int dy = g; dy < g + 3 && dy <= R; ++dy)
#pragma unroll
        for (int dx = -R; dx <= R; ++dx) {
          const int yy = y + dy, xx = x + dx;
          const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
          u[(dy - g) * D + dx + R] = ld_relaxed(map + (ok ? yy * W + xx : (int)ci));
        }
    }
#pragma unroll
    for (int dy = g; dy < g + 3 && dy <= R; ++dy)
#pragma unroll
      for (int dx = -R; dx <= R; ++dx) {
        if (dy == 0 && dx == 0) continue;
        const int yy = y + dy, xx = x + dx;
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const uint32_t q = yy * W + xx, uu = u[(dy - g) * D + dx + R];
        k |= ok && (uu & 0x80000000u);
        if (ok && !(uu & 0x80000000u) && (uu > v || (uu == v && q < ci))) {
          if (cnt < 8) l |= (unsigned long long)((dy + R) * D + (dx + R) + 1) << (8 * cnt);
          ++cnt;
        }
      }
  }
  *kept_nb = k;
  *wait = cnt > 0;
  *list = cnt <= 8 ? l : 0ull;
}

template <int R>
__global__ __launch_bounds__(NMS_ROUNDS_THREADS) void nms_rounds_kernel(const NmsArgs a) {
  const int b = blockIdx.y;
  const size_t HW = (size_t)a.H * a.W;
  const int n = a.ncand[b];
  if (n <= 1 || (int)(blockIdx.x * NMS_ROUNDS_THREADS) >= n) return;  // n == 1 is settled by nms_finish_kernel
  uint32_t* map = a.nmsmap + b * HW;
  uint32_t* cand = a.cand + b * HW;
  const int first = blockIdx.x * NMS_ROUNDS_THREADS + threadIdx.x, stride = gridDim.x * NMS_ROUNDS_THREADS;
  if constexpr (R == 0) {
    nms_run_rounds<0>(map, cand, n, first, stride, a.H, a.W, a.r, false);
  } else {
    constexpr int D = 2 * R + 1;
    __shared__ unsigned long long lists[NMS_LIST_SLOTS][NMS_ROUNDS_THREADS];  // slot k of a thread: its k-th candidate
    static_assert(sizeof(lists) <= 64 * 1024, "static LDS of a kernel is limited to 64 KiB: fewer list slots or dynamic LDS");
    const int W = a.W, H = a.H;
#pragma unroll
    for (int k = 0; k < NMS_LIST_SLOTS; ++k) lists[k][threadIdx.x] = 0ull;
    int idle = 0;
    for (int round = 0;; ++round) {  // waves run their rounds independently: nothing below is shared between threads
      bool pending = false, progress = false;
      int k = 0;
      for (int i = first; i < n; i += stride, ++k) {
        const uint32_t ci = cand[i];
        if (ci & 0x80000000u) continue;  // decided earlier (only the owning thread writes cand[i])
        bool kept_nb = false, wait = false;
        unsigned long long l = k < NMS_LIST_SLOTS ? lists[k][threadIdx.x] : 0ull;
        uint32_t v = 0;
        if (l) {
          unsigned long long keep_l = 0ull;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int code = (int)((l >> (8 * e)) & 0xffu);
            if (code) {
              const int dy = (code - 1) / D - R, dx = (code - 1) % D - R;
              const uint32_t uu = ld_relaxed(map + (int)ci + dy * W + dx);
              kept_nb |= (uu & 0x80000000u) != 0;
              if (uu && !(uu & 0x80000000u)) keep_l |= (unsigned long long)code << (8 * e);
            }
          }
          wait = keep_l != 0ull;
          if (keep_l != l) {
            lists[k][threadIdx.x] = keep_l;
            if (!kept_nb && wait) progress = true;  // a neighbour was suppressed: the chain moves
          }
          if (!kept_nb && !wait) v = ld_relaxed(map + ci);
        } else {
          v = ld_relaxed(map + ci);
          if (round == 0) nms_scan_list<R, true>(map, H, W, ci, v, &kept_nb, &wait, &l);
          else nms_scan_list<R, false>(map, H, W, ci, v, &kept_nb, &wait, &l);
          if (k < NMS_LIST_SLOTS && !kept_nb && wait) lists[k][threadIdx.x] = l;
        }
        if (kept_nb) {
          st_relaxed(map + ci, 0u);
          cand[i] = ci | 0x80000000u;
          progress = true;
        } else if (!wait) {
          st_relaxed(map + ci, v | 0x80000000u);
          cand[i] = ci | 0x80000000u;
          progress = true;
        } else {
          pending = true;
        }
      }
      if (!__any(pending)) break;                       // nothing of this wave's left
      idle = __any(progress) ? 0 : idle + 1;
      if (idle >= NMS_IDLE_ROUNDS) break;               // stuck on other workgroups' candidates: let the launch end
    }
  }
}

// Survivors inside the border -> sort -> outputs.  One workgroup per frame.
// 64-bit keys (conf bits, ~index): a descending sort of unique keys gives
// (confidence desc, index asc) -- netutils.py:92-99.
template <typename KeyPtr>
__device__ __forceinline__ void nms_sort_body(const NmsArgs& a, KeyPtr keys, int P, int K, int n, const uint32_t* map,
                                              const uint32_t* cand, int* s_count) {
  const int b = blockIdx.x, tid = threadIdx.x;  // nms_sort_kernel's grid is (B)
  const int H = a.H, W = a.W, bw = a.border;
  for (int i = tid; i < P; i += 1024) keys[i] = 0ull;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    const uint32_t ci = i < n ? cand[i] & 0x7fffffffu : 0u;
    const int y = ci / W, x = ci - y * W;
    const uint32_t u = i < n ? map[ci] : 0u;
    const bool keep = (u & 0x80000000u) && x >= bw && x < W - bw && y >= bw && y < H - bw;
    // one LDS atomic per wave, not per survivor (same-address atomics serialise)
    const unsigned long long mask = __ballot(keep);
    if (mask) {
      int base = 0;
      if ((tid & 63) == 0) base = atomicAdd(s_count, __popcll(mask));
      base = __shfl(base, 0);
      if (keep)
        keys[base + __popcll(mask & ((1ull << (tid & 63)) - 1))] =
            ((unsigned long long)(u & 0x7fffffffu) << 32) | (0xffffffffu - ci);
    }
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1)  // bitonic sort, descending; one thread per compare-exchange
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (P >> 1); t += 1024) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        const unsigned long long ki = keys[i], kl = keys[l];
        const bool desc = (i & k) == 0;
        if (desc ? ki < kl : ki > kl) {
          keys[i] = kl;
          keys[l] = ki;
        }
      }
      __syncthreads();
    }
  // a caller-chosen capacity (fpc_config.max_keypoints) below K keeps the `cap` most confident points
  if (tid == 0) a.count[b] = K < a.cap ? K : a.cap;
  for (int i = tid; i < K && i < a.cap; i += 1024) {
    const unsigned long long k = keys[i];
    const uint32_t ci = 0xffffffffu - (uint32_t)(k & 0xffffffffu);
    const int y = ci / W, x = ci - y * W;
    a.xy[((size_t)b * a.cap + i) * 2 + 0] = x;
    a.xy[((size_t)b * a.cap + i) * 2 + 1] = y;
    a.conf[(size_t)b * a.cap + i] = nms_state_conf((uint32_t)(k >> 32));
  }
}

// The usual case, after nms_finish_kernel has settled what the round launches left: the frame's candidate list in
// at most NMS_MAX_CHUNKS slices (SLICE candidates each, longer ones when the frame has more than NMS_MAX_CHUNKS x
// SLICE), one workgroup per slice (grid (NMS_MAX_CHUNKS, B)).  Survivors of the slice inside the border -> LDS ->
// bitonic sort, descending -> a run of sort_scratch claimed with one atomicAdd on count[b] (which thereby ends up as
// the frame's K).  nms_merge_kernel then places every key at (its index in its run) + (the number of larger keys in
// every other run): keys are unique, so that is its position in the frame's order.  A slice with more than SLICE
// survivors (LDS holds no more) raises aux[0]: nms_sort_kernel's single workgroup then redoes that frame.
__device__ __forceinline__ void nms_slices(int n, int SLICE, int* S, int* C) {
  const int s = n <= NMS_MAX_CHUNKS * SLICE ? SLICE : ((n + NMS_MAX_CHUNKS - 1) / NMS_MAX_CHUNKS + 1023) / 1024 * 1024;
  *S = s;
  *C = (n + s - 1) / s;
}

__device__ __forceinline__ unsigned long long nms_key(uint32_t state, uint32_t ci) {
  return ((unsigned long long)(state & 0x7fffffffu) << 32) | (0xffffffffu - ci);
}

// One workgroup per frame owns all of the frame's candidates, so running rounds to completion cannot wait on anybody:
// the few candidates at the ends of long chains that the parallel launches gave up on are decided here -- collected
// in LDS first, so that a round costs their scans only, not a pass over the frame's whole candidate list.
constexpr int NMS_FINISH_LIST = 4096;
constexpr int NMS_FINISH_THREADS = 512;  // the unrolled window scan needs > 128 VGPRs

template <int R>
__device__ __forceinline__ void nms_finish_listed(uint32_t* map, uint32_t* cand, const int* list, int m, int H, int W,
                                                  int r) {
  while (true) {
    int pending = 0;
    for (int j = threadIdx.x; j < m; j += NMS_FINISH_THREADS) {
      const int i = list[j];
      const uint32_t ci = cand[i];
      if (ci & 0x80000000u) continue;
      const uint32_t v = ld_relaxed(map + ci);
      bool kept_nb, wait;
      nms_scan<R>(map, H, W, r, ci, v, &kept_nb, &wait);
      if (kept_nb) {
        st_relaxed(map + ci, 0u);
        cand[i] = ci | 0x80000000u;
      } else if (!wait) {
        st_relaxed(map + ci, v | 0x80000000u);
        cand[i] = ci | 0x80000000u;
      } else {
        pending = 1;
      }
    }
    if (!__syncthreads_or(pending)) break;  // (the barrier also orders this round's stores before the next round's loads)
  }
}

__global__ __launch_bounds__(NMS_FINISH_THREADS) void nms_finish_kernel(const NmsArgs a) {
  __shared__ int s_list[NMS_FINISH_LIST];
  __shared__ int s_n;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = a.ncand[b];
  uint32_t* wmap = a.nmsmap + (size_t)b * a.H * a.W;
  uint32_t* wcand = a.cand + (size_t)b * a.H * a.W;
  if (tid == 0) {
    a.count[b] = 0;  // nms_chunk_sort_kernel's allocation counter
    a.aux[(size_t)b * NMS_AUX_INTS] = 0;
    s_n = 0;
  }
  if (n == 1) {  // nms.py:23-25: a single candidate is returned as is
    if (tid == 0) {
      wmap[wcand[0] & 0x7fffffffu] |= 0x80000000u;
      wcand[0] |= 0x80000000u;
    }
    return;
  }
  __syncthreads();
  for (int i = tid; i < n; i += NMS_FINISH_THREADS)
    if (!(wcand[i] & 0x80000000u)) {
      const int j = atomicAdd(&s_n, 1);
      if (j < NMS_FINISH_LIST) s_list[j] = i;
    }
  __syncthreads();
  const int m = s_n;
  if (m == 0) return;
  if (m > NMS_FINISH_LIST) nms_run_rounds<0>(wmap, wcand, n, tid, NMS_FINISH_THREADS, a.H, a.W, a.r, true);
  else if (a.r == 4) nms_finish_listed<4>(wmap, wcand, s_list, m, a.H, a.W, a.r);
  else nms_finish_listed<0>(wmap, wcand, s_list, m, a.H, a.W, a.r);
}

template <int SLICE>
__global__ __launch_bounds__(1024) void nms_chunk_sort_kernel(const NmsArgs a) {
  static_assert(SLICE >= 2048 && SLICE <= 16384 && (SLICE & (SLICE - 1)) == 0, "slice: a power of two that LDS holds");
  extern __shared__ unsigned long long keys[];  // [SLICE]
  __shared__ int s_count, s_base;
  const int b = blockIdx.y, tid = threadIdx.x;
  const int n = a.ncand[b];
  if (n <= 0) return;
  int32_t* aux = a.aux + (size_t)b * NMS_AUX_INTS;
  int S, C;
  nms_slices(n, SLICE, &S, &C);
  const int H = a.H, W = a.W, bw = a.border;
  const uint32_t* map = a.nmsmap + (size_t)b * H * W;
  const uint32_t* cand = a.cand + (size_t)b * H * W;
  for (int c = blockIdx.x; c < C; c += gridDim.x) {
    __syncthreads();  // the previous slice's keys have left LDS
    if (tid == 0) s_count = 0;
#pragma unroll
    for (int it = 0; it < SLICE / 1024; ++it) keys[it * 1024 + tid] = 0ull;
    __syncthreads();
    for (int i0 = c * S; i0 < (c + 1) * S; i0 += 1024) {
      const int i = i0 + tid;
      const uint32_t ci = i < n ? cand[i] & 0x7fffffffu : 0u;
      const int y = ci / W, x = ci - y * W;
      const uint32_t u = i < n ? map[ci] : 0u;
      const bool keep = (u & 0x80000000u) && x >= bw && x < W - bw && y >= bw && y < H - bw;
      const unsigned long long mask = __ballot(keep);
      if (mask) {
        int base = 0;
        if ((tid & 63) == 0) base = atomicAdd(&s_count, __popcll(mask));
        base = __shfl(base, 0);
        const int at = base + __popcll(mask & ((1ull << (tid & 63)) - 1));
        if (keep && at < SLICE) keys[at] = nms_key(u, ci);
      }
    }
    __syncthreads();
    if (s_count > SLICE) {  // uniform
      if (tid == 0) aux[0] = 1;
      return;
    }
    const int K = s_count;
    int P = 128;
    while (P < K) P <<= 1;
    // bitonic sort, descending.  Pair t of a pass with distance j is elements i(t) and i(t) + j; for j <= 64 the pairs
    // t = 64 w .. 64 w + 63 of a wave lie in the same 128 elements in every such pass, so only the passes with
    // j >= 128 need the workgroup barrier.
    for (int k = 2; k <= P; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        if (j >= 128) __syncthreads();
        else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int t = tid; t < (P >> 1); t += 1024) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
          const unsigned long long ki = keys[i], kl = keys[l];
          const bool desc = (i & k) == 0;
          if (desc ? ki < kl : ki > kl) {
            keys[i] = kl;
            keys[l] = ki;
          }
        }
        if (j >= 128) __syncthreads();
        else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    __syncthreads();
    if (tid == 0) {
      const int base = K ? atomicAdd(&a.count[b], K) : 0;
      s_base = base;
      aux[2 + 2 * c] = base;
      aux[3 + 2 * c] = K;
    }
    __syncthreads();
    unsigned long long* run = a.sort_scratch + (size_t)b * a.sort_cap + s_base;
    for (int i = tid; i < K; i += 1024) run[i] = keys[i];
  }
}

// grid (G, B): a workgroup ranks the keys of runs blockIdx.x, + G, ... against the other runs and writes the outputs.
// The searches are what this kernel costs, and as one-key requests to L2 they were bound by the request rate of the
// vector memory pipeline (HD: 82 M of them per 64 frames): each other run is staged in LDS with coalesced loads and
// searched there; a thread keeps its SLICE / 1024 keys and their ranks in registers.
template <int SLICE>
__global__ __launch_bounds__(1024) void nms_merge_kernel(const NmsArgs a) {
  constexpr int E = SLICE / 1024;
  extern __shared__ unsigned long long other[];  // [SLICE]
  __shared__ int s_off[NMS_MAX_CHUNKS], s_cnt[NMS_MAX_CHUNKS];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int n = a.ncand[b];
  const int32_t* aux = a.aux + (size_t)b * NMS_AUX_INTS;
  if (n > 0 && aux[0]) return;  // nms_sort_kernel redoes this frame
  int S = 0, C = 0;
  if (n > 0) nms_slices(n, SLICE, &S, &C);
  if ((int)blockIdx.x >= C && blockIdx.x > 0) return;
  for (int q = tid; q < C; q += 1024) {
    s_off[q] = aux[2 + 2 * q];
    s_cnt[q] = aux[3 + 2 * q];
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) {  // count[b] was the allocation counter = K; the caller sees min(K, cap)
    int K = 0;
    for (int q = 0; q < C; ++q) K += s_cnt[q];
    a.count[b] = K < a.cap ? K : a.cap;
  }
  const int W = a.W;
  const unsigned long long* base = a.sort_scratch + (size_t)b * a.sort_cap;
  for (int c = blockIdx.x; c < C; c += gridDim.x) {
    const int Kc = s_cnt[c];
    if (Kc == 0) continue;
    unsigned long long key[E];
    int pos[E];
#pragma unroll
    for (int u = 0; u < E; ++u) {
      const int e = tid + u * 1024;
      key[u] = e < Kc ? base[s_off[c] + e] : ~0ull;  // (a key that is larger than every real one: its searches end at 0)
      pos[u] = e;
    }
    for (int q = 0; q < C; ++q) {
      const int m = s_cnt[q];
      if (q == c || m == 0) continue;
      __syncthreads();  // the previous run has been searched
      for (int i = tid; i < m; i += 1024) other[i] = base[s_off[q] + i];
      __syncthreads();
      // number of keys of the (descending) run larger than `key` = the largest idx with other[idx - 1] > key
      int lo[E];
#pragma unroll
      for (int u = 0; u < E; ++u) lo[u] = 0;
#pragma unroll 1
      for (int s = SLICE; s >= 1; s >>= 1) {
#pragma unroll
        for (int u = 0; u < E; ++u) {
          const int idx = lo[u] + s;
          if (idx <= m && other[idx - 1] > key[u]) lo[u] = idx;
        }
      }
#pragma unroll
      for (int u = 0; u < E; ++u) pos[u] += lo[u];
    }
#pragma unroll
    for (int u = 0; u < E; ++u) {
      const int e = tid + u * 1024;
      if (e < Kc && pos[u] < a.cap) {
        const uint32_t ci = 0xffffffffu - (uint32_t)(key[u] & 0xffffffffu);
        const int y = ci / W, x = ci - y * W;
        a.xy[((size_t)b * a.cap + pos[u]) * 2 + 0] = x;
        a.xy[((size_t)b * a.cap + pos[u]) * 2 + 1] = y;
        a.conf[(size_t)b * a.cap + pos[u]] = nms_state_conf((uint32_t)(key[u] >> 32));
      }
    }
  }
}

// The general path, one workgroup per frame; frames the two kernels above have dealt with are skipped
// (a.aux == nullptr: every frame is done here).
__global__ __launch_bounds__(1024) void nms_sort_kernel(const NmsArgs a) {
  extern __shared__ unsigned long long keys_lds[];
  __shared__ int s_count, s_total;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (a.aux && (a.ncand[b] <= 0 || !a.aux[(size_t)b * NMS_AUX_INTS])) return;  // nms_merge_kernel has done the frame
  const int H = a.H, W = a.W;
  const uint32_t* map = a.nmsmap + (size_t)b * H * W;
  const uint32_t* cand = a.cand + (size_t)b * H * W;
  const int n = a.ncand[b];
  if (tid == 0) {
    s_count = 0;
    s_total = 0;
  }
  // whatever the parallel launches left undecided (normally nothing): this workgroup owns all of
  // the frame's candidates, so running rounds to completion cannot wait on anybody
  uint32_t* wmap = a.nmsmap + (size_t)b * H * W;
  uint32_t* wcand = a.cand + (size_t)b * H * W;
  if (n == 1) {  // nms.py:23-25: a single candidate is returned as is
    if (tid == 0) wmap[wcand[0]] |= 0x80000000u;
  } else if (n > 1) {
    nms_run_rounds<0>(wmap, wcand, n, tid, 1024, H, W, a.r, true);
  }
  __syncthreads();
  // count first so that the LDS / scratch choice is uniform over the workgroup
  const int bw = a.border;
  int mine = 0;
  for (int i = tid; i < n; i += 1024) {
    const uint32_t ci = cand[i] & 0x7fffffffu;
    const int y = ci / W, x = ci - y * W;
    mine += (map[ci] & 0x80000000u) && x >= bw && x < W - bw && y >= bw && y < H - bw;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
  if ((tid & 63) == 0 && mine) atomicAdd(&s_total, mine);
  __syncthreads();
  const int K = s_total;
  int P = 1;
  while (P < K) P <<= 1;
  if (!a.aux && P <= NMS_LDS_KEYS)  // (as the chunked path's fallback the kernel is launched without dynamic LDS)
    nms_sort_body(a, keys_lds, P, K, n, map, cand, &s_count);
  else  // more survivors than LDS holds (large frames): same code on a global scratch buffer
    nms_sort_body(a, a.sort_scratch + (size_t)b * a.sort_cap, P, K, n, map, cand, &s_count);
}

// ---------------------------------------------------------------------------------
// get_descriptors (python/src/netutils.py:103-121): bilinear grid_sample with
// align_corners=True and zero padding at gx = x/(W/2) - 1, gy = y/(H/2) - 1, then
// division by the L2 norm (no epsilon).  One wave per keypoint, lane = 2 channels of
// the NHWC descriptor map (D = 128): each corner is one 512-byte coalesced read.
// ---------------------------------------------------------------------------------
// One keypoint per wave: `base` = the frame's NHWC map + VPL * lane, (gx, gy) the grid coordinates in [-1, 1].
template <int VPL>  // values per lane: 2 (D = 128, python net) or 4 (D = 256, cpp/src/settings.h:25)
__device__ __forceinline__ void descriptor_sample(const float* base, int cs, int Hc, int Wc, float gx, float gy, float* dst) {
#pragma clang fp contract(off)   // products and sums rounded separately, as grid_sample's C++ and the oracle do
  const float ix = ((gx + 1.f) / 2.f) * (float)(Wc - 1);
  const float iy = ((gy + 1.f) / 2.f) * (float)(Hc - 1);
  const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), x1 = x0 + 1, y1 = y0 + 1;
  const float wnw = ((float)x1 - ix) * ((float)y1 - iy), wne = (ix - (float)x0) * ((float)y1 - iy);
  const float wsw = ((float)x1 - ix) * (iy - (float)y0), wse = (ix - (float)x0) * (iy - (float)y0);
  const bool vx0 = x0 >= 0 && x0 < Wc, vx1 = x1 >= 0 && x1 < Wc, vy0 = y0 >= 0 && y0 < Hc, vy1 = y1 >= 0 && y1 < Hc;
  float v[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) v[i] = 0.f;
  // (no branch: a corner outside the map is read at a clamped position with weight 0 -- the same sums, and the loads of
  // two keypoints handled back to back can be in flight together)
  auto corner = [&](bool ok, int yy, int xx, float wgt) {
    const int yc = min(max(yy, 0), Hc - 1), xc = min(max(xx, 0), Wc - 1);
    const float wg = ok ? wgt : 0.f;
    const float* q = base + (size_t)(yc * Wc + xc) * cs;
    if (VPL == 2) {
      const float2 t = *reinterpret_cast<const float2*>(q);
      v[0] += t.x * wg;
      v[1] += t.y * wg;
    } else {
      const float4 t = *reinterpret_cast<const float4*>(q);
      v[0] += t.x * wg;
      v[1] += t.y * wg;
      v[VPL - 2] += t.z * wg;
      v[VPL - 1] += t.w * wg;
    }
  };
  corner(vy0 && vx0, y0, x0, wnw);
  corner(vy0 && vx1, y0, x1, wne);
  corner(vy1 && vx0, y1, x0, wsw);
  corner(vy1 && vx1, y1, x1, wse);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) ss += v[i] * v[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float nrm = sqrtf(ss);
  if (VPL == 2)
    *reinterpret_cast<float2*>(dst) = make_float2(v[0] / nrm, v[1] / nrm);
  else
    *reinterpret_cast<float4*>(dst) = make_float4(v[0] / nrm, v[1] / nrm, v[VPL - 2] / nrm, v[VPL - 1] / nrm);
}

// grid (G), G a multiple of 8: a persistent walk.  The keypoints of a frame come in order of confidence, i.e. in no
// spatial order, and every one reads four 512-byte cells of the frame's descriptor map: with the workgroups of ALL
// eight XCDs on the same frame, each XCD's L2 fetched nearly the whole map of every frame (2.1 GB of L2 misses per 64 HD
// frames by the PMC counters for 0.47 GB of maps), and the launch held one workgroup per four slots of the CAPACITY
// (590 k workgroups at HD, most of them leaving at once).  Here XCD k -- the workgroups with blockIdx.x & 7 == k --
// takes the frames b = k, k + 8, ... one after the other (by_xcd; with fewer than 8 frames all XCDs share each frame),
// and a workgroup strides over the frame's actual keypoints.
// SIXTEEN lanes per keypoint (CPL = D / 16 channels each, 16-byte loads) and four keypoints per wave: a quarter of the
// instructions per keypoint (one wave per keypoint, descriptor_sample above, spends them on 8-byte loads and a 6-step
// cross-lane sum for one norm) and four keypoints' requests in flight per wave.  Same per-channel sums in the same order; the sum of
// squares is taken over a lane's channels first, then a 4-step butterfly inside the sixteen lanes.
// BF16MAP (FPC_BF16, round 3): the descriptor map is bf16 -- the kernel runs at the fabric's rate for its bytes (1.7 GB
// in 0.34 ms per 64 HD frames with an fp32 map), and the map is half of them; a lane's eight channels are ONE 16-byte
// load per corner.  The arithmetic on the converted values is the same.
template <int CPL, bool BF16MAP = false>
__global__ __launch_bounds__(256) void descriptor16_kernel(const float* dmap, int cs, int Hc, int Wc, int H, int W,
                                                           const int32_t* count, const int32_t* xy, int cap,
                                                           float* out, int nframes, int by_xcd) {
#pragma clang fp contract(off)   // products and sums rounded separately, as grid_sample's C++ and the oracle do
  static_assert(!BF16MAP || CPL == 8, "a lane's channels of a bf16 map are one 16-byte load");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & 15, sub = lane >> 4;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int b0 = by_xcd ? xcd : 0, bstep = by_xcd ? 8 : 1;
  const int k0 = ((by_xcd ? slot : (int)blockIdx.x) * 4 + wave) * 4, kstep = (by_xcd ? nslots : (int)gridDim.x) * 16;
  for (int b = b0; b < nframes; b += bstep) {
    const int K = min(count[b], cap);
    const float* base = dmap + (size_t)b * Hc * Wc * cs + CPL * q;
    const unsigned short* base16 = reinterpret_cast<const unsigned short*>(dmap) + (size_t)b * Hc * Wc * cs + CPL * q;
    for (int kw = k0; kw < K; kw += kstep) {
      const bool live = kw + sub < K;
      const int k = live ? kw + sub : K - 1;
      const int2 p = *reinterpret_cast<const int2*>(xy + ((size_t)b * cap + k) * 2);
      const float gx = (float)((double)p.x / ((double)W / 2.) - 1.), gy = (float)((double)p.y / ((double)H / 2.) - 1.);
      const float ix = ((gx + 1.f) / 2.f) * (float)(Wc - 1);
      const float iy = ((gy + 1.f) / 2.f) * (float)(Hc - 1);
      const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), x1 = x0 + 1, y1 = y0 + 1;
      const float wnw = ((float)x1 - ix) * ((float)y1 - iy), wne = (ix - (float)x0) * ((float)y1 - iy);
      const float wsw = ((float)x1 - ix) * (iy - (float)y0), wse = (ix - (float)x0) * (iy - (float)y0);
      const bool vx0 = x0 >= 0 && x0 < Wc, vx1 = x1 >= 0 && x1 < Wc, vy0 = y0 >= 0 && y0 < Hc, vy1 = y1 >= 0 && y1 < Hc;
      float v[CPL];
#pragma unroll
      for (int i = 0; i < CPL; ++i) v[i] = 0.f;
      const int cy[4] = {y0, y0, y1, y1}, cx[4] = {x0, x1, x0, x1};
      const float cw[4] = {(vy0 && vx0) ? wnw : 0.f, (vy0 && vx1) ? wne : 0.f, (vy1 && vx0) ? wsw : 0.f, (vy1 && vx1) ? wse : 0.f};
      float4 t[4][CPL / 4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {   // a corner outside the map: a clamped position with weight 0
        const int yc = min(max(cy[c], 0), Hc - 1), xc = min(max(cx[c], 0), Wc - 1);
        if constexpr (BF16MAP) {
          const uint4 u = *reinterpret_cast<const uint4*>(base16 + (size_t)(yc * Wc + xc) * cs);
          t[c][0] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
          t[c][1] = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u));
        } else {
          const float4* qp = reinterpret_cast<const float4*>(base + (size_t)(yc * Wc + xc) * cs);
#pragma unroll
          for (int i = 0; i < CPL / 4; ++i) t[c][i] = qp[i];
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < CPL / 4; ++i) {
          v[4 * i + 0] += t[c][i].x * cw[c];
          v[4 * i + 1] += t[c][i].y * cw[c];
          v[4 * i + 2] += t[c][i].z * cw[c];
          v[4 * i + 3] += t[c][i].w * cw[c];
        }
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < CPL; ++i) ss += v[i] * v[i];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
      const float nrm = sqrtf(ss);
      if (live) {
        float4* dst = reinterpret_cast<float4*>(out + ((size_t)b * cap + k) * (16 * CPL) + CPL * q);
#pragma unroll
        for (int i = 0; i < CPL / 4; ++i) dst[i] = make_float4(v[4 * i] / nrm, v[4 * i + 1] / nrm, v[4 * i + 2] / nrm, v[4 * i + 3] / nrm);
      }
    }
  }
}

// get_descriptors on its own (netutils.py:103-121): K caller-provided points (x, y) as float64 -- the reference's
// `points` array -- on ONE descriptor map; the normalisation runs in double and is rounded to float once, as
// `sample_points.float()` does there.  out [K][D].
template <int VPL>
__global__ __launch_bounds__(256) void descriptor_at_points_kernel(const float* dmap, int cs, int Hc, int Wc, int H, int W,
                                                                   const double* xy, int K, float* out) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  const float gx = (float)(xy[(size_t)k * 2] / ((double)W / 2.) - 1.);
  const float gy = (float)(xy[(size_t)k * 2 + 1] / ((double)H / 2.) - 1.);
  descriptor_sample<VPL>(dmap + VPL * lane, cs, Hc, Wc, gx, gy, out + (size_t)k * (64 * VPL) + VPL * lane);
}

// NHWC (pixel stride cs, first C channels) -> NCHW, for the reference-layout dense outputs.
// `range` (nullable): the frame's largest |value| goes to its word `word` (fpc_forward's descriptor map: fpc_output_range)
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* in, int cs, int C, int HW, int B, float* out,
                                                           uint32_t* range = nullptr, int word = 0) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool in_range = i < (size_t)B * C * HW;
  const int p = i % HW;
  const int c = (i / HW) % C;
  const int b = in_range ? (int)(i / ((size_t)HW * C)) : B - 1;
  float v = 0.f;
  if (in_range) out[i] = v = in[((size_t)b * HW + p) * cs + c];
  if (range) {
    // (a wave may straddle two frames: the maximum goes to the frame of its first lane's element and of its last one's --
    // a frame's word may then carry its neighbour's maximum: the words bound the tensor, which is all they are for)
    float m = fabsf(v);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const int lane = threadIdx.x & 63;
    if (lane == 0 || lane == 63) {
      uint32_t* w = range + b * FPC_RANGE_WORDS + word;
      if (__float_as_uint(m) > __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(w, __float_as_uint(m));
    }
  }
}

// the same from a bf16 NHWC tensor (FPC_BF16 mode; fpc_read_activation)
__global__ __launch_bounds__(256) void nhwc_bf16_to_nchw_kernel(const unsigned short* in, int cs, int C, int HW, int B, float* out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)B * C * HW) return;
  const int p = i % HW;
  const int c = (i / HW) % C;
  const int b = i / ((size_t)HW * C);
  out[i] = __uint_as_float((unsigned)in[((size_t)b * HW + p) * cs + c] << 16);
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* in, int C, int HW, int B, float* out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)B * C * HW) return;
  const int c = i % C;
  const int p = (i / C) % HW;
  const int b = i / ((size_t)HW * C);
  out[i] = in[((size_t)b * C + c) * HW + p];
}

// ---------------------------------------------------------------------------------
// 8-bit camera frames -> planar float frames (fpc_detect_u8; python/src/camera.py:31,
// inferencewrapper.py:70-81, inference.py:79, cpp/src/camera.cc:17-18).  HBM-bound:
// 1 or 3 bytes in, 4 or 12 bytes out per pixel; four pixels per thread.
// layout: 0 gray, 1 RGB HWC, 2 BGR HWC (swap), 3 BGR HWC -> gray (OpenCV 8-bit fixed point)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void u8_to_float_kernel(const unsigned char* __restrict__ in, float* __restrict__ out,
                                                           int n, int HW, int layout) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;  // over n * HW / 4 pixel quads (HW % 4 == 0)
  const size_t quads = (size_t)n * (HW / 4);
  if (q >= quads) return;
  const size_t f = q / (HW / 4), p = (q - f * (HW / 4)) * 4;
  if (layout == 0) {
    const uchar4 v = *reinterpret_cast<const uchar4*>(in + f * HW + p);
    *reinterpret_cast<float4*>(out + f * HW + p) = make_float4((float)v.x / 255.0f, (float)v.y / 255.0f, (float)v.z / 255.0f, (float)v.w / 255.0f);
    return;
  }
  const unsigned char* src = in + (f * HW + p) * 3;
  const uint3 w = *reinterpret_cast<const uint3*>(src);  // 12 bytes = 4 pixels x 3 channels (12 | offset)
  unsigned char b[12];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b[i] = (w.x >> (8 * i)) & 0xff;
    b[4 + i] = (w.y >> (8 * i)) & 0xff;
    b[8 + i] = (w.z >> (8 * i)) & 0xff;
  }
  if (layout == 3) {
    const float k = (float)(1.0 / 255.0);
    float g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned y = (b[3 * i] * 1868u + b[3 * i + 1] * 9617u + b[3 * i + 2] * 4899u + 8192u) >> 14;
      g[i] = __fmul_rn((float)y, k);
    }
    *reinterpret_cast<float4*>(out + f * HW + p) = make_float4(g[0], g[1], g[2], g[3]);
    return;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sc = layout == 2 ? 2 - c : c;
    *reinterpret_cast<float4*>(out + (f * 3 + c) * HW + p) =
        make_float4((float)b[sc] / 255.0f, (float)b[3 + sc] / 255.0f, (float)b[6 + sc] / 255.0f, (float)b[9 + sc] / 255.0f);
  }
}

// ---------------------------------------------------------------------------------
// Camera frame -> network input in one pass (python/src/inference.py:72-85 make_query_image after camera.py:31):
// float32(u8) / 255, optional BGR -> RGB, ratio-preserving bilinear resize to cover the target, centre crop,
// HWC -> CHW.  Geometry exactly as the reference computes it:
//   scale = max(H / src_h, W / src_w);  new_w = int(src_w * scale), new_h = int(src_h * scale)   (host, double)
//   resized = cv2.resize(img, (new_w, new_h), INTER_LINEAR);  crop at x0 = new_w // 2 - W // 2, y0 = new_h // 2 - H // 2
// cv2.resize on float32 data (OpenCV resize.cpp, INTER_LINEAR): fx = float((dx + 0.5) * (src_w / new_w) - 0.5) with the
// ratio in double, sx = floor(fx), clamped at both edges with weight 0; horizontal interpolation first, then vertical,
// all in fp32.  OpenCV is absent from this image: the restatement is pinned against torch's
// F.interpolate(bilinear, align_corners=False) -- the same sampling rule with the scale in fp32 -- within 1e-5
// (fixture F9); PARITY with cv2 itself is unpinned.
// ---------------------------------------------------------------------------------
struct ResizeArgs {
  const unsigned char* in;   // [n][src_h][src_w][3] u8
  float* out;                // [n][3 or 1][H][W]
  int n, src_h, src_w, H, W;
  int new_w, new_h, x0, y0;  // resized size and crop origin
  double scale_x, scale_y;   // src_w / new_w, src_h / new_h
  int swap_rb;               // 1: BGR input -> RGB planes
};

__global__ __launch_bounds__(256) void resize_crop_u8_kernel(const ResizeArgs a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t HW = (size_t)a.H * a.W;
  if (i >= (size_t)a.n * HW) return;
  const int f = i / HW;
  const int r = i - (size_t)f * HW;
  const int oy = r / a.W, ox = r - oy * a.W;
  const int dx = ox + a.x0, dy = oy + a.y0;
  float fx = (float)(((double)dx + 0.5) * a.scale_x - 0.5), fy = (float)(((double)dy + 0.5) * a.scale_y - 0.5);
  int sx = (int)floorf(fx), sy = (int)floorf(fy);
  fx -= (float)sx;
  fy -= (float)sy;
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= a.src_w - 1) { fx = 0.f; sx = a.src_w - 1; }
  if (sy < 0) { fy = 0.f; sy = 0; }
  if (sy >= a.src_h - 1) { fy = 0.f; sy = a.src_h - 1; }
  const int sx1 = sx + 1 < a.src_w ? sx + 1 : sx, sy1 = sy + 1 < a.src_h ? sy + 1 : sy;
  const unsigned char* base = a.in + (size_t)f * a.src_h * a.src_w * 3;
  const unsigned char* p00 = base + ((size_t)sy * a.src_w + sx) * 3;
  const unsigned char* p01 = base + ((size_t)sy * a.src_w + sx1) * 3;
  const unsigned char* p10 = base + ((size_t)sy1 * a.src_w + sx) * 3;
  const unsigned char* p11 = base + ((size_t)sy1 * a.src_w + sx1) * 3;
  const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sc = a.swap_rb ? 2 - c : c;
    const float v00 = (float)p00[sc] / 255.0f, v01 = (float)p01[sc] / 255.0f;
    const float v10 = (float)p10[sc] / 255.0f, v11 = (float)p11[sc] / 255.0f;
    const float r0 = __fmaf_rn(v01, a1, v00 * a0), r1 = __fmaf_rn(v11, a1, v10 * a0);   // HResizeLinear
    a.out[((size_t)f * 3 + c) * HW + r] = __fmaf_rn(r1, b1, r0 * b0);                      // VResizeLinear
  }
}

// ---------------------------------------------------------------------------------
// Kernels only the reference's C++ network needs (superpoint::SPModel, cpp/src/model.cc).
// ---------------------------------------------------------------------------------
// encoder_conv0_a: Conv2d(1, 64, 3, padding 1) + bias + ReLU on a gray frame (model.cc:66-68).  K = 9 is no
// GEMM: plain fp32 FMAs, one thread = one pixel x 16 output channels; HBM-bound on its 256 B/pixel output.
struct VggConv0Args {
  const float* in;   // [B,1,H,W]
  const float* w;    // [9][64] (tap-major), bias [64] behind it
  float* out;        // [B,H,W,64]
  int H, W, frame0;
};
__global__ __launch_bounds__(256) void vgg_conv0_kernel(const VggConv0Args a) {
  __shared__ float wl[10 * 64];
  for (int i = threadIdx.x; i < 640; i += 256) wl[i] = a.w[i];
  __syncthreads();
  const int cg = threadIdx.x & 3;                                   // 16-channel group
  const size_t pix = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2);  // over n*H*W of this launch
  const size_t HW = (size_t)a.H * a.W;
  const int bl = pix / HW;
  const int r = pix - bl * HW;
  const int y = r / a.W, x = r - y * a.W;
  const float* src = a.in + (size_t)(a.frame0 + bl) * HW;
  float v[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = y + ky - 1, ix = x + kx - 1;
      const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const float t = src[ok ? (size_t)iy * a.W + ix : 0];
      v[ky * 3 + kx] = ok ? t : 0.f;
    }
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = fmaf(v[t], wl[t * 64 + cg * 16 + j], acc[j]);
  float* dst = a.out + ((size_t)(a.frame0 + bl) * HW + r) * 64 + cg * 16;
#pragma unroll
  for (int j = 0; j < 16; j += 4) {
    float4 o;
    o.x = fmaxf(acc[j] + wl[576 + cg * 16 + j], 0.f);
    o.y = fmaxf(acc[j + 1] + wl[576 + cg * 16 + j + 1], 0.f);
    o.z = fmaxf(acc[j + 2] + wl[576 + cg * 16 + j + 2], 0.f);
    o.w = fmaxf(acc[j + 3] + wl[576 + cg * 16 + j + 3], 0.f);
    *reinterpret_cast<float4*>(dst + j) = o;
  }
}

// torch::max_pool2d(x, 2, 2) on NHWC (model.cc:74); C4 = channels / 4
__global__ __launch_bounds__(256) void maxpool2_kernel(const float4* in, float4* out, int n, int H, int W, int C4,
                                                       int frame0) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int Ho = H / 2, Wo = W / 2;
  if (i >= (size_t)n * Ho * Wo * C4) return;
  const int c = i % C4;
  size_t p = i / C4;
  const int x = p % Wo;
  p /= Wo;
  const int y = p % Ho;
  const int b = frame0 + (int)(p / Ho);
  const float4* s = in + (((size_t)b * H + 2 * y) * W + 2 * x) * C4 + c;
  const float4 q0 = s[0], q1 = s[C4], q2 = s[(size_t)W * C4], q3 = s[(size_t)W * C4 + C4];
  float4 m;
  m.x = fmaxf(fmaxf(q0.x, q1.x), fmaxf(q2.x, q3.x));
  m.y = fmaxf(fmaxf(q0.y, q1.y), fmaxf(q2.y, q3.y));
  m.z = fmaxf(fmaxf(q0.z, q1.z), fmaxf(q2.z, q3.z));
  m.w = fmaxf(fmaxf(q0.w, q1.w), fmaxf(q2.w, q3.w));
  out[(((size_t)b * Ho + y) * Wo + x) * C4 + c] = m;
}

// desc / norm(desc, 2, dim = 1) (model.cc:90-91): one wave per pixel of the 256-channel NHWC map, in place
__global__ __launch_bounds__(256) void l2norm256_kernel(float* d, size_t npix) {
  const size_t pix = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= npix) return;
  float4* p = reinterpret_cast<float4*>(d + pix * 256) + (threadIdx.x & 63);
  float4 v = *p;
  float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float nrm = sqrtf(ss);
  *p = make_float4(v.x / nrm, v.y / nrm, v.z / nrm, v.w / nrm);
}

}  // namespace fpc
