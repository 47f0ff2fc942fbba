// stem_wino.h -- the stem (7x7 stride-2 convolution + folded BN + ReLU + 3x3/2 max-pool, superpoint.py:12-15,20-23) with
// HALF the matrix work: polyphase + Winograd F(2x2, 4x4).
//
//   out[y][x] = sum_{ky,kx,c} w[ky][kx][c] in[2y + ky - 3][2x + kx - 3].   Pad the filter with a zero row / column in front
//   (w8[ky + 1][kx + 1] = w[ky][kx]) and split ky8 = 2u + a, kx8 = 2v + b: input row 2y + ky8 - 4 = 2(y + u - 2) + a, so with
//   the four PHASE images P_ab[i][j] = in[2i + a][2j + b]
//       out[y][x] = sum_{a,b,c} sum_{u,v<4} w8[2u + a][2v + b][c] P_ab,c[y + u - 2][x + v - 2]
//   -- four stride-1 correlations with 4x4 filters on 3 channels each: 12 "channels" of a 4x4 convolution.  F(2x2, 4x4)
//   (points 0, 1, -1, 2, inf) computes a 2x2 output tile from 25 products per channel instead of 64:
//       V = B^T d B (5x5 patch of a phase image),  M_xi = sum_k V_xi[k] U_xi[k][n],  Y = A^T M A.
//   Per 2x2 outputs and output channel: 25 x 12 = 300 multiply-adds where the direct form has 4 x 147 = 588 (152 x 4 = 608 as
//   padded on the MFMAs): the matrix cores do half the work, K = 12 is exactly three v_mfma_f32_16x16x4_f32 per position.
//   fp32 error against fp64 measured before anything was built (experiments/harness/stem_wino_numerics.py): 1.7e-6 at an
//   output scale of 3 -- the direct fp32 form's order, two orders inside the 1e-4 bar.
//
// Since K is only three MFMAs deep, a position's accumulator is finished three instructions after it is started: the
// kernel walks the 5x5 positions COLUMN BY COLUMN -- five positions x two 16-tile blocks = 40 accumulator registers --
// applies A^T along the column at once and adds the column's share to the 2x2 outputs (32 registers): no 25-position
// accumulator array, so several workgroups fit a CU.
#pragma once
#include "../feature-point-cnn_amd/csrc/kernels_misc.h"

namespace fpc {

struct StemWinoArgs {
  const float* in;      // [B,3,H,W]
  const float4* u;      // [25 positions][4 channel blocks][64 lanes] float4: lane (kq = l >> 4, n = l & 15) holds
                        // U[pos][kq][n'], U[pos][4 + kq][n'], U[pos][8 + kq][n'], 0 with n' = 16 block + n, k = (2a + b) * 3 + c
  const float* bias;    // [64] folded BN bias
  float* out;           // [B,Hp,Wp,64]; cells with pooled row % 4 == 0 or column % 8 == 0 zeroed by the host (atomicMax targets)
  int H, W, Ho, Wo, Hp, Wp, tiles_x, tiles_y;
  int total;            // tiles_x * tiles_y * frames: the grid is persistent (a multiple of 8 workgroups, XCD k walks tiles [k T / 8, (k + 1) T / 8))
};

constexpr int SW_TH = 8, SW_TW = 16;                 // conv-output pixels of a tile: 4 x 8 Winograd tiles of 2 x 2
constexpr int SW_ROWS = 2 * SW_TH + 6, SW_LW = 40;   // input window: 22 rows x 40 columns (38 used) per channel
constexpr int SW_WIN = 3 * SW_ROWS * SW_LW;          // floats
constexpr int SW_V = 25 * 2 * 3 * 64;                // V[pos][tile block][k step][lane = kq * 16 + m]
constexpr int SW_TROW = 65;                          // floats per pixel of the conv tile in LDS: 64 channels + 1 skew
constexpr int SW_TILE = SW_TH * SW_TW * SW_TROW;     // the conv tile (all 64 channels), followed by one row of "no pixel" values
constexpr int SW_LDS_FLOATS = SW_WIN + (SW_V > SW_TILE + SW_TW * SW_TROW ? SW_V : SW_TILE + SW_TW * SW_TROW);

__device__ __forceinline__ float sw_fma(float k, float x, float y) { return __builtin_fmaf(k, x, y); }
typedef float sw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ sw_f32x2 sw_fma(float k, sw_f32x2 x, sw_f32x2 y) { return __builtin_elementwise_fma(sw_f32x2{k, k}, x, y); }
// B^T d for five values (in place): points 0, 1, -1, 2, inf -- nine operations
template <class T>
__device__ __forceinline__ void sw_bt5(T& d0, T& d1, T& d2, T& d3, T& d4) {
  const T a = d3 - d1, t = d3 - d2, e0 = d0 - d2, e4 = d4 - d2;
  const T v0 = sw_fma(2.f, e0, a);                           // 2 d0 - d1 - 2 d2 + d3
  const T v1 = sw_fma(-2.f, d1, t);                          // -2 d1 - d2 + d3
  const T v2 = sw_fma(2.f, d1, sw_fma(-3.f, d2, d3));        // 2 d1 - 3 d2 + d3
  const T v4 = sw_fma(-2.f, a, e4);                          // 2 d1 - d2 - 2 d3 + d4
  d0 = v0;
  d1 = v1;
  d2 = v2;
  d3 = a;
  d4 = v4;
}

// MaxPool2d(3, stride 2, padding 1) of one 32-channel half of the SW_TH x SW_TW conv tile in LDS (pre-ReLU values; the ReLU
// is applied to the pooled value).  Thread (c = tid & 31, j = tid >> 5) owns pooled row j of the tile (rows 0 .. SW_TH / 2:
// the last one holds conv row SW_TH - 1 only, the rest of its windows belongs to the tile below).  Cells whose window
// straddles tiles -- pooled row 0 or SW_TH / 2, pooled column 0 or 8 -- are completed with atomicMax on the float bits.
// (v_max3_f32 / v_max_f32 as the instructions they are: `fmaxf` canonicalises both operands first -- a v_max_f32 x, x, x in
// front of every maximum, which doubled the pooling's VALU count; the operands here are finite convolution outputs)
__device__ __forceinline__ float sw_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float sw_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ void sw_pool_emit(const float* tile, float* out, int b, int ty, int tx, int nb, int Ho, int Wo, int Hp,
                                             int Wp, int tid) {
  const int c = nb * 32 + (tid & 31), py = tid >> 5;
  if (py > SW_TH / 2) return;
  const int gy0 = ty * SW_TH, gx0 = tx * SW_TW;
  const int gpy = ty * (SW_TH / 2) + py;
  if (gpy >= Hp) return;
  constexpr float NONE = -3.0e38f;
  // the (up to) three conv rows of this pooled row: a row outside the tile or the frame is read from the NONE row behind
  // the tile (no select per element)
  float cm[SW_TW];
  {
    const int r0 = 2 * py - 1;
    const bool ok0 = r0 >= 0 && gy0 + r0 < Ho, ok1 = r0 + 1 <= SW_TH - 1 && gy0 + r0 + 1 < Ho, ok2 = r0 + 2 <= SW_TH - 1 && gy0 + r0 + 2 < Ho;
    const float* t0 = tile + (ok0 ? r0 * SW_TW * SW_TROW : SW_TILE) + c;
    const float* t1 = tile + (ok1 ? (r0 + 1) * SW_TW * SW_TROW : SW_TILE) + c;
    const float* t2 = tile + (ok2 ? (r0 + 2) * SW_TW * SW_TROW : SW_TILE) + c;
#pragma unroll
    for (int cc = 0; cc < SW_TW; ++cc) {
      cm[cc] = sw_max3(t0[cc * SW_TROW], t1[cc * SW_TROW], t2[cc * SW_TROW]);
      if (cc % 4 == 3) __builtin_amdgcn_sched_barrier(0);      // (twelve reads in flight, not forty-eight: registers)
    }
  }
  const int wlim = Wo - gx0;      // conv columns of the tile inside the frame
  if (wlim < SW_TW) {             // (uniform: the frame's last tile column only)
#pragma unroll
    for (int cc = 0; cc < SW_TW; ++cc) cm[cc] = cc < wlim ? cm[cc] : NONE;
  }
  float* row = out + ((size_t)(b * Hp + gpy) * Wp + tx * 8) * 64 + c;
#pragma unroll
  for (int px = 0; px < 9; ++px) {
    if (tx * 8 + px >= Wp) continue;
    const float l = px > 0 ? cm[px > 0 ? 2 * px - 1 : 0] : NONE;
    const float m = px < 8 ? cm[px < 8 ? 2 * px : 0] : NONE;
    const float r = px < 8 ? cm[px < 8 ? 2 * px + 1 : 0] : NONE;
    float mx = sw_max3(l, m, r);
    if (mx < -1.0e38f) continue;      // no pixel of this window lies in this tile
    mx = sw_max(mx, 0.f);             // ReLU (>= +0 from here on: the atomicMax on the float bits relies on it)
    if (py >= 1 && py <= SW_TH / 2 - 1 && px >= 1 && px <= 7)
      row[px * 64] = mx;
    else
      atomicMax(reinterpret_cast<unsigned int*>(row + px * 64), __float_as_uint(mx));
  }
}

__global__ __launch_bounds__(256, 3) void stem_wino_kernel(const StemWinoArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[SW_LDS_FLOATS];
  float* const win = lds;
  float* const vbuf = lds + SW_WIN;      // V, then the conv tile of one 32-channel half
  typedef float f32x4w __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = a.tiles_x * a.tiles_y;
  // Persistent grid: a tile's input window is requested a whole tile ahead (into registers, stored to LDS once the current
  // tile's transform has read the previous one) -- with one tile per workgroup the window's trip to HBM (2-3 us) was as
  // long as the tile's arithmetic and three workgroups per CU did not cover it.
  const int T = a.total, per = gridDim.x >> 3, xcd = blockIdx.x & 7;
  const int t_first = (int)(((long long)xcd * T) >> 3), t_end = (int)(((long long)(xcd + 1) * T) >> 3);

  // this wave's filter fragments, a column of positions ahead (L2-resident: 25 KB per channel block)
  const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.u), 0, 25 * 4 * 64 * 16, 0x00020000);
  const int ulane = (wave * 64 + lane) * 16;
  auto ldu = [&](int pos) { return __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(ursrc, ulane, pos * 4096, 0)); };

  constexpr int NQ = SW_LW / 4, NE = 3 * SW_ROWS * NQ, IT = (NE + 255) / 256;
  float4 wv[IT];
  auto load_window = [&](int tt) {      // aligned float4 row segments (ix0 is a multiple of 4, W of 8: a float4 is inside or outside)
    const bool live = tt < t_end;
    const int tc = live ? tt : t_first;
    const int b_ = tc / tiles, t_ = tc - b_ * tiles;
    const int ty_ = t_ / a.tiles_x, tx_ = t_ - ty_ * a.tiles_x;
    const int iy0 = ty_ * SW_TH * 2 - 4, ix0 = tx_ * SW_TW * 2 - 4;      // window row 0 / column 0 in the frame
    int tl = tid;
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int e = tl + i * 256;
      const int row = e / NQ, q = e - row * NQ;
      const int c = row / SW_ROWS, hy = row - c * SW_ROWS;
      const int iy = iy0 + hy, ix = ix0 + 4 * q;
      const bool ok = live && e < NE && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const float4 x = *reinterpret_cast<const float4*>(a.in + (ok ? ((size_t)(b_ * 3 + c) * a.H + iy) * a.W + ix : 0));
      wv[i] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_window = [&]() {
    int tl = tid;
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int e = tl + i * 256;
      if (e < NE) *reinterpret_cast<float4*>(win + e * 4) = wv[i];
    }
  };
  load_window(t_first + (int)(blockIdx.x >> 3));
  store_window();
  __syncthreads();

  for (int tcur = t_first + (int)(blockIdx.x >> 3); tcur < t_end; tcur += per) {
  const int b = tcur / tiles;
  const int t = tcur - b * tiles;
  const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
  load_window(tcur + per);      // in flight during this tile's transform and GEMMs
  f32x4w ucol[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) ucol[i] = ldu(i * 5);

  // ---- input transform: item = (Winograd tile wt, phase-channel k): 32 x 12 = 384 items, V = B^T d B of its 5x5 patch.
  // A thread takes the PAIR k = 2 kk, 2 kk + 1 of one tile as the two halves of v_pk_* operands (192 threads; every
  // operation of the transform is one packed instruction for both items).
  typedef float f32x2w __attribute__((ext_vector_type(2)));
  int tid_x = tid;      // (a copy the optimiser cannot see through: per-thread indices are invariants of the tile loop otherwise)
  asm volatile("" : "+v"(tid_x));
  if (tid_x < 192) {
    const int wt = tid_x & 31, kk = tid_x >> 5;
    const int ty_t = wt >> 3, tx_t = wt & 7;
    int base[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = 2 * kk + e, ph = k / 3, c = k - ph * 3;
      base[e] = c * (SW_ROWS * SW_LW) + (4 * ty_t + (ph >> 1)) * SW_LW + 4 * tx_t + (ph & 1);
    }
    f32x2w d[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) d[i][j] = f32x2w{win[base[0] + i * 2 * SW_LW + j * 2], win[base[1] + i * 2 * SW_LW + j * 2]};
#pragma unroll
    for (int j = 0; j < 5; ++j) sw_bt5(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j]);
#pragma unroll
    for (int i = 0; i < 5; ++i) sw_bt5(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4]);
    // k = 2 kk (+ 1): k step kk >> 1, kq = 2 (kk & 1) (+ 1)
    const int vb = (wt >> 4) * 192 + (kk >> 1) * 64 + (2 * (kk & 1)) * 16 + (wt & 15);
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        vbuf[(i * 5 + j) * 384 + vb] = d[i][j][0];
        vbuf[(i * 5 + j) * 384 + vb + 16] = d[i][j][1];
      }
  }
  __syncthreads();

  // ---- 25 GEMMs of K = 12, column by column (a position's accumulator is finished after three MFMAs): thirty MFMAs, then
  // A^T along the column at once and the column's share added to the 2x2 outputs -- on four-wide vectors (v_pk_*: two
  // accumulator rows per instruction).  (Issuing the next column's MFMAs in front of this column's transform needs a second
  // accumulator set: 198 registers, two workgroups per CU instead of three, measured 13 % slower.)
  const float bias = a.bias[16 * wave + (lane & 15)];
  f32x4w y[2][2][2];      // [output row][output column][tile block], starting at the folded-BN bias
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) y[p][q][mb] = f32x4w{bias, bias, bias, bias};
  // (a real loop over the columns: unrolled, the scheduler reads later columns' operands ahead until the register file is
  // full and the allocator spills the outputs -- 40 registers to scratch at three workgroups per CU)
  f32x4w unext[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) unext[i] = ldu(i * 5 + 1);
#pragma nounroll
  for (int j = 0; j < 5; ++j) {
    f32x4w acc[5][2];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) acc[i][mb] = f32x4w{0.f, 0.f, 0.f, 0.f};
    const float* vj = vbuf + j * 384 + lane;
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_)       // ten independent accumulators in turn: no MFMA waits for its predecessor
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
          acc[i][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(vj[i * 5 * 384 + mb * 192 + s_ * 64], ucol[i][s_], acc[i][mb], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 5; ++i) ucol[i] = unext[i];
    {
      const int jn = j + 2 < 5 ? j + 2 : 4;      // (the last two trips re-read column 4: in range, unused)
#pragma unroll
      for (int i = 0; i < 5; ++i) unext[i] = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(ursrc, ulane, (i * 5 + jn) * 4096, 0));
    }
    // Y[.][0] += A^T[0][j] T, Y[.][1] += A^T[1][j] T with A^T = (1, 1, 1, 1, 0; 0, 1, -1, 2, 1)
    const float c0 = j < 4 ? 1.f : 0.f, c1 = j == 0 ? 0.f : j == 2 ? -1.f : j == 3 ? 2.f : 1.f;
    const f32x4w k0 = {c0, c0, c0, c0}, k1 = {c1, c1, c1, c1}, two = {2.f, 2.f, 2.f, 2.f};
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      // T = A^T M along the column: t0 = m0 + m1 + m2 + m3, t1 = m1 - m2 + 2 m3 + m4
      const f32x4w t0 = (acc[0][mb] + acc[1][mb]) + (acc[2][mb] + acc[3][mb]);
      const f32x4w t1 = __builtin_elementwise_fma(two, acc[3][mb], acc[1][mb] - acc[2][mb]) + acc[4][mb];
      y[0][0][mb] = __builtin_elementwise_fma(k0, t0, y[0][0][mb]);
      y[1][0][mb] = __builtin_elementwise_fma(k0, t1, y[1][0][mb]);
      y[0][1][mb] = __builtin_elementwise_fma(k1, t0, y[0][1][mb]);
      y[1][1][mb] = __builtin_elementwise_fma(k1, t1, y[1][1][mb]);
    }
  }

  store_window();       // the next tile's window (every wave has read this tile's: the barrier behind the transform)
  // ---- epilogue: conv + bias -> LDS tile -> max-pool (ReLU on the pooled value)
  __syncthreads();      // V has been read by every wave: its LDS becomes the conv tile [128 pixels][64 channels + 1]
  // (every per-thread index of the epilogue comes from a copy of the thread id the optimiser cannot see through: computed
  // from threadIdx.x they are scheduled in front of the GEMMs -- fifty address registers alive across them, and spills)
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  for (int i = tid_e; i < SW_TW * SW_TROW; i += 256) vbuf[SW_TILE + i] = -3.0e38f;      // the "no pixel" row of sw_pool_emit
  {
    const int n16 = tid_e & 15, kq = (tid_e >> 4) & 3;
    // accumulator row r of block mb, lane (n16, kq): Winograd tile 16 mb + 4 kq + r = (row 2 mb + (kq >> 1), column 4 (kq & 1) + r)
    // -> conv pixel (4 mb + 2 (kq >> 1) + p, 8 (kq & 1) + 2 r + q): one per-lane base, the rest compile-time offsets
    float* const tb = vbuf + ((2 * (kq >> 1)) * SW_TW + 8 * (kq & 1)) * SW_TROW + 16 * wave + n16;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int q = 0; q < 2; ++q) tb[((4 * mb + p) * SW_TW + 2 * r + q) * SW_TROW] = y[p][q][mb][r];
  }
  __syncthreads();
  sw_pool_emit(vbuf, a.out, b, ty, tx, 0, a.Ho, a.Wo, a.Hp, a.Wp, tid_e);
  asm volatile("" : "+v"(tid_e));
  sw_pool_emit(vbuf, a.out, b, ty, tx, 1, a.Ho, a.Wo, a.Hp, a.Wp, tid_e);
  __syncthreads();      // the tile has been pooled: its LDS becomes the next tile's V; the next window is in place
  }  // persistent tile loop
}

}  // namespace fpc
