// wblock_mfma.h -- a stride-1 ResNetBlock (python/src/resnet_blocks.py:14-27) per launch with the
// 3x3 convolution computed by Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
//     V = B^T d B   (4x4 input patch of every 2x2 output tile, per channel)      VALU, in LDS
//     M_xi = V_xi . U_xi   for the 16 patch positions xi, U = G g G^T (host)      16 GEMMs on MFMA
//     Y = A^T M A   (+ folded-BN bias, ReLU)  -> h                                VALU, in LDS
//     out = relu(conv1x1(h) + shortcut(x))                                       as block_mfma.h
//
// 16 multiplications per output tile and channel pair instead of 36: the 3x3 costs 2.25x fewer
// MFMAs.  The transforms only use the constants 0, +-1 (input, output) and 1/2 (filters, folded on
// the host in double), so the result stays within a few 1e-6 of the direct fp32 convolution.
//
// One workgroup = 8 waves = an 8x16 pixel tile (32 Winograd tiles = exactly one 32-row MFMA block
// per position xi) x ALL output channels (64 or 128).  Wave (gx, gn): positions gx*PX..gx*PX+PX-1,
// N blocks gn*NBW..gn*NBW+NBW-1, so every wave owns PX*NBW accumulator blocks.
#pragma once
#include "block_mfma.h"

namespace fpc {

struct WBlockArgs {
  const float* x;        // NHWC input, already offset to its first channel
  int csx, nchunk;       // pixel stride (floats), Cin / KC
  int H, W;              // input == output size (stride 1)
  const float4* w1;      // Winograd-domain fragments: [chunk][xi][k8][nb][64] float4 (+2 steps of padding)
  const float* b1;       // [NBT*32]
  const float4* w2;      // 1x1 fragments: k8_h steps over h, then k8_x steps over x
  const float* b2;
  int k8_h, k8_x;        // k8_x == 0: identity shortcut
  float* out;
  int cso, tiles_x, tiles_y, frame0;
  int total;             // tiles_x * tiles_y * frames of this launch; the grid is persistent
  int xcd_order;         // 1: XCD-aware tile order (see the kernel)
  int ysplit_floats;     // wblock16_kernel, conv_only: blockIdx.y = 1 computes the NEXT 16 NCG output channels -- its fragments and
                         // bias lie this many floats behind w1 / b1, its outputs 16 NCG channels behind `out` (a 256-wide
                         // layer is one launch of two 128-channel halves)
  unsigned x_bytes;      // wblock16_kernel: bytes from `x` to the end of the input tensor's frames of this launch (buffer descriptor range)
  int conv_only;         // 1: stop after h = relu(conv3x3(x) + b1) and store it (a plain Conv2d + bias/BN + ReLU:
                         // the layers of the C++ network, conv1 of a block too wide to fuse); w2 / b2 unused
  const float* dust;     // wblock36_kernel<..., DUST>: the 65th channel's weights (W36Dust, wblock36_mfma.h); else unused
  int dust_in;           // ... 1: the INPUT has a 65th channel too (x[..][64]: detector.layer.1), not covered by nchunk
#ifdef FPC_DIAG
  unsigned long long* stamps;
#endif
};

template <int KC, int NBT, int CMID_>
struct WBlockCfg {
  static constexpr int TH = 8, TW = 16, NT = 512;
  static constexpr int HW = TW + 2, HH = TH + 2;
  static constexpr int ROW4 = KC / 4 + 1;                       // float4 per halo pixel / per V row
  static constexpr int CMID = CMID_;                             // real (8-padded) channel count, <= NBT*32
  static constexpr int ROWH4 = CMID / 4 + 1;
  static constexpr int HALO_BYTES = HH * HW * ROW4 * 16;
  static constexpr int V_BYTES = 16 * 32 * ROW4 * 16;
  static constexpr int M_BYTES = 16 * 32 * 36 * 4;              // one 32-channel quarter (+4 skew), all 16 positions
  static constexpr int H_BYTES = 128 * ROWH4 * 16;
  static constexpr int LDS_BYTES = (HALO_BYTES + V_BYTES) > (M_BYTES + H_BYTES) ? (HALO_BYTES + V_BYTES) : (M_BYTES + H_BYTES);
  static constexpr int GN = NBT >= 4 ? 2 : 1;                   // wave grid: GX position groups x GN channel groups
  static constexpr int GX = 8 / GN;
  static constexpr int PX = 16 / GX;                            // positions per wave
  static constexpr int NBW = NBT / GN;                          // N blocks per wave
};

// Uniform base (SGPR pair) + 32-bit per-lane byte offset.  The asm pins the uniform part in SGPRs; otherwise the
// compiler folds it into 64-bit per-lane addresses, hoists one per K step out of the tile loop and spills them.
__device__ __forceinline__ float4 fpc_ldg_su(const float4* ubase, unsigned lane_bytes) {
  typedef const char __attribute__((address_space(1))) * gptr;
  typedef float f4v __attribute__((ext_vector_type(4)));
  gptr b = (gptr) reinterpret_cast<const char*>(ubase);
  asm("" : "+s"(b));
  const f4v v = *reinterpret_cast<const f4v __attribute__((address_space(1)))*>(b + lane_bytes);
  return make_float4(v.x, v.y, v.z, v.w);
}

template <int KC, int NBT, int CMID_>
__global__ __launch_bounds__(512, 2) void wblock_mfma_kernel(const WBlockArgs a) {
  using C = WBlockCfg<KC, NBT, CMID_>;
  constexpr int NT = C::NT, TH = C::TH, TW = C::TW, HW = C::HW, HH = C::HH, ROW4 = C::ROW4, K8 = KC / 8, KC4 = KC / 4;
  constexpr int NV = HH * HW * KC4, ITER = (NV + NT - 1) / NT, ROWH4 = C::ROWH4, CMID = C::CMID;
  constexpr int GN = C::GN, PX = C::PX, NBW = C::NBW;
  extern __shared__ float4 lds4[];
  float4* halo4 = lds4;
  float4* v4 = lds4 + C::HALO_BYTES / 16;

  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform on purpose: what derives from it stays in SGPRs
  const int gx = wave / GN, gn = wave % GN;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y;
  FPC_STAMP(0)

  // Persistent workgroups: tile indices blockIdx.x, blockIdx.x + gridDim.x, ...  The first halo
  // chunk of the NEXT tile is fetched while the current one is in its last GEMM / transforms /
  // epilogue, so only the very first tile of a workgroup waits for global memory.
  float4 stage[ITER];
  auto load_chunk = [&](int wg, int chunk) {
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
      float4 v = *reinterpret_cast<const float4*>(a.x + off);
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      stage[i] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int e = tid + i * NT;
      const int pix = e / KC4, c4 = e - pix * KC4;
      if (NV % NT == 0 || e < NV) halo4[pix * ROW4 + c4] = stage[i];
    }
  };

  // B fragments of this wave: step s = (chunk, local position p, k8), laid out
  // [chunk][xi][k8][nb][lane]; consecutive (p, k8) of one chunk are contiguous.
  constexpr int STEPS = PX * K8;                  // per chunk
  constexpr int stepstride = NBT * 64;
  // Weight pointers: a uniform base (SGPR pair) plus the lane offset per load, or -- for the three-block kernels, where
  // it measures 5 % faster -- per-lane 64-bit pointers that the compiler hoists out of the tile loop.
  constexpr bool PIN = NBT != 3;
  const float4* wbase = a.w1 + (size_t)(gx * PX * K8 * NBT + gn * NBW) * 64 + (PIN ? 0 : lane);
  auto ldw = [&](const float4* pw) { return PIN ? fpc_ldg_su(pw, lane16) : *pw; };
  auto wptr = [&](int s) {                        // s = step index of this wave within a tile
    const int chunk = s / STEPS, ls = s - chunk * STEPS;
    return wbase + (size_t)(chunk * 16 * K8 + ls) * stepstride;
  };

  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (blockIdx.x & 7), each with its own L2.
  // XCD k walks the contiguous tile range [k * chunk, (k + 1) * chunk): neighbouring tiles -- which share halo rows
  // -- and the frames' weights stay in ONE L2 instead of being fetched by all eight.  (Only when the grid is a
  // multiple of 8, i.e. in the persistent case; otherwise the plain order.)
  const bool xcd_order = a.xcd_order && (gridDim.x & 7) == 0;
  const int wg_step = xcd_order ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int xchunk = (a.total + 7) >> 3;
  const int wg_first = xcd_order ? (int)(blockIdx.x & 7) * xchunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int wg_end = xcd_order ? min(a.total, ((int)(blockIdx.x & 7) + 1) * xchunk) : a.total;
  if (wg_first < wg_end) load_chunk(wg_first, 0);
  // diagnostic stamps describe a workgroup's THIRD tile (steady state: weights in L2, halo prefetched)
  const int wg_stamp = wg_first + 2 * wg_step < wg_end ? wg_first + 2 * wg_step : wg_first;
  for (int wg = wg_first; wg < wg_end; wg += wg_step) {
  const int bl = wg / tiles;
  const int b = a.frame0 + bl;
  const int t = wg - bl * tiles;
  const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
  if (wg == wg_stamp && wg != wg_first) { FPC_STAMP(0) }
  // per-thread index math below is cheap; recompute it per tile rather than let the compiler hoist it
  // out of the persistent loop and keep dozens of values alive (they spilled to scratch)
  int tid_t = tid;
  asm volatile("" : "+v"(tid_t));

  f32x16 acc[PX][NBW];
#pragma unroll
  for (int p = 0; p < PX; ++p)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][nb][r] = 0.f;

  // ---------------------------------------------------------------- phase 1: Winograd 3x3
  float4 b0[NBW], b1[NBW];
  {
    const float4* p0 = wptr(0);
    const float4* p1 = wptr(1);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      b0[nb] = ldw(p0 + nb * 64);
      b1[nb] = ldw(p1 + nb * 64);
    }
  }
  int gs = 0;
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    FPC_LDS_BARRIER();   // previous chunk's GEMM is done with V / previous tile's epilogue with the LDS
    store_chunk();
    FPC_LDS_BARRIER();
    if (chunk == 0 && wg == wg_stamp) { FPC_STAMP(6) }
    if (chunk + 1 < a.nchunk) load_chunk(wg, chunk + 1);
    else if (wg + wg_step < wg_end) load_chunk(wg + wg_step, 0);
    {
      // input transform V = B^T d B for (tile, channel pair): 512 items = 32 tiles x KC/2 pairs (KC = 32)
      const float2* halo2 = reinterpret_cast<const float2*>(halo4);
      float2* v2 = reinterpret_cast<float2*>(v4);
      // lanes run over channel pairs first (16 lanes = one pixel's 128 contiguous bytes): with the
      // tile index fastest, the 288-byte tile pitch put 32 lanes on 8 bank groups (4-way conflicts)
      for (int item = tid_t; item < 32 * (KC / 2); item += NT) {
        const int c2 = item % (KC / 2), wt = item / (KC / 2);
        const int ty2 = wt >> 3, tx2 = wt & 7;
        const int base = ((2 * ty2) * HW + 2 * tx2) * (ROW4 * 2) + c2;
        float2 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) d[i][j] = halo2[base + (i * HW + j) * (ROW4 * 2)];
        float2 r[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // B^T d (rows)
          r[0][j] = make_float2(d[0][j].x - d[2][j].x, d[0][j].y - d[2][j].y);
          r[1][j] = make_float2(d[1][j].x + d[2][j].x, d[1][j].y + d[2][j].y);
          r[2][j] = make_float2(d[2][j].x - d[1][j].x, d[2][j].y - d[1][j].y);
          r[3][j] = make_float2(d[1][j].x - d[3][j].x, d[1][j].y - d[3][j].y);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // (B^T d) B (columns)
          const float2 q0 = make_float2(r[i][0].x - r[i][2].x, r[i][0].y - r[i][2].y);
          const float2 q1 = make_float2(r[i][1].x + r[i][2].x, r[i][1].y + r[i][2].y);
          const float2 q2 = make_float2(r[i][2].x - r[i][1].x, r[i][2].y - r[i][1].y);
          const float2 q3 = make_float2(r[i][1].x - r[i][3].x, r[i][1].y - r[i][3].y);
          v2[((i * 4 + 0) * 32 + wt) * (ROW4 * 2) + c2] = q0;
          v2[((i * 4 + 1) * 32 + wt) * (ROW4 * 2) + c2] = q1;
          v2[((i * 4 + 2) * 32 + wt) * (ROW4 * 2) + c2] = q2;
          v2[((i * 4 + 3) * 32 + wt) * (ROW4 * 2) + c2] = q3;
        }
      }
    }
    FPC_LDS_BARRIER();
    if (chunk == 0 && wg == wg_stamp) { FPC_STAMP(1) }
    // 16 GEMMs, one 32-row block each: this wave's PX positions x NBW channel blocks
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      const int abase = ((gx * PX + p) * 32 + l31) * ROW4 + half;
#pragma unroll
      for (int k8 = 0; k8 < K8; ++k8) {
        float4 b2[NBW];
        const float4* pn = wptr(gs + 2);
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) b2[nb] = ldw(pn + nb * 64);
        ++gs;
        __builtin_amdgcn_sched_barrier(0);
        const float4 av = v4[abase + k8 * 2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb) {
            const float af = j == 0 ? av.x : j == 1 ? av.y : j == 2 ? av.z : av.w;
            const float bf = j == 0 ? b0[nb].x : j == 1 ? b0[nb].y : j == 2 ? b0[nb].z : b0[nb].w;
            acc[p][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[p][nb], 0, 0, 0);
          }
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
          b0[nb] = b1[nb];
          b1[nb] = b2[nb];
        }
      }
    }
    if (chunk == 0 && wg == wg_stamp) { FPC_STAMP(7) }
  }
  if (wg == wg_stamp) { FPC_STAMP(2) }

  // ---------------------------------------------------------------- output transform -> h (LDS)
  const int lane_t = tid_t & 63, l31_t = lane_t & 31, half_t = lane_t >> 5, wave_t = __builtin_amdgcn_readfirstlane(tid_t >> 6);
  // per 32-channel quarter: M[xi][tile][c] of all 16 positions -> LDS, then Y = A^T M A, + bias, ReLU
  const float4* wq = a.w2;
  float* mreg = reinterpret_cast<float*>(lds4);                       // [16][32][36]
  float4* h4w = lds4 + C::M_BYTES / 16;                               // [128][ROWH4] float4
  for (int q = 0; q < NBT; ++q) {
    FPC_LDS_BARRIER();  // q == 0: GEMMs done with V; q > 0: previous quarter's transform done with M
    if (q / NBW == gn) {
      const int nb = q - gn * NBW;
#pragma unroll
      for (int nbi = 0; nbi < NBW; ++nbi)
        if (nbi == nb) {
#pragma unroll
          for (int p = 0; p < PX; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int wt = (r & 3) + 8 * (r >> 2) + 4 * half_t;
              mreg[((gx * PX + p) * 32 + wt) * 36 + l31_t] = acc[p][nbi][r];
            }
        }
    }
    FPC_LDS_BARRIER();
    // Y = A^T M A on 2 channels at a time: 512 items = 32 tiles x 16 channel pairs (float2 keeps the
    // register footprint next to the live accumulators small; float4 spilled)
    if (q * 32 + (tid_t & 15) * 2 < CMID) {
      const int c2 = tid_t & 15, wt = tid_t >> 4;
      const float2* m2 = reinterpret_cast<const float2*>(mreg);
      float2 m[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = m2[((i * 4 + j) * 32 + wt) * 18 + c2];
      const float2 bias = *reinterpret_cast<const float2*>(a.b1 + q * 32 + c2 * 2);
      float2 y[4];
#define FPC_WOUT(comp)                                                                         \
      {                                                                                        \
        float s0[4], s1[4];                                                                    \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                        \
          s0[j] = m[0][j].comp + m[1][j].comp + m[2][j].comp;                                  \
          s1[j] = m[1][j].comp - m[2][j].comp - m[3][j].comp;                                  \
        }                                                                                      \
        const float y00 = s0[0] + s0[1] + s0[2] + bias.comp, y01 = s0[1] - s0[2] - s0[3] + bias.comp; \
        const float y10 = s1[0] + s1[1] + s1[2] + bias.comp, y11 = s1[1] - s1[2] - s1[3] + bias.comp; \
        y[0].comp = y00 > 0.f ? y00 : 0.f;                                                     \
        y[1].comp = y01 > 0.f ? y01 : 0.f;                                                     \
        y[2].comp = y10 > 0.f ? y10 : 0.f;                                                     \
        y[3].comp = y11 > 0.f ? y11 : 0.f;                                                     \
      }
      FPC_WOUT(x) FPC_WOUT(y)
#undef FPC_WOUT
      const int ty2 = wt >> 3, tx2 = wt & 7;
      const int pm = (2 * ty2) * TW + 2 * tx2;
      float2* h2w = reinterpret_cast<float2*>(h4w);
      const int cq = q * 16 + c2;
      h2w[(pm) * (ROWH4 * 2) + cq] = y[0];
      h2w[(pm + 1) * (ROWH4 * 2) + cq] = y[1];
      h2w[(pm + TW) * (ROWH4 * 2) + cq] = y[2];
      h2w[(pm + TW + 1) * (ROWH4 * 2) + cq] = y[3];
    }
  }
  FPC_LDS_BARRIER();
  if (wg == wg_stamp) { FPC_STAMP(3) }
  if (a.conv_only) {  // h is the result: [128 px][CMID] in LDS -> 16-byte stores
    constexpr int C4S = CMID / 4;
    constexpr int NES = TH * TW * C4S, EITS = (NES + NT - 1) / NT;
    const float4* h4r = lds4 + C::M_BYTES / 16;
#pragma unroll
    for (int i = 0; i < EITS; ++i) {
      const int e = tid_t + i * NT;
      const int m = e / C4S, c4 = e - m * C4S;
      const int py = m / TW, px = m - py * TW;
      const int y = ty * TH + py, x = tx * TW + px;
      if ((NES % NT == 0 || e < NES) && y < a.H && x < a.W)
        *reinterpret_cast<float4*>(a.out + ((size_t)(b * a.H + y) * a.W + x) * a.cso + c4 * 4) = h4r[m * ROWH4 + c4];
    }
    continue;  // next tile: its first barrier orders these LDS reads before the V region is rewritten
  }

  // ---------------------------------------------------------------- phase 2: 1x1 over h (+ projection over x)
  // 4 M blocks (128 pixels) x NBT channel blocks over 8 waves
  constexpr int NB2 = (NBT + 1) / 2;             // channel blocks per wave_t: M block mw, blocks nb0..nb0+NB2-1 (< NBT)
  const int mw = wave_t & 3, nb0 = (wave_t >> 2) * NB2;
  // The accumulators of the 1x1 start from the identity shortcut (or zero when the shortcut is a projection, which
  // is more K below): x is read in the accumulator layout -- register r of lanes 0..31 is 128 contiguous bytes of one
  // pixel -- straight into the registers the MFMAs need anyway.  Held in separate registers until the epilogue, the
  // identity was spilled value by value (load, wait, scratch store) and fetched back one round trip at a time.
  f32x16 acc2[NB2];
  if (a.k8_x == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int py = mw * 2 + (r >> 3), px = 8 * ((r >> 2) & 1) + 4 * half_t + (r & 3);
      int y = ty * TH + py, x = tx * TW + px;
      y = y < a.H ? y : a.H - 1;
      x = x < a.W ? x : a.W - 1;
      const float* xp = a.x + ((size_t)(b * a.H + y) * a.W + x) * a.csx;
#pragma unroll
      for (int nb = 0; nb < NB2; ++nb) {
        const int n = (nb0 + nb) * 32 + l31_t;
        acc2[nb][r] = xp[n < CMID ? n : 0];
        if (n >= CMID) acc2[nb][r] = 0.f;
      }
    }
  } else {
#pragma unroll
    for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[nb][r] = 0.f;
  }
  const float4* h4 = lds4 + C::M_BYTES / 16;
  const int hbase = (mw * 32 + l31_t) * ROWH4 + half_t;
  // B fragments of the 1x1: a ring of four steps with STATIC slot indices (the K loop over h is unrolled, the one
  // over x runs four steps per iteration).  A ring rotated by register moves in a rolled loop made every step
  // wait for the load it had just issued -- a full L2 round trip per 8 MFMAs, 2 to 4 times the MFMA time.
  constexpr int KH = CMID / 8, RING = 4;           // a.k8_h == KH
  const int nsteps2 = KH + a.k8_x;                 // the fragment array is padded by two steps
  const float4* wq2 = wq + (size_t)nb0 * 64;
  float4 cb[RING][NB2];
  auto load_b = [&](int slot, int step) {          // step is clamped into the padded array (uniform)
    const float4* pw = wq2 + (size_t)min(step, nsteps2 + 1) * stepstride;
#pragma unroll
    for (int nb = 0; nb < NB2; ++nb) cb[slot][nb] = fpc_ldg_su(pw + nb * 64, lane16);
  };
#pragma unroll
  for (int s_ = 0; s_ < RING - 1; ++s_) load_b(s_, s_);
  auto mfma_step = [&](const float4& av, int slot) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < NB2; ++nb) {
        if (NBT % 2 && nb0 + nb >= NBT) continue;  // odd NBT: the last wave_t group has one block less
        const float af = j == 0 ? av.x : j == 1 ? av.y : j == 2 ? av.z : av.w;
        const float bf = j == 0 ? cb[slot][nb].x : j == 1 ? cb[slot][nb].y : j == 2 ? cb[slot][nb].z : cb[slot][nb].w;
        acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc2[nb], 0, 0, 0);
      }
  };
  // projection shortcut: this thread's share of the first 128 channels of the centre pixels, consumed after the
  // GEMM over h (see below)
  constexpr int XROW4 = 33, XIT = 128 * 32 / NT;
  static_assert(128 * XROW4 * 16 <= C::M_BYTES, "x staging fits the M region");
  float4* xs4 = lds4;
  float4 xst[XIT];
  auto load_x = [&](int pass) {
    const int kx4 = min(32, a.k8_x * 2 - pass * 32);   // float4 per pixel in this pass
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
      const int e = tid_t + i * NT;
      const int m = e >> 5, c4 = e & 31;
      const int py = m / TW, px = m - py * TW;
      int y = ty * TH + py, x = tx * TW + px;
      y = y < a.H ? y : a.H - 1;
      x = x < a.W ? x : a.W - 1;
      const bool ok = c4 < kx4;
      // (through a vector value: a struct-to-struct copy becomes a memcpy into a private array that is never promoted)
      typedef float f4v __attribute__((ext_vector_type(4)));
      const f4v v = *reinterpret_cast<const f4v*>(a.x + ((size_t)(b * a.H + y) * a.W + x) * a.csx + (ok ? pass * 128 + c4 * 4 : 0));
      xst[i] = make_float4(v.x, v.y, v.z, v.w);
    }
  };
  if (a.k8_x > 0) load_x(0);
  {
    float4 av = h4[hbase];
#pragma unroll
    for (int k8 = 0; k8 < KH; ++k8) {
      load_b((k8 + RING - 1) % RING, k8 + RING - 1);
      const float4 an = h4[hbase + (k8 + 1 < KH ? k8 + 1 : k8) * 2];
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(av, k8 % RING);
      av = an;
    }
  }
  if (a.k8_x > 0) {
    // Projection shortcut: more K into the same accumulators, A = the tile's centre pixels of x.  Reading them
    // straight into the MFMA layout (lane = pixel) touches 64 different cache lines per load; instead 128 channels
    // at a time are loaded as whole pixels (coalesced, requested before the GEMM over h), go through the dead M
    // region of the LDS and come back as ds_read_b128.
    const int npass = (a.k8_x + 15) >> 4;
    for (int pass = 0; pass < npass; ++pass) {
      if (pass > 0) FPC_LDS_BARRIER();   // the previous pass's fragments have been read
#pragma unroll
      for (int i = 0; i < XIT; ++i) {
        const int e = tid_t + i * NT;
        xs4[(e >> 5) * XROW4 + (e & 31)] = xst[i];
      }
      FPC_LDS_BARRIER();
      if (pass + 1 < npass) load_x(pass + 1);
      const int steps = min(16, a.k8_x - pass * 16);   // a multiple of 4
      const int xbase = (mw * 32 + l31_t) * XROW4 + half_t;
      for (int k8 = 0; k8 < steps; k8 += 4) {
        const int sg = KH + pass * 16 + k8;            // global step of this group; sg - KH is a multiple of 4
        float4 av[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) av[u] = xs4[xbase + (k8 + u) * 2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          load_b((KH + u + RING - 1) % RING, sg + u + RING - 1);
          __builtin_amdgcn_sched_barrier(0);
          mfma_step(av[u], (KH + u) % RING);
        }
      }
    }
  }
  if (wg == wg_stamp) { FPC_STAMP(4) }

  // ---------------------------------------------------------------- epilogue (as block_mfma.h)
  FPC_LDS_BARRIER();  // every wave_t is done reading h
  {
    float* ol = reinterpret_cast<float*>(lds4);  // [128][ROWH4*4], over the (dead) M region
#pragma unroll
    for (int nb = 0; nb < NB2; ++nb) {
      const int n = (nb0 + nb) * 32 + l31_t;
      if (n >= CMID) continue;
      const float bias = a.b2[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mw * 32 + (r & 3) + 8 * (r >> 2) + 4 * half_t;
        ol[m * (ROWH4 * 4) + n] = acc2[nb][r] + bias;
      }
    }
  }
  FPC_LDS_BARRIER();
  {
    constexpr int C4 = CMID / 4;
    constexpr int NE = TH * TW * C4, EIT = (NE + NT - 1) / NT;
    const int oyb = ty * TH, oxb = tx * TW;
#pragma unroll
    for (int i = 0; i < EIT; ++i) {
      const int e = tid_t + i * NT;
      const int m = e / C4, c4 = e - m * C4;
      const int py = m / TW, px = m - py * TW;
      const int y = oyb + py, x = oxb + px;
      if ((NE % NT == 0 || e < NE) && y < a.H && x < a.W) {
        float4 v = lds4[m * ROWH4 + c4];
        v.x = v.x > 0.f ? v.x : 0.f;
        v.y = v.y > 0.f ? v.y : 0.f;
        v.z = v.z > 0.f ? v.z : 0.f;
        v.w = v.w > 0.f ? v.w : 0.f;
        *reinterpret_cast<float4*>(a.out + ((size_t)(b * a.H + y) * a.W + x) * a.cso + c4 * 4) = v;
      }
    }
  }
  if (wg == wg_stamp) { FPC_STAMP(5) }
  }  // persistent tile loop
}

}  // namespace fpc
