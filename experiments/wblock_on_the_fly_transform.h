// wblock_mfma.h -- a stride-1 ResNetBlock (python/src/resnet_blocks.py:14-27) per launch with the
// 3x3 convolution computed by Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
//     V = B^T d B   (4x4 input patch of every 2x2 output tile, per channel)      VALU, straight into the A operand
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
  int conv_only;         // 1: stop after h = relu(conv3x3(x) + b1) and store it (a plain Conv2d + bias/BN + ReLU:
                         // the layers of the C++ network, conv1 of a block too wide to fuse); w2 / b2 unused
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
  static constexpr int HALO_BYTES = HH * HW * ROW4 * 16;       // one channel chunk of the halo tile; two buffers
  static constexpr int M_BYTES = 16 * 32 * 36 * 4;              // one 32-channel quarter (+4 skew), all 16 positions
  static constexpr int H_BYTES = 128 * ROWH4 * 16;
  static constexpr int LDS_BYTES = (2 * HALO_BYTES) > (M_BYTES + H_BYTES) ? (2 * HALO_BYTES) : (M_BYTES + H_BYTES);
  static constexpr int GN = NBT >= 4 ? 2 : 1;                   // wave grid: GX position groups x GN channel groups
  static constexpr int GX = 8 / GN;
  static constexpr int PX = 16 / GX;                            // positions per wave
  static constexpr int NBW = NBT / GN;                          // N blocks per wave
};

// uniform base (SGPR pair) + 32-bit per-lane byte offset: the form `global_load_dwordx4 v, v_off, s[base]` takes
__device__ __forceinline__ float4 fpc_ldg_su(const float4* ubase, unsigned lane_bytes) {
  typedef const char __attribute__((address_space(1))) * gptr;
  gptr b = (gptr) reinterpret_cast<const char*>(ubase);
  asm("" : "+s"(b));   // pin the uniform part in an SGPR pair so that it is not folded into a 64-bit per-lane address
  typedef float f4v __attribute__((ext_vector_type(4)));
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
  constexpr int HALO4 = C::HALO_BYTES / 16;

  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform on purpose: everything derived from it
  const int gx = wave / GN, gn = wave % GN;                    // (weight pointers, transform signs) lives in SGPRs
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
  auto store_chunk = [&](int buf) {
    float4* hb = halo4 + buf * HALO4;
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int e = tid + i * NT;
      const int pix = e / KC4, c4 = e - pix * KC4;
      if (NV % NT == 0 || e < NV) hb[pix * ROW4 + c4] = stage[i];
    }
  };

  // B fragments of this wave, laid out [chunk][xi][k8][nb][lane] (+ one zero chunk, so reading ahead of the last
  // step stays inside the array).  The GEMM walks a chunk k8-major: step ls = k8 * PX + p.
  constexpr int STEPS = PX * K8;                  // per chunk
  constexpr int stepstride = NBT * 64;
  const float4* wbase = a.w1 + (size_t)(gx * PX * K8 * NBT + gn * NBW) * 64;   // uniform; the lane is added per load
  auto wptr = [&](int chunk, int ls) {            // ls may run past the chunk: it continues in the next one
    if (ls >= STEPS) {
      ls -= STEPS;
      ++chunk;
    }
    const int k8 = ls / PX, p = ls - k8 * PX;
    return wbase + (size_t)(chunk * 16 * K8 + p * K8 + k8) * stepstride;
  };

  // The input transform V = B^T d B is never materialised: B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], so the
  // element (i, j) of V is a signed sum of four halo pixels.  A wave owns positions of ONE transform row i (PX = 4:
  // j = 0..3; PX = 2: j = j0, j0 + 1), forms r[c] = +-d[ra][c] +- d[rb][c] for the columns it needs and then the
  // column combinations, all on the 16-byte fragment it is about to feed to the MFMAs -- 6 to 8 ds_read_b128 and a
  // dozen VALU operations per 16 to 32 MFMAs, no V array in LDS, no transform phase, no barrier for it.
  const int ti = (gx * PX) >> 2;
  const int ra = ti == 0 ? 0 : 1, rb = ti == 3 ? 3 : 2;
  const float ca = ti == 2 ? -1.f : 1.f, cb = (ti == 1 || ti == 2) ? 1.f : -1.f;
  const bool jlo = ((gx * PX) & 3) == 0;          // PX == 2: this wave's columns are j = 0, 1 (else 2, 3)
  const int lbase = ((2 * (l31 >> 3)) * HW + 2 * (l31 & 7)) * ROW4 + half;   // lane's tile: patch origin, its K half
  const int offa = lbase + ra * HW * ROW4, offb = lbase + rb * HW * ROW4;
  // PX == 2 reads three columns X, Y, Z = (0, 1, 2) or (3, 2, 1): p0 = (jlo ? X : Y) - Z, p1 = Z + (jlo ? Y : -X)
  const int cx = jlo ? 0 : 3 * ROW4, cy = jlo ? ROW4 : 2 * ROW4, cz = jlo ? 2 * ROW4 : ROW4;

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
  int par = 0;   // halo buffer of the next chunk; alternates across chunks AND tiles
  for (int wg = wg_first; wg < wg_end; wg += wg_step) {
  const int bl = wg / tiles;
  const int b = a.frame0 + bl;
  const int t = wg - bl * tiles;
  const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
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
    const float4* p0 = wptr(0, 0);
    const float4* p1 = wptr(0, 1);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      b0[nb] = fpc_ldg_su(p0 + nb * 64, lane16);
      b1[nb] = fpc_ldg_su(p1 + nb * 64, lane16);
    }
  }
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    // Two halo buffers: chunk c + 1 is stored while other waves are still in the GEMM of chunk c; the buffer it
    // goes to was last read two chunks ago, before the barrier every wave has passed since.  Only a tile's first
    // store has to wait for the previous tile's epilogue, which used the whole LDS.
    if (chunk == 0) FPC_LDS_BARRIER();
    store_chunk(par);
    FPC_LDS_BARRIER();
    if (chunk == 0 && wg == wg_first) { FPC_STAMP(6) FPC_STAMP(1) }
    if (chunk + 1 < a.nchunk) load_chunk(wg, chunk + 1);
    else if (wg + wg_step < wg_end) load_chunk(wg + wg_step, 0);
    const float4* hb = halo4 + par * HALO4;
    par ^= 1;
    // 16 GEMMs, one 32-row block each: this wave's PX positions x NBW channel blocks
#pragma unroll
    for (int k8 = 0; k8 < K8; ++k8) {
      float4 frag[PX];
      auto row = [&](int coff) {   // r[c] of this wave's transform row, channels k8*8 + half*4 .. +3
        const float4 da = hb[offa + coff + k8 * 2];
        const float4 db = hb[offb + coff + k8 * 2];
        return make_float4(ca * da.x + cb * db.x, ca * da.y + cb * db.y, ca * da.z + cb * db.z, ca * da.w + cb * db.w);
      };
      if constexpr (PX == 4) {
        const float4 r0 = row(0), r1 = row(ROW4), r2 = row(2 * ROW4), r3 = row(3 * ROW4);
        frag[0] = make_float4(r0.x - r2.x, r0.y - r2.y, r0.z - r2.z, r0.w - r2.w);
        frag[1] = make_float4(r1.x + r2.x, r1.y + r2.y, r1.z + r2.z, r1.w + r2.w);
        frag[2] = make_float4(r2.x - r1.x, r2.y - r1.y, r2.z - r1.z, r2.w - r1.w);
        frag[3] = make_float4(r1.x - r3.x, r1.y - r3.y, r1.z - r3.z, r1.w - r3.w);
      } else {
        static_assert(PX == 2, "positions per wave");
        const float4 X = row(cx), Y = row(cy), Z = row(cz);
        const float4 S = jlo ? X : Y;
        const float4 T = jlo ? Y : make_float4(-X.x, -X.y, -X.z, -X.w);
        frag[0] = make_float4(S.x - Z.x, S.y - Z.y, S.z - Z.z, S.w - Z.w);
        frag[1] = make_float4(Z.x + T.x, Z.y + T.y, Z.z + T.z, Z.w + T.w);
      }
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        float4 b2[NBW];
        const float4* pn = wptr(chunk, k8 * PX + p + 2);
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) b2[nb] = fpc_ldg_su(pn + nb * 64, lane16);
        __builtin_amdgcn_sched_barrier(0);
        const float4 av = frag[p];
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
    if (chunk == 0 && wg == wg_first) { FPC_STAMP(7) }
  }
  if (wg == wg_first) { FPC_STAMP(2) }

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
  if (wg == wg_first) { FPC_STAMP(3) }
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
  // identity shortcut: fetch this thread's share of x now, it is consumed in the epilogue
  constexpr int C4E = CMID / 4;
  constexpr int NEE = TH * TW * C4E, EITE = (NEE + NT - 1) / NT;
  float4 idv[EITE];
  if (a.k8_x == 0) {
#pragma unroll
    for (int i = 0; i < EITE; ++i) {
      const int e = tid_t + i * NT;
      const int m = e / C4E, c4 = e - m * C4E;
      const int py = m / TW, px = m - py * TW;
      const int y = ty * TH + py, x = tx * TW + px;
      const bool ok = (NEE % NT == 0 || e < NEE) && y < a.H && x < a.W;
      idv[i] = *reinterpret_cast<const float4*>(a.x + (ok ? ((size_t)(b * a.H + y) * a.W + x) * a.csx + c4 * 4 : 0));
    }
  }
  constexpr int NB2 = (NBT + 1) / 2;             // channel blocks per wave_t: M block mw, blocks nb0..nb0+NB2-1 (< NBT)
  const int mw = wave_t & 3, nb0 = (wave_t >> 2) * NB2;
  f32x16 acc2[NB2];
#pragma unroll
  for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[nb][r] = 0.f;
  const float4* h4 = lds4 + C::M_BYTES / 16;
  const int hbase = (mw * 32 + l31_t) * ROWH4 + half_t;
  const float4* wq2 = wq + (size_t)nb0 * 64;
  float4 c0[NB2], c1[NB2];
#pragma unroll
  for (int nb = 0; nb < NB2; ++nb) {
    c0[nb] = fpc_ldg_su(wq2 + nb * 64, lane16);
    c1[nb] = fpc_ldg_su(wq2 + stepstride + nb * 64, lane16);
  }
  wq2 += 2 * stepstride;
  // projection shortcut: the first four A fragments (centre pixels of x, straight from global) are requested now,
  // so that their latency hides behind the GEMM over h
  const float* xrow = a.x;
  float4 xa[4];
  if (a.k8_x > 0) {
    const int m = mw * 32 + l31_t;
    const int py = m / TW, px = m - py * TW;
    int y = ty * TH + py, x = tx * TW + px;
    y = y < a.H ? y : a.H - 1;
    x = x < a.W ? x : a.W - 1;
    xrow = a.x + ((size_t)(b * a.H + y) * a.W + x) * a.csx + half_t * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) xa[u] = *reinterpret_cast<const float4*>(xrow + u * 8);
  }
  {
    float4 av = h4[hbase];
    for (int k8 = 0; k8 < a.k8_h; ++k8) {
      float4 c2[NB2];
#pragma unroll
      for (int nb = 0; nb < NB2; ++nb) c2[nb] = fpc_ldg_su(wq2 + nb * 64, lane16);
      wq2 += stepstride;
      const float4 an = h4[hbase + (k8 + 1 < a.k8_h ? k8 + 1 : k8) * 2];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb) {
          if (NBT % 2 && nb0 + nb >= NBT) continue;  // odd NBT: the last wave_t group has one block less
          const float af = j == 0 ? av.x : j == 1 ? av.y : j == 2 ? av.z : av.w;
          const float bf = j == 0 ? c0[nb].x : j == 1 ? c0[nb].y : j == 2 ? c0[nb].z : c0[nb].w;
          acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc2[nb], 0, 0, 0);
        }
#pragma unroll
      for (int nb = 0; nb < NB2; ++nb) {
        c0[nb] = c1[nb];
        c1[nb] = c2[nb];
      }
      av = an;
    }
  }
  if (a.k8_x > 0) {  // projection shortcut: A straight from global (centre pixels of x), four steps in flight
    for (int k8 = 0; k8 < a.k8_x; k8 += 4) {   // k8_x is a multiple of 4 (checked when the plan is built)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float4 c2[NB2];
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb) c2[nb] = fpc_ldg_su(wq2 + nb * 64, lane16);
        wq2 += stepstride;
        const float4 av = xa[u];
        xa[u] = *reinterpret_cast<const float4*>(xrow + (k8 + u + 4 < a.k8_x ? k8 + u + 4 : a.k8_x - 1) * 8);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < NB2; ++nb) {
            if (NBT % 2 && nb0 + nb >= NBT) continue;  // odd NBT: the last wave_t group has one block less
            const float af = j == 0 ? av.x : j == 1 ? av.y : j == 2 ? av.z : av.w;
            const float bf = j == 0 ? c0[nb].x : j == 1 ? c0[nb].y : j == 2 ? c0[nb].z : c0[nb].w;
            acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc2[nb], 0, 0, 0);
          }
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb) {
          c0[nb] = c1[nb];
          c1[nb] = c2[nb];
        }
      }
    }
  }
  if (wg == wg_first) { FPC_STAMP(4) }

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
    const bool ident = a.k8_x == 0;
    static_assert(EIT == EITE, "epilogue partition");
#pragma unroll
    for (int i = 0; i < EIT; ++i) {
      const int e = tid_t + i * NT;
      const int m = e / C4, c4 = e - m * C4;
      const int py = m / TW, px = m - py * TW;
      const int y = oyb + py, x = oxb + px;
      if ((NE % NT == 0 || e < NE) && y < a.H && x < a.W) {
        float4 v = lds4[m * ROWH4 + c4];
        if (ident) {
          v.x += idv[i].x;
          v.y += idv[i].y;
          v.z += idv[i].z;
          v.w += idv[i].w;
        }
        v.x = v.x > 0.f ? v.x : 0.f;
        v.y = v.y > 0.f ? v.y : 0.f;
        v.z = v.z > 0.f ? v.z : 0.f;
        v.w = v.w > 0.f ? v.w : 0.f;
        *reinterpret_cast<float4*>(a.out + ((size_t)(b * a.H + y) * a.W + x) * a.cso + c4 * 4) = v;
      }
    }
  }
  if (wg == wg_first) { FPC_STAMP(5) }
  }  // persistent tile loop
}

}  // namespace fpc
