// conv_mfma.h -- halo-tile direct convolution on the fp32 matrix cores of gfx950.
//
// One workgroup computes a TH x TW tile of output pixels times NB*WN*32 output
// channels of one frame.  The input window of the tile (its "halo", (TH-1)*S+EXT
// rows by (TW-1)*S+EXT columns, KC channels at a time) is staged ONCE per channel
// chunk in LDS in NHWC order and then read back shifted by each filter tap: no
// im2col copy is ever materialised.  Weights do not pass through LDS at all: they
// are pre-packed on the host into the exact per-lane operand order of
// v_mfma_f32_32x32x2_f32 (one coalesced 1 KiB float4 load per 32 output channels
// and 8 input channels), stay L2/L1 resident, and are prefetched one step ahead.
//
// Arithmetic: D = A*B + C with A = pixels x channels (LDS), B = channels x output
// channels (global), f32 in / f32 accumulate -- bit-for-bit an fmaf chain, so the
// path is "plain fp32" (MI355X_MICROARCH.md, FP32-input MFMA).
//
// The same kernel serves, by arguments only:
//   3x3 stride 1/2 (+folded BN +ReLU)                  resnet_blocks.py:16-18
//   1x1 over [h | x] with the K dimension concatenated  resnet_blocks.py:19-26
//        (conv2+bn2 and identity_downsample conv+bn in ONE GEMM; or +identity)
//   the four output-parity phases of ConvTranspose2d    superpoint.py:55-57
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fpc {

// XCD-aware workgroup order for one-tile-per-workgroup launches: the dispatcher deals workgroups round-robin to the
// 8 XCDs (blockIdx.x & 7), each with its own L2.  Returning (blockIdx.x & 7) * (grid / 8) + (blockIdx.x >> 3) gives
// XCD k the contiguous tile range [k * grid / 8, (k + 1) * grid / 8): neighbouring tiles -- which share halo rows --
// meet in one L2.  A bijection when the grid is a multiple of 8; otherwise the plain order.
__device__ __forceinline__ int fpc_xcd_tile_index() {
  const unsigned g = gridDim.x, b = blockIdx.x;
  return (g & 7u) == 0u ? (int)((b & 7u) * (g >> 3) + (b >> 3)) : (int)b;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic, not for its global
// loads / stores.  __syncthreads() also drains vmcnt, which would stall every barrier behind the
// prefetched next halo chunk and -- in the persistent kernels -- behind the previous tile's output
// stores (vmcnt counts stores on gfx950).
#define FPC_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct ConvSub {               // one output-parity phase (plain convs use sub[0] only)
  const float4* wfrag;         // packed B fragments, see pack_conv_weights()
  int ntaps;
  int tapoff4[9];              // halo offset of each tap in float4 units
  int oy0, ox0;                // output pixel = (y*oys + oy0, x*oxs + ox0)
};

struct ConvArgs {
  const float* in0;            // NHWC, already offset to its first channel
  int cs0;                     // floats between consecutive pixels of in0
  int nchunk0;                 // Cin0 / KC
  const float* in1;            // optional second K-source (1x1 only), sampled with stride s1
  int cs1, nchunk1, s1, H1, W1;
  int H, W;                    // spatial size of in0
  int pad;                     // halo origin = tile origin * S - pad
  const float* bias;           // [NBT*32]
  const float* res;            // optional identity added before ReLU; NHWC at output resolution
  int csr;
  float* out;                  // NHWC, already offset to its first channel
  int cso;
  int OH, OW, oys, oxs;        // output buffer geometry
  int Ho, Wo;                  // tile iteration space (== OH, OW except for ConvTranspose phases)
  int tiles_x, tiles_y;
  int nstore;                  // output channels actually stored
  int relu;
  int nbt;                     // 32-wide N blocks in the packed weights
  int frame0;                  // first frame of this launch (sub-batches run on separate streams)
  ConvSub sub[4];
};

template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB>
struct ConvCfg {
  static constexpr int NT = WM * WN * 64;
  static constexpr int HW = (TW - 1) * S + EXT;       // halo width  (pixels)
  static constexpr int HH = (TH - 1) * S + EXT;       // halo height (pixels)
  static constexpr int ROW4 = KC / 4 + 1;             // float4 per halo pixel (+1 = bank skew)
  static constexpr int LDS_BYTES = HH * HW * ROW4 * 16;
  static constexpr int M = WM * MB * 32;
  static constexpr int N = WN * NB * 32;
  static_assert(TH * TW <= M && TH * TW > M - 32, "tile must fill the M blocks");
  static_assert(KC % 8 == 0, "KC must be a multiple of 8");
};

template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB>
__global__ __launch_bounds__(WM* WN * 64, 2) void conv_mfma_kernel(const ConvArgs a) {
  using C = ConvCfg<TH, TW, S, EXT, KC, WM, WN, MB, NB>;
  constexpr int NT = C::NT, HW = C::HW, HH = C::HH, ROW4 = C::ROW4, K8 = KC / 8, KC4 = KC / 4;
  constexpr int NV = HH * HW * KC4;                   // float4 per halo chunk
  constexpr int ITER = (NV + NT - 1) / NT;
  extern __shared__ float4 lds4[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;

  const int tiles = a.tiles_x * a.tiles_y;
  const int bidx = fpc_xcd_tile_index();
  const int bl = bidx / tiles;
  const int b = a.frame0 + bl;
  const int t = bidx - bl * tiles;
  const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
  // sub-problems (ConvTranspose parity phases) are dispatched heaviest first: blockIdx.z = 0 is
  // the last sub-problem (4 taps), so the short ones fill the tail of the launch
  const int subi = gridDim.z - 1 - blockIdx.z;
  const ConvSub& sp = a.sub[subi];

  // LDS read base of this lane for each of its M blocks (float4 units)
  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    int m = (wm * MB + mb) * 32 + l31;
    m = m < TH * TW ? m : TH * TW - 1;
    const int py = m / TW, px = m - py * TW;
    abase[mb] = ((py * S) * HW + px * S) * ROW4 + half;
  }
  const int stepstride = a.nbt * 64;
  // the fragments through a buffer descriptor (round 3; as block_mfma.h: no 64-bit pointer arithmetic between fp32 MFMAs)
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float4*>(sp.wfrag), 0, (int)((unsigned)((a.nchunk0 * sp.ntaps + a.nchunk1) * K8 + 2) * (unsigned)(stepstride * 16)), 0x00020000);
  const int wlane = (((blockIdx.y * WN + wn) * NB) * 64 + lane) * 16;
  int wstep = 0;   // (scalar) byte offset of the next step to request
  // (EXT == 1, the 1x1 instances -- few steps per chunk -- measured 12 % SLOWER with it and keep the pointer)
  const float4* wp = sp.wfrag + (size_t)((blockIdx.y * WN + wn) * NB) * 64 + lane;
  auto wfrag = [&](int nb) {
    if constexpr (EXT == 1) {
      return wp[(wstep >> 4) + nb * 64];
    } else {
      typedef float f32x4w __attribute__((ext_vector_type(4)));
      const f32x4w r = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + nb * 1024, wstep, 0));
      return make_float4(r.x, r.y, r.z, r.w);
    }
  };

  const int iy0 = ty * TH * S - a.pad, ix0 = tx * TW * S - a.pad;
  float4 stage[ITER];
  auto load_chunk = [&](int chunk) {
    const bool second = chunk >= a.nchunk0;
    const float* src = second ? a.in1 : a.in0;
    const int cs = second ? a.cs1 : a.cs0;
    const int c0 = (second ? chunk - a.nchunk0 : chunk) * KC;
    const int sH = second ? a.H1 : a.H, sW = second ? a.W1 : a.W;
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int e = tid + i * NT;
      const int pix = e / KC4, c4 = e - pix * KC4;
      const int hy = pix / HW, hx = pix - hy * HW;
      int iy, ix;
      if (second) {  // 1x1 only: halo == tile
        iy = (ty * TH + hy) * a.s1;
        ix = (tx * TW + hx) * a.s1;
      } else {
        iy = iy0 + hy;
        ix = ix0 + hx;
      }
      const bool ok = (NV % NT == 0 || e < NV) && iy >= 0 && iy < sH && ix >= 0 && ix < sW;
      // unconditional load from a clamped (always valid) address, then select: a
      // branch per element would serialise the loads (one vmcnt(0) each)
      const size_t off = ok ? ((size_t)(b * sH + iy) * sW + ix) * cs + c0 + c4 * 4 : 0;
      float4 v = *reinterpret_cast<const float4*>(src + off);
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      stage[i] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int e = tid + i * NT;
      const int pix = e / KC4, c4 = e - pix * KC4;
      if (NV % NT == 0 || e < NV) lds4[pix * ROW4 + c4] = stage[i];
    }
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  const int nchunks = a.nchunk0 + a.nchunk1;
  const int ntaps = sp.ntaps;
  load_chunk(0);
  // B operands run two steps ahead of the MFMAs that consume them (the blob is
  // padded by two steps so the tail prefetch stays in bounds).
  float4 b0[NB], b1[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) b0[nb] = wfrag(nb);
  wstep += stepstride * 16;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) b1[nb] = wfrag(nb);
  wstep += stepstride * 16;

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    if (chunk) __syncthreads();
    store_chunk();
    __syncthreads();
    if (chunk + 1 < nchunks) load_chunk(chunk + 1);
    for (int tap = 0; tap < ntaps; ++tap) {
      const int toff = sp.tapoff4[tap];
#pragma unroll
      for (int k8 = 0; k8 < K8; ++k8) {
        float4 b2[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b2[nb] = wfrag(nb);
        wstep += stepstride * 16;
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch issue up here
        float4 av[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[mb] = lds4[abase[mb] + toff + k8 * 2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              const float af = j == 0 ? av[mb].x : j == 1 ? av[mb].y : j == 2 ? av[mb].z : av[mb].w;
              const float bf = j == 0 ? b0[nb].x : j == 1 ? b0[nb].y : j == 2 ? b0[nb].z : b0[nb].w;
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[mb][nb], 0, 0, 0);
            }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          b0[nb] = b1[nb];
          b1[nb] = b2[nb];
        }
      }
    }
  }

  // Epilogue: + folded-BN bias (+ identity) -> ReLU -> NHWC store.
  // C/D map of the 32x32 MFMA: column (N) = lane & 31, row (M) = (r&3) + 8*(r>>2) + 4*half.
  const int oyb = ty * TH, oxb = tx * TW;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = ((blockIdx.y * WN + wn) * NB + nb) * 32 + l31;
    const float bias = a.bias[n];
    const bool nok = n < a.nstore;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (wm * MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int py = m / TW, px = m - py * TW;
        const int y = oyb + py, x = oxb + px;
        if (nok && m < TH * TW && y < a.Ho && x < a.Wo) {
          const size_t opix = (size_t)(b * a.OH + y * a.oys + sp.oy0) * a.OW + x * a.oxs + sp.ox0;
          float v = acc[mb][nb][r] + bias;
          if (a.res) v += a.res[opix * a.csr + n];
          if (a.relu) v = v > 0.f ? v : 0.f;
          a.out[opix * a.cso + n] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// Round 5: conv_mfma_kernel on a VALU diet (kernels_misc.h, stem_pool2_kernel, has the why: on gfx950 a VALU instruction
// is an fp32 MFMA slot of its SIMD, and the shipped conv_mfma_kernel instances issue ~1 300 of them per wave and tile --
// 64-bit addresses and two divisions per staged element and chunk, one four-byte store + one four-byte identity load with
// a 64-bit address and a branch per accumulator register -- beside 256 MFMAs in the 1x1 of descriptor.layer_in.1).  For
// launches with ONE K source whose workgroups store all their N channels (the host checks; the others keep
// conv_mfma_kernel).  Same arithmetic in the same order: bit-identical results.
//   * staging: a buffer descriptor of the FRAME, per-element byte offsets computed once per workgroup (out-of-frame,
//     padding: a marker that stays out of range under the saturating addition of the chunk's offset) -- one VALU
//     instruction per element and chunk;
//   * a step's pixels are read from LDS while the previous step's MFMAs run (the old loop read them in front of their MFMAs);
//   * epilogue: acc + bias -> LDS tile -> float4 elements: identity (requested before the tile is turned around), ReLU,
//     16-byte stores through a descriptor of the output frame.
// ---------------------------------------------------------------------------------
#define CONV2_MARKER 0x7fffff00

template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB>
struct Conv2Cfg {
  using C = ConvCfg<TH, TW, S, EXT, KC, WM, WN, MB, NB>;
  static constexpr int ROWO4 = C::N / 4 + 1;                          // float4 per pixel of the output tile (+1 skew)
  static constexpr int OUT_BYTES = C::M * ROWO4 * 16;
  static constexpr int LDS_BYTES = C::LDS_BYTES > OUT_BYTES ? C::LDS_BYTES : OUT_BYTES;
};

template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB>
__global__ __launch_bounds__(WM* WN * 64, 2) void conv2_mfma_kernel(const ConvArgs a) {
  using C = ConvCfg<TH, TW, S, EXT, KC, WM, WN, MB, NB>;
  using C2 = Conv2Cfg<TH, TW, S, EXT, KC, WM, WN, MB, NB>;
  constexpr int NT = C::NT, HW = C::HW, HH = C::HH, ROW4 = C::ROW4, K8 = KC / 8, KC4 = KC / 4, N = C::N;
  constexpr int NV = HH * HW * KC4, ITER = (NV + NT - 1) / NT, ROWO4 = C2::ROWO4;
  extern __shared__ float4 lds4[];
  typedef float f32x4w __attribute__((ext_vector_type(4)));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y;
  const int bidx = fpc_xcd_tile_index();
  const int bl = bidx / tiles;
  const int b = a.frame0 + bl;
  const int t = bidx - bl * tiles;
  const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
  const int subi = gridDim.z - 1 - blockIdx.z;
  const ConvSub& sp = a.sub[subi];

  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    int m = (wm * MB + mb) * 32 + l31;
    m = m < TH * TW ? m : TH * TW - 1;
    const int py = m / TW, px = m - py * TW;
    abase[mb] = ((py * S) * HW + px * S) * ROW4 + half;
  }
  const int stepstride = a.nbt * 64;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float4*>(sp.wfrag), 0, (int)((unsigned)(a.nchunk0 * sp.ntaps * K8 + 2) * (unsigned)(stepstride * 16)), 0x00020000);
  const int wlane = (((blockIdx.y * WN + wn) * NB) * 64 + lane) * 16;
  int wstep = 0;
  auto wfrag = [&](int nb) {
    const f32x4w r = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + nb * 1024, wstep, 0));
    return make_float4(r.x, r.y, r.z, r.w);
  };

  // ---- staging: per-element offsets inside the frame, once per workgroup
  const int iy0 = ty * TH * S - a.pad, ix0 = tx * TW * S - a.pad;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in0) + (size_t)b * a.H * a.W * a.cs0, 0, a.H * a.W * a.cs0 * 4, 0x00020000);
  int eoff[ITER], eslot[ITER];
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int e = tid + i * NT;
    int pix = e / KC4, c4 = e - pix * KC4;
    const int hy = pix / HW, hx = pix - hy * HW;
    const int iy = iy0 + hy, ix = ix0 + hx;
    const bool ok = (NV % NT == 0 || e < NV) & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
    eoff[i] = ok ? ((iy * a.W + ix) * a.cs0 + c4 * 4) * 4 : CONV2_MARKER;
    if (NV % NT != 0) {   // elements past the halo go to the skew slot of its last pixel (never read)
      const bool in = e < NV;
      pix = in ? pix : HH * HW - 1;
      c4 = in ? c4 : KC4;
    }
    eslot[i] = pix * ROW4 + c4;
  }
  f32x4w stage[ITER];
  auto load_chunk = [&](int chunk) {
    const int cbase = chunk < a.nchunk0 ? chunk * (KC * 4) : CONV2_MARKER;   // nothing is in range past the last chunk
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      int voff;
      asm("v_add_i32 %0, %1, %2 clamp" : "=v"(voff) : "v"(eoff[i]), "s"(cbase));
      stage[i] = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 0, 0));
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < ITER; ++i) *reinterpret_cast<f32x4w*>(&lds4[eslot[i]]) = stage[i];
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  const int nchunks = a.nchunk0, ntaps = sp.ntaps;
  load_chunk(0);
  float4 b0[NB], b1[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) b0[nb] = wfrag(nb);
  wstep += stepstride * 16;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) b1[nb] = wfrag(nb);
  wstep += stepstride * 16;

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    FPC_LDS_BARRIER();
    store_chunk();
    FPC_LDS_BARRIER();
    load_chunk(chunk + 1);
    float4 av[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) av[mb] = lds4[abase[mb] + sp.tapoff4[0]];
    for (int tap = 0; tap < ntaps; ++tap) {
      const int toff = sp.tapoff4[tap];
      const int toffn = sp.tapoff4[tap + 1 < ntaps ? tap + 1 : tap];
#pragma unroll
      for (int k8 = 0; k8 < K8; ++k8) {
        float4 b2[NB], an[MB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b2[nb] = wfrag(nb);
        wstep += stepstride * 16;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) an[mb] = lds4[abase[mb] + (k8 + 1 < K8 ? toff + (k8 + 1) * 2 : toffn)];   // (last step of a chunk: unused)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              const float af = j == 0 ? av[mb].x : j == 1 ? av[mb].y : j == 2 ? av[mb].z : av[mb].w;
              const float bf = j == 0 ? b0[nb].x : j == 1 ? b0[nb].y : j == 2 ? b0[nb].z : b0[nb].w;
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[mb][nb], 0, 0, 0);
            }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          b0[nb] = b1[nb];
          b1[nb] = b2[nb];
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[mb] = an[mb];
      }
    }
  }

  // ---- epilogue: + bias -> LDS tile -> (+ identity) -> ReLU -> 16-byte stores
  constexpr int C4 = N / 4, NE = TH * TW * C4, EIT = (NE + NT - 1) / NT;
  const int oyb = ty * TH, oxb = tx * TW;
  const int nbase = blockIdx.y * N;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b * a.OH * a.OW * a.cso, 0, a.OH * a.OW * a.cso * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.res ? a.res : a.out) + (size_t)b * a.OH * a.OW * (a.res ? a.csr : a.cso), 0, a.res ? a.OH * a.OW * a.csr * 4 : 0, 0x00020000);
  int ooff[EIT], olds[EIT];
  f32x4w idv[EIT];
#pragma unroll
  for (int i = 0; i < EIT; ++i) {
    const int e = tid + i * NT;
    const int m = e / C4, c4 = e - m * C4;
    const int py = m / TW, px = m - py * TW;
    const int y = oyb + py, x = oxb + px;
    const bool ok = (NE % NT == 0 || e < NE) & (y < a.Ho) & (x < a.Wo);
    const int opix = (y * a.oys + sp.oy0) * a.OW + x * a.oxs + sp.ox0;
    ooff[i] = ok ? (opix * a.cso + nbase + c4 * 4) * 4 : CONV2_MARKER;
    olds[i] = (NE % NT == 0 || e < NE) ? m * ROWO4 + c4 : 0;
    // (the identity: zeros from the bounds check where there is none -- the descriptor's range is 0 then)
    idv[i] = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ok ? (opix * a.csr + nbase + c4 * 4) * 4 : CONV2_MARKER, 0, 0));
  }
  FPC_LDS_BARRIER();   // every wave is done reading the last chunk
  {
    float* ol = reinterpret_cast<float*>(lds4);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int nl = (wn * NB + nb) * 32 + l31;
      const float bias = a.bias[nbase + nl];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (wm * MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          ol[m * (ROWO4 * 4) + nl] = acc[mb][nb][r] + bias;
        }
    }
  }
  FPC_LDS_BARRIER();
  const float floor_ = a.relu ? 0.f : -3.0e38f;
#pragma unroll
  for (int i = 0; i < EIT; ++i) {
    f32x4w v = *reinterpret_cast<const f32x4w*>(&lds4[olds[i]]);
    v += idv[i];
    v.x = fmaxf(v.x, floor_);
    v.y = fmaxf(v.y, floor_);
    v.z = fmaxf(v.z, floor_);
    v.w = fmaxf(v.w, floor_);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), orsrc, ooff[i], 0, 0);
  }
}

}  // namespace fpc
