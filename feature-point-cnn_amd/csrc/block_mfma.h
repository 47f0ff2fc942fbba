// block_mfma.h -- one whole ResNetBlock (python/src/resnet_blocks.py:14-27) per launch:
//
//     h   = relu(bn1(conv3x3_s(x)))                    phase 1  (halo-tile conv, as conv_mfma.h)
//     out = relu(bn2(conv1x1(h)) + shortcut(x))        phase 2  (GEMM over h, which never leaves LDS)
//     shortcut(x) = bn(conv1x1_s(x))  (first block of a stage)  or  x  (second block)
//
// Phase 1 ends by writing relu(acc + b1) into LDS as the [pixel][channel] A operand of phase
// 2 (the accumulator layout has the channel on the lane, the A operand wants the pixel on the
// lane: the LDS round trip IS the transpose).  The projection shortcut is more K for the same
// accumulators; its A operand (the centre pixels of x, strided for stride-2 blocks) is read
// straight from global memory -- those lines were just streamed through L2 by phase 1.
// Compared with separate launches this removes the HBM write + read of h, the re-read of x and
// one launch (with its own tail) per block.
//
// A workgroup owns ALL output channels of its pixel tile (phase 2 needs every channel of h).
#pragma once
#include "conv_mfma.h"

namespace fpc {

#ifdef FPC_DIAG
// In-kernel stamps (diagnostic build, `make diag`): wave 0 of every workgroup records the shader
// clock at its phase boundaries plus where it ran.  Never compiled into libfpc.so.
#define FPC_STAMP(i)                                                                     \
  if (a.stamps && threadIdx.x == 0) {                                                    \
    unsigned long long t_;                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
    a.stamps[(size_t)blockIdx.x * 8 + (i)] = t_;                                         \
  }
// the constant 100 MHz clock beside it (slot i): in-kernel shader clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
#define FPC_RSTAMP(i)                                                                    \
  if (a.stamps && threadIdx.x == 0) {                                                    \
    unsigned long long t_;                                                               \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
    a.stamps[(size_t)blockIdx.x * 8 + (i)] = t_;                                         \
  }
#else
#define FPC_STAMP(i)
#define FPC_RSTAMP(i)
#endif

struct BlockArgs {
  const float* x;        // NHWC input, already offset to its first channel
  int csx, nchunk;       // pixel stride (floats), Cin_pad / KC
  int H, W;              // input size
  const float4* w1;      // packed 3x3 fragments (steps = nchunk*9*KC/8, nbt blocks)
  const float* b1;       // [nbt*32]
  int tapoff4[9];
  const float4* w2;      // packed 1x1 fragments: k8_h steps over h, then k8_x steps over x
  const float* b2;       // bn2 bias (+ shortcut bn bias)
  int k8_h, k8_x;        // k8_x == 0: identity shortcut
  float* out;
  int cso, Ho, Wo, tiles_x, tiles_y, nstore, frame0;
  unsigned x_bytes;      // bytes of the input tensor from x on (buffer descriptor range)
#ifdef FPC_DIAG
  unsigned long long* stamps;  // diagnostic build only: 8 words per workgroup
#endif
};

template <int TH, int TW, int S, int KC, int WM, int WN, int MB, int NB, int CMIDP>
struct BlockCfg {
  using C1 = ConvCfg<TH, TW, S, 3, KC, WM, WN, MB, NB>;
  static constexpr int ROWH4 = CMIDP / 4 + 1;
  static constexpr int H_BYTES = C1::M * ROWH4 * 16;
  static constexpr int LDS_BYTES = C1::LDS_BYTES > H_BYTES ? C1::LDS_BYTES : H_BYTES;
  static_assert(CMIDP % 8 == 0 && CMIDP <= C1::N, "mid channels must fit the workgroup's N");
};

template <int TH, int TW, int S, int KC, int WM, int WN, int MB, int NB, int CMIDP>
__global__ __launch_bounds__(WM* WN * 64, 2) void block_mfma_kernel(const BlockArgs a) {
  using C = ConvCfg<TH, TW, S, 3, KC, WM, WN, MB, NB>;
  using BC = BlockCfg<TH, TW, S, KC, WM, WN, MB, NB, CMIDP>;
  constexpr int NT = C::NT, HW = C::HW, HH = C::HH, ROW4 = C::ROW4, K8 = KC / 8, KC4 = KC / 4;
  constexpr int NV = HH * HW * KC4, ITER = (NV + NT - 1) / NT, ROWH4 = BC::ROWH4, NBT = WN * NB;
  extern __shared__ float4 lds4[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y;
  const int bidx = fpc_xcd_tile_index();
  const int bl = bidx / tiles;
  const int b = a.frame0 + bl;
  const int t = bidx - bl * tiles;
  const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;

  FPC_STAMP(0)
#ifdef FPC_DIAG
  if (a.stamps && threadIdx.x == 0) {
    a.stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg(63492);          // HW_ID
    a.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
  }
#endif
  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    int m = (wm * MB + mb) * 32 + l31;
    m = m < TH * TW ? m : TH * TW - 1;
    const int py = m / TW, px = m - py * TW;
    abase[mb] = ((py * S) * HW + px * S) * ROW4 + half;
  }
  constexpr int stepstride = NBT * 64;
  // conv1's fragments through a buffer descriptor (round 3): the lane's offset in a VGPR that never changes, the step
  // in the SCALAR offset.  As wp[nb * 64], wp += stepstride every step was a 64-bit pointer addition -- two VALU
  // instructions between fp32 MFMAs, each of which holds the matrix core up (generation 3).
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float4*>(a.w1), 0, (int)((unsigned)(a.nchunk * 9 * K8 + 2) * (unsigned)(stepstride * 16)), 0x00020000);
  const int wlane = ((wn * NB) * 64 + lane) * 16;
  int wstep = 0;   // (scalar) byte offset of the next step to request
  auto wfrag = [&](int nb) {
    typedef float f32x4w __attribute__((ext_vector_type(4)));
    const f32x4w r = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + nb * 1024, wstep, 0));
    return make_float4(r.x, r.y, r.z, r.w);
  };

  const int iy0 = ty * TH * S - 1, ix0 = tx * TW * S - 1;
  // The halo chunk through a buffer descriptor: a position outside the frame (or a chunk past the last one) gets an
  // offset beyond the descriptor's range and comes back as zeros from the hardware's bounds check, and an element past
  // the halo is stored to the skew column of its last pixel -- no branch between a request and its use (with one
  // `if` per element the compiler waited for EVERY outstanding request at each of them: block_bf16.h has the numbers).
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v stage[ITER];   // (vector values: an array of float4 structs is not promoted to registers)
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const unsigned xbase = (unsigned)((b * a.H + iy0) * a.W + ix0) * (unsigned)(a.csx * 4);   // unsigned: mod 2^32, exact for in-frame pixels
  auto load_chunk = [&](int chunk) {
    const int wlim = chunk < a.nchunk ? a.W : 0;   // nothing is in range past the last chunk
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int e = tid + i * NT;
      const int pix = e / KC4, c4 = e - pix * KC4;
      const int hy = pix / HW, hx = pix - hy * HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)wlim) & (hy < HH);   // (& not &&: no branch)
      unsigned off = xbase + (unsigned)(((hy * a.W + hx) * a.csx + chunk * KC + c4 * 4) * 4);
      asm volatile("" : "+v"(off));   // computed for every lane: as a conditional the compiler branches around it
      stage[i] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(ok ? off : 0xfffffff0u), 0, 0));
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int e = tid + i * NT;
      int pix = e / KC4, c4 = e - pix * KC4;
      if (NV % NT != 0) {
        const bool in = e < NV;
        pix = in ? pix : HH * HW - 1;
        c4 = in ? c4 : KC4;
      }
      *reinterpret_cast<f32x4v*>(&lds4[pix * ROW4 + c4]) = stage[i];
    }
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  // ---------------------------------------------------------------- phase 1: 3x3 conv
  load_chunk(0);
  float4 b0[NB], b1[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) b0[nb] = wfrag(nb);
  wstep += stepstride * 16;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) b1[nb] = wfrag(nb);
  wstep += stepstride * 16;
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    FPC_LDS_BARRIER();
    store_chunk();
    FPC_LDS_BARRIER();
    load_chunk(chunk + 1);
    if (chunk == 0) { FPC_STAMP(1) }
    // A fragments run one step ahead of the MFMAs that consume them (LDS latency would
    // otherwise sit between every group of MFMAs), B fragments two steps ahead.
    float4 av[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) av[mb] = lds4[abase[mb] + a.tapoff4[0]];
#pragma unroll   // rolled, the B ring was rotated by register moves that wait for the load just issued
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = a.tapoff4[tap];
      const int toffn = a.tapoff4[tap < 8 ? tap + 1 : 8];
#pragma unroll
      for (int k8 = 0; k8 < K8; ++k8) {
        float4 b2[NB], an[MB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b2[nb] = wfrag(nb);
        wstep += stepstride * 16;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          an[mb] = lds4[abase[mb] + (k8 + 1 < K8 ? toff + (k8 + 1) * 2 : toffn)];  // last step of a chunk: unused
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

  FPC_STAMP(2)
  // ---------------------------------------------------------------- h = relu(acc + b1) -> LDS
  // start streaming phase 2's weights while the tile is being turned around: a ring of four K steps with STATIC slot
  // indices (the loop over h is unrolled, the one over x runs four steps per iteration) -- in a rolled loop the ring
  // is rotated by register moves, and a move of the fragment just requested waits for its whole L2 round trip
  constexpr int KH = CMIDP / 8, RING = 4;   // a.k8_h == KH
  const int nsteps2 = KH + a.k8_x;           // the fragment array is padded by two steps
  const float4* wq = a.w2 + (size_t)(wn * NB) * 64 + lane;
  float4 cb[RING][NB];
  auto load_b = [&](int slot, int step) {
    const float4* pw = wq + (size_t)min(step, nsteps2 + 1) * stepstride;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) cb[slot][nb] = pw[nb * 64];
  };
#pragma unroll
  for (int s_ = 0; s_ < RING - 1; ++s_) load_b(s_, s_);
  FPC_LDS_BARRIER();  // every wave is done reading the halo: its LDS becomes the h tile
  {
    float* hl = reinterpret_cast<float*>(lds4);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = (wn * NB + nb) * 32 + l31;
      const float bias = a.b1[n];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (wm * MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const float v = acc[mb][nb][r] + bias;
          if (n < CMIDP) hl[m * (ROWH4 * 4) + n] = v > 0.f ? v : 0.f;
          acc[mb][nb][r] = 0.f;
        }
    }
  }
  FPC_LDS_BARRIER();

  FPC_STAMP(3)
  // ---------------------------------------------------------------- phase 2a: K over h (LDS)
  int hbase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) hbase[mb] = ((wm * MB + mb) * 32 + l31) * ROWH4 + half;
  auto mfma_step = [&](const float4 (&av)[MB], int slot) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float af = j == 0 ? av[mb].x : j == 1 ? av[mb].y : j == 2 ? av[mb].z : av[mb].w;
          const float bf = j == 0 ? cb[slot][nb].x : j == 1 ? cb[slot][nb].y : j == 2 ? cb[slot][nb].z : cb[slot][nb].w;
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[mb][nb], 0, 0, 0);
        }
  };
  // ---------------------------------------------------------------- phase 2b operands: K over x (projection)
  // A operand straight from global: lane (pixel, half) reads 4 channels of its own pixel per step; the first four
  // steps are requested before the GEMM over h, the next four during each group of four.
  const float* xrow[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    int m = (wm * MB + mb) * 32 + l31;
    m = m < TH * TW ? m : TH * TW - 1;
    const int py = m / TW, px = m - py * TW;
    int y = (ty * TH + py) * S, x = (tx * TW + px) * S;
    y = y < a.H ? y : a.H - 1;  // rows of partial tiles: any valid address, results are not stored
    x = x < a.W ? x : a.W - 1;
    xrow[mb] = a.x + ((size_t)(b * a.H + y) * a.W + x) * a.csx + half * 4;
  }
  typedef float f4v __attribute__((ext_vector_type(4)));
  auto ldx = [&](int mb, int k8) {
    const f4v v = *reinterpret_cast<const f4v*>(xrow[mb] + (k8 < a.k8_x ? k8 : 0) * 8);
    return make_float4(v.x, v.y, v.z, v.w);
  };
  float4 xa[4][MB];
  if (a.k8_x > 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) xa[u][mb] = ldx(mb, u);
  }
  {
    float4 av[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) av[mb] = lds4[hbase[mb]];
#pragma unroll
    for (int k8 = 0; k8 < KH; ++k8) {
      float4 an[MB];
      load_b((k8 + RING - 1) % RING, k8 + RING - 1);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) an[mb] = lds4[hbase[mb] + (k8 + 1 < KH ? k8 + 1 : k8) * 2];
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(av, k8 % RING);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) av[mb] = an[mb];
    }
  }

  // ---------------------------------------------------------------- phase 2b: K over x (projection)
  for (int k8 = 0; k8 < a.k8_x; k8 += 8) {   // k8_x is a multiple of 8 (checked when the plan is built)
    float4 xb[4][MB];                         // two groups of four per iteration: no register moves at the back edge
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      load_b((KH + u + RING - 1) % RING, KH + k8 + u + RING - 1);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) xb[u][mb] = ldx(mb, k8 + 4 + u);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(xa[u], (KH + u) % RING);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      load_b((KH + u + RING - 1) % RING, KH + k8 + 4 + u + RING - 1);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) xa[u][mb] = ldx(mb, k8 + 8 + u);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(xb[u], (KH + u) % RING);
    }
  }

  // ---------------------------------------------------------------- epilogue
  // The accumulators hold one channel per lane: stored as they are, every lane would issue 16*MB*NB
  // four-byte stores (and as many loads of the identity).  Instead the tile goes through LDS once
  // more ([pixel][channel], as h did) and leaves as whole 16-byte channel vectors: bias, identity
  // (read as float4, coalesced), ReLU, store.
  FPC_STAMP(4)
  FPC_LDS_BARRIER();  // every wave is done reading h
  {
    float* ol = reinterpret_cast<float*>(lds4);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = (wn * NB + nb) * 32 + l31;
      const float bias = a.b2[n];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (wm * MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (n < CMIDP) ol[m * (ROWH4 * 4) + n] = acc[mb][nb][r] + bias;
        }
    }
  }
  FPC_LDS_BARRIER();
  {
    constexpr int C4 = CMIDP / 4;              // CMIDP == padded output channels of the block
    constexpr int NE = TH * TW * C4, EIT = (NE + NT - 1) / NT;
    const int oyb = ty * TH, oxb = tx * TW;
    const bool ident = a.k8_x == 0;
    float4 idv[EIT];
    if (ident) {
#pragma unroll
      for (int i = 0; i < EIT; ++i) {
        const int e = tid + i * NT;
        const int m = e / C4, c4 = e - m * C4;
        const int py = m / TW, px = m - py * TW;
        const int y = oyb + py, x = oxb + px;
        const bool ok = (NE % NT == 0 || e < NE) && y < a.Ho && x < a.Wo;
        idv[i] = *reinterpret_cast<const float4*>(a.x + (ok ? ((size_t)(b * a.Ho + y) * a.Wo + x) * a.csx + c4 * 4 : 0));
      }
    }
#pragma unroll
    for (int i = 0; i < EIT; ++i) {
      const int e = tid + i * NT;
      const int m = e / C4, c4 = e - m * C4;
      const int py = m / TW, px = m - py * TW;
      const int y = oyb + py, x = oxb + px;
      if ((NE % NT == 0 || e < NE) && y < a.Ho && x < a.Wo) {
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
        *reinterpret_cast<float4*>(a.out + ((size_t)(b * a.Ho + y) * a.Wo + x) * a.cso + c4 * 4) = v;
      }
    }
  }
  FPC_STAMP(5)
}

// (Round 5 measured this kernel on the diet that paid for the stem and for conv_mfma_kernel -- staging offsets once per
// workgroup, descriptors instead of 64-bit pointers, phase 2's accumulators from the MFMA's inline 0, 32-bit epilogue addresses:
// 1 039 -> 589 and 875 -> 472 static VALU instructions for the two default-plan instances -- and found nothing: layer2.0 0.263 ->
// 0.268 ms, layer_in.0 0.244 -> 0.243; with conv2's fragments or the projection's pixels through descriptors + scalar step
// offsets layer_in.0 was 6-12 % SLOWER.  The variant is not kept; HISTORY.md R5.10.)

}  // namespace fpc
