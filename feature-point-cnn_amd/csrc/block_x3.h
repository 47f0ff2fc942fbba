// block_x3.h -- fp32 ResNetBlock / convolution kernels whose matrix products run on the bf16 MFMA pipe
// with SPLIT OPERANDS (`dtype = FPC_F32_SPLIT`).
//
// Every fp32 operand x is split exactly into three bf16 terms x = x0 + x1 + x2 (8 significant bits each,
// by truncation: x0 = top 16 bits of x, x1 = top 16 bits of x - x0, x2 = x - x0 - x1, which then has at most
// 8 significant bits and is exact in bf16).  A product a*b is evaluated as the six cross terms of weight
// >= 2^-16:  a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0, each exact in the fp32 accumulator of
// v_mfma_f32_32x32x16_bf16 (8 x 8 significant bits).  The three dropped terms (a1 b2, a2 b1, a2 b2) are
// below 2^-23 of |a b| -- the size of one fp32 rounding -- so the result carries fp32 accuracy, at 6 bf16
// MFMAs (6 x 32 cycles per 32x32x16) instead of 8 fp32 MFMAs (8 x 64 cycles) per 16 channels: 2.67x the
// fp32 matrix rate.  Tensors in HBM stay fp32 NHWC exactly as in the default path; activations are split
// while they are staged into LDS, weights are split on the host (weights.h: pack_conv_x3).
//
// Structure as block_mfma.h / block_bf16.h: halo tile in LDS (three bf16 planes), weights pre-packed in
// MFMA lane order and read straight from global memory one step ahead, h = relu(bn1(conv1 x)) stays in LDS
// (fp32, split when read), shortcut (projection as extra K, or identity) and ReLU in the epilogue.
#pragma once
#include "block_bf16.h"
#include "kernels_misc.h"

namespace fpc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NP>
struct SplitN {
  uint4 p[NP];
};
typedef SplitN<3> Split3;

// 8 fp32 values -> NP planes of 8 sixteen-bit terms (16 bytes each).
//   NP = 3: bf16 terms by truncation (exact 24 = 3 x 8 bit split, any fp32 magnitude)
//   NP = 2: fp16 terms, round-to-nearest: x = h + l + e with |e| <= 2^-24 |x| for 6.1e-5 <= |x| <= 65504 (fp16's
//           normal range; below it the error is <= 3e-8 absolute, above it the value saturates -- block_x3.h header)
template <int NP>
__device__ __forceinline__ SplitN<NP> splitN(const float4& a, const float4& b) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  SplitN<NP> s;
  if constexpr (NP == 3) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned u = __float_as_uint(v[i]);
      const unsigned hi = u & 0xffff0000u;
      const float r = v[i] - __uint_as_float(hi);
      const unsigned mi = __float_as_uint(r) & 0xffff0000u;
      const float q = r - __uint_as_float(mi);
      h[i] = hi;
      m[i] = mi;
      l[i] = __float_as_uint(q);  // <= 8 significant bits: its low 16 bits are zero
    }
    // pack pairs: element 2j in the low half, 2j+1 in the high half
    s.p[0] = make_uint4((h[0] >> 16) | h[1], (h[2] >> 16) | h[3], (h[4] >> 16) | h[5], (h[6] >> 16) | h[7]);
    s.p[1] = make_uint4((m[0] >> 16) | m[1], (m[2] >> 16) | m[3], (m[4] >> 16) | m[5], (m[6] >> 16) | m[7]);
    s.p[2] = make_uint4((l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u),
                        (l[4] >> 16) | (l[5] & 0xffff0000u), (l[6] >> 16) | (l[7] & 0xffff0000u));
  } else {
    unsigned hh[8], ll[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const _Float16 h = (_Float16)__builtin_amdgcn_fmed3f(v[i], -65504.f, 65504.f);
      const _Float16 l = (_Float16)__builtin_amdgcn_fmed3f(v[i] - (float)h, -65504.f, 65504.f);
      hh[i] = __builtin_bit_cast(unsigned short, h);
      ll[i] = __builtin_bit_cast(unsigned short, l);
    }
    s.p[0] = make_uint4(hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16), hh[4] | (hh[5] << 16), hh[6] | (hh[7] << 16));
    s.p[1] = make_uint4(ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16), ll[4] | (ll[5] << 16), ll[6] | (ll[7] << 16));
  }
  return s;
}
__device__ __forceinline__ Split3 split8(const float4& a, const float4& b) { return splitN<3>(a, b); }

#define FPC_X3_MFMA(ACC, A, B)                                                                                          \
  do {                                                                                                                  \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (A)[2]), __builtin_bit_cast(bf16x8, (B)[0]), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (A)[0]), __builtin_bit_cast(bf16x8, (B)[2]), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (A)[1]), __builtin_bit_cast(bf16x8, (B)[1]), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (A)[1]), __builtin_bit_cast(bf16x8, (B)[0]), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (A)[0]), __builtin_bit_cast(bf16x8, (B)[1]), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (A)[0]), __builtin_bit_cast(bf16x8, (B)[0]), ACC, 0, 0, 0); \
  } while (0)

// acc += a * b on split operands: NP = 3 six bf16 MFMAs, NP = 2 three fp16 MFMAs (small terms first)
template <int NP>
__device__ __forceinline__ void mfma_split(f32x16& acc, const uint4* A, const uint4* B) {
  if constexpr (NP == 1) {  // plain bf16 operands (FPC_BF16's stem)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[0]), __builtin_bit_cast(bf16x8, B[0]), acc, 0, 0, 0);
  } else if constexpr (NP == 3) {
    FPC_X3_MFMA(acc, A, B);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[1]), __builtin_bit_cast(f16x8, B[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[0]), __builtin_bit_cast(f16x8, B[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[0]), __builtin_bit_cast(f16x8, B[0]), acc, 0, 0, 0);
  }
}

template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB, int CMIDP, int NP>
struct BlockSplitCfg {
  static constexpr int NT = WM * WN * 64;
  static constexpr int HW = (TW - 1) * S + EXT, HH = (TH - 1) * S + EXT;
  static constexpr int ROW16 = KC / 8 + 1;            // 16-byte units per halo pixel and plane (+1 skew)
  static constexpr int PLANE16 = HH * HW * ROW16;
  static constexpr int ROWO4 = CMIDP / 4 + 1;         // float4 per row of the fp32 h / output tile
  static constexpr int M = WM * MB * 32, N = WN * NB * 32;
  static constexpr int HALO_BYTES = NP * PLANE16 * 16;
  static constexpr int O_BYTES = M * ROWO4 * 16;
  static constexpr int LDS_BYTES = HALO_BYTES > O_BYTES ? HALO_BYTES : O_BYTES;
  static_assert(KC % 16 == 0 && CMIDP % 16 == 0 && CMIDP <= N, "bf16 MFMA consumes 16 channels per step");
};

// BlockBfArgs as in block_bf16.h; x and out are fp32 (in_f32 / out_f32 are ignored), csx / cso in floats.
// Weight fragments: [step][plane][nb][lane] uint4.
template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB, int CMIDP>
using BlockX3Cfg = BlockSplitCfg<TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP, 3>;
template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB, int CMIDP>
using BlockH2Cfg = BlockSplitCfg<TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP, 2>;

template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB, int CMIDP, int NP>
__device__ __forceinline__ void block_split_body(const BlockBfArgs& a) {
  using C = BlockSplitCfg<TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP, NP>;
  constexpr int NT = C::NT, HW = C::HW, HH = C::HH, ROW16 = C::ROW16, K16 = KC / 16, KC8 = KC / 8;
  constexpr int NV = HH * HW * KC8, ITER = (NV + NT - 1) / NT, PLANE16 = C::PLANE16, ROWO4 = C::ROWO4, NBT = WN * NB;
  extern __shared__ uint4 lds16[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y;
  const int bidx = fpc_xcd_tile_index();
  const int bl = bidx / tiles;
  const int b = a.frame0 + bl;
  const int t = bidx - bl * tiles;
  const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
  const float* __restrict__ xin = static_cast<const float*>(a.x);

  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    int m = (wm * MB + mb) * 32 + l31;
    m = m < TH * TW ? m : TH * TW - 1;
    const int py = m / TW, px = m - py * TW;
    abase[mb] = ((py * S) * HW + px * S) * ROW16 + half;
  }
  constexpr int planestride = NBT * 64, stepstride = NP * NBT * 64;
  const uint4* wp = a.w1 + (size_t)(wn * NB) * 64 + lane;

  const int iy0 = ty * TH * S - a.pad, ix0 = tx * TW * S - a.pad;
  float4 stage[ITER][2];
  auto load_chunk = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int e = tid + i * NT;
      const int pix = e / KC8, c8 = e - pix * KC8;
      const int hy = pix / HW, hx = pix - hy * HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = (NV % NT == 0 || e < NV) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const size_t off = ok ? ((size_t)(b * a.H + iy) * a.W + ix) * a.csx + chunk * KC + c8 * 8 : 0;
      const float4* p = reinterpret_cast<const float4*>(xin + off);
      float4 v0 = p[0], v1 = p[1];
      if (!ok) v0 = v1 = make_float4(0.f, 0.f, 0.f, 0.f);
      stage[i][0] = v0;
      stage[i][1] = v1;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int e = tid + i * NT;
      const int pix = e / KC8, c8 = e - pix * KC8;
      if (NV % NT == 0 || e < NV) {
        const SplitN<NP> s = splitN<NP>(stage[i][0], stage[i][1]);
#pragma unroll
        for (int p = 0; p < NP; ++p) lds16[p * PLANE16 + pix * ROW16 + c8] = s.p[p];
      }
    }
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
  bool bad = false;  // NP == 2: a value outside fp16's range was produced (it saturates when the next layer splits it)

  FPC_STAMP(0)
  // ---------------------------------------------------------------- phase 1: KxK conv
  load_chunk(0);
  uint4 bc[NB][NP];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int p = 0; p < NP; ++p) bc[nb][p] = wp[p * planestride + nb * 64];
  wp += stepstride;
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    if (chunk) FPC_LDS_BARRIER();
    store_chunk();
    FPC_LDS_BARRIER();
    if (chunk == 0) { FPC_STAMP(1) }
    if (chunk + 1 < a.nchunk) load_chunk(chunk + 1);
    for (int tap = 0; tap < a.ntaps; ++tap) {
      const int toff = a.tapoff16[tap];
#pragma unroll
      for (int k = 0; k < K16; ++k) {
        uint4 bn[NB][NP];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int p = 0; p < NP; ++p) bn[nb][p] = wp[p * planestride + nb * 64];
        wp += stepstride;
        __builtin_amdgcn_sched_barrier(0);
        uint4 av[MB][NP];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int p = 0; p < NP; ++p) av[mb][p] = lds16[p * PLANE16 + abase[mb] + toff + k * 2];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) mfma_split<NP>(acc[mb][nb], av[mb], bc[nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int p = 0; p < NP; ++p) bc[nb][p] = bn[nb][p];
      }
    }
  }

  FPC_STAMP(2)
  if (!a.conv_only) {
    // -------------------------------------------------------------- h = relu(acc + b1) -> LDS (fp32)
    const uint4* wq = a.w2 + (size_t)(wn * NB) * 64 + lane;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int p = 0; p < NP; ++p) bc[nb][p] = wq[p * planestride + nb * 64];
    wq += stepstride;
    FPC_LDS_BARRIER();
    {
      float* hl = reinterpret_cast<float*>(lds16);
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
            if (n < CMIDP) hl[m * (ROWO4 * 4) + n] = v > 0.f ? v : 0.f;
            if (NP == 2) bad |= v > 65504.f;
            acc[mb][nb][r] = 0.f;
          }
      }
    }
    FPC_LDS_BARRIER();
    FPC_STAMP(3)
    // -------------------------------------------------------------- phase 2a: K over h (LDS, split on read)
    const float4* hl4 = reinterpret_cast<const float4*>(lds16);
    int hbase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) hbase[mb] = ((wm * MB + mb) * 32 + l31) * ROWO4 + half * 2;
    for (int k = 0; k < a.k16_h; ++k) {
      uint4 bn[NB][NP];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int p = 0; p < NP; ++p) bn[nb][p] = wq[p * planestride + nb * 64];
      wq += stepstride;
      __builtin_amdgcn_sched_barrier(0);
      SplitN<NP> av[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) av[mb] = splitN<NP>(hl4[hbase[mb] + k * 4], hl4[hbase[mb] + k * 4 + 1]);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) mfma_split<NP>(acc[mb][nb], av[mb].p, bc[nb]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int p = 0; p < NP; ++p) bc[nb][p] = bn[nb][p];
    }
    // -------------------------------------------------------------- phase 2b: K over x (projection)
    if (a.k16_x > 0) {
      size_t xoff[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        int m = (wm * MB + mb) * 32 + l31;
        m = m < TH * TW ? m : TH * TW - 1;
        const int py = m / TW, px = m - py * TW;
        int y = (ty * TH + py) * S, x = (tx * TW + px) * S;
        y = y < a.H ? y : a.H - 1;
        x = x < a.W ? x : a.W - 1;
        xoff[mb] = ((size_t)(b * a.H + y) * a.W + x) * a.csx + half * 8;
      }
      float4 an[MB][2];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const float4* p = reinterpret_cast<const float4*>(xin + xoff[mb]);
        an[mb][0] = p[0];
        an[mb][1] = p[1];
      }
      for (int k = 0; k < a.k16_x; ++k) {
        uint4 bn[NB][NP];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int p = 0; p < NP; ++p) bn[nb][p] = wq[p * planestride + nb * 64];
        wq += stepstride;
        SplitN<NP> av[MB];
        const int kn = k + 1 < a.k16_x ? k + 1 : k;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          av[mb] = splitN<NP>(an[mb][0], an[mb][1]);
          const float4* p = reinterpret_cast<const float4*>(xin + xoff[mb] + kn * 16);
          an[mb][0] = p[0];
          an[mb][1] = p[1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) mfma_split<NP>(acc[mb][nb], av[mb].p, bc[nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int p = 0; p < NP; ++p) bc[nb][p] = bn[nb][p];
      }
    }
  }

  FPC_STAMP(4)
  // ---------------------------------------------------------------- epilogue: fp32 tile -> LDS -> 8-channel vectors
  FPC_LDS_BARRIER();
  {
    float* ol = reinterpret_cast<float*>(lds16);
    const float* bptr = a.conv_only ? a.b1 : a.b2;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = (wn * NB + nb) * 32 + l31;
      const float bias = bptr[n];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (wm * MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (n < CMIDP) ol[m * (ROWO4 * 4) + n] = acc[mb][nb][r] + bias;
        }
    }
  }
  {
    constexpr int C4 = CMIDP / 4;
    constexpr int NE = TH * TW * C4, EIT = (NE + NT - 1) / NT;
    const float4* ol4 = reinterpret_cast<const float4*>(lds16);
    const int oyb = ty * TH, oxb = tx * TW;
    const bool ident = !a.conv_only && a.k16_x == 0;
    float* __restrict__ outp = static_cast<float*>(a.out);
    // identity shortcut (same geometry as the output, stride 1): the global loads are issued BEFORE the barrier --
    // the accumulators are dead after the LDS writes above, so their registers carry the loads across it
    float4 idv[EIT];
    if (ident) {
#pragma unroll
      for (int i = 0; i < EIT; ++i) {
        const int e = tid + i * NT;
        const int m = e / C4, c4 = e - m * C4;
        const int py = m / TW, px = m - py * TW;
        const int y = oyb + py, x = oxb + px;
        const bool ok = (NE % NT == 0 || e < NE) && y < a.Ho && x < a.Wo;
        idv[i] = *reinterpret_cast<const float4*>(xin + (ok ? ((size_t)(b * a.H + y) * a.W + x) * a.csx + c4 * 4 : 0));
      }
    }
    FPC_LDS_BARRIER();
#pragma unroll
    for (int i = 0; i < EIT; ++i) {
      const int e = tid + i * NT;
      const int m = e / C4, c4 = e - m * C4;
      const int py = m / TW, px = m - py * TW;
      const int y = oyb + py, x = oxb + px;
      if ((NE % NT == 0 || e < NE) && y < a.Ho && x < a.Wo) {
        float4 v = ol4[m * ROWO4 + c4];
        if (ident) {
          v.x += idv[i].x; v.y += idv[i].y; v.z += idv[i].z; v.w += idv[i].w;
        }
        if (!a.norelu) {
          v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        }
        if (NP == 2) bad |= fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) > 65504.f;
        const size_t opix = (size_t)(b * a.OH + y * a.oys + a.oy0) * a.OW + x * a.oxs + a.ox0;
        *reinterpret_cast<float4*>(outp + opix * a.cso + c4 * 4) = v;
      }
    }
  }
  FPC_STAMP(5)
  if (NP == 2 && bad && a.range_flag) atomicOr(a.range_flag, 1);
}

template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB, int CMIDP>
__global__ __launch_bounds__(WM* WN * 64, 2) void block_x3_kernel(const BlockBfArgs a) {
  block_split_body<TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP, 3>(a);
}

// The same kernel on two fp16 terms per operand and three fp16 MFMAs per product (dtype = FPC_F32_SPLIT_F16)
template <int TH, int TW, int S, int EXT, int KC, int WM, int WN, int MB, int NB, int CMIDP>
__global__ __launch_bounds__(WM* WN * 64, 2) void block_h2_kernel(const BlockBfArgs a) {
  block_split_body<TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP, 2>(a);
}

// ---------------------------------------------------------------------------------
// Stem + max-pool on split operands (stem_pool_kernel of kernels_misc.h with the 7x7/2 convolution as six
// bf16 MFMAs per product).  K is laid out as (input channel, filter row) x 8 filter columns (7 real + one
// zero weight): a lane's 8 values of one K16 half are 8 consecutive pixels of one input row, read from the
// LDS window (three bf16 planes) as four dwords.  K16 step s covers rows 2s (lanes 0-31) and 2s+1 (32-63).
// ---------------------------------------------------------------------------------
constexpr int STEMX_LW = 40;                                   // bf16 per LDS row (37 real + zero pad)
constexpr int STEMX_PLANE = 3 * STEM_HALO * STEMX_LW;          // bf16 per plane (CIN = 3)
constexpr int STEMX_LDS_BYTES = (3 * STEMX_PLANE * 2 > 256 * STEM_TROW * 4) ? 3 * STEMX_PLANE * 2 : 256 * STEM_TROW * 4;

struct StemX3Args {
  const float* in;      // [B,CIN,H,W]
  const uint4* wfrag;   // [steps + 2][NP planes][2 nb][64] uint4
  const float* bias;    // [64]
  float* out;           // [B,Hp,Wp,64], zero-filled
  int H, W, Ho, Wo, Hp, Wp, tiles_x, tiles_y;
  int frames;           // stem_pool_bf16_kernel (persistent grid): frames of the launch
  int* range_flag;      // NP == 2: raised when an output leaves fp16's range
#ifdef FPC_DIAG
  unsigned long long* stamps;
#endif
};

template <int CIN, int NP>
__global__ __launch_bounds__(256) void stem_pool_x3_kernel(const StemX3Args a) {
  constexpr int ROWS = CIN * 7, STEPS = (ROWS + 1) / 2;  // 21 -> 11 steps; 7 -> 4
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[STEMX_LDS_BYTES];
  const unsigned* lds32 = reinterpret_cast<const unsigned*>(lds_raw);
  float* lds = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y;
  const int bidx = fpc_xcd_tile_index();   // neighbouring tiles complete each other's border windows with atomics: keep them in one L2
  const int b = bidx / tiles;
  const int t = bidx - b * tiles;
  const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
  const int iy0 = ty * STEM_T * 2 - 3, ix0 = tx * STEM_T * 2 - 3;

  FPC_STAMP(0)
  const uint4* wp = a.wfrag + lane;
  uint4 bc[2][NP];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int p = 0; p < NP; ++p) bc[nb][p] = wp[(p * 2 + nb) * 64];
  wp += 2 * NP * 64;
  uint4 b1[2][NP];  // fragments run two steps ahead (the blob carries two zero steps of padding)
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int p = 0; p < NP; ++p) b1[nb][p] = wp[(p * 2 + nb) * 64];
  wp += 2 * NP * 64;

  {  // input window -> NP sixteen-bit planes in LDS, as aligned float4 row segments: LDS column 0 is image column
     // ix0 - 1 (= 32 tx - 4; W is a multiple of 8, so a float4 is entirely inside or outside the frame); the 8 K-values
     // of a lane then start at the even column 2 px with the zero-weight pad in FRONT (kx' = kx + 1).
    constexpr int NQ = STEMX_LW / 4, NE = CIN * STEM_HALO * NQ, IT = (NE + 255) / 256;
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
    uint2* lds64 = reinterpret_cast<uint2*>(lds_raw);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int e = tid + i * 256;
      if (e < NE) {
        const float f[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        unsigned short t[NP][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if constexpr (NP == 1) {
            t[0][k] = f2bf(f[k]);
          } else if constexpr (NP == 3) {
            const unsigned u = __float_as_uint(f[k]);
            const unsigned hi = u & 0xffff0000u;
            const float r = f[k] - __uint_as_float(hi);
            const unsigned mi = __float_as_uint(r) & 0xffff0000u;
            const float qq = r - __uint_as_float(mi);
            t[0][k] = (unsigned short)(hi >> 16);
            t[1][k] = (unsigned short)(mi >> 16);
            t[NP - 1][k] = (unsigned short)(__float_as_uint(qq) >> 16);
          } else {  // two fp16 terms (frame values lie in [0, 1]; clamped like every other operand)
            const _Float16 hh = (_Float16)__builtin_amdgcn_fmed3f(f[k], -65504.f, 65504.f);
            const _Float16 ll = (_Float16)__builtin_amdgcn_fmed3f(f[k] - (float)hh, -65504.f, 65504.f);
            t[0][k] = __builtin_bit_cast(unsigned short, hh);
            t[NP - 1][k] = __builtin_bit_cast(unsigned short, ll);
          }
        }
#pragma unroll
        for (int p = 0; p < NP; ++p)  // element e * 4 of plane p = 8-byte unit p * STEMX_PLANE / 4 + e
          lds64[p * (STEMX_PLANE / 4) + e] = make_uint2(t[p][0] | ((unsigned)t[p][1] << 16), t[p][2] | ((unsigned)t[p][3] << 16));
      }
    }
  }
  __syncthreads();
  FPC_STAMP(1)

  int abase[2];  // dword index of (row 0, 2*px) of this lane's output pixel
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int m = (wave * 2 + mb) * 32 + l31;
    abase[mb] = ((2 * (m / STEM_T)) * STEMX_LW + 2 * (m % STEM_T)) / 2;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    uint4 bn[2][NP];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int p = 0; p < NP; ++p) bn[nb][p] = wp[(p * 2 + nb) * 64];
    wp += 2 * NP * 64;
    __builtin_amdgcn_sched_barrier(0);
    // this lane's filter row: (c, ky); a padded row (weights zero) re-reads the last real one
    const int r0 = 2 * s < ROWS ? 2 * s : ROWS - 1, r1 = 2 * s + 1 < ROWS ? 2 * s + 1 : ROWS - 1;
    const int off0 = ((r0 / 7) * STEM_HALO + (r0 % 7)) * (STEMX_LW / 2), off1 = ((r1 / 7) * STEM_HALO + (r1 % 7)) * (STEMX_LW / 2);
    const int off = half ? off1 : off0;
    uint4 av[2][NP];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const unsigned* q = lds32 + p * (STEMX_PLANE / 2) + abase[mb] + off;
        av[mb][p] = make_uint4(q[0], q[1], q[2], q[3]);
      }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) mfma_split<NP>(acc[mb][nb], av[mb], bc[nb]);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        bc[nb][p] = b1[nb][p];
        b1[nb][p] = bn[nb][p];
      }
  }

  FPC_STAMP(2)
  // epilogue: per 32-channel half, tile -> LDS -> 3x3/2 max-pool (as stem_pool_kernel)
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    __syncthreads();
    const float bias = a.bias[nb * 32 + l31];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (wave * 2 + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float v = acc[mb][nb][r] + bias;
        lds[m * STEM_TROW + l31] = v > 0.f ? v : 0.f;
      }
    __syncthreads();
    if (nb == 0) { FPC_STAMP(3) } else { FPC_STAMP(5) }
    stem_pool_emit(lds, a.out, b, ty, tx, nb, a.Ho, a.Wo, a.Hp, a.Wp, tid, NP == 2 ? a.range_flag : nullptr);
    if (nb == 0) { FPC_STAMP(4) } else { FPC_STAMP(6) }
  }
}

}  // namespace fpc
