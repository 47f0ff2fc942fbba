// weights.h -- host side: checkpoint entries -> BatchNorm-folded, MFMA-fragment-packed blob.
//
// Input is the reference's ckpt['model_state_dict'] (python/src/saveutils.py:57-62,
// SURVEY.md table W) as name -> float tensor.  BatchNorm is evaluated in inference
// form (running statistics, eps 1e-5):  y = conv(x) * s + t  with
// s = gamma / sqrt(var + eps), t = beta - mean * s  (python/src/resnet_blocks.py:17,20,23).
// s is folded into the convolution weights in double precision, t becomes the bias.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

namespace fpc {

struct HostTensor {
  const float* data = nullptr;
  std::vector<int64_t> shape;
};
using TensorMap = std::unordered_map<std::string, HostTensor>;

struct Fold {
  std::vector<double> s, t;
};

inline bool has_shape(const TensorMap& m, const std::string& k, std::initializer_list<int64_t> shp) {
  auto it = m.find(k);
  if (it == m.end() || it->second.data == nullptr) return false;
  return it->second.shape == std::vector<int64_t>(shp);
}

// BatchNorm2d(C) entries under `prefix` -> (s, t); returns false when an entry is missing / misshapen.
inline bool fold_bn(const TensorMap& m, const std::string& prefix, int C, Fold* f, std::string* missing) {
  static const char* leaf[4] = {".weight", ".bias", ".running_mean", ".running_var"};
  const float* p[4];
  for (int i = 0; i < 4; ++i) {
    if (!has_shape(m, prefix + leaf[i], {C})) {
      *missing = prefix + leaf[i];
      return false;
    }
    p[i] = m.at(prefix + leaf[i]).data;
  }
  f->s.resize(C);
  f->t.resize(C);
  for (int c = 0; c < C; ++c) {
    const double s = (double)p[0][c] / std::sqrt((double)p[3][c] + 1e-5);
    f->s[c] = s;
    f->t[c] = (double)p[1][c] - (double)p[2][c] * s;
  }
  return true;
}

// One K-source of a packed GEMM-B: `cin_pad` channels (a multiple of KC) of which the
// first `cin` are real, visited for each tap; w(n, c, tap) is the UNSCALED weight.
struct PackSource {
  int cin, cin_pad, ntaps;
  std::function<double(int n, int c, int tap)> w;
  const std::vector<double>* scale;  // per output channel
};

// B fragments in the exact order conv_mfma_kernel consumes them:
//   step = ((chunk * ntaps + tap) * K8 + k8), chunks of source 0 first, then source 1;
//   frag[(step * nbt + nb) * 64 + lane] = { B[step*8 + 4*half + j][nb*32 + (lane&31)] }, j = 0..3
// followed by two all-zero steps (the kernel prefetches two steps ahead).
inline std::vector<float> pack_conv(const std::vector<PackSource>& srcs, int cout, int nbt, int KC) {
  const int K8 = KC / 8;
  size_t nsteps = 0;
  for (auto& s : srcs) nsteps += (size_t)(s.cin_pad / KC) * s.ntaps * K8;
  std::vector<float> out((nsteps + 2) * nbt * 64 * 4, 0.f);
  size_t step = 0;
  for (auto& s : srcs)
    for (int chunk = 0; chunk < s.cin_pad / KC; ++chunk)
      for (int tap = 0; tap < s.ntaps; ++tap)
        for (int k8 = 0; k8 < K8; ++k8, ++step)
          for (int nb = 0; nb < nbt; ++nb)
            for (int lane = 0; lane < 64; ++lane) {
              const int n = nb * 32 + (lane & 31), half = lane >> 5;
              float* dst = &out[((step * nbt + nb) * 64 + lane) * 4];
              for (int j = 0; j < 4; ++j) {
                const int c = chunk * KC + k8 * 8 + 4 * half + j;
                dst[j] = (n < cout && c < s.cin) ? (float)(s.w(n, c, tap) * (*s.scale)[n]) : 0.f;
              }
            }
  return out;
}

// float -> bf16, round to nearest even (what v_cvt_pk_bf16_f32 does on the device)
inline uint16_t host_f2bf(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// Exact three-way split of a float into bf16 terms by truncation (block_x3.h): x = t[0] + t[1] + t[2].
inline void host_split3(float x, uint16_t t[3]) {
  uint32_t u, v;
  memcpy(&u, &x, 4);
  u &= 0xffff0000u;
  float hi;
  memcpy(&hi, &u, 4);
  const float r = x - hi;
  memcpy(&v, &r, 4);
  v &= 0xffff0000u;
  float mid;
  memcpy(&mid, &v, 4);
  const float q = r - mid;
  uint32_t w;
  memcpy(&w, &q, 4);
  t[0] = (uint16_t)(u >> 16);
  t[1] = (uint16_t)(v >> 16);
  t[2] = (uint16_t)(w >> 16);
}

// float -> IEEE half, round to nearest even, saturating at +-65504 (the device's v_cvt_f16_f32 of a clamped value)
inline uint16_t host_f2h(float x) {
  if (x > 65504.f) x = 65504.f;
  if (x < -65504.f) x = -65504.f;
  uint32_t u;
  memcpy(&u, &x, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const int32_t e = (int32_t)((u >> 23) & 0xff) - 127 + 15;
  uint32_t m = u & 0x7fffffu;
  if (((u >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7e00u);  // NaN
  if (e >= 31) return (uint16_t)(sign | 0x7bffu);
  if (e <= 0) {  // subnormal half (or zero)
    if (e < -10) return (uint16_t)sign;
    m |= 0x800000u;
    const int shift = 14 - e;  // 24-bit significand -> 10 bits, plus the subnormal shift
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (r & 1))) ++r;
    return (uint16_t)(sign | r);
  }
  uint32_t r = ((uint32_t)e << 10) | (m >> 13);
  const uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;  // may carry into the exponent: still correct
  if (r >= 0x7c00u) r = 0x7bffu;
  return (uint16_t)(sign | r);
}
inline float host_h2f(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu, u;
  if (e == 0) {
    if (m == 0) {
      u = sign;
    } else {
      int sh = 0;
      while (!(m & 0x400u)) {
        m <<= 1;
        ++sh;
      }
      u = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((m & 0x3ffu) << 13);
    }
  } else {
    u = sign | ((e - 15 + 127) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// Two-term fp16 split (block_x3.h, NP = 2): x ~ t[0] + t[1]
inline void host_split2_f16(float x, uint16_t t[2]) {
  t[0] = host_f2h(x);
  t[1] = host_f2h(x - host_h2f(t[0]));
}

// The same for block_bf16_kernel / block_x3_kernel: 16 channels per step, 8 bf16 (16 bytes) per lane;
//   planes = 1: frag[(step * nbt + nb) * 64 + lane] = { bf16(B[step*16 + 8*half + j][nb*32 + (lane&31)]) }, j = 0..7
//   planes = 3: frag[((step * 3 + p) * nbt + nb) * 64 + lane] = term p of the exact split of the same element
// Returned as floats (4 per lane) so that it sits in the same blob.
inline std::vector<float> pack_conv_bf16(const std::vector<PackSource>& srcs, int cout, int nbt, int KC, int planes = 1,
                                         bool* out_of_fp16_range = nullptr) {
  const int K16 = KC / 16;
  size_t nsteps = 0;
  for (auto& s : srcs) nsteps += (size_t)(s.cin_pad / KC) * s.ntaps * K16;
  std::vector<float> out((nsteps + 2) * planes * nbt * 64 * 4, 0.f);
  uint16_t* o16 = reinterpret_cast<uint16_t*>(out.data());
  size_t step = 0;
  for (auto& s : srcs)
    for (int chunk = 0; chunk < s.cin_pad / KC; ++chunk)
      for (int tap = 0; tap < s.ntaps; ++tap)
        for (int k16 = 0; k16 < K16; ++k16, ++step)
          for (int nb = 0; nb < nbt; ++nb)
            for (int lane = 0; lane < 64; ++lane) {
              const int n = nb * 32 + (lane & 31), half = lane >> 5;
              for (int j = 0; j < 8; ++j) {
                const int c = chunk * KC + k16 * 16 + 8 * half + j;
                const float v = (n < cout && c < s.cin) ? (float)(s.w(n, c, tap) * (*s.scale)[n]) : 0.f;
                if (planes == 1) {
                  o16[((step * nbt + nb) * 64 + lane) * 8 + j] = host_f2bf(v);
                } else if (planes == 2) {  // block_h2_kernel: two fp16 terms
                  if (out_of_fp16_range && !(std::fabs(v) <= 65504.f)) *out_of_fp16_range = true;
                  uint16_t t[2];
                  host_split2_f16(v, t);
                  for (int p = 0; p < 2; ++p) o16[(((step * 2 + p) * nbt + nb) * 64 + lane) * 8 + j] = t[p];
                } else {
                  uint16_t t[3];
                  host_split3(v, t);
                  for (int p = 0; p < 3; ++p) o16[(((step * 3 + p) * nbt + nb) * 64 + lane) * 8 + j] = t[p];
                }
              }
            }
  return out;
}

}  // namespace fpc
