// stem_pool_kernel<3> (csrc/kernels_misc.h: conv 7x7/2 + BN + ReLU + max-pool 3x3/2 in one launch, fp32 MFMA) alone on B
// frames: checks the pooled map against a CPU restatement (double accumulation) and times it -- `-DSTEM_VARIANT=2`: the
// round-5 instance (stem_pool2_kernel: bias as a K step, v_max3 pooling, 32-bit staging addresses).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DSTEM_VARIANT=2] -o sp stem_pool_bench.hip && ./sp 32 480 640
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../feature-point-cnn_amd/csrc/kernels_misc.h"
using namespace fpc;
#ifndef STEM_VARIANT
#define STEM_VARIANT 1
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, H = argc > 2 ? atoi(argv[2]) : 480, W = argc > 3 ? atoi(argv[3]) : 640;
  const int Ho = H / 2, Wo = W / 2, Hp = H / 4, Wp = W / 4;
  srand(1);
  std::vector<float> in((size_t)B * 3 * H * W), w(64 * 3 * 49), scale(64), bias(64);
  for (auto& v : in) v = (float)rand() / (float)RAND_MAX;
  for (auto& v : w) v = ((float)rand() / (float)RAND_MAX - 0.5f) * 0.4f;
  for (int n = 0; n < 64; ++n) { scale[n] = 0.75f + 0.5f * rand() / (float)RAND_MAX; bias[n] = ((float)rand() / (float)RAND_MAX - 0.5f) * 0.4f; }
  // fragments in the kernels' K order (fpc_api.hip: pack of the fp32 stem)
  constexpr int KG = 19;
  std::vector<float> frag((size_t)(KG + 2) * 2 * 64 * 4, 0.f);
  for (int g = 0; g < KG; ++g)
    for (int nb = 0; nb < 2; ++nb)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
#if STEM_VARIANT == 2
          const StemPair sp = stem_pair2(3, g * 4 + j);
#else
          const StemPair sp = stem_pair(3, g * 4 + j);
#endif
          const int k = (lane >> 5) ? sp.wb : sp.wa, n = nb * 32 + (lane & 31);
          double v = 0.0;
          if (k >= 0 && k < 147) v = (double)w[n * 147 + k] * scale[n];
#if STEM_VARIANT == 2
          if (!(lane >> 5) && sp.wa == STEM_BIAS_TAP) v = bias[n];
#endif
          frag[(((size_t)g * 2 + nb) * 64 + lane) * 4 + j] = (float)v;
        }
  float *din, *dout, *dbias; float4* df;
  const size_t nout = (size_t)B * Hp * Wp * 64;
  CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dout, nout * 4)); CK(hipMalloc(&dbias, 256)); CK(hipMalloc(&df, frag.size() * 4));
  CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(df, frag.data(), frag.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dbias, bias.data(), 256, hipMemcpyHostToDevice));
  StemPoolArgs a{};
  a.in = din; a.wfrag = df; a.bias = dbias; a.out = dout; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.Hp = Hp; a.Wp = Wp;
  a.tiles_x = (Wo + STEM_T - 1) / STEM_T; a.tiles_y = (Ho + STEM_T - 1) / STEM_T;
  a.range = nullptr;
  const int grid = a.tiles_x * a.tiles_y * B;
  hipFuncAttributes fa;
#if STEM_VARIANT == 2
  auto kern = stem_pool2_kernel<3>;
#else
  auto kern = stem_pool_kernel<3>;
#endif
  CK(hipFuncGetAttributes(&fa, (const void*)kern));
  int nb_ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, (const void*)kern, 256, 0));
  printf("variant %d: regs %d, static LDS %zu B, scratch %zu B, %d workgroups/CU\n", STEM_VARIANT, fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes, nb_);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9, best_clear = 1e9;
  for (int rep = 0; rep < 10; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(stem_border_clear_kernel, dim3((B * Hp + 7) / 8), dim3(256), 0, 0, reinterpret_cast<float4*>(dout), Hp, Wp, B * Hp);
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1); best_clear = std::min(best_clear, ms);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a);
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  printf("stem (variant %d): %d frames %dx%d, grid %d: best %.4f ms (border clear %.4f)\n", STEM_VARIANT, B, H, W, grid, best, best_clear);
  std::vector<float> out(nout);
  CK(hipMemcpy(out.data(), dout, nout * 4, hipMemcpyDeviceToHost));
  double maxerr = 0, maxref = 0;
  long checked = 0;
  auto conv = [&](int b, int n, int y, int x) {
    double s = 0;
    for (int c = 0; c < 3; ++c)
      for (int ky = 0; ky < 7; ++ky)
        for (int kx = 0; kx < 7; ++kx) {
          const int iy = 2 * y + ky - 3, ix = 2 * x + kx - 3;
          if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
          s += (double)w[((size_t)n * 3 + c) * 49 + ky * 7 + kx] * in[(((size_t)b * 3 + c) * H + iy) * W + ix];
        }
    return s * scale[n] + bias[n];
  };
  for (int b : {0, B - 1})
    for (int py = 0; py < Hp; py += (py < 10 || py > Hp - 4 ? 1 : 7))
      for (int px = 0; px < Wp; px += (px < 18 || px > Wp - 4 ? 1 : 5))
        for (int n = 0; n < 64; n += 5) {
          double m = -1e30;
          for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
              const int y = 2 * py + dy, x = 2 * px + dx;
              if (y < 0 || y >= Ho || x < 0 || x >= Wo) continue;
              m = std::max(m, conv(b, n, y, x));
            }
          m = std::max(m, 0.0);
          const double got = out[(((size_t)b * Hp + py) * Wp + px) * 64 + n];
          maxerr = std::max(maxerr, std::fabs(got - m));
          maxref = std::max(maxref, m);
          ++checked;
        }
  printf("checked %ld pooled values: max |err| %.3e (max value %.3f)\n", checked, maxerr, maxref);
  return maxerr < 1e-4 ? 0 : 1;
}
