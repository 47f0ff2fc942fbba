// Stand-alone check + timing of wblock36p_kernel (the 64-channel Winograd F(4x4,3x3) ResNetBlock at two waves per SIMD,
// csrc/wblock36p_mfma.h) beside wblock36_kernel<1, ...> (one wave per SIMD) on the same problems.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o w36p w36p_bench.hip
//   ./w36 check            small problems against a double-precision CPU block (all instances, all modes)
//   ./w36 time             the layers of the VGA path at B = 32
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <algorithm>
#include "../../feature-point-cnn_amd/csrc/wblock36p_mfma.h"
using namespace fpc;
#ifndef W36P_RING
#define W36P_RING 9
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

static const double G43[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};

struct Problem {
  int B, H, W, Cin, N, Cx;     // Cx: 0 = identity shortcut (Cin == N), else projection over Cx = Cin channels
  bool conv_only; int ysplit;  // conv_only with ysplit = 2: 2 N output channels as two halves
};

template <int NBP, int TYT, int TXT>      // NBP = 0: wblock36p_kernel<TYT, TXT, W36P_RING>; 1, 2: wblock36_kernel<NBP, TYT, TXT>
static double run(const Problem& P, bool check, int reps, int grid_override = 0) {
  constexpr int NB = NBP == 0 ? 1 : NBP;
  constexpr bool PAIRED = NBP == 0;
  using C = W36Cfg<NB, TYT, TXT>;
  constexpr int LDSB = PAIRED ? W36PCfg<TYT, TXT>::LDS_BYTES : C::LDS_BYTES;
  constexpr int THREADS = PAIRED ? 512 : 256;
  const void* fn = PAIRED ? (const void*)wblock36p_kernel<TYT, TXT, W36P_RING> : (const void*)wblock36_kernel<NB, TYT, TXT>;
  auto launch = [&](dim3 g, const WBlockArgs& aa) {
    if constexpr (PAIRED) hipLaunchKernelGGL((wblock36p_kernel<TYT, TXT, W36P_RING>), g, dim3(512), LDSB, 0, aa);
    else hipLaunchKernelGGL((wblock36_kernel<NB, TYT, TXT>), g, dim3(256), LDSB, 0, aa);
  };
  (void)THREADS;
  const int N = C::N, ncg = N / 16, nchunk = P.Cin / 16, nout = P.conv_only ? N * P.ysplit : N;
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> x((size_t)P.B * P.H * P.W * P.Cin);
  for (auto& v : x) v = std::max(0.f, nd(rng));
  std::vector<double> w1((size_t)nout * P.Cin * 9), w2((size_t)N * N), wp((size_t)N * std::max(1, P.Cx));
  std::vector<float> b1(nout), b2(N);
  for (auto& v : w1) v = nd(rng) * std::sqrt(2.0 / (P.Cin * 9));
  for (auto& v : w2) v = nd(rng) * std::sqrt(1.0 / N);
  for (auto& v : wp) v = nd(rng) * std::sqrt(1.0 / std::max(1, P.Cx));
  for (auto& v : b1) v = 0.1f * nd(rng);
  for (auto& v : b2) v = 0.1f * nd(rng);
  // ---- pack w1: per half (ysplit), per channel group: [chunk][pos][lane] float4 + WPAD positions
  const size_t gstride = ((size_t)nchunk * 36 + C::WPAD) * 256;            // floats per group
  const size_t half_floats = (size_t)ncg * gstride + N;                    // fragments + bias of one half (ysplit_floats)
  const int nh = P.conv_only ? P.ysplit : 1;
  std::vector<float> w1p(half_floats * nh, 0.f);
  for (int hf = 0; hf < nh; ++hf)
    for (int cg = 0; cg < ncg; ++cg)
      for (int ch = 0; ch < nchunk; ++ch)
        for (int pos = 0; pos < 36; ++pos)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
              const int n = hf * N + 16 * cg + (lane & 15), cc = 16 * ch + 4 * (lane >> 4) + j;
              double u = 0;
              for (int p = 0; p < 3; ++p)
                for (int q = 0; q < 3; ++q) u += G43[pos / 6][p] * w1[((size_t)n * P.Cin + cc) * 9 + p * 3 + q] * G43[pos % 6][q];
              w1p[hf * half_floats + cg * gstride + (((size_t)ch * 36 + pos) * 64 + lane) * 4 + j] = (float)u;
            }
  for (int hf = 0; hf < nh; ++hf)
    for (int n = 0; n < N; ++n) w1p[hf * half_floats + (size_t)ncg * gstride + n] = b1[hf * N + n];
  // ---- pack w2: per group [step of 16][lane] float4, h steps then x steps, + WPAD2 steps
  const int k8_h = N / 8, k8_x = P.Cx / 8;
  const size_t g2 = ((size_t)(k8_h + k8_x) / 2 + C::WPAD2) * 256;
  std::vector<float> w2p((size_t)ncg * g2, 0.f);
  for (int cg = 0; cg < ncg; ++cg)
    for (int st = 0; st < (k8_h + k8_x) / 2; ++st)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
          const int n = 16 * cg + (lane & 15), cc = 16 * st + 4 * (lane >> 4) + j;
          w2p[cg * g2 + ((size_t)st * 64 + lane) * 4 + j] = cc < N ? (float)w2[(size_t)n * N + cc] : (float)wp[(size_t)n * P.Cx + (cc - N)];
        }
  float *dx, *dw1, *dw2, *db2, *dout;
  const size_t nout_el = (size_t)P.B * P.H * P.W * nout;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dw1, w1p.size() * 4)); CK(hipMalloc(&dw2, w2p.size() * 4));
  CK(hipMalloc(&db2, N * 4)); CK(hipMalloc(&dout, nout_el * 4));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw1, w1p.data(), w1p.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw2, w2p.data(), w2p.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db2, b2.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dout, 0xff, nout_el * 4));
  if (getenv("W36_PTRS")) printf("x %p..%p  w1 %p..%p  w2 %p..%p  b2 %p  out %p..%p\n", dx, dx + x.size(), dw1, dw1 + w1p.size(), dw2, dw2 + w2p.size(), db2, dout, dout + nout_el);
  WBlockArgs a{};
  a.x = dx; a.csx = P.Cin; a.nchunk = nchunk; a.H = P.H; a.W = P.W;
  a.w1 = (const float4*)dw1; a.b1 = dw1 + (size_t)ncg * gstride; a.w2 = (const float4*)dw2; a.b2 = db2;
  a.k8_h = k8_h; a.k8_x = k8_x; a.out = dout; a.cso = nout;
  a.tiles_x = (P.W + C::TW - 1) / C::TW; a.tiles_y = (P.H + C::TH - 1) / C::TH; a.frame0 = 0;
  a.total = a.tiles_x * a.tiles_y * P.B; a.xcd_order = 1; a.ysplit_floats = (int)half_floats;
  a.x_bytes = (unsigned)(x.size() * 4); a.conv_only = P.conv_only ? 1 : 0;
#ifdef FPC_DIAG
  unsigned long long* dst_; CK(hipMalloc(&dst_, 1024 * 8 * 8 + 1024 * 4 * 16 * 8)); CK(hipMemset(dst_, 0, 1024 * 8 * 8 + 1024 * 4 * 16 * 8)); a.stamps = dst_;
#endif
  CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  int grid = grid_override ? grid_override : 256;
  if (grid > a.total) grid = a.total >= 8 ? (a.total / 8) * 8 : a.total;
  const dim3 g(grid, nh);
  launch(g, a);
  CK(hipDeviceSynchronize());
  double result = 0;
  if (check) {
    std::vector<float> out(nout_el);
    CK(hipMemcpy(out.data(), dout, nout_el * 4, hipMemcpyDeviceToHost));
    // CPU block in double
    double maxerr = 0, scale = 0;
    std::vector<double> h((size_t)P.H * P.W * nout);
    for (int b = 0; b < P.B; ++b) {
      for (int y = 0; y < P.H; ++y)
        for (int xx = 0; xx < P.W; ++xx)
          for (int n = 0; n < nout; ++n) {
            double s = b1[n];
            for (int p = 0; p < 3; ++p)
              for (int q = 0; q < 3; ++q) {
                const int iy = y + p - 1, ix = xx + q - 1;
                if (iy < 0 || iy >= P.H || ix < 0 || ix >= P.W) continue;
                const float* xp = &x[((size_t)(b * P.H + iy) * P.W + ix) * P.Cin];
                const double* wq = &w1[(size_t)n * P.Cin * 9 + p * 3 + q];
                for (int c = 0; c < P.Cin; ++c) s += xp[c] * wq[(size_t)c * 9];
              }
            h[((size_t)y * P.W + xx) * nout + n] = s > 0 ? s : 0;
          }
      for (int y = 0; y < P.H; ++y)
        for (int xx = 0; xx < P.W; ++xx)
          for (int n = 0; n < nout; ++n) {
            double want;
            if (P.conv_only) want = h[((size_t)y * P.W + xx) * nout + n];
            else {
              double s = b2[n];
              for (int c = 0; c < N; ++c) s += h[((size_t)y * P.W + xx) * N + c] * w2[(size_t)n * N + c];
              const float* xp = &x[((size_t)(b * P.H + y) * P.W + xx) * P.Cin];
              if (P.Cx) for (int c = 0; c < P.Cx; ++c) s += xp[c] * wp[(size_t)n * P.Cx + c];
              else s += xp[n];
              want = s > 0 ? s : 0;
            }
            const double got = out[((size_t)(b * P.H + y) * P.W + xx) * nout + n];
            const double e = std::fabs(got - want);
            if (!(e <= maxerr)) maxerr = std::isnan(e) ? 1e30 : e;
            if (std::fabs(want) > scale) scale = std::fabs(want);
          }
    }
    printf("  check %s NB=%d %dx%d tiles  B=%d %dx%d Cin=%d N=%d Cx=%d conv_only=%d ysplit=%d: max err %.3g (scale %.3g, rel %.3g) %s\n", PAIRED ? "PAIRED" : "one-wave", NB, TYT, TXT,
           P.B, P.H, P.W, P.Cin, N, P.Cx, (int)P.conv_only, P.ysplit, maxerr, scale, maxerr / scale, maxerr / scale < 1e-4 ? "OK" : "FAIL");
    result = maxerr / scale;
  } else {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < reps; ++rep) {
      hipEventRecord(e0);
      launch(g, a);
      hipEventRecord(e1);
      CK(hipDeviceSynchronize());
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; sum += ms;
    }
    const double macs = (double)P.B * P.H * P.W * ((double)nout * P.Cin * 9 + (P.conv_only ? 0.0 : (double)N * (N + P.Cx)));
    printf("  time %s NB=%d %dx%d tiles  B=%d %dx%d Cin=%d N=%d Cx=%d conv_only=%d grid=%d: best %.4f ms, mean %.4f ms, %d tiles, %.1f TFLOP/s algorithmic\n", PAIRED ? "PAIRED" : "one-wave", NB, TYT, TXT,
           P.B, P.H, P.W, P.Cin, N, P.Cx, (int)P.conv_only, grid, best, sum / reps, a.total, 2 * macs / (best * 1e-3) / 1e12);
    result = best;
#ifdef FPC_DIAG
    {
      std::vector<unsigned long long> st(1024 * 8);
      CK(hipMemcpy(st.data(), a.stamps, st.size() * 8, hipMemcpyDeviceToHost));
      // stamps 0..5 of a workgroup's third tile: tile start, chunk loop end, h written (half 0), 1x1 over h done, GEMMs done, half 0 stored
      const char* nm[5] = {"chunk loop", "out transform + h", "1x1 over h", "projection", "epilogue half 0"};
      for (int k = 0; k < 5; ++k) {
        std::vector<double> d;
        for (int g2 = 0; g2 < grid; ++g2) if (st[g2 * 8 + k + 1] > st[g2 * 8 + k] && st[g2 * 8 + k]) d.push_back((double)(st[g2 * 8 + k + 1] - st[g2 * 8 + k]));
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        printf("      stamp %-20s median %8.0f ticks  (min %.0f, max %.0f, n %zu)\n", nm[k], d[d.size() / 2], d.front(), d.back(), d.size());
      }
      // inside chunk 2 of that tile, per wave: stamps at every (STEPS / 9)-th step, loop end, behind the barrier
      std::vector<unsigned long long> tq(1024 * 4 * 16);
      CK(hipMemcpy(tq.data(), a.stamps + 1024 * 8, tq.size() * 8, hipMemcpyDeviceToHost));
      for (int w = 0; w < 4; ++w) {
        printf("      chunk 2, wave %d:", w);
        for (int k = 0; k < 10; ++k) {
          std::vector<double> d;
          for (int g2 = 0; g2 < grid; ++g2) { const unsigned long long* q = &tq[(g2 * 4 + w) * 16]; if (q[k + 1] > q[k] && q[k]) d.push_back((double)(q[k + 1] - q[k])); }
          if (d.empty()) { printf("     -"); continue; }
          std::sort(d.begin(), d.end());
          printf(" %5.0f", d[d.size() / 2]);
        }
        printf("\n");
      }
    }
#endif
  }
  hipFree(dx); hipFree(dw1); hipFree(dw2); hipFree(db2); hipFree(dout);
  return result;
}

int main(int argc, char** argv) {
  const bool check = argc < 2 || !strcmp(argv[1], "check");
  if (check) {
    if (argc <= 2) {
      run<0, 2, 8>({2, 24, 70, 64, 64, 0, false, 1}, true, 0);
      run<0, 2, 8>({1, 13, 40, 64, 64, 64, false, 1}, true, 0);
      run<0, 4, 4>({1, 30, 30, 64, 64, 0, false, 1}, true, 0);
      run<0, 4, 4>({2, 20, 36, 128, 64, 128, false, 1}, true, 0);
      run<0, 4, 4>({1, 16, 48, 64, 64, 0, true, 2}, true, 0);
      run<0, 4, 4>({1, 30, 40, 256, 64, 0, true, 4}, true, 0);
      run<0, 2, 8>({1, 17, 33, 128, 64, 0, true, 1}, true, 0);
    }
    if (argc > 2) {   // tiny maps (tiles larger than the frame), one case per process
      const int k = atoi(argv[2]);
      if (k == 0) run<0, 2, 8>({2, 16, 24, 64, 64, 64, false, 1}, true, 0);
      if (k == 1) run<0, 4, 4>({1, 8, 12, 64, 64, 0, false, 1}, true, 0);
      if (k == 2) run<0, 4, 4>({2, 4, 6, 64, 64, 0, false, 1}, true, 0);
      if (k == 3) run<0, 2, 8>({1, 4, 6, 128, 64, 128, false, 1}, true, 0);
      if (k == 4) run<0, 4, 4>({1, 2, 3, 256, 64, 0, true, 4}, true, 0);
      return 0;
    }
  } else {
    const int B = argc > 2 ? atoi(argv[2]) : 32;
    run<1, 2, 8>({B, 120, 160, 64, 64, 64, false, 1}, false, 10);      // layer1.0
    run<0, 2, 8>({B, 120, 160, 64, 64, 64, false, 1}, false, 10);
    run<1, 2, 8>({B, 120, 160, 64, 64, 0, false, 1}, false, 10);       // layer1.1
    run<0, 2, 8>({B, 120, 160, 64, 64, 0, false, 1}, false, 10);
    run<1, 4, 4>({B, 30, 40, 256, 64, 0, true, 4}, false, 10);         // layer_in.1 conv1, four parts
    run<0, 4, 4>({B, 30, 40, 256, 64, 0, true, 4}, false, 10);
    run<1, 4, 4>({B, 60, 80, 128, 64, 128, false, 1}, false, 10);      // (the detector's 64 channels, first block)
    run<0, 4, 4>({B, 60, 80, 128, 64, 128, false, 1}, false, 10);
  }
  return 0;
}
