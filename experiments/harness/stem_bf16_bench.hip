// stem_bf16_kernel<3> (round 3) alone on B frames, beside round 2's stem_pool_bf16_kernel<3> and the wave-specialised
// variant (experiments/stem_bf16_wave_specialised.h) on the same input: times the three and compares the pooled maps
// bit for bit (same fragments, same K order).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DSTEM_ABL=2] -o st stem_bf16_bench.hip && ./st 64 960 1280 512 256
// STEM_ABL removes phases of the round-3 kernels (STEMB_ABL_LOAD = 1, _K = 2, _TILE = 4, _POOL = 8, _STORE = 16; with
// the tile write removed the compiler drops the K loop as well).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../feature-point-cnn_amd/csrc/block_mfma.h"
#include "../../feature-point-cnn_amd/csrc/block_bf16.h"
#include "../../feature-point-cnn_amd/csrc/block_x3.h"
#include "../../feature-point-cnn_amd/csrc/stem_bf16.h"
#include "../stem_pool_bf16_round2.h"
#include "../stem_bf16_wave_specialised.h"
using namespace fpc;
#ifndef STEM_ABL
#define STEM_ABL 0
#endif
#ifndef STEM_CIN
#define STEM_CIN 3
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 960, W = argc > 3 ? atoi(argv[3]) : 1280;
  const int G = argc > 4 ? atoi(argv[4]) : 512;
  constexpr int CIN = STEM_CIN;
  float* in; uint4* wf; float* bias; unsigned short *out0, *out1, *out2;
  const size_t nin = (size_t)B * CIN * H * W, nout = (size_t)B * (H / 4) * (W / 4) * 64;
  CK(hipMalloc(&in, nin * 4)); CK(hipMalloc(&wf, 13 * 2 * 64 * 16)); CK(hipMalloc(&bias, 256));
  CK(hipMalloc(&out0, nout * 2)); CK(hipMalloc(&out1, nout * 2)); CK(hipMalloc(&out2, nout * 2));
  std::vector<float> h(nin); for (auto& x : h) x = (float)rand() / RAND_MAX - 0.3f;
  CK(hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice));
  std::vector<unsigned short> w(13 * 2 * 64 * 8);
  for (size_t i = 0; i < w.size(); ++i) w[i] = (i % 8 == 0) ? 0 : (unsigned short)(0x3c00 + (rand() & 0xff) + ((rand() & 1) << 15));  // j = 0: the zero-weight pad
  CK(hipMemcpy(wf, w.data(), w.size() * 2, hipMemcpyHostToDevice));
  std::vector<float> hb(64); for (auto& x : hb) x = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(bias, hb.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemset(out0, 0xff, nout * 2)); CK(hipMemset(out1, 0xee, nout * 2)); CK(hipMemset(out2, 0xdd, nout * 2));
  StemX3Args a{}; a.in = in; a.wfrag = wf; a.bias = bias; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2; a.Hp = H / 4; a.Wp = W / 4; a.frames = B;
  StemX3Args a0 = a, a1 = a;
  const int G3 = argc > 5 ? atoi(argv[5]) : 256;
  a0.out = (float*)out0; a0.tiles_x = (a.Wp + 7) / 8; a0.tiles_y = (a.Hp + 7) / 8;
  a1.out = (float*)out1; a1.tiles_x = (a.Wp + SB2_PW - 1) / SB2_PW; a1.tiles_y = (a.Hp + SB2_PH - 1) / SB2_PH;
  CK(hipFuncSetAttribute((const void*)stem_pool_bf16_kernel<CIN, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, StemBCfg<CIN>::LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)stem_bf16_kernel<CIN, STEM_ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, StemB2Cfg<CIN>::LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)stem_bf16_ws_kernel<CIN, STEM_ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, StemB3Cfg<CIN>::LDS_BYTES));
  StemX3Args a2 = a1; a2.out = (float*)out2;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best0 = 1e9, best1 = 1e9, best2 = 1e9;
  for (int rep = 0; rep < 6; ++rep) {
    float ms;
    hipEventRecord(e0);
    hipLaunchKernelGGL((stem_pool_bf16_kernel<CIN, 0>), dim3(G), dim3(STEMB_THREADS), StemBCfg<CIN>::LDS_BYTES, 0, a0);
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    hipEventElapsedTime(&ms, e0, e1); if (ms < best0) best0 = ms;
    hipEventRecord(e0);
    hipLaunchKernelGGL((stem_bf16_kernel<CIN, STEM_ABL>), dim3(G), dim3(SB2_THREADS), StemB2Cfg<CIN>::LDS_BYTES, 0, a1);
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    hipEventElapsedTime(&ms, e0, e1); if (ms < best1) best1 = ms;
    hipEventRecord(e0);
    hipLaunchKernelGGL((stem_bf16_ws_kernel<CIN, STEM_ABL>), dim3(G3), dim3(SB3_THREADS), StemB3Cfg<CIN>::LDS_BYTES, 0, a2);
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    hipEventElapsedTime(&ms, e0, e1); if (ms < best2) best2 = ms;
  }
#ifdef FPC_DIAG
  {   // phase times of every workgroup's tenth tile (wave 0's s_memtime stamps), two workgroups per CU at work
    unsigned long long* st;
    CK(hipMalloc(&st, (size_t)G * 8 * 8)); CK(hipMemset(st, 0, (size_t)G * 8 * 8));
    StemX3Args as = a1; as.stamps = st;
    hipLaunchKernelGGL((stem_bf16_kernel<CIN, STEM_ABL>), dim3(G), dim3(SB2_THREADS), StemB2Cfg<CIN>::LDS_BYTES, 0, as);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs((size_t)G * 8);
    CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
    const char* names[6] = {"wait for the window + convert + stage", "barrier", "request + acc init + K loop", "tile write", "barrier", "pool + store"};
    for (int ph = 0; ph < 6; ++ph) {
      std::vector<double> d;
      for (int g = 0; g < G; ++g) if (hs[(size_t)g * 8] && hs[(size_t)g * 8 + 6]) d.push_back((double)(hs[(size_t)g * 8 + ph + 1] - hs[(size_t)g * 8 + ph]));
      if (d.empty()) continue;
      std::sort(d.begin(), d.end());
      printf("  %-40s median %7.0f  p90 %7.0f cycles (%zu workgroups)\n", names[ph], d[d.size() / 2], d[d.size() * 9 / 10], d.size());
    }
  }
#endif
  printf("B=%d %dx%d cin=%d G=%d/%d: round 2 %.3f ms (%d tiles)  stem_bf16 %.3f ms  stem_bf16_ws %.3f ms (%d tiles, abl %d)\n", B, H, W, CIN, G, G3, best0,
         a0.tiles_x * a0.tiles_y * B, best1, best2, a1.tiles_x * a1.tiles_y * B, STEM_ABL);
  if (STEM_ABL == 0) {
    std::vector<unsigned short> o0(nout), o1(nout), o2(nout);
    CK(hipMemcpy(o0.data(), out0, nout * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), out1, nout * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o2.data(), out2, nout * 2, hipMemcpyDeviceToHost));
    size_t bad = 0, bad2 = 0, nz = 0, first = (size_t)-1;
    for (size_t i = 0; i < nout; ++i) { if (o0[i] != o1[i]) { if (!bad) first = i; ++bad; } if (o0[i] != o2[i]) { if (!bad && !bad2) { first = i; o1[i] = o2[i]; } ++bad2; } if (o0[i]) ++nz; }
    printf("compare: stem_bf16 %zu, stem_bf16_ws %zu of %zu differ (%zu non-zero)\n", bad, bad2, nout, nz);
    bad += bad2;
    if (bad) {
      const size_t px = first / 64; const int Wp = W / 4, Hp = H / 4;
      printf("first: frame %zu row %zu col %zu ch %zu old %04x new %04x\n", px / ((size_t)Hp * Wp), (px / Wp) % Hp, px % Wp, first % 64, o0[first], o1[first]);
      return 1;
    }
  }
  return 0;
}
