#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../feature-point-cnn_amd/csrc/block_mfma.h"
#include "../../feature-point-cnn_amd/csrc/block_bf16.h"
#include "../../feature-point-cnn_amd/csrc/block_x3.h"
using namespace fpc;
#ifndef STEM_ABL
#define STEM_ABL 0   // -DSTEM_ABL=8 (STEMB_ABL_POOL) etc.: one phase removed, see block_x3.h
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 720, W = argc > 3 ? atoi(argv[3]) : 1280;
  const int G = argc > 4 ? atoi(argv[4]) : 512;
  float* in; uint4* wf; float* bias; unsigned short* out;
  const size_t nin = (size_t)B * 3 * H * W;
  CK(hipMalloc(&in, nin * 4)); CK(hipMalloc(&wf, 13 * 2 * 64 * 16)); CK(hipMalloc(&bias, 256)); CK(hipMalloc(&out, (size_t)B * (H / 4) * (W / 4) * 64 * 2));
  std::vector<float> h(nin); for (auto& x : h) x = (float)rand() / RAND_MAX;
  CK(hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice));
  std::vector<unsigned short> w(13 * 2 * 64 * 8); for (auto& x : w) x = 0x3c00 + (rand() & 0xff);
  CK(hipMemcpy(wf, w.data(), w.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, 256));
  StemX3Args a{}; a.in = in; a.wfrag = wf; a.bias = bias; a.out = (float*)out; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2; a.Hp = H / 4; a.Wp = W / 4;
  a.tiles_x = (a.Wp + 7) / 8; a.tiles_y = (a.Hp + 7) / 8; a.frames = B;
  CK(hipFuncSetAttribute((const void*)stem_pool_bf16_kernel<3, STEM_ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, StemBCfg<3>::LDS_BYTES));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stem_pool_bf16_kernel<3, STEM_ABL>), dim3(G), dim3(STEMB_THREADS), StemBCfg<3>::LDS_BYTES, 0, a);
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("%s B=%d %dx%d G=%d: %.3f ms (%d tiles)\n", argv[0], B, H, W, G, best, a.tiles_x * a.tiles_y * B);
  return 0;
}
