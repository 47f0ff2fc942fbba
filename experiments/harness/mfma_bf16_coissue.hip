// The same question as mfma_f32_coissue.hip for v_mfma_f32_32x32x16_bf16 (32 cycles on the SIMD's matrix pipe): how many
// non-MFMA instructions fit beside one?  (round 5: whether a VALU diet can pay in the bf16 kernels)
// One or two waves per SIMD; fillers: independent v_fma_f32, ds_read_b32, ds_write_b32, ds_read_b128.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o coissue mfma_bf16_coissue.hip && ./coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

enum { F_VALU = 0, F_LDSR = 1, F_LDSW = 2, F_LDSR128 = 3 };

template <int K, int KIND, int NACC>
__global__ __launch_bounds__(512) void bench(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x;
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x4 a = {1.0f + tid * 1e-6f, 2.f, 3.f, 4.f}, b = {1.0f - tid * 1e-6f, .5f, .25f, .125f};   // (8 bf16 each, as bits)
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = tid * 0.001f + i;
  f32x4 r4[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  const float c1 = 0.999f, c2 = 0.001f;
  lds[tid] = (float)tid;
  lds[tid + 512] = 1.f;
  __syncthreads();
  unsigned lofs = (unsigned)tid * 4u;
  unsigned lofs16 = (unsigned)(tid & 255) * 16u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NACC; ++m) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (KIND == F_VALU) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x[(m * K + k) & 7]) : "v"(c1), "v"(c2));
        else if (KIND == F_LDSR) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x[(m * K + k) & 7]) : "v"(lofs), "n"(((0) & 7) * 2048));
        else if (KIND == F_LDSW) asm volatile("ds_write_b32 %0, %1 offset:8192" : : "v"(lofs), "v"(x[(m * K + k) & 7]));
        else asm volatile("ds_read_b128 %0, %1" : "=v"(r4[(m * K + k) & 1]) : "v"(lofs16));
      }
    }
    if (KIND == F_LDSR || KIND == F_LDSR128 || KIND == F_LDSW) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += x[i];
  s += r4[0][0] + r4[1][1];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int KIND, int NACC>
static void run(const char* name, int threads) {
  const int G = 256, iters = 2000;
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, G * 512 * 4)); CK(hipMalloc(&cyc, G * 8));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((bench<K, KIND, NACC>), dim3(G), dim3(threads), 0, 0, out, cyc, 10);
  CK(hipDeviceSynchronize());
  hipEventRecord(e0);
  hipLaunchKernelGGL((bench<K, KIND, NACC>), dim3(G), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(G); CK(hipMemcpy(h.data(), cyc, G * 8, hipMemcpyDeviceToHost));
  double mean = 0; for (auto v : h) mean += v; mean /= G;
  const int wps = threads / 256;   // waves per SIMD
  // s_memtime ticks at 100 MHz on gfx950? report both: ticks per (MFMA per SIMD) and wall ns per MFMA-per-SIMD
  const double mfma_per_simd = (double)iters * NACC * wps;
  printf("%-10s K=%d acc=%2d waves/SIMD=%d: %.3f ms, %.2f ns per MFMA per SIMD, memtime ticks/MFMA %.2f\n", name, K, NACC, wps, ms,
         ms * 1e6 / mfma_per_simd, mean / mfma_per_simd);
  hipFree(out); hipFree(cyc);
}

int main() {
#define ROW(KIND, name) \
  run<0, KIND, 4>(name, 256); run<1, KIND, 4>(name, 256); run<2, KIND, 4>(name, 256); run<3, KIND, 4>(name, 256); run<4, KIND, 4>(name, 256); run<6, KIND, 4>(name, 256); run<8, KIND, 4>(name, 256); \
  run<0, KIND, 4>(name, 512); run<1, KIND, 4>(name, 512); run<2, KIND, 4>(name, 512); run<3, KIND, 4>(name, 512); run<4, KIND, 4>(name, 512); run<6, KIND, 4>(name, 512); run<8, KIND, 4>(name, 512);
  ROW(F_VALU, "valu")
  ROW(F_LDSR, "ds_read32")
  ROW(F_LDSW, "ds_write32")
  ROW(F_LDSR128, "ds_read128")
  // dependent chains: 4 accumulators only (each MFMA depends on the one 4 back), and 2
  run<0, F_VALU, 4>("valu", 256); run<2, F_VALU, 4>("valu", 256); run<0, F_VALU, 2>("valu", 256); run<2, F_VALU, 2>("valu", 256);
  run<0, F_VALU, 2>("valu", 512); run<2, F_VALU, 2>("valu", 512); run<0, F_VALU, 1>("valu", 256); run<0, F_VALU, 1>("valu", 512);
  return 0;
}
