// One wave per SIMD: what does a GAP of non-MFMA instructions between two blocks of eight v_mfma_f32_16x16x4_f32 cost?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gap mfma_f32_gap.hip && ./gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define MF8 "v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n" \
            "v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_mfma_f32_16x16x4_f32 %1, %2, %3, %1"

// MODE bits: 1 = two buffer loads (1 KiB each, L2-resident), 2 = one ds_read_b128, 4 = one ds_write_b32, 8 = two s_add, 16 = s_waitcnt on everything but the newest,
//            32 = four v_fma (a VALU burst), 64 = ONE buffer load, 128 = global_load_dwordx4 x2 instead of buffer
template <int MODE>
__global__ __launch_bounds__(256, 1) void bench(const float* w, float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int tid = threadIdx.x;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = 1.0f + tid * 1e-6f, b = 1.0f - tid * 1e-6f;
  lds[tid] = (float)tid; lds[tid + 256] = 1.f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 1 << 24, 0x00020000);
  unsigned voff = (unsigned)(tid & 63) * 16u + (unsigned)(tid >> 6) * 262144u;
  unsigned lofs = (unsigned)tid * 16u;
  f32x4 ld[4] = {f32x4{0,0,0,0}, f32x4{0,0,0,0}, f32x4{0,0,0,0}, f32x4{0,0,0,0}};
  f32x4 lr = f32x4{0,0,0,0};
  float x[4] = {1.f, 2.f, 3.f, 4.f};
  int soff = 0, sdummy = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (MODE & 8) { asm volatile("s_add_i32 %0, %0, 0x400" : "+s"(soff)); asm volatile("s_add_i32 %0, %0, 0x400" : "+s"(sdummy)); }
      if (MODE & 1) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ld[(2 * m) & 3]) : "v"(voff), "s"(rs), "s"(soff & 0xffff));
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=v"(ld[(2 * m + 1) & 3]) : "v"(voff), "s"(rs), "s"(soff & 0xffff));
      }
      if (MODE & 64) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ld[m & 3]) : "v"(voff), "s"(rs), "s"(soff & 0xffff));
      if (MODE & 2) asm volatile("ds_read_b128 %0, %1" : "=v"(lr) : "v"(lofs));
      if (MODE & 4) asm volatile("ds_write_b32 %0, %1 offset:8192" : : "v"(lofs), "v"(x[0]));
      if (MODE & 32) { for (int k = 0; k < 4; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[k]) : "v"(a)); }
      if (MODE & 16) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(2)");
      asm volatile(MF8 : "+a"(acc[2 * m]), "+a"(acc[2 * m + 1]) : "v"(a), "v"(b));
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 4; ++i) s += ld[i][0] + ld[i][3] + x[i];
  s += lr[0];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const float* w, const char* name) {
  const int G = 256, iters = 500;
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, G * 256 * 4)); CK(hipMalloc(&cyc, G * 8));
  hipLaunchKernelGGL((bench<MODE>), dim3(G), dim3(256), 0, 0, w, out, cyc, 10);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((bench<MODE>), dim3(G), dim3(256), 0, 0, w, out, cyc, iters);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(G); CK(hipMemcpy(h.data(), cyc, G * 8, hipMemcpyDeviceToHost));
  double mean = 0; for (auto v : h) mean += v; mean /= G;
  printf("%-60s: %.1f cycles per block of 8 MFMAs (256 = back to back)\n", name, mean / (iters * 4.0));
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  float* w; CK(hipMalloc(&w, 1 << 24)); CK(hipMemset(w, 0, 1 << 24));
  run<0>(w, "nothing");
  run<8>(w, "2 s_add");
  run<2>(w, "1 ds_read_b128");
  run<6>(w, "1 ds_read_b128 + 1 ds_write_b32");
  run<64>(w, "1 buffer_load_dwordx4");
  run<1>(w, "2 buffer_load_dwordx4");
  run<9>(w, "2 s_add + 2 buffer_load");
  run<11>(w, "2 s_add + 2 buffer_load + ds_read_b128");
  run<15>(w, "2 s_add + 2 buffer_load + ds_read_b128 + ds_write_b32");
  run<31>(w, "... + s_waitcnt");
  run<32>(w, "4 v_fma");
  run<16>(w, "s_waitcnt only");
  run<18>(w, "ds_read_b128 + s_waitcnt");
  return 0;
}
