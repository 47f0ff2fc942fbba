// What does rocprofv3's FETCH_SIZE report for the access shapes of the Winograd kernels' halo requests?
//   hipcc -O3 --offload-arch=gfx950 -o cal fetch_size_calibration.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o p -- ./cal
// Kernels (each reads exactly the stated bytes from a 1 GiB buffer, far larger than the 256 MiB Infinity Cache):
//   stream16      every lane 16 consecutive bytes, fully coalesced (the guide's case: FETCH_SIZE = bytes / 2)
//   seg64_once    4 lanes x 16 B = one 64-byte segment per 512-byte pixel, each pixel touched ONCE (half of every 128-byte line is never used)
//   seg64_pairs   the same, then the OTHER 64 bytes of the same lines right afterwards by the same workgroup (the chunk c / chunk c + 1 pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void stream16(const f32x4* p, size_t n, float* out) {
  f32x4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s.x == 12345.f) out[0] = s.y + s.z + s.w;
}
// pixel = 512 bytes = 32 float4; segment k (0..7) = float4 4k .. 4k+3
__global__ void seg64(const f32x4* p, size_t npix, int k0, int nk, float* out) {
  f32x4 s = {0, 0, 0, 0};
  const size_t lane4 = threadIdx.x & 3;
  for (size_t px = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; px < npix; px += ((size_t)gridDim.x * blockDim.x) >> 2)
    for (int k = k0; k < k0 + nk; ++k) s += p[px * 32 + 4 * k + lane4];
  if (s.x == 12345.f) out[0] = s.y + s.z + s.w;
}
int main() {
  const size_t bytes = 1ull << 30;
  f32x4* p; float* out;
  CK(hipMalloc(&p, bytes)); CK(hipMalloc(&out, 16)); CK(hipMemset(p, 0, bytes));
  const size_t n = bytes / 16, npix = bytes / 512;
  hipLaunchKernelGGL(stream16, dim3(2048), dim3(256), 0, 0, p, n, out);           // 1 GiB
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(seg64, dim3(2048), dim3(256), 0, 0, p, npix, 0, 1, out);      // 128 MiB requested, one segment per pixel
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(seg64, dim3(2048), dim3(256), 0, 0, p, npix, 2, 2, out);      // 256 MiB requested: segments 2 and 3 = one whole 128-byte line per pixel, back to back
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(seg64, dim3(2048), dim3(256), 0, 0, p, npix, 0, 8, out);      // 1 GiB requested: all eight segments of every pixel
  CK(hipDeviceSynchronize());
  printf("requested bytes: stream16 %zu, seg64 x1 %zu, seg64 x2 (one line) %zu, seg64 x8 %zu\n", bytes, npix * 64, npix * 128, npix * 512);
  return 0;
}
