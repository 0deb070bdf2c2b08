#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <chrono>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void gather(const float* __restrict__ host_rows, const uint32_t* __restrict__ idx, uint64_t n, float* __restrict__ out) {
  const int lig = threadIdx.x & 15;
  const uint64_t groups = (uint64_t)gridDim.x * (blockDim.x / 16);
  for (uint64_t j = (uint64_t)blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4); j < n; j += groups) {
    const float* src = host_rows + (uint64_t)idx[j] * 128;
    float* dst = out + j * 128;
    f4 a = *reinterpret_cast<const f4*>(src + lig * 4);
    f4 b = *reinterpret_cast<const f4*>(src + 64 + lig * 4);
    *reinterpret_cast<f4*>(dst + lig * 4) = a;
    *reinterpret_cast<f4*>(dst + 64 + lig * 4) = b;
  }
}
int main() {
  const uint64_t rows = 16ull << 20;  // 16M rows x 512 B = 8 GB
  float* h = nullptr;
  auto t0 = std::chrono::steady_clock::now();
  if (hipHostMalloc((void**)&h, rows * 512, hipHostMallocDefault) != hipSuccess) { printf("hostmalloc failed\n"); return 1; }
  printf("hipHostMalloc 8GB: %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  for (uint64_t i = 0; i < rows * 128; i += 1024) h[i] = (float)i;
  for (uint64_t n : {8192ull, 85000ull, 400000ull}) {
    std::vector<uint32_t> hi(n);
    uint64_t x = 88172645463325252ull;
    for (auto& v : hi) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x % rows); }
    uint32_t* di; float* dout;
    hipMalloc((void**)&di, n * 4); hipMalloc((void**)&dout, n * 512);
    hipMemcpy(di, hi.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 1024, 2048}) {
      gather<<<blocks, 256>>>(h, di, n, dout);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int it = 0; it < 5; ++it) gather<<<blocks, 256>>>(h, di, n, dout);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      printf("n=%lu blocks=%d: %.3f ms  %.1f GB/s\n", (unsigned long)n, blocks, ms, n * 512 / ms / 1e6);
    }
    // memcpy baseline
    float* hp; hipHostMalloc((void**)&hp, n * 512, hipHostMallocDefault);
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) hipMemcpyAsync(dout, hp, n * 512, hipMemcpyHostToDevice, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("n=%lu hipMemcpyAsync H2D contiguous: %.3f ms  %.1f GB/s\n", (unsigned long)n, ms, n * 512 / ms / 1e6);
    hipFree(di); hipFree(dout); hipHostFree(hp);
  }
  return 0;
}
