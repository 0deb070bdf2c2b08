// Why does the hit-gather kernel take 211-217 us on some boxes of the pool and 224-238 us on others, with the same device-to-device copy
// rate (profiles/round5/box_lottery_gather_kernel.txt)?  Random 512-B row reads over footprints of different size, with and without the
// gather's streaming row writes: if only the LARGE footprints are slow on a slow box, the difference is address translation (page-table
// fragment size / TLB reach), not HBM.
//   hipcc --offload-arch=gfx950 -O3 -o random_rows.bin tools/micro/random_rows.hip && ./random_rows.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// one 16-lane group per row, eight rows in flight per group (the gather kernel's shape); rows = footprint / 512 B
template <bool WRITE>
__global__ __launch_bounds__(256) void rows_kernel(const float* __restrict__ src, uint64_t rows, uint64_t n, uint64_t seed, float* __restrict__ out, float* __restrict__ sink) {
  const int lig = threadIdx.x & 15;
  const uint64_t groups = (uint64_t)gridDim.x * 16;
  f4 acc = {0, 0, 0, 0};
  for (uint64_t i0 = ((uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4)) * 8; i0 < n; i0 += groups * 8) {
    f4 v[8][2];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint64_t i = i0 + r < n ? i0 + r : n - 1;
      const float* s = src + (mix(seed + i) % rows) * 128;
      v[r][0] = *reinterpret_cast<const f4*>(s + lig * 4);
      v[r][1] = *reinterpret_cast<const f4*>(s + 64 + lig * 4);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (WRITE) {
        const uint64_t i = i0 + r < n ? i0 + r : n - 1;
        __builtin_nontemporal_store(v[r][0], reinterpret_cast<f4*>(out + i * 128 + lig * 4));
        __builtin_nontemporal_store(v[r][1], reinterpret_cast<f4*>(out + i * 128 + 64 + lig * 4));
      } else {
        acc += v[r][0] + v[r][1];
      }
    }
  }
  if (!WRITE && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = 1.f;
}

int main() {
  const uint64_t n = 1703936;                       // rows per "call"
  float *out, *sink;
  CK(hipMalloc(&out, n * 512));
  CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double gbs[] = {0.25, 1, 4, 16, 27, 64};
  printf("footprint   read-only us (GB/s)      read+write us (GB/s of 2 x 872 MB)\n");
  for (double gb : gbs) {
    const uint64_t rows = (uint64_t)(gb * (1ull << 30) / 512);
    float* src = nullptr;
    if (hipMalloc(&src, rows * 512) != hipSuccess) { printf("%5.2f GB: allocation failed\n", gb); (void)hipGetLastError(); continue; }
    CK(hipMemset(src, 0, rows * 512));
    double res[2][2];
    for (int w = 0; w < 2; ++w) {
      std::vector<float> ms;
      for (int it = 0; it < 14; ++it) {
        CK(hipEventRecord(e0));
        if (w) hipLaunchKernelGGL(rows_kernel<true>, dim3(6656), dim3(256), 0, 0, src, rows, n, (uint64_t)it * 7919, out, sink);
        else hipLaunchKernelGGL(rows_kernel<false>, dim3(6656), dim3(256), 0, 0, src, rows, n, (uint64_t)it * 7919, out, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); if (it >= 4) ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      res[w][0] = ms[ms.size() / 2] * 1e3;
      res[w][1] = (w ? 2.0 : 1.0) * n * 512 / (ms[ms.size() / 2] * 1e-3) / 1e9;
    }
    printf("%6.2f GB   %7.1f (%5.0f)            %7.1f (%5.0f)\n", gb, res[0][0], res[0][1], res[1][0], res[1][1]);
    CK(hipFree(src));
  }
  // the same 27 GB as 26 separate allocations (the cache: one row array per table)
  {
    const int T = 26; const uint64_t rows_t = (uint64_t)(27.0 / T * (1ull << 30) / 512);
    std::vector<float*> parts(T, nullptr);
    bool ok = true;
    for (int t = 0; t < T && ok; ++t) ok = hipMalloc(&parts[t], rows_t * 512) == hipSuccess;
    if (ok) {
      std::vector<float> ms;
      for (int it = 0; it < 12; ++it) {
        CK(hipEventRecord(e0));
        for (int t = 0; t < T; ++t)
          hipLaunchKernelGGL(rows_kernel<true>, dim3(256), dim3(256), 0, 0, parts[t], rows_t, n / T, (uint64_t)(it * 31 + t), out + (uint64_t)t * (n / T) * 128, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t2; CK(hipEventElapsedTime(&t2, e0, e1)); if (it >= 2) ms.push_back(t2);
      }
      std::sort(ms.begin(), ms.end());
      printf("27 GB in 26 allocations, 26 launches: read+write %7.1f us\n", ms[ms.size() / 2] * 1e3);
    }
    for (auto p : parts) if (p) (void)hipFree(p);
  }
  return 0;
}
