// How many PCIe reads in flight saturate the link, and what do they cost a concurrent HBM-bound kernel?
//   gather<W>: 16-lane group per 512-B row read out of pinned host memory; `blocks` x 256 threads
//   stream   : HBM copy (the stand-in for the probe+gather kernel), 2 GB per launch, runs on a second stream
// For each grid size: gather alone, stream alone, both together.
// Build: hipcc --offload-arch=gfx950 -O3 -o pcie_contention.bin pcie_contention.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gather(const float* __restrict__ host_rows, const uint32_t* __restrict__ idx,
                                              uint64_t n, float* __restrict__ out) {
  const int lig = threadIdx.x & 15;
  const uint64_t groups = (uint64_t)gridDim.x * 16;
  for (uint64_t j = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4); j < n; j += groups) {
    const float* src = host_rows + (uint64_t)idx[j] * 128;
    float* dst = out + j * 128;
    f4 a = *reinterpret_cast<const f4*>(src + lig * 4);
    f4 b = *reinterpret_cast<const f4*>(src + 64 + lig * 4);
    *reinterpret_cast<f4*>(dst + lig * 4) = a;
    *reinterpret_cast<f4*>(dst + 64 + lig * 4) = b;
  }
}

// one wave per row: 64 lanes x 8 B = 512 B in a single wave-level request
__global__ __launch_bounds__(256) void gather_wave(const float* __restrict__ host_rows, const uint32_t* __restrict__ idx,
                                                   uint64_t n, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const uint64_t waves = (uint64_t)gridDim.x * 4;
  for (uint64_t j = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); j < n; j += waves) {
    const float2* src = reinterpret_cast<const float2*>(host_rows + (uint64_t)idx[j] * 128);
    reinterpret_cast<float2*>(out + j * 128)[lane] = src[lane];
  }
}

__global__ __launch_bounds__(256) void stream(const f4* __restrict__ a, f4* __restrict__ b, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(a[i], &b[i]);
}

static float elapsed(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
  const uint64_t rows = 16ull << 20;  // 8 GB of pinned rows
  float* h = nullptr;
  if (hipHostMalloc((void**)&h, rows * 512, hipHostMallocDefault) != hipSuccess) { printf("hostmalloc failed\n"); return 1; }
  for (uint64_t i = 0; i < rows * 128; i += 1024) h[i] = (float)i;
  const uint64_t n = 82000;  // unique misses of one BASELINE config-2 batch
  std::vector<uint32_t> hi(n);
  uint64_t x = 88172645463325252ull;
  for (auto& v : hi) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x % rows); }
  uint32_t* di; float* dout;
  hipMalloc((void**)&di, n * 4); hipMalloc((void**)&dout, n * 512);
  hipMemcpy(di, hi.data(), n * 4, hipMemcpyHostToDevice);
  const uint64_t sn = (1ull << 30) / 16;  // 1 GB read + 1 GB written per launch
  f4 *sa, *sb;
  hipMalloc((void**)&sa, sn * 16); hipMalloc((void**)&sb, sn * 16);
  hipMemset(sa, 1, sn * 16);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t g0, g1, c0, c1; hipEventCreate(&g0); hipEventCreate(&g1); hipEventCreate(&c0); hipEventCreate(&c1);
  // stream alone
  stream<<<2048, 256, 0, s2>>>(sa, sb, sn); hipDeviceSynchronize();
  hipEventRecord(c0, s2);
  for (int it = 0; it < 4; ++it) stream<<<2048, 256, 0, s2>>>(sa, sb, sn);
  hipEventRecord(c1, s2); hipEventSynchronize(c1);
  const float stream_alone = elapsed(c0, c1) / 4;
  printf("stream alone: %.3f ms  %.0f GB/s\n", stream_alone, 2.0 * sn * 16 / stream_alone / 1e6);
  for (int variant = 0; variant < 2; ++variant) {
    for (int blocks : {8, 16, 32, 64, 128, 256, 512, 2048}) {
      auto launch = [&](hipStream_t s) {
        if (variant == 0) gather<<<blocks, 256, 0, s>>>(h, di, n, dout);
        else gather_wave<<<blocks, 256, 0, s>>>(h, di, n, dout);
      };
      launch(s1); hipDeviceSynchronize();
      hipEventRecord(g0, s1);
      for (int it = 0; it < 3; ++it) launch(s1);
      hipEventRecord(g1, s1); hipEventSynchronize(g1);
      const float alone = elapsed(g0, g1) / 3;
      // together: 3 gathers on s1 while stream launches keep s2 busy for at least as long
      const int ns = (int)(alone * 3 / stream_alone) + 2;
      hipEventRecord(c0, s2);
      for (int it = 0; it < ns; ++it) stream<<<2048, 256, 0, s2>>>(sa, sb, sn);
      hipEventRecord(c1, s2);
      hipEventRecord(g0, s1);
      for (int it = 0; it < 3; ++it) launch(s1);
      hipEventRecord(g1, s1);
      hipDeviceSynchronize();
      const float both_g = elapsed(g0, g1) / 3, both_s = elapsed(c0, c1) / ns;
      printf("%s blocks=%4d rows_in_flight=%5d: alone %.3f ms %.1f GB/s | with stream %.3f ms %.1f GB/s, stream %.3f ms (x%.2f)\n",
             variant == 0 ? "group16" : "wave64 ", blocks, blocks * (variant == 0 ? 16 : 4), alone, n * 512 / alone / 1e6, both_g,
             n * 512 / both_g / 1e6, both_s, both_s / stream_alone);
    }
  }
  return 0;
}
