// Does it matter WHERE on the chip the PCIe-reading waves sit?  A gather of random 512-B rows out of pinned host memory next to
// an HBM-bound copy, with the gather's working waves (a) spread over all 8 XCDs, (b) confined to 1, 2 or 4 XCDs (the grid covers
// the chip; workgroups that find themselves on another XCD exit; rows are claimed from an atomic counter).
// Build: hipcc --offload-arch=gfx950 -O3 -o pcie_xcd.bin pcie_xcd.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

// xcd_mask: bit x set = workgroups on XCD x work.  Rows claimed 16 at a time per 16-lane group.
__global__ __launch_bounds__(256) void gather(const float* __restrict__ rows, const uint32_t* __restrict__ idx, uint32_t n,
                                              float* __restrict__ out, uint32_t xcd_mask, uint32_t* counter) {
  const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
  if (!((xcd_mask >> xcc) & 1u)) return;
  const int lig = threadIdx.x & 15;
  for (;;) {
    uint32_t j0 = 0;
    if (lig == 0) j0 = atomicAdd(counter, 4u);
    j0 = __shfl(j0, (threadIdx.x & 63) & ~15, 64);
    if (j0 >= n) break;
    f4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t j = j0 + u < n ? j0 + u : n - 1;
      const float* src = rows + (uint64_t)idx[j] * 128;
      a[u] = *reinterpret_cast<const f4*>(src + lig * 4);
      b[u] = *reinterpret_cast<const f4*>(src + 64 + lig * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (j0 + u >= n) break;
      float* dst = out + (uint64_t)(j0 + u) * 128;
      *reinterpret_cast<f4*>(dst + lig * 4) = a[u];
      *reinterpret_cast<f4*>(dst + 64 + lig * 4) = b[u];
    }
  }
}

__global__ __launch_bounds__(256) void stream_static(const f4* __restrict__ a, f4* __restrict__ b, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(a[i], &b[i]);
}

static float elapsed(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
  const uint64_t rows = 16ull << 20;  // 8 GB
  float* h = nullptr;
  if (hipHostMalloc((void**)&h, rows * 512, hipHostMallocDefault) != hipSuccess) { printf("hostmalloc failed\n"); return 1; }
  for (uint64_t i = 0; i < rows * 128; i += 1024) h[i] = (float)i;
  const uint32_t n = 87000;
  std::vector<uint32_t> hi(n);
  uint64_t x = 88172645463325252ull;
  for (auto& v : hi) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x % rows); }
  uint32_t* di; float* dout; uint32_t* dctr;
  hipMalloc((void**)&di, n * 4); hipMalloc((void**)&dout, (uint64_t)n * 512); hipMalloc((void**)&dctr, 256);
  hipMemcpy(di, hi.data(), n * 4, hipMemcpyHostToDevice);
  const uint64_t sn = (1ull << 30) / 16;
  f4 *sa, *sb;
  hipMalloc((void**)&sa, sn * 16); hipMalloc((void**)&sb, sn * 16);
  hipMemset(sa, 1, sn * 16);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t g0, g1, c0, c1; hipEventCreate(&g0); hipEventCreate(&g1); hipEventCreate(&c0); hipEventCreate(&c1);
  stream_static<<<2048, 256, 0, s2>>>(sa, sb, sn); hipDeviceSynchronize();
  hipEventRecord(c0, s2);
  for (int it = 0; it < 4; ++it) stream_static<<<2048, 256, 0, s2>>>(sa, sb, sn);
  hipEventRecord(c1, s2); hipEventSynchronize(c1);
  const float stream_alone = elapsed(c0, c1) / 4;
  printf("HBM copy alone: %.3f ms  %.0f GB/s\n", stream_alone, 2.0 * sn * 16 / stream_alone / 1e6);
  struct Cfg { const char* name; uint32_t mask; int grid; };
  const Cfg cfgs[] = {
      {"all 8 XCDs, 128 workgroups", 0xFF, 128},  {"all 8 XCDs, 256 workgroups", 0xFF, 256},
      {"XCD 0 only, 128 working of 1024", 0x01, 1024}, {"XCD 0 only, 256 working of 2048", 0x01, 2048},
      {"XCDs 0-1, 128 working of 512", 0x03, 512},   {"XCDs 0-3, 128 working of 256", 0x0F, 256},
      {"XCD 7 only, 128 working of 1024", 0x80, 1024},
  };
  for (const Cfg& c : cfgs) {
    auto launch = [&]() { hipMemsetAsync(dctr, 0, 4, s1); gather<<<c.grid, 256, 0, s1>>>(h, di, n, dout, c.mask, dctr); };
    launch(); hipDeviceSynchronize();
    hipEventRecord(g0, s1);
    for (int it = 0; it < 3; ++it) launch();
    hipEventRecord(g1, s1); hipEventSynchronize(g1);
    const float alone = elapsed(g0, g1) / 3;
    const int ns = (int)(alone * 3 / stream_alone) + 2;
    hipEventRecord(c0, s2);
    for (int it = 0; it < ns; ++it) stream_static<<<2048, 256, 0, s2>>>(sa, sb, sn);
    hipEventRecord(c1, s2);
    hipEventRecord(g0, s1);
    for (int it = 0; it < 3; ++it) launch();
    hipEventRecord(g1, s1);
    hipDeviceSynchronize();
    const float both_g = elapsed(g0, g1) / 3, both_s = elapsed(c0, c1) / ns;
    printf("%-36s alone %.3f ms %5.1f GB/s | next to the HBM copy: gather %.3f ms %5.1f GB/s, copy x%.2f\n", c.name, alone,
           (double)n * 512 / alone / 1e6, both_g, (double)n * 512 / both_g / 1e6, both_s / stream_alone);
  }
  return 0;
}
