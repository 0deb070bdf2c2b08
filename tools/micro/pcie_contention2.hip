// Follow-up to pcie_contention.hip: WHY does a small PCIe-reading kernel slow an HBM-bound neighbour by 1.3-1.7x
// whatever its size?  Hypothesis: the damage is per CU (the CUs that host PCIe-reading waves serve their other waves
// late), and a statically partitioned neighbour finishes with its slowest CU.  Tests:
//   stream_static : grid-stride copy (equal share per block)            -> duration = slowest CU
//   stream_dynamic: blocks grab 256-KB tiles from an atomic counter     -> duration = average throughput
//   gather from pinned host memory vs the same gather from HBM (control), 256- and 1024-thread blocks
// Build: hipcc --offload-arch=gfx950 -O3 -o pcie_contention2.bin pcie_contention2.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void gather(const float* __restrict__ rows, const uint32_t* __restrict__ idx, uint64_t n, float* __restrict__ out) {
  const int lig = threadIdx.x & 15;
  const uint64_t gpb = blockDim.x / 16;
  const uint64_t groups = (uint64_t)gridDim.x * gpb;
  for (uint64_t j = (uint64_t)blockIdx.x * gpb + (threadIdx.x >> 4); j < n; j += groups) {
    const float* src = rows + (uint64_t)idx[j] * 128;
    float* dst = out + j * 128;
    f4 a = *reinterpret_cast<const f4*>(src + lig * 4);
    f4 b = *reinterpret_cast<const f4*>(src + 64 + lig * 4);
    *reinterpret_cast<f4*>(dst + lig * 4) = a;
    *reinterpret_cast<f4*>(dst + 64 + lig * 4) = b;
  }
}

__global__ __launch_bounds__(256) void stream_static(const f4* __restrict__ a, f4* __restrict__ b, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(a[i], &b[i]);
}

// tile = 16384 f4 = 256 KB; one atomic per tile per block
__global__ __launch_bounds__(256) void stream_dynamic(const f4* __restrict__ a, f4* __restrict__ b, uint64_t n, uint32_t* counter) {
  __shared__ uint32_t tile_s;
  const uint64_t tiles = (n + 16383) / 16384;
  for (;;) {
    if (threadIdx.x == 0) tile_s = atomicAdd(counter, 1u);
    __syncthreads();
    const uint64_t tile = tile_s;
    __syncthreads();
    if (tile >= tiles) break;
    const uint64_t lo = tile * 16384, hi = lo + 16384 < n ? lo + 16384 : n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) __builtin_nontemporal_store(a[i], &b[i]);
  }
}

static float elapsed(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
  const uint64_t rows = 16ull << 20;  // 8 GB
  float* h = nullptr; float* dtab = nullptr;
  if (hipHostMalloc((void**)&h, rows * 512, hipHostMallocDefault) != hipSuccess) { printf("hostmalloc failed\n"); return 1; }
  for (uint64_t i = 0; i < rows * 128; i += 1024) h[i] = (float)i;
  hipMalloc((void**)&dtab, rows * 512);
  hipMemset(dtab, 0, rows * 512);
  const uint64_t n = 82000;
  std::vector<uint32_t> hi(n);
  uint64_t x = 88172645463325252ull;
  for (auto& v : hi) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x % rows); }
  uint32_t* di; float* dout; uint32_t* dctr;
  hipMalloc((void**)&di, n * 4); hipMalloc((void**)&dout, n * 512); hipMalloc((void**)&dctr, 64 * 4);
  hipMemcpy(di, hi.data(), n * 4, hipMemcpyHostToDevice);
  const uint64_t sn = (1ull << 30) / 16;
  f4 *sa, *sb;
  hipMalloc((void**)&sa, sn * 16); hipMalloc((void**)&sb, sn * 16);
  hipMemset(sa, 1, sn * 16);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t g0, g1, c0, c1; hipEventCreate(&g0); hipEventCreate(&g1); hipEventCreate(&c0); hipEventCreate(&c1);
  float alone_ms[2];
  for (int dyn = 0; dyn < 2; ++dyn) {
    auto sl = [&]() {
      if (dyn) { hipMemsetAsync(dctr, 0, 4, s2); stream_dynamic<<<2048, 256, 0, s2>>>(sa, sb, sn, dctr); }
      else stream_static<<<2048, 256, 0, s2>>>(sa, sb, sn);
    };
    sl(); hipDeviceSynchronize();
    hipEventRecord(c0, s2);
    for (int it = 0; it < 4; ++it) sl();
    hipEventRecord(c1, s2); hipEventSynchronize(c1);
    alone_ms[dyn] = elapsed(c0, c1) / 4;
    printf("stream %s alone: %.3f ms  %.0f GB/s\n", dyn ? "dynamic" : "static ", alone_ms[dyn], 2.0 * sn * 16 / alone_ms[dyn] / 1e6);
  }
  for (int src = 0; src < 2; ++src) {            // 0 = pinned host rows, 1 = HBM rows (control)
    for (int threads : {256, 1024}) {
      for (int blocks : {8, 32, 128, 512}) {
        if (threads == 1024 && blocks > 128) continue;
        const float* tab = src == 0 ? h : dtab;
        auto launch = [&](hipStream_t s) { gather<<<blocks, threads, 0, s>>>(tab, di, n, dout); };
        launch(s1); hipDeviceSynchronize();
        hipEventRecord(g0, s1);
        for (int it = 0; it < 3; ++it) launch(s1);
        hipEventRecord(g1, s1); hipEventSynchronize(g1);
        const float alone = elapsed(g0, g1) / 3;
        printf("%s thr=%4d blocks=%4d rows_in_flight=%5d: alone %.3f ms %5.1f GB/s |", src == 0 ? "host" : "hbm ", threads, blocks,
               blocks * threads / 16, alone, n * 512 / alone / 1e6);
        for (int dyn = 0; dyn < 2; ++dyn) {
          const int ns = (int)(alone * 3 / alone_ms[dyn]) + 2;
          hipEventRecord(c0, s2);
          for (int it = 0; it < ns; ++it) {
            if (dyn) { hipMemsetAsync(dctr, 0, 4, s2); stream_dynamic<<<2048, 256, 0, s2>>>(sa, sb, sn, dctr); }
            else stream_static<<<2048, 256, 0, s2>>>(sa, sb, sn);
          }
          hipEventRecord(c1, s2);
          hipEventRecord(g0, s1);
          for (int it = 0; it < 3; ++it) launch(s1);
          hipEventRecord(g1, s1);
          hipDeviceSynchronize();
          const float both_g = elapsed(g0, g1) / 3, both_s = elapsed(c0, c1) / ns;
          printf(" with %s stream: gather %.3f ms %5.1f GB/s, stream x%.2f |", dyn ? "dynamic" : "static", both_g, n * 512 / both_g / 1e6,
                 both_s / alone_ms[dyn]);
        }
        printf("\n");
      }
    }
  }
  return 0;
}
