// Does the NUMA node of the pinned host table matter for the device-driven fetch?  Random 512-B row gather from
// pinned memory placed (a) where hipHostMalloc puts it by default, (b) on node 0, (c) on node 1
// (hipHostMallocNumaUser + set_mempolicy(MPOL_BIND)).  Prints the node the pages really landed on.
// Build: hipcc --offload-arch=gfx950 -O3 -o pcie_numa.bin pcie_numa.hip
#include <hip/hip_runtime.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void gather(const float* __restrict__ rows, const uint32_t* __restrict__ idx, uint64_t n, float* __restrict__ out) {
  const int lig = threadIdx.x & 15;
  const uint64_t groups = (uint64_t)gridDim.x * 16;
  for (uint64_t j = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4); j < n; j += groups) {
    const float* src = rows + (uint64_t)idx[j] * 128;
    float* dst = out + j * 128;
    f4 a = *reinterpret_cast<const f4*>(src + lig * 4);
    f4 b = *reinterpret_cast<const f4*>(src + 64 + lig * 4);
    *reinterpret_cast<f4*>(dst + lig * 4) = a;
    *reinterpret_cast<f4*>(dst + 64 + lig * 4) = b;
  }
}

static int node_of(void* p) {
  int node = -1;
  // get_mempolicy(&node, NULL, 0, addr, MPOL_F_NODE | MPOL_F_ADDR)
  if (syscall(SYS_get_mempolicy, &node, nullptr, 0, p, 1 | 2) != 0) return -1;
  return node;
}

int main() {
  const uint64_t rows = 8ull << 20;  // 4 GB
  const uint64_t n = 400000;
  std::vector<uint32_t> hi(n);
  uint64_t x = 88172645463325252ull;
  for (auto& v : hi) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x % rows); }
  uint32_t* di; float* dout;
  hipMalloc((void**)&di, n * 4); hipMalloc((void**)&dout, n * 512);
  hipMemcpy(di, hi.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    float* h = nullptr;
    unsigned flags = hipHostMallocDefault;
    if (mode > 0) {
      unsigned long mask = 1ul << (mode - 1);
      if (syscall(SYS_set_mempolicy, 2 /*MPOL_BIND*/, &mask, 64) != 0) { printf("set_mempolicy failed\n"); continue; }
      flags = hipHostMallocNumaUser;
    }
    if (hipHostMalloc((void**)&h, rows * 512, flags) != hipSuccess) { printf("mode %d: hostmalloc failed\n", mode); continue; }
    for (uint64_t i = 0; i < rows * 128; i += 1024) h[i] = (float)i;
    unsigned long none = 0;
    syscall(SYS_set_mempolicy, 0 /*MPOL_DEFAULT*/, &none, 64);
    const int nd0 = node_of(h), nd1 = node_of(h + rows * 64), nd2 = node_of(h + rows * 128 - 1024);
    gather<<<128, 256>>>(h, di, n, dout);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) gather<<<128, 256>>>(h, di, n, dout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-28s pages on nodes %d/%d/%d: %.3f ms  %.1f GB/s\n",
           mode == 0 ? "hipHostMalloc default" : (mode == 1 ? "NumaUser + bind node 0" : "NumaUser + bind node 1"), nd0, nd1, nd2, ms,
           n * 512 / ms / 1e6);
    hipHostFree(h);
  }
  return 0;
}
