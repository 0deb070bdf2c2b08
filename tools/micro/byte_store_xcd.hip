// Do plain 1-byte stores from different XCDs into the SAME dword / 8-byte word / 128-B line all survive the kernel boundary?
// (Per-XCD L2s are not coherent with each other; if an L2 wrote dirty data back at anything coarser than byte granularity,
//  one XCD's write-back would revert its neighbour's byte.)  The bucket lines of the cache carry one recency byte per slot
// that probes running on any XCD rewrite (csrc/cache/kernels.hip, hps_probe_tile_kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/byte_store_xcd.hip -o tools/micro/byte_store_xcd.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// block b writes byte (b % stride_bytes) of every `stride_bytes`-byte unit it is given; blocks b, b+1, ... run on different XCDs.
// `pre_read`: every thread first reads the unit (so that the line sits in its XCD's L2, as a probed bucket line does).
__global__ void store_bytes(uint8_t* a, uint64_t units, int stride_bytes, int pre_read, uint32_t* sink, uint8_t val) {
  const int which = blockIdx.x % stride_bytes;
  uint32_t acc = 0;
  for (uint64_t u = threadIdx.x + (uint64_t)(blockIdx.x / stride_bytes) * blockDim.x; u < units; u += (uint64_t)blockDim.x * (gridDim.x / stride_bytes)) {
    if (pre_read) acc += a[u * stride_bytes + ((which + 1) % stride_bytes)];
    a[u * stride_bytes + which] = (uint8_t)(val + which + (pre_read ? (acc & 0) : 0));
  }
  if (acc == 0xFFFFFFFFu) *sink = acc;
}

int main() {
  const uint64_t bytes = 64ull << 20;
  uint8_t* d; uint32_t* sink;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4));
  std::vector<uint8_t> h(bytes);
  for (int stride : {4, 8, 16}) {
    for (int pre : {0, 1}) {
      uint64_t lost_total = 0;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(d, 0, bytes));
        CK(hipDeviceSynchronize());
        const uint64_t units = bytes / stride;
        hipLaunchKernelGGL(store_bytes, dim3(stride * 256), dim3(256), 0, 0, d, units, stride, pre, sink, (uint8_t)(10 + rep));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost));
        uint64_t lost = 0;
        for (uint64_t i = 0; i < bytes; ++i) lost += h[i] != (uint8_t)(10 + rep + (i % stride));
        lost_total += lost;
      }
      printf("bytes of one %2d-byte unit written by %2d different blocks (XCDs), pre-read %d: %llu of %llu byte stores lost\n", stride, stride, pre,
             (unsigned long long)lost_total, (unsigned long long)(5 * bytes));
    }
  }
  return 0;
}
