// Store pattern for the multi-GPU self-test (multi_gpu_probe.h): a kernel on GPU a writing into memory of GPU b over the peer
// mapping — the mechanism of the peer_store transport (shard_entry.h) — as a bandwidth probe and as a 4-KB correctness check.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "multi_gpu_probe.h"

namespace hps {

typedef uint32_t u4p __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void hps_probe_store_kernel(uint32_t* __restrict__ dst, uint64_t words, uint32_t seed) {
  const uint64_t quads = words / 4;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += stride) {
    const uint32_t b = seed + (uint32_t)(q * 4);
    u4p v = {b, b + 1, b + 2, b + 3};
    __builtin_nontemporal_store(v, reinterpret_cast<u4p*>(dst) + q);
  }
  if (blockIdx.x == 0 && threadIdx.x < (words & 3)) dst[quads * 4 + threadIdx.x] = seed + (uint32_t)(quads * 4 + threadIdx.x);
}

hipError_t LaunchProbeStore(uint32_t* dst, uint64_t words, uint32_t seed, hipStream_t stream) {
  if (words == 0) return hipSuccess;
  uint64_t want = (words / 4 + 255) / 256;
  if (want < 1) want = 1;
  if (want > 2048) want = 2048;
  hipLaunchKernelGGL(hps_probe_store_kernel, dim3((uint32_t)want), dim3(256), 0, stream, dst, words, seed);
  return hipGetLastError();
}

}  // namespace hps
