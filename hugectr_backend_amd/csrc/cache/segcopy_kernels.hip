// hps_segcopy_kernel: see segcopy_kernels.h.  HBM-bound byte mover: one workgroup per 16-KB chunk, 16-byte accesses when the
// segment allows (rows of D % 4 == 0 floats out of 16-byte aligned buffers), 4-byte accesses otherwise.
#include <hip/hip_runtime.h>

#include "segcopy_kernels.h"

namespace hps {

typedef uint32_t u4s __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void hps_segcopy_kernel(const SegCopyArgs a) {
  for (uint32_t c = blockIdx.x; c < a.num_chunks; c += gridDim.x) {
    int lo = 0, hi = (int)a.num_segments;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.first_chunk[mid] <= c) lo = mid; else hi = mid;
    }
    const uint64_t off = (uint64_t)(c - a.first_chunk[lo]) * kSegCopyChunk;
    const uint64_t left = a.bytes[lo] - off;
    const uint32_t n = left < kSegCopyChunk ? (uint32_t)left : kSegCopyChunk;
    const char* s = a.src[lo] + off;
    char* d = a.dst[lo] + off;
    if ((((uintptr_t)s | (uintptr_t)d) & 15u) == 0) {
      const uint32_t q = n / 16;
      for (uint32_t i = threadIdx.x; i < q; i += blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const u4s*>(s) + i), reinterpret_cast<u4s*>(d) + i);
      for (uint32_t i = q * 4 + threadIdx.x; i < n / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(d)[i] = reinterpret_cast<const uint32_t*>(s)[i];
    } else {
      for (uint32_t i = threadIdx.x; i < n / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(d)[i] = reinterpret_cast<const uint32_t*>(s)[i];
    }
  }
}

hipError_t LaunchSegmentedCopy(const void* const* src, void* const* dst, const uint64_t* bytes, size_t n, hipStream_t stream) {
  size_t i = 0;
  while (i < n) {
    SegCopyArgs a;
    a.num_segments = 0;
    a.num_chunks = 0;
    for (; i < n && a.num_segments < (uint32_t)kSegCopyMax; ++i) {
      if (bytes[i] == 0) continue;
      if ((bytes[i] & 3u) || !src[i] || !dst[i]) return hipErrorInvalidValue;
      const uint32_t g = a.num_segments++;
      a.src[g] = (const char*)src[i];
      a.dst[g] = (char*)dst[i];
      a.bytes[g] = bytes[i];
      a.first_chunk[g] = a.num_chunks;
      a.num_chunks += (uint32_t)((bytes[i] + kSegCopyChunk - 1) / kSegCopyChunk);
    }
    if (a.num_segments == 0) break;
    a.first_chunk[a.num_segments] = a.num_chunks;
    const uint32_t grid = a.num_chunks < 4096u ? a.num_chunks : 4096u;
    hipLaunchKernelGGL(hps_segcopy_kernel, dim3(grid), dim3(256), 0, stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace hps
