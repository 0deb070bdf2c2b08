// Segmented device-to-device copy: the rows of ONE engine call that served several requests of a
// TRITONBACKEND_ModelInstanceExecute call go from the instance's result buffer to every request's own output buffer
// (csrc/triton/model_instance_state.cpp: ProcessCoalesced) — one launch instead of requests x tables hipMemcpyAsync calls.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace hps {

constexpr int kSegCopyMax = 96;   // segments per launch (the argument block travels by value: 96 x 28 B)
struct SegCopyArgs {
  uint32_t num_segments;
  uint32_t num_chunks;                        // 16-KB chunks over all segments
  const char* src[kSegCopyMax];
  char* dst[kSegCopyMax];
  uint64_t bytes[kSegCopyMax];                // multiples of 4
  uint32_t first_chunk[kSegCopyMax + 1];      // segment g owns chunks [first_chunk[g], first_chunk[g + 1])
};
constexpr uint32_t kSegCopyChunk = 16384;

// Enqueues the copies of `n` segments (any n: several launches when n > kSegCopyMax); src / dst / bytes are host arrays.
hipError_t LaunchSegmentedCopy(const void* const* src, void* const* dst, const uint64_t* bytes, size_t n, hipStream_t stream);

}  // namespace hps
