// Micro-benchmark behind the round-3 bucket layout: what does ONE random probe of a key bucket cost on gfx950
// as a function of the bytes it touches and of the lanes that share it?  The probe kernel of the cache (K_P) issues ~1.1 M
// of them per config-2 call into ~9 GB of bucket lines (far beyond L2 + MALL), each followed by a recency update.
//
//   variant          lanes x bytes   touches      recency update
//   L128x16          16 x 8 B        128-B line   none / 4-B store into a separate array / 1-B store into the line
//   L128x8            8 x 16 B       128-B line   "
//   L64x8             8 x 8 B        64-B half    "
//   L64x4             4 x 16 B       64-B half    "
//
// Every workgroup (512 threads) takes a run of probes the way a tile does; kU probes in flight per lane group.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/probe_width.hip -o tools/micro/probe_width.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// kLanes lanes share a probe; each loads kBytes (8 or 16).  Granule = kLanes*kBytes bytes, aligned.
// mode 0: no update; 1: 4-B store into `stamps` (separate array, one word per 8-B key slot); 2: 1-B store into the granule's last 8 bytes
template <int kLanes, int kBytes, int kU, int kMode>
__global__ __launch_bounds__(512) void probe(const unsigned long long* __restrict__ lines, uint64_t num_granules, uint32_t* __restrict__ stamps,
                                              int32_t* __restrict__ slot, uint64_t n, uint64_t salt, uint32_t epoch) {
  constexpr int kGroups = 512 / kLanes;
  const int lig = threadIdx.x % kLanes;
  const int g = threadIdx.x / kLanes;
  const uint64_t per_block = (n + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = per_block * blockIdx.x, hi = lo + per_block < n ? lo + per_block : n;
  constexpr int kWordsPerGranule = kLanes * kBytes / 8;
  for (uint64_t r0 = lo + (uint64_t)g * kU; r0 < hi; r0 += (uint64_t)kGroups * kU) {
    uint64_t gr[kU];
    unsigned long long v0[kU], v1[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const uint64_t r = r0 + u < hi ? r0 + u : r0;
      gr[u] = ((mix64(r ^ salt) >> 32) * num_granules) >> 32;
      const unsigned long long* p = lines + gr[u] * kWordsPerGranule + (uint64_t)lig * (kBytes / 8);
      if (kBytes == 16) { const u64x2 t = *reinterpret_cast<const u64x2*>(p); v0[u] = t.x; v1[u] = t.y; }
      else { v0[u] = *p; v1[u] = 1; }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      // "match": pretend the key sits where a hash of the probe says (so most probes hit, as at 95 % hit)
      const uint32_t want = (uint32_t)(mix64(gr[u]) % (uint32_t)kWordsPerGranule);
      const uint32_t mine = (uint32_t)lig * (kBytes / 8);
      const bool match = (v0[u] != 0x123456789ull && mine == want) || (kBytes == 16 && v1[u] != 0x123456789ull && mine + 1 == want);
      if (match && r0 + u < hi) {
        const uint32_t s = (uint32_t)(gr[u] * kWordsPerGranule + want);
        slot[r0 + u] = (int32_t)s;
        if (kMode == 1) stamps[s] = epoch;
        if (kMode == 2) reinterpret_cast<uint8_t*>(const_cast<unsigned long long*>(lines))[(gr[u] + 1) * (uint64_t)(kWordsPerGranule * 8) - 8 + (want & 7)] = (uint8_t)epoch;
      }
    }
  }
}

template <int kLanes, int kBytes, int kU, int kMode>
float run(const char* name, unsigned long long* lines, uint64_t bytes, uint32_t* stamps, int32_t* slot, uint64_t n, int blocks) {
  const uint64_t granules = bytes / (kLanes * kBytes);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int it = 0; it < 12; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<kLanes, kBytes, kU, kMode>), dim3(blocks), dim3(512), 0, 0, lines, granules, stamps, slot, n, 0x1234567ull * (it + 1), (uint32_t)it + 2);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    if (it >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const float med = ms[ms.size() / 2];
  printf("%-28s granule %3d B  U=%d  mode %d  blocks %4d: median %7.1f us  min %7.1f us  (%5.2f G probes/s, %6.1f GB/s of granules)\n", name,
         kLanes * kBytes, kU, kMode, blocks, med * 1e3, ms[0] * 1e3, n / (med * 1e-3) / 1e9, n * (double)(kLanes * kBytes) / (med * 1e-3) / 1e9);
  return med;
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1100000ull;
  const uint64_t bytes = (argc > 2 ? strtoull(argv[2], nullptr, 10) : 8192ull) << 20;   // MB of bucket lines
  unsigned long long* lines; uint32_t* stamps; int32_t* slot;
  CK(hipMalloc(&lines, bytes));
  CK(hipMalloc(&stamps, bytes / 2));
  CK(hipMalloc(&slot, n * sizeof(int32_t)));
  CK(hipMemset(lines, 0x5A, bytes));
  CK(hipMemset(stamps, 0, bytes / 2));
  CK(hipDeviceSynchronize());
  printf("probes per launch %llu, bucket lines %llu MB\n", (unsigned long long)n, (unsigned long long)(bytes >> 20));
  for (int blocks : {1664, 1024, 2048}) {
#define R(L, B, U, M) run<L, B, U, M>(#L "x" #B, lines, bytes, stamps, slot, n, blocks)
    R(16, 8, 4, 0); R(16, 8, 8, 0); R(8, 16, 4, 0); R(8, 16, 8, 0); R(8, 8, 4, 0); R(8, 8, 8, 0); R(4, 16, 4, 0); R(4, 16, 8, 0);
    R(16, 8, 4, 1); R(8, 16, 4, 1); R(8, 8, 4, 1); R(4, 16, 8, 1);
    R(16, 8, 4, 2); R(8, 16, 4, 2); R(8, 8, 4, 2); R(4, 16, 8, 2);
#undef R
  }
  return 0;
}
