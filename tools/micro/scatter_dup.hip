// Round 6, review item 5: what would "the scatter kernel also writes each unique missed row into its cache slot" buy?
// (the one insert variant not tried: claim phase with CAS only, then ONE kernel that copies every staged row to its cache slot AND
//  to the positions that asked for it — the staged rows are read once instead of twice.)
// This measures the DATA MOVEMENT of the three candidates on the headline's shape, nothing else (no claims, no lists to walk):
//   scatter       U staged rows of 512 B -> S output positions (what hps_miss_scatter_kernel moves)
//   copy          U staged rows -> U cache slots              (what hps_cache_insert_kernel moves besides its claim chain)
//   scatter+dup   U staged rows -> S output positions AND U cache slots, the row read once
// 16 lanes per row, four rows in flight per group, non-temporal stores — the product kernels' arrangement.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/scatter_dup tools/micro/scatter_dup.hip && /tmp/scatter_dup
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool OUT, bool SLOT>
__global__ __launch_bounds__(256) void mv(const float* __restrict__ staging, const uint32_t* __restrict__ src_of, const uint32_t* __restrict__ out_pos,
                                          const uint32_t* __restrict__ slot_pos, uint32_t n, float* __restrict__ out, float* __restrict__ cache) {
  const int lig = threadIdx.x & 15;
  constexpr int R = 4;
  const uint32_t groups = gridDim.x * 16;
  for (uint32_t q0 = (blockIdx.x * 16 + (threadIdx.x >> 4)) * R; q0 < n; q0 += groups * R) {
    f4 v[R][2];
    uint32_t u[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t q = q0 + r;
      u[r] = q < n ? src_of[q] : 0xFFFFFFFFu;
      if (u[r] != 0xFFFFFFFFu) {
        const float* s = staging + (size_t)u[r] * 128;
        v[r][0] = *reinterpret_cast<const f4*>(s + lig * 4);
        v[r][1] = *reinterpret_cast<const f4*>(s + 64 + lig * 4);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (u[r] == 0xFFFFFFFFu) continue;
      const uint32_t q = q0 + r;
      if (OUT) {
        float* d = out + (size_t)out_pos[q] * 128;
        __builtin_nontemporal_store(v[r][0], reinterpret_cast<f4*>(d + lig * 4));
        __builtin_nontemporal_store(v[r][1], reinterpret_cast<f4*>(d + 64 + lig * 4));
      }
      if (SLOT) {
        float* d = cache + (size_t)slot_pos[u[r]] * 128;
        __builtin_nontemporal_store(v[r][0], reinterpret_cast<f4*>(d + lig * 4));
        __builtin_nontemporal_store(v[r][1], reinterpret_cast<f4*>(d + 64 + lig * 4));
      }
    }
  }
}

int main() {
  const uint32_t U = 73000, N = 1703936;                 // unique missed rows of a headline call; output rows
  const size_t slots = (size_t)16 << 20;                 // 16 M slots of 512 B = 8 GB of "cache rows" to scatter into
  std::vector<uint32_t> src(U), opos(U), spos(U);
  uint64_t s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
  for (uint32_t i = 0; i < U; ++i) { src[i] = i; opos[i] = rnd() % N; spos[i] = rnd() % slots; }
  // (the product's lists are per tile: sources nearly in order inside a table, positions anywhere)
  float *staging, *out, *cache; uint32_t *d_src, *d_opos, *d_spos;
  CK(hipMalloc(&staging, (size_t)U * 512 * 8));          // eight staging buffers, rotated (a fresh upload is not in any cache)
  CK(hipMalloc(&out, (size_t)N * 512));
  CK(hipMalloc(&cache, slots * 512));
  CK(hipMalloc(&d_src, U * 4)); CK(hipMalloc(&d_opos, U * 4)); CK(hipMalloc(&d_spos, U * 4));
  CK(hipMemcpy(d_src, src.data(), U * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_opos, opos.data(), U * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_spos, spos.data(), U * 4, hipMemcpyHostToDevice));
  CK(hipMemset(staging, 1, (size_t)U * 512 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t grid = (U / 4 + 15) / 16;               // one group per four rows, as the product's scatter launches
  auto time = [&](const char* name, auto kernel, double bytes) {
    std::vector<float> ms;
    for (int it = 0; it < 24; ++it) {
      const float* st = staging + (size_t)(it % 8) * U * 128;
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, st, d_src, d_opos, d_spos, U, out, cache);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); if (it >= 4) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-14s median %6.1f us  min %6.1f us   %5.0f GB/s (bytes moved / median)\n", name, ms[ms.size() / 2] * 1e3, ms[0] * 1e3, bytes / (ms[ms.size() / 2] * 1e-3) / 1e9);
  };
  const double row = 512.0 * U;
  time("scatter", mv<true, false>, 2 * row);
  time("copy", mv<false, true>, 2 * row);
  time("scatter+dup", mv<true, true>, 3 * row);
  printf("(hipEvent pairs around single launches: add ~5 us of launch hand-off to each figure against a profiler's kernel time)\n");
  return 0;
}
