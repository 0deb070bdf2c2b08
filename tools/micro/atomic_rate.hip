// Micro-benchmark behind the round-2 probe design: what do device-scope atomics on DISTINCT random
// addresses cost on gfx950, next to the one-word figure of the guide (≈88/µs)?
//   (1) 64-bit CAS into a 32 MB open-addressing set (MALL-resident)         -> global dedup of block-unique keys
//   (2) returning 32-bit atomicExch on a 277 MB array (HBM-resident)        -> "first toucher of a slot this call"
//   (3) returning 32-bit atomicOr on an 8.7 MB bitmap                        -> unique-hit count by slot bitmap
//   (4) wave-aggregated append (one atomicAdd per wave on 26 counters)       -> miss list
//   (5) LDS 64-bit CAS dedup of 1024-key tiles (Zipf-like duplicates)        -> block-local input dedup
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void cas64(unsigned long long* set, uint64_t mask, uint64_t n, uint32_t* winners, uint64_t salt) {
  uint32_t w = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t key = mix64(i ^ salt) | 1;
    uint64_t h = mix64(key) & mask;
    for (;;) {
      const unsigned long long prev = atomicCAS(&set[h], 0ull, (unsigned long long)key);
      if (prev == 0ull) { ++w; break; }
      if (prev == key) break;
      h = (h + 1) & mask;
    }
  }
  if (w) atomicAdd(winners, w);
}

__global__ void exch32(uint32_t* arr, uint64_t words, uint64_t n, uint32_t epoch, uint32_t* firsts, uint64_t salt) {
  uint32_t w = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t a = mix64(i ^ salt) % words;
    const uint32_t old = atomicExch(&arr[a], epoch);
    w += old != epoch;
  }
  if (w) atomicAdd(firsts, w);
}

__global__ void store32(uint32_t* arr, uint64_t words, uint64_t n, uint32_t epoch, uint64_t salt) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t a = mix64(i ^ salt) % words;
    arr[a] = epoch;
  }
}

__global__ void or32(uint32_t* bm, uint64_t bits, uint64_t n, uint32_t* firsts, uint64_t salt) {
  uint32_t w = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t b = mix64(i ^ salt) % bits;
    const uint32_t m = 1u << (b & 31);
    const uint32_t old = atomicOr(&bm[b >> 5], m);
    w += (old & m) == 0;
  }
  if (w) atomicAdd(firsts, w);
}

__global__ void wave_append(uint32_t* counters, uint64_t n, uint64_t* list, uint64_t per_table, uint32_t miss_per_1024) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t t = (uint32_t)(i / per_table);
    const bool miss = (mix64(i) & 1023) < miss_per_1024;
    const uint64_t bal = __ballot(miss);
    if (bal) {
      const int lane = threadIdx.x & 63;
      uint32_t base = 0;
      if (lane == __builtin_ctzll(bal)) base = atomicAdd(&counters[t], (uint32_t)__popcll(bal));
      base = __shfl(base, __builtin_ctzll(bal), 64);
      if (miss) list[(uint64_t)t * per_table + base + __popcll(bal & ((1ull << lane) - 1))] = i;
    }
  }
}

// tile dedup in LDS: 1024 keys per block pass, set of 2048 64-bit entries; duplicates ~50 %
template <int kTile>
__global__ __launch_bounds__(256) void lds_dedup(const int64_t* __restrict__ keys, uint64_t n, int32_t* __restrict__ rep_out,
                                                 uint32_t* uniq_total) {
  constexpr int kSet = kTile * 2;
  __shared__ unsigned long long set[kSet];
  __shared__ uint16_t owner[kSet];
  __shared__ uint32_t cnt;
  const uint64_t tiles = (n + kTile - 1) / kTile;
  for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    for (int e = threadIdx.x; e < kSet; e += 256) set[e] = 0x8000000000000000ull;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (int j = threadIdx.x; j < kTile; j += 256) {
      const uint64_t i = tile * kTile + j;
      if (i >= n) break;
      const unsigned long long key = (unsigned long long)keys[i];
      uint32_t h = (uint32_t)(mix64(key) >> 40) & (kSet - 1);
      int32_t rep;
      for (;;) {
        const unsigned long long prev = atomicCAS(&set[h], 0x8000000000000000ull, key);
        if (prev == 0x8000000000000000ull) { owner[h] = (uint16_t)j; rep = -1; ++mine; break; }
        if (prev == key) { rep = (int32_t)h; break; }
        h = (h + 1) & (kSet - 1);
      }
      rep_out[i] = rep;
    }
    if (mine) atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0 && cnt) atomicAdd(uniq_total, cnt);
    __syncthreads();
  }
}

template <typename F>
static float time_ms(F f, int iters = 5) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  uint32_t* d_cnt; hipMalloc((void**)&d_cnt, 4096); hipMemset(d_cnt, 0, 4096);
  const uint64_t n = 920000;
  // (1)
  {
    const uint64_t cap = 4ull << 20;
    unsigned long long* set; hipMalloc((void**)&set, cap * 8);
    uint64_t salt = 1;
    for (int blocks : {512, 2048}) {
      float ms = time_ms([&] { hipMemsetAsync(set, 0, cap * 8, 0); cas64<<<blocks, 256>>>(set, cap - 1, n, d_cnt, salt++); });
      float ms0 = time_ms([&] { hipMemsetAsync(set, 0, cap * 8, 0); });
      printf("(1) cas64 n=%lu into 32MB set, %d blocks: %.1f us (memset alone %.1f us) -> %.1f G atomics/s\n", (unsigned long)n, blocks,
             ms * 1e3, ms0 * 1e3, n / ((ms - ms0) * 1e6));
    }
    hipFree(set);
  }
  // (2)
  {
    const uint64_t words = 277ull << 18;  // 277 MB
    uint32_t* arr; hipMalloc((void**)&arr, words * 4); hipMemset(arr, 0, words * 4);
    uint32_t epoch = 1; uint64_t salt = 100;
    for (int blocks : {512, 2048}) {
      float ms = time_ms([&] { exch32<<<blocks, 256>>>(arr, words, n, ++epoch, d_cnt, salt++); });
      printf("(2) atomicExch32 (returning) n=%lu on 277MB, %d blocks: %.1f us -> %.1f G/s\n", (unsigned long)n, blocks, ms * 1e3, n / (ms * 1e6));
      float ms2 = time_ms([&] { store32<<<blocks, 256>>>(arr, words, n, ++epoch, salt++); });
      printf("    plain 4-B store same pattern: %.1f us\n", ms2 * 1e3);
    }
    hipFree(arr);
  }
  // (3)
  {
    const uint64_t bits = 26ull * 2670000;
    uint32_t* bm; hipMalloc((void**)&bm, bits / 8 + 64);
    uint64_t salt = 1000;
    float ms = time_ms([&] { hipMemsetAsync(bm, 0, bits / 8 + 64, 0); or32<<<2048, 256>>>(bm, bits, n, d_cnt, salt++); });
    float ms0 = time_ms([&] { hipMemsetAsync(bm, 0, bits / 8 + 64, 0); });
    printf("(3) atomicOr32 bitmap 8.7MB n=%lu: %.1f us (memset alone %.1f)\n", (unsigned long)n, ms * 1e3, ms0 * 1e3);
    hipFree(bm);
  }
  // (4)
  {
    const uint64_t N = 1703936, per = 65536;
    uint64_t* list; hipMalloc((void**)&list, N * 8);
    for (uint32_t mp : {51u, 512u}) {
      float ms = time_ms([&] { hipMemsetAsync(d_cnt, 0, 4096, 0); wave_append<<<2048, 256>>>(d_cnt, N, list, per, mp); });
      printf("(4) wave-aggregated append, N=%lu, miss %u/1024: %.1f us\n", (unsigned long)N, mp, ms * 1e3);
    }
    hipFree(list);
  }
  // (5)
  {
    const uint64_t N = 1703936;
    std::vector<int64_t> hk(N);
    uint64_t x = 88172645463325252ull;
    for (uint64_t i = 0; i < N; ++i) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      const uint64_t t = i / 65536;
      // ~59 % of the draws from 1000 hot keys (crudely zipf-ish by squaring a uniform), rest from 2 M
      const double u = (double)(x >> 11) / 9007199254740992.0;
      int64_t k;
      if ((x & 1023) < 604) k = (int64_t)(u * u * u * 1000.0); else k = 1000 + (int64_t)(u * 2e6);
      hk[i] = k + (int64_t)t * 100000000ll;
    }
    int64_t* dk; int32_t* rep; hipMalloc((void**)&dk, N * 8); hipMalloc((void**)&rep, N * 4);
    hipMemcpy(dk, hk.data(), N * 8, hipMemcpyHostToDevice);
    hipMemset(d_cnt, 0, 4096);
    lds_dedup<1024><<<1664, 256>>>(dk, N, rep, d_cnt); hipDeviceSynchronize();
    uint32_t u = 0; hipMemcpy(&u, d_cnt, 4, hipMemcpyDeviceToHost);
    float ms = time_ms([&] { lds_dedup<1024><<<1664, 256>>>(dk, N, rep, d_cnt); });
    printf("(5) LDS tile dedup 1024 keys/tile, N=%lu: %.1f us, block-unique fraction %.3f\n", (unsigned long)N, ms * 1e3, (double)u / N);
    hipMemset(d_cnt, 0, 4096);
    lds_dedup<2048><<<832, 256>>>(dk, N, rep, d_cnt); hipDeviceSynchronize();
    hipMemcpy(&u, d_cnt, 4, hipMemcpyDeviceToHost);
    ms = time_ms([&] { lds_dedup<2048><<<832, 256>>>(dk, N, rep, d_cnt); });
    printf("(5) LDS tile dedup 2048 keys/tile: %.1f us, block-unique fraction %.3f\n", ms * 1e3, (double)u / N);
  }
  return 0;
}
