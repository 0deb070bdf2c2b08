// Kernels of the ENTRY side of a table-sharded lookup (shard_entry.h) — BASELINE config 3 served through ONE instance:
// the whole request arrives on one GPU (Triton hands one request to one instance and blocks on it,
// /root/reference/hps_backend/src/hps.cc:353-369, 406), that GPU buckets the keys by owner, every owner looks its bucket up
// and writes the rows STRAIGHT into the entry GPU's output (peer-mapped pointers, xGMI stores; kernels.hip: CallDesc::dst_index),
// and rows of keys the request repeats are copied locally afterwards.  No collective, no lock-step between instances.
//
//   hps_entry_dedup     call-wide representatives per (table, key): a repeated key travels to its owner once
//   hps_entry_hist      per tile (<= 1,024 keys of one table): representatives per owner
//   hps_entry_scan      one workgroup: place of every (tile, owner) inside the owner's bucket, keys per (owner, table)
//   hps_entry_scatter   stable scatter: bucket keys (owner-major, table-major inside an owner, input order inside a table)
//                       + for every bucket key the row position it has in its table's output slice
//   hps_entry_expand    out[i] = out[rep[i]] for the request's repeated keys (local HBM copy on the entry GPU)
//
// All of it is integer / byte work bound by HBM and LDS atomics: 8 B read per key and pass, 12 B written per representative.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common/hps_hash.h"
#include "entry_kernels.h"

namespace hps {

typedef float f4e __attribute__((ext_vector_type(4)));
constexpr int kEntryTile = kTileKeys;   // 1,024 keys = 1,024 threads
constexpr int kEntryMaxShards = 64;

__device__ __forceinline__ uint32_t entry_owner(int64_t key, uint32_t P) { return (uint32_t)(hps_mix64((uint64_t)key) % P); }

// largest t with ks[t] <= i (ks[T] = N > i); skips empty tables
__device__ __forceinline__ int entry_find_table(const uint64_t* ks, int T, uint64_t i) {
  int lo = 0, hi = T;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ks[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// narrowed host keys -> int64 (one workgroup per tile: the table, and with it the base, is the tile's)
__global__ __launch_bounds__(kEntryTile) void hps_entry_widen_kernel(const EntryDesc* __restrict__ d, const TileDesc* __restrict__ tiles,
                                                                    const void* __restrict__ narrow, uint32_t key_bytes,
                                                                    int64_t* __restrict__ keys) {
  const TileDesc td = tiles[blockIdx.x];
  if (threadIdx.x >= td.count) return;
  const uint64_t i = td.begin + threadIdx.x;
  const int64_t base = d->key_base[td.table];
  uint32_t v;
  if (key_bytes == 4) v = reinterpret_cast<const uint32_t*>(narrow)[i];
  else {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(narrow) + 3 * i;
    v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
  }
  keys[i] = base + (int64_t)(uint64_t)v;
}

// rep[i] = index of the representative of (table(i), key(i)).  Two levels, like the probe kernel's input dedup (kernels.hip):
//   tile   one workgroup per tile of <= 1,024 keys of one table: a 32-bit LDS CAS set gives every key its tile-local
//          representative — the hot head of a Zipf request (one key = 7 % of 1.7 M keys) collapses to one key per tile HERE, in
//          LDS, instead of 120 K same-address atomics on one word of the call-wide set (0.35 ms of a 0.7-ms bucket step)
//   call   the tile representatives claim entries (tag << 32 | index) of the call-wide open-addressing set; entries of
//          earlier calls (other tags) are free, so the set is never cleared.  A loser compares against the INPUT array, which no
//          kernel of the call writes: no intra-launch hand-off of data.
__global__ __launch_bounds__(kEntryTile) void hps_entry_dedup_kernel(const EntryDesc* __restrict__ d, const TileDesc* __restrict__ tiles,
                                                                    const int64_t* __restrict__ keys, unsigned long long* __restrict__ set,
                                                                    uint64_t mask, uint32_t tag, uint32_t* __restrict__ rep) {
  __shared__ int64_t sh_key[kEntryTile];
  __shared__ uint32_t sh_set[2 * kEntryTile];
  __shared__ uint32_t sh_grep[kEntryTile];
  const TileDesc td = tiles[blockIdx.x];
  const uint32_t j = threadIdx.x;
  const bool inb = j < td.count;
  const uint64_t i = td.begin + j;
  const int64_t key = inb ? keys[i] : 0;
  const uint32_t t = td.table;
  const uint64_t h0 = hps_mix64((uint64_t)key ^ ((uint64_t)(t + 1) * 0x9E3779B97F4A7C15ull));
  sh_key[j] = key;
  sh_set[j] = 0xFFFFFFFFu;
  sh_set[j + kEntryTile] = 0xFFFFFFFFu;
  __syncthreads();
  uint32_t lrep = j;
  if (inb) {
    uint32_t e = (uint32_t)h0 & (2 * kEntryTile - 1);
    for (;;) {
      const uint32_t prev = atomicCAS(&sh_set[e], 0xFFFFFFFFu, j);
      if (prev == 0xFFFFFFFFu) break;
      if (sh_key[prev] == key) { lrep = prev; break; }
      e = (e + 1) & (2 * kEntryTile - 1);
    }
  }
  if (inb && lrep == j && set == nullptr) sh_grep[j] = (uint32_t)i;   // tile level only: the tile's representative travels
  if (inb && lrep == j && set != nullptr) {
    const uint64_t lo = d->key_start[t], hi = d->key_start[t + 1];
    uint64_t h = (h0 >> 11) & mask;
    const unsigned long long mine = ((unsigned long long)tag << 32) | (unsigned long long)(uint32_t)i;
    unsigned long long cur = __hip_atomic_load(&set[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t r = (uint32_t)i;
    for (;;) {
      if ((uint32_t)(cur >> 32) != tag) {
        const unsigned long long prev = atomicCAS(&set[h], cur, mine);
        if (prev == cur) break;
        cur = prev;
        continue;
      }
      const uint32_t g = (uint32_t)cur;
      if (g >= lo && g < hi && keys[g] == key) { r = g; break; }   // same table, same key
      h = (h + 1) & mask;
      cur = __hip_atomic_load(&set[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    sh_grep[j] = r;
  }
  __syncthreads();
  if (inb) rep[i] = sh_grep[lrep];
}

__global__ __launch_bounds__(kEntryTile) void hps_entry_hist_kernel(const EntryDesc* __restrict__ d, const TileDesc* __restrict__ tiles,
                                                                   const int64_t* __restrict__ keys, const uint32_t* __restrict__ rep,
                                                                   uint32_t* __restrict__ hist /*[tiles][P]*/) {
  __shared__ uint32_t sh[kEntryMaxShards];
  const uint32_t P = d->num_shards;
  if (threadIdx.x < P) sh[threadIdx.x] = 0;
  __syncthreads();
  const TileDesc td = tiles[blockIdx.x];
  if (threadIdx.x < td.count) {
    const uint64_t i = td.begin + threadIdx.x;
    if (!rep || rep[i] == (uint32_t)i) atomicAdd(&sh[entry_owner(keys[i], P)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < P) hist[(uint64_t)blockIdx.x * P + threadIdx.x] = sh[threadIdx.x];
}

// exclusive scan of one value per thread over a 1,024-thread workgroup; *total = the sum
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* sh_wave /*[17]*/, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
    if (lane >= off) inc += o;
  }
  __syncthreads();   // (sh_wave may still be read from the previous round)
  if (lane == 63) sh_wave[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int w = 0; w < 16; ++w) { const uint32_t c = sh_wave[w]; sh_wave[w] = run; run += c; }
    sh_wave[16] = run;
  }
  __syncthreads();
  *total = sh_wave[16];
  return sh_wave[wave] + inc - v;
}

// One workgroup.  within[tile][p] = representatives of owner p in the tiles before `tile`; base[p] = first bucket position of
// owner p (base[P] = all representatives); counts[p][t] = keys owner p gets of table t (tiles are table-major and the
// scatter is stable, so an owner's bucket is table-major too).
__global__ __launch_bounds__(1024) void hps_entry_scan_kernel(const EntryDesc* __restrict__ d, const uint32_t* __restrict__ hist,
                                                              uint32_t* __restrict__ within, uint32_t* __restrict__ base /*[P+1]*/,
                                                              uint32_t* __restrict__ counts /*[P][T]*/) {
  __shared__ uint32_t sh_wave[17];
  __shared__ uint32_t sh_total[kEntryMaxShards + 1];
  const uint32_t P = d->num_shards, tiles = d->num_tiles, T = d->num_tables;
  const uint32_t per = (tiles + 1023) / 1024;
  const uint32_t t0 = threadIdx.x * per, t1 = t0 + per < tiles ? t0 + per : tiles;
  for (uint32_t p = 0; p < P; ++p) {
    uint32_t sum = 0;
    for (uint32_t b = t0; b < t1; ++b) sum += hist[(uint64_t)b * P + p];
    uint32_t total = 0;
    uint32_t run = block_excl_scan_1024(sum, sh_wave, &total);
    for (uint32_t b = t0; b < t1; ++b) { within[(uint64_t)b * P + p] = run; run += hist[(uint64_t)b * P + p]; }
    if (threadIdx.x == 0) sh_total[p] = total;
  }
  __syncthreads();
  __threadfence_block();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (uint32_t p = 0; p < P; ++p) { base[p] = run; run += sh_total[p]; }
    base[P] = run;
  }
  // (within[] was written by this workgroup's own threads: visible after the barrier above)
  for (uint32_t e = threadIdx.x; e < P * T; e += 1024) {
    const uint32_t p = e / T, t = e - p * T;
    const uint32_t a = d->first_tile[t], b = d->first_tile[t + 1];
    const uint32_t wa = a < tiles ? within[(uint64_t)a * P + p] : sh_total[p];
    const uint32_t wb = b < tiles ? within[(uint64_t)b * P + p] : sh_total[p];
    counts[e] = wb - wa;
  }
}

__global__ __launch_bounds__(kEntryTile) void hps_entry_scatter_kernel(const EntryDesc* __restrict__ d, const TileDesc* __restrict__ tiles,
                                                                      const int64_t* __restrict__ keys, const uint32_t* __restrict__ rep,
                                                                      const uint32_t* __restrict__ within, const uint32_t* __restrict__ base,
                                                                      int64_t* __restrict__ bkeys, uint32_t* __restrict__ bidx) {
  // stable inside the tile: rank of a key among the tile's earlier keys with the same owner = one ballot per owner value
  // present in the wave + the earlier waves' counts from LDS
  __shared__ uint32_t wave_cnt[kEntryTile / 64][kEntryMaxShards];
  const uint32_t P = d->num_shards;
  const TileDesc td = tiles[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t i = td.begin + threadIdx.x;
  const bool inb = threadIdx.x < td.count;
  const int64_t key = inb ? keys[i] : 0;
  const bool valid = inb && (!rep || rep[i] == (uint32_t)i);
  const uint32_t own = valid ? entry_owner(key, P) : 0xFFFFFFFFu;
  uint32_t rank_in_wave = 0;
  for (uint32_t s = 0; s < P; ++s) {
    const uint64_t m = __ballot(own == s);
    if (own == s) rank_in_wave = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave][s] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  if (valid) {
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w][own];
    const uint32_t pos = base[own] + within[(uint64_t)blockIdx.x * P + own] + before + rank_in_wave;
    bkeys[pos] = key;
    bidx[pos] = (uint32_t)(i - d->key_start[td.table]);   // the row's place in its table's output slice
  }
}

// "staged_copy" transport: rows of one piece of an owner's bucket, shipped into the entry GPU by a copy engine, go to their places
// in OUTPUT0.  A 16-lane group moves one row (a 512-B row = two float4 per lane); 512 B read + 512 B written + 4 B of index per
// key at D = 128, all local HBM of the entry GPU.
__global__ __launch_bounds__(256) void hps_entry_place_kernel(const EntryDesc* __restrict__ d, const PlaceArgs a,
                                                              const uint32_t* __restrict__ bidx, const float* __restrict__ block) {
  const int lig = threadIdx.x & 15;
  const uint32_t groups = gridDim.x * (blockDim.x / 16);
  for (uint32_t j = blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4); j < a.num_keys; j += groups) {
    int lo = 0, hi = (int)a.num_segments;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.start[mid] <= j) lo = mid; else hi = mid;
    }
    const uint32_t t = a.table[lo];
    const uint32_t D = d->dim[t];
    const float* s = block + a.src_off[lo] + (uint64_t)(j - a.start[lo]) * D;
    float* o = d->out[t] + (uint64_t)bidx[j] * D;
    if ((D & 3u) == 0 && (((uintptr_t)d->out[t]) & 15u) == 0) {
      for (uint32_t e = (uint32_t)lig * 4; e < D; e += 64)
        __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const f4e*>(s + e)), reinterpret_cast<f4e*>(o + e));
    } else {
      for (uint32_t e = (uint32_t)lig; e < D; e += 16) o[e] = s[e];
    }
  }
}

// out[i] = out[rep[i]] for the keys the request repeats (their representative's row has arrived from its owner by now).
// A wave takes 64 keys (one coalesced load of their rep words), each 16-lane group copies the repeated ones among its 16.
__global__ __launch_bounds__(256) void hps_entry_expand_kernel(const EntryDesc* __restrict__ d, const uint32_t* __restrict__ rep) {
  __shared__ uint64_t sh_ks[kMaxTables + 1];
  const int T = (int)d->num_tables;
  for (int t = threadIdx.x; t <= T; t += blockDim.x) sh_ks[t] = d->key_start[t];
  __syncthreads();
  const uint64_t n = d->total_keys;
  const int lane = threadIdx.x & 63, g = lane >> 4, lig = lane & 15;
  const uint64_t chunks = (n + 63) / 64, waves = (uint64_t)gridDim.x * (blockDim.x / 64);
  for (uint64_t c = (uint64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6); c < chunks; c += waves) {
    const uint64_t i = c * 64 + (uint64_t)lane;
    const uint32_t r = i < n ? rep[i] : (uint32_t)i;
    const bool dup = i < n && r != (uint32_t)i;
    if (__ballot(dup) == 0) continue;
    for (int j = 0; j < 16; ++j) {
      const int src = g * 16 + j;
      const uint32_t rr = (uint32_t)__shfl((int)r, src, 64);
      const bool dd = __shfl((int)dup, src, 64) != 0;
      if (!dd) continue;
      const uint64_t ii = c * 64 + (uint64_t)src;
      const int t = entry_find_table(sh_ks, T, ii);
      const uint32_t D = d->dim[t];
      const float* s = d->out[t] + ((uint64_t)rr - sh_ks[t]) * D;
      float* o = d->out[t] + (ii - sh_ks[t]) * D;
      if ((D & 3u) == 0 && (((uintptr_t)d->out[t]) & 15u) == 0) {
        for (uint32_t e = (uint32_t)lig * 4; e < D; e += 64)
          __builtin_nontemporal_store(*reinterpret_cast<const f4e*>(s + e), reinterpret_cast<f4e*>(o + e));
      } else {
        for (uint32_t e = (uint32_t)lig; e < D; e += 16) o[e] = s[e];
      }
    }
  }
}

hipError_t LaunchEntryWiden(const EntryDesc* d_desc, const TileDesc* d_tiles, uint32_t num_tiles, const void* d_narrow, uint32_t key_bytes,
                            int64_t* d_keys, hipStream_t stream) {
  if (num_tiles == 0) return hipSuccess;
  if (key_bytes != 3 && key_bytes != 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(hps_entry_widen_kernel, dim3(num_tiles), dim3(kEntryTile), 0, stream, d_desc, d_tiles, d_narrow, key_bytes, d_keys);
  return hipGetLastError();
}

hipError_t LaunchEntryDedup(const EntryDesc* d_desc, const TileDesc* d_tiles, uint32_t num_tiles, const int64_t* d_keys, uint64_t n,
                            unsigned long long* d_set, uint64_t set_mask, uint32_t tag, uint32_t* d_rep, hipStream_t stream) {
  if (n == 0 || num_tiles == 0) return hipSuccess;
  if (d_set && (tag == 0 || (set_mask & (set_mask + 1)) != 0 || set_mask + 1 < 2 * n)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(hps_entry_dedup_kernel, dim3(num_tiles), dim3(kEntryTile), 0, stream, d_desc, d_tiles, d_keys, d_set, set_mask, tag, d_rep);
  return hipGetLastError();
}

hipError_t LaunchEntryBucket(const EntryDesc* d_desc, const TileDesc* d_tiles, uint32_t num_tiles, uint32_t num_shards,
                             const int64_t* d_keys, const uint32_t* d_rep, uint32_t* d_hist, uint32_t* d_within, uint32_t* d_base,
                             uint32_t* d_counts, int64_t* d_bkeys, uint32_t* d_bidx, hipStream_t stream) {
  if (num_shards == 0 || num_shards > (uint32_t)kEntryMaxShards) return hipErrorInvalidValue;
  if (num_tiles)
    hipLaunchKernelGGL(hps_entry_hist_kernel, dim3(num_tiles), dim3(kEntryTile), 0, stream, d_desc, d_tiles, d_keys, d_rep, d_hist);
  hipLaunchKernelGGL(hps_entry_scan_kernel, dim3(1), dim3(1024), 0, stream, d_desc, d_hist, d_within, d_base, d_counts);
  if (num_tiles)
    hipLaunchKernelGGL(hps_entry_scatter_kernel, dim3(num_tiles), dim3(kEntryTile), 0, stream, d_desc, d_tiles, d_keys, d_rep, d_within,
                       d_base, d_bkeys, d_bidx);
  return hipGetLastError();
}

hipError_t LaunchEntryPlace(const EntryDesc* d_desc, const PlaceArgs& args, const uint32_t* d_bidx, const float* d_block, hipStream_t stream) {
  if (args.num_keys == 0) return hipSuccess;
  if (args.num_segments == 0 || args.num_segments > (uint32_t)kPlaceMaxSegments) return hipErrorInvalidValue;
  uint32_t want = (args.num_keys + 15) / 16;
  if (want > 8192) want = 8192;
  hipLaunchKernelGGL(hps_entry_place_kernel, dim3(want), dim3(256), 0, stream, d_desc, args, d_bidx, d_block);
  return hipGetLastError();
}

hipError_t LaunchEntryExpand(const EntryDesc* d_desc, const uint32_t* d_rep, uint64_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  uint64_t want = (n + 255) / 256;
  if (want > 4096) want = 4096;
  hipLaunchKernelGGL(hps_entry_expand_kernel, dim3((uint32_t)want), dim3(256), 0, stream, d_desc, d_rep);
  return hipGetLastError();
}

}  // namespace hps
