// Kernels of the table-sharded (model-parallel) lookup — BASELINE config 3, a north-star addition (the
// reference itself is replicas-only).  Rows of one table are partitioned over P ranks by
// owner(key) = mix64(key) mod P; a rank buckets its local keys by owner, exchanges keys and rows with RCCL
// send/recv groups (csrc/cache/shard_session.cpp; the torch.distributed variant in hugectr_backend_amd/sharded.py serves
// host-tier shards over gloo) and restores the input order.
//
//   hps_shard_hist      per-block histogram of owners (1024 keys per block)
//   hps_shard_scan      exclusive scan -> write offset of every (shard, block) pair; totals per shard
//   hps_shard_scatter   stable counting-sort scatter: keys grouped by owner + permutation
//   hps_shard_unpermute out[perm[j]] = rows[j]   (16-lane group per row, 16 B per lane)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common/hps_hash.h"
#include "shard_kernels.h"

namespace hps {

typedef float f4s __attribute__((ext_vector_type(4)));
constexpr int kShardBlock = 1024;
constexpr int kMaxShards = 64;

__device__ __forceinline__ uint32_t owner_of(int64_t key, uint32_t P) { return (uint32_t)(hps_mix64((uint64_t)key) % P); }
// keys of the padded exchange arrive as int64 or, when a host request was narrowed while it was staged, as uint32
__device__ __forceinline__ int64_t load_key(const void* keys, uint32_t key_bytes, uint64_t i) {
  return key_bytes == 4 ? (int64_t)(uint64_t)reinterpret_cast<const uint32_t*>(keys)[i] : reinterpret_cast<const int64_t*>(keys)[i];
}
constexpr uint32_t kNoOwner = 0xFFFFFFFFu;
constexpr uint32_t kPosDefault = 0xFFFFFFFFu;   // pos[] of a key that is never sent (the cache's reserved key): default vector

// K1 in front of the exchange: the request's keys are deduplicated call-wide BEFORE they are bucketed, so that a key a rank
// asks for a thousand times (Zipf traffic) crosses the links once and its row comes back once.  rep[i] = index of the key's
// representative (itself for the first claimant of its set entry; a duplicate learns it from the entry and compares against
// the INPUT array, which no kernel of the call writes — no intra-launch hand-off of data).  Open addressing, entries
// (tag << 32 | index), entries of earlier calls (other tags) are free: the set is never cleared.
__global__ __launch_bounds__(256) void hps_shard_dedup_kernel(const void* __restrict__ keys, uint32_t key_bytes, uint64_t n,
                                                              unsigned long long* __restrict__ set, uint64_t mask, uint32_t tag,
                                                              uint32_t* __restrict__ rep) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const int64_t key = load_key(keys, key_bytes, i);
    uint32_t r = (uint32_t)i;
    if (key != HPS_EMPTY_KEY) {
      uint64_t h = (hps_mix64((uint64_t)key) >> 13) & mask;   // (the owner is the same hash modulo P: other bits here)
      const unsigned long long mine = ((unsigned long long)tag << 32) | (unsigned long long)(uint32_t)i;
      unsigned long long cur = __hip_atomic_load(&set[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        if ((uint32_t)(cur >> 32) != tag) {
          const unsigned long long prev = atomicCAS(&set[h], cur, mine);
          if (prev == cur) break;
          cur = prev;
          continue;
        }
        const uint32_t j = (uint32_t)cur;
        if (load_key(keys, key_bytes, j) == key) { r = j; break; }
        h = (h + 1) & mask;
        cur = __hip_atomic_load(&set[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    rep[i] = r;
  }
}

__global__ __launch_bounds__(kShardBlock) void hps_shard_hist_padded_kernel(const void* __restrict__ keys, uint32_t key_bytes, uint64_t n,
                                                                           uint32_t P, const uint32_t* __restrict__ rep /*optional*/,
                                                                           uint32_t* __restrict__ hist /*[blocks][P]*/) {
  __shared__ uint32_t sh[kMaxShards];
  if (threadIdx.x < P) sh[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * kShardBlock + threadIdx.x;
  if (i < n) {
    const int64_t key = load_key(keys, key_bytes, i);
    if (key != HPS_EMPTY_KEY && (!rep || rep[i] == (uint32_t)i)) atomicAdd(&sh[owner_of(key, P)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < P) hist[(uint64_t)blockIdx.x * P + threadIdx.x] = sh[threadIdx.x];
}

__global__ __launch_bounds__(kShardBlock) void hps_shard_hist_kernel(const int64_t* __restrict__ keys, uint64_t n, uint32_t P,
                                                                    uint32_t* __restrict__ hist /*[blocks][P]*/) {
  __shared__ uint32_t sh[kMaxShards];
  if (threadIdx.x < P) sh[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * kShardBlock + threadIdx.x;
  if (i < n) atomicAdd(&sh[owner_of(keys[i], P)], 1u);
  __syncthreads();
  if (threadIdx.x < P) hist[(uint64_t)blockIdx.x * P + threadIdx.x] = sh[threadIdx.x];
}

// one block; offsets[b][s] = sum_{s'<s} total[s'] + sum_{b'<b} hist[b'][s]
__global__ __launch_bounds__(64) void hps_shard_scan_kernel(const uint32_t* __restrict__ hist, uint32_t blocks, uint32_t P,
                                                            uint64_t* __restrict__ offsets, uint64_t* __restrict__ totals) {
  __shared__ uint64_t tot[kMaxShards];
  const uint32_t s = threadIdx.x;
  if (s < P) {
    uint64_t run = 0;
    for (uint32_t b = 0; b < blocks; ++b) { offsets[(uint64_t)b * P + s] = run; run += hist[(uint64_t)b * P + s]; }
    tot[s] = run;
    totals[s] = run;
  }
  __syncthreads();
  if (s < P) {
    uint64_t base = 0;
    for (uint32_t q = 0; q < s; ++q) base += tot[q];
    for (uint32_t b = 0; b < blocks; ++b) offsets[(uint64_t)b * P + s] += base;
  }
}

__global__ __launch_bounds__(kShardBlock) void hps_shard_scatter_kernel(const int64_t* __restrict__ keys, uint64_t n, uint32_t P,
                                                                       const uint64_t* __restrict__ offsets,
                                                                       int64_t* __restrict__ keys_sorted,
                                                                       int32_t* __restrict__ perm) {
  // stable inside the block: rank of a key among the block's earlier keys with the same owner, computed with
  // one ballot per shard value present in the wave + per-wave counts in LDS
  __shared__ uint32_t wave_cnt[kShardBlock / 64][kMaxShards];
  const uint64_t i = (uint64_t)blockIdx.x * kShardBlock + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool valid = i < n;
  const int64_t key = valid ? keys[i] : 0;
  const uint32_t own = valid ? owner_of(key, P) : 0xFFFFFFFFu;
  for (uint32_t s = lane; s < P; s += 64) wave_cnt[wave][s] = 0;
  __syncthreads();
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
    const uint64_t pos = offsets[(uint64_t)blockIdx.x * P + own] + before + rank_in_wave;
    keys_sorted[pos] = key;
    perm[pos] = (int32_t)i;
  }
}

__global__ __launch_bounds__(256) void hps_shard_unpermute_kernel(const float* __restrict__ rows, const int32_t* __restrict__ perm,
                                                                  uint64_t n, uint32_t D, float* __restrict__ out, int vec) {
  const int lig = threadIdx.x & 15;
  const uint64_t groups_total = (uint64_t)gridDim.x * 16;
  for (uint64_t j = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4); j < n; j += groups_total) {
    const float* src = rows + j * D;
    float* dst = out + (uint64_t)(uint32_t)perm[j] * D;
    if (vec) {
      for (uint32_t c = (uint32_t)lig * 4; c < D; c += 64)
        __builtin_nontemporal_store(*reinterpret_cast<const f4s*>(src + c), reinterpret_cast<f4s*>(dst + c));
    } else {
      for (uint32_t c = (uint32_t)lig; c < D; c += 16) dst[c] = src[c];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fixed-capacity ("padded") exchange of the native sharded session (shard_session.cpp): every rank sends every peer a
// block of `stride` = 2 + cap int64 words — [0] keys in the block, [1] 1 if this rank overflowed some block, then the
// keys — so the all-to-all needs no count exchange, no host read-back and no stream synchronisation.
// ------------------------------------------------------------------------------------------------
// one block: totals + headers of the P send blocks + base offsets of the stable scatter (as hps_shard_scan)
__global__ __launch_bounds__(64) void hps_shard_scan_padded_kernel(const uint32_t* __restrict__ hist, uint32_t blocks, uint32_t P,
                                                                   uint64_t* __restrict__ offsets /*[blocks][P]: rank of the block's first key inside its shard*/,
                                                                   int64_t* __restrict__ send, uint64_t stride, uint64_t cap,
                                                                   uint64_t* __restrict__ totals) {
  __shared__ unsigned long long need;   // the largest block this rank would have liked to send
  if (threadIdx.x == 0) need = 0;
  __syncthreads();
  const uint32_t s = threadIdx.x;
  uint64_t run = 0;
  if (s < P) {
    for (uint32_t b = 0; b < blocks; ++b) { offsets[(uint64_t)b * P + s] = run; run += hist[(uint64_t)b * P + s]; }
    totals[s] = run;
    atomicMax(&need, (unsigned long long)run);
  }
  __syncthreads();
  if (s < P) {
    send[(uint64_t)s * stride] = (int64_t)(run < cap ? run : cap);
    send[(uint64_t)s * stride + 1] = (int64_t)need;   // > cap: this rank overflowed; the group retries with the largest need seen
  }
}

// stable scatter into the padded send blocks; pos[i] = where key i's row will sit in the returned padded row layout
__global__ __launch_bounds__(kShardBlock) void hps_shard_scatter_padded_kernel(const void* __restrict__ keys, uint32_t key_bytes, uint64_t n, uint32_t P,
                                                                              const uint64_t* __restrict__ offsets,
                                                                              int64_t* __restrict__ send, uint64_t stride, uint64_t cap,
                                                                              const uint32_t* __restrict__ rep /*optional*/,
                                                                              uint32_t* __restrict__ pos) {
  __shared__ uint32_t wave_cnt[kShardBlock / 64][kMaxShards];
  const uint64_t i = (uint64_t)blockIdx.x * kShardBlock + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool inb = i < n;
  const int64_t key = inb ? load_key(keys, key_bytes, i) : 0;
  const bool dup = inb && rep && rep[i] != (uint32_t)i;   // travels as its representative; the gather reads pos[rep[i]]
  const bool valid = inb && !dup && key != HPS_EMPTY_KEY;   // the cache's reserved key never travels: its answer is the default vector
  const uint32_t own = valid ? owner_of(key, P) : kNoOwner;
  for (uint32_t s = lane; s < P; s += 64) wave_cnt[wave][s] = 0;
  __syncthreads();
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
    const uint64_t r = offsets[(uint64_t)blockIdx.x * P + own] + before + rank_in_wave;   // rank inside the shard's block
    if (r < cap) send[(uint64_t)own * stride + 2 + r] = key;
    pos[i] = (uint32_t)((uint64_t)own * cap + (r < cap ? r : cap - 1));   // overflowed keys: a valid slot; the call is retried
  } else if (inb && !dup) {
    pos[i] = kPosDefault;
  }
}

// received blocks -> one contiguous padded key array [P][cap] for the local lookup.  Unused slots carry HPS_EMPTY_KEY, which
// the probe kernel skips when the call says so (CallDesc::skip_empty_keys): no bucket probe, no row, no statistics.
// flags[0] = the largest block any rank of the group needed (> cap: the call is repeated with that capacity);
// flags[1] = keys received (the local lookup's real size)
__global__ __launch_bounds__(256) void hps_shard_prepare_kernel(const int64_t* __restrict__ recv, uint32_t P, uint64_t stride, uint64_t cap,
                                                                int64_t* __restrict__ keys_pad, uint32_t* __restrict__ flags) {
  const uint64_t total = (uint64_t)P * cap;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t p = e / cap, j = e - p * cap;
    const uint64_t cnt = (uint64_t)recv[p * stride];
    keys_pad[e] = j < cnt ? recv[p * stride + 2 + j] : HPS_EMPTY_KEY;
  }
  if (blockIdx.x == 0 && threadIdx.x < P) {
    atomicMax(&flags[0], (uint32_t)recv[(uint64_t)threadIdx.x * stride + 1]);
    atomicAdd(&flags[1], (uint32_t)recv[(uint64_t)threadIdx.x * stride]);
  }
}

// out[i] = rows[pos[i]]   (16-lane group per row, 16 B per lane; input order restored by construction)
__global__ __launch_bounds__(256) void hps_shard_gather_back_kernel(const float* __restrict__ rows, const uint32_t* __restrict__ pos,
                                                                    const uint32_t* __restrict__ rep /*optional*/,
                                                                    uint64_t n, uint32_t D, float* __restrict__ out, int vec, float default_value) {
  const int lig = threadIdx.x & 15;
  const uint64_t groups_total = (uint64_t)gridDim.x * 16;
  for (uint64_t i = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4); i < n; i += groups_total) {
    float* dst = out + i * D;
    const uint32_t p = pos[rep ? rep[i] : i];   // a duplicate's row sits where its representative's does
    if (p == kPosDefault) {   // the reserved key: in no table by construction (docs/hierarchical_parameter_server.md:244-246)
      for (uint32_t c = (uint32_t)lig; c < D; c += 16) dst[c] = default_value;
      continue;
    }
    const float* src = rows + (uint64_t)p * D;
    if (vec) {
      for (uint32_t c = (uint32_t)lig * 4; c < D; c += 64)
        __builtin_nontemporal_store(*reinterpret_cast<const f4s*>(src + c), reinterpret_cast<f4s*>(dst + c));
    } else {
      for (uint32_t c = (uint32_t)lig; c < D; c += 16) dst[c] = src[c];
    }
  }
}

uint32_t ShardOwnerHost(int64_t key, uint32_t P) { return (uint32_t)(hps_mix64((uint64_t)key) % P); }

size_t ShardBucketWorkspaceBytes(uint64_t n, uint32_t P) {
  const uint64_t blocks = (n + kShardBlock - 1) / kShardBlock;
  return (size_t)(blocks * P * (sizeof(uint32_t) + sizeof(uint64_t)) + 256);
}

hipError_t LaunchShardBucket(const int64_t* d_keys, uint64_t n, uint32_t P, int64_t* d_keys_sorted, int32_t* d_perm,
                             uint64_t* d_totals, void* d_workspace, hipStream_t stream) {
  if (P == 0 || P > (uint32_t)kMaxShards) return hipErrorInvalidValue;
  const uint32_t blocks = (uint32_t)((n + kShardBlock - 1) / kShardBlock);
  if (blocks == 0) return hipMemsetAsync(d_totals, 0, sizeof(uint64_t) * P, stream);
  uint32_t* hist = reinterpret_cast<uint32_t*>(d_workspace);
  uint64_t* offsets = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(d_workspace) + (((size_t)blocks * P * sizeof(uint32_t) + 15) & ~(size_t)15));
  hipLaunchKernelGGL(hps_shard_hist_kernel, dim3(blocks), dim3(kShardBlock), 0, stream, d_keys, n, P, hist);
  hipLaunchKernelGGL(hps_shard_scan_kernel, dim3(1), dim3(64), 0, stream, hist, blocks, P, offsets, d_totals);
  hipLaunchKernelGGL(hps_shard_scatter_kernel, dim3(blocks), dim3(kShardBlock), 0, stream, d_keys, n, P, offsets, d_keys_sorted,
                     d_perm);
  return hipGetLastError();
}

hipError_t LaunchShardDedup(const void* d_keys, uint32_t key_bytes, uint64_t n, unsigned long long* d_set, uint64_t set_mask, uint32_t tag,
                            uint32_t* d_rep, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  if ((key_bytes != 8 && key_bytes != 4) || tag == 0 || (set_mask & (set_mask + 1)) != 0 || set_mask + 1 < 2 * n) return hipErrorInvalidValue;
  uint64_t want = (n + 255) / 256;
  if (want > 4096) want = 4096;
  hipLaunchKernelGGL(hps_shard_dedup_kernel, dim3((uint32_t)want), dim3(256), 0, stream, d_keys, key_bytes, n, d_set, set_mask, tag, d_rep);
  return hipGetLastError();
}

hipError_t LaunchShardBucketPadded(const void* d_keys, uint32_t key_bytes, uint64_t n, uint32_t P, uint64_t cap, int64_t* d_send, uint32_t* d_pos,
                                   uint64_t* d_totals, void* d_workspace, hipStream_t stream, const uint32_t* d_rep) {
  if (key_bytes != 8 && key_bytes != 4) return hipErrorInvalidValue;
  if (P == 0 || P > (uint32_t)kMaxShards || cap == 0) return hipErrorInvalidValue;
  const uint64_t stride = cap + 2;
  uint32_t blocks = (uint32_t)((n + kShardBlock - 1) / kShardBlock);
  uint32_t* hist = reinterpret_cast<uint32_t*>(d_workspace);
  uint64_t* offsets = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(d_workspace) + (((size_t)(blocks ? blocks : 1) * P * sizeof(uint32_t) + 15) & ~(size_t)15));
  if (blocks) hipLaunchKernelGGL(hps_shard_hist_padded_kernel, dim3(blocks), dim3(kShardBlock), 0, stream, d_keys, key_bytes, n, P, d_rep, hist);
  hipLaunchKernelGGL(hps_shard_scan_padded_kernel, dim3(1), dim3(64), 0, stream, hist, blocks, P, offsets, d_send, stride, cap, d_totals);
  if (blocks)
    hipLaunchKernelGGL(hps_shard_scatter_padded_kernel, dim3(blocks), dim3(kShardBlock), 0, stream, d_keys, key_bytes, n, P, offsets, d_send,
                       stride, cap, d_rep, d_pos);
  return hipGetLastError();
}

hipError_t LaunchShardPrepare(const int64_t* d_recv, uint32_t P, uint64_t cap, int64_t* d_keys_pad, uint32_t* d_flags,
                              hipStream_t stream) {
  uint64_t want = ((uint64_t)P * cap + 255) / 256;
  if (want > 2048) want = 2048;
  if (want == 0) want = 1;
  hipLaunchKernelGGL(hps_shard_prepare_kernel, dim3((uint32_t)want), dim3(256), 0, stream, d_recv, P, cap + 2, cap, d_keys_pad, d_flags);
  return hipGetLastError();
}

hipError_t LaunchShardGatherBack(const float* d_rows, const uint32_t* d_pos, uint64_t n, uint32_t D, float* d_out, float default_value,
                                 hipStream_t stream, const uint32_t* d_rep) {
  if (n == 0) return hipSuccess;
  uint64_t want = (n + 15) / 16;
  if (want > 2048) want = 2048;
  const int vec = ((D & 3u) == 0 && ((uintptr_t)d_rows & 15u) == 0 && ((uintptr_t)d_out & 15u) == 0) ? 1 : 0;
  hipLaunchKernelGGL(hps_shard_gather_back_kernel, dim3((uint32_t)want), dim3(256), 0, stream, d_rows, d_pos, d_rep, n, D, d_out, vec, default_value);
  return hipGetLastError();
}

hipError_t LaunchShardUnpermute(const float* d_rows, const int32_t* d_perm, uint64_t n, uint32_t D, float* d_out,
                                hipStream_t stream) {
  if (n == 0) return hipSuccess;
  uint64_t want = (n + 15) / 16;
  if (want > 2048) want = 2048;
  const int vec = ((D & 3u) == 0 && ((uintptr_t)d_rows & 15u) == 0 && ((uintptr_t)d_out & 15u) == 0) ? 1 : 0;
  hipLaunchKernelGGL(hps_shard_unpermute_kernel, dim3((uint32_t)want), dim3(256), 0, stream, d_rows, d_perm, n, D, d_out, vec);
  return hipGetLastError();
}

}  // namespace hps
