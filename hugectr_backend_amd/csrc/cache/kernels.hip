// HIP kernels of the GPU embedding cache, written for gfx950 (MI355X, wave64, 256 CUs / 8 XCDs).
//
// There is nothing to port: /root/reference contains no device code; the kernels are
// defined by function, from docs/hierarchical_parameter_server.md:65-78 (dedup -> cache query -> miss ->
// parameter server -> insert) and the north star in BASELINE.json.
//
//   K_A  hps_probe_gather      fused cache probe + hit-row gather           HBM-bound, the roofline kernel
//   K_B0 hps_miss_begin        sum per-block miss counts, clear dedup set
//   K_B1 hps_miss_dedup        unique missed keys per table (hash set + block prefix sum)
//   K_B2 hps_miss_resolve      duplicates pick up their representative's index
//   K_C1 hps_miss_scatter      missed rows: staging -> output
//   K_C2 hps_cache_insert      unique missed (key,row) -> bucket, LRU victim claimed by CAS
//   K_D  hps_miss_fill_default async-insert mode: missed rows = default vector
//
// Work decomposition everywhere: one 16-lane group per key (4 keys per wave at a time).  A 16-lane group
// reads one 128-B key bucket with a single 8-B load per lane and moves a D=128 row as 2 x 16 B per lane
// (two fully coalesced 256-B segments), so every HBM request is a whole number of 64/128-B lines.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "device_types.h"
#include "kernels.h"

namespace hps {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// largest t with ks[t] <= i (ks[T] = N > i).  Skips empty tables.
__device__ __forceinline__ int find_table(const uint64_t* ks, int T, uint64_t i) {
  int lo = 0, hi = T;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ks[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ int64_t shfl_i64(int64_t v, int src) {
  const int lo = __shfl((int)(uint32_t)(uint64_t)v, src, 64);
  const int hi = __shfl((int)(uint32_t)((uint64_t)v >> 32), src, 64);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  return ((uint64_t)uniform_u32((uint32_t)(v >> 32)) << 32) | uniform_u32((uint32_t)v);
}

// Per-table values every probe step needs; kept in LDS once per block.
struct __attribute__((aligned(16))) TableLds {
  const int64_t* bucket_keys;
  uint32_t* stamps;
  const float* rows;
  float* out;          // output slice of this table for this call
  uint64_t key_start;  // first global key index of this table
  uint32_t num_buckets;
  uint32_t dim;
  uint32_t flags;      // bit0 static cache, bit1 vec_ok
  uint32_t pad;
};

__device__ __forceinline__ void load_tables_to_lds(TableLds* sh, uint64_t* sh_ks, const CallDesc* call,
                                                   const TableCacheDev* tables, int T) {
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    TableLds e;
    e.bucket_keys = tables[t].bucket_keys;
    e.stamps = tables[t].stamps;
    e.rows = tables[t].rows;
    e.out = call->out[t];
    e.key_start = call->key_start[t];
    e.num_buckets = tables[t].num_buckets;
    e.dim = tables[t].dim;
    e.flags = (tables[t].flags & 1u) | (call->vec_ok[t] ? 2u : 0u);
    e.pad = 0;
    sh[t] = e;
  }
  for (int t = threadIdx.x; t <= T; t += blockDim.x) sh_ks[t] = call->key_start[t];
}

// Copy one row of D floats with a 16-lane group.  vec: 16 B per lane per step (D%4==0, both sides
// 16-B aligned); otherwise 4 B per lane.  Output goes out with non-temporal stores: it is written
// once and never re-read by this path, so it should not displace cached rows/buckets in L2.
template <bool kNtStore>
__device__ __forceinline__ void copy_row(const float* __restrict__ src, float* __restrict__ dst, uint32_t D,
                                         int lig, bool vec) {
  if (vec) {
    for (uint32_t c = (uint32_t)lig * 4; c < D; c += 64) {
      const f4 v = *reinterpret_cast<const f4*>(src + c);
      if (kNtStore) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst + c));
      else *reinterpret_cast<f4*>(dst + c) = v;
    }
  } else {
    for (uint32_t c = (uint32_t)lig; c < D; c += 16) {
      const float v = src[c];
      if (kNtStore) __builtin_nontemporal_store(v, dst + c);
      else dst[c] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K_A  fused probe + gather.
// Each wave owns 64 consecutive keys per iteration (one coalesced 512-B key load); the 16 keys of
// each quarter are then walked by that quarter's 16-lane group, kUnroll keys at a time so that
// kUnroll bucket loads and then 2*kUnroll row loads per lane are in flight together.
// Algorithmic bytes per key: 8 (key) + 4D (row read) + 4D (row write).
// Overhead traffic: 128-B bucket line per key, 4-B slot index write, 4-B stamp write per hit.
// ------------------------------------------------------------------------------------------------
template <int U>
struct ProbeGroup {  // one key group in flight: keys, their tables, bucket ids and the loaded bucket lane
  int64_t k[U];
  int tt[U];
  uint32_t b[U];
  int64_t bk[U];
};

// kOuter: unroll factor of the loop over the 16/kUnroll key groups of a chunk (1 = rolled: fewer VGPRs,
// more waves per SIMD; 16/kUnroll = fully unrolled: the compiler overlaps consecutive groups;
// 0 = rolled and software-pipelined by hand: next group's bucket loads issued before this group's rows).
// kStampShift: the LRU stamp of a hit slot is rewritten for 1 in 2^kStampShift hits (hashed on key and
// epoch): a blind 4-B store per hit is a read-modify-write of a whole DRAM sector.
template <int kUnroll, int kOuter, int kStampShift>
__global__ __launch_bounds__(kProbeBlockThreads) void hps_probe_gather_kernel(
    const CallDesc* __restrict__ call, const TableCacheDev* __restrict__ tables,
    int32_t* __restrict__ slot_out, uint32_t* __restrict__ block_miss) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = (int)call->num_tables;
  TableLds* sh_tab = reinterpret_cast<TableLds*>(smem);
  uint64_t* sh_ks = reinterpret_cast<uint64_t*>(smem + sizeof(TableLds) * (size_t)T);
  uint32_t* sh_cnt = reinterpret_cast<uint32_t*>(sh_ks + (T + 1));
  load_tables_to_lds(sh_tab, sh_ks, call, tables, T);
  __syncthreads();

  const uint64_t N = call->total_keys;
  const uint32_t epoch = call->epoch;
  const int64_t* __restrict__ keys = call->keys;
  const int lane = lane_id();
  const int g = lane >> 4;    // which 16-lane group of the wave
  const int lig = lane & 15;  // lane in group
  const uint64_t waves_total = (uint64_t)gridDim.x * (kProbeBlockThreads / 64);
  const uint64_t wave_global = (uint64_t)blockIdx.x * (kProbeBlockThreads / 64) + (threadIdx.x >> 6);
  const uint64_t chunks = (N + 63) / 64;
  uint32_t my_misses = 0;

  for (uint64_t chunk = wave_global; chunk < chunks; chunk += waves_total) {
    const uint64_t i = chunk * 64 + (uint64_t)lane;
    const bool valid = i < N;
    const int64_t key = valid ? keys[i] : HPS_EMPTY_KEY;
    // table of this lane's key: search once per wave, walk forward for lanes past a table boundary
    const int t0 = (int)uniform_u32((uint32_t)find_table(sh_ks, T, chunk * 64));
    int t = t0;
    if (valid) { while (i >= sh_ks[t + 1]) ++t; }
    int32_t my_slot = kSlotMiss;
    // every lane hashes its OWN key once (64 hashes per chunk in one pass); the groups then pick the bucket
    // index up with one cross-lane read instead of re-hashing the shuffled key in all 16 lanes of every step
    const uint32_t my_bucket = hps_bucket_of(key, sh_tab[t].num_buckets);

    // The body is instantiated twice: `uniform` = all 64 keys of the chunk belong to one table (every chunk
    // except the ones that straddle a table boundary): the table descriptor sits in scalar registers, no
    // per-key LDS reads and no table-id shuffle.
    auto run = [&](auto uniform_tag, const TableLds& du) {
      constexpr bool kUniform = decltype(uniform_tag)::value;
      // phase 1 of a key group: kUnroll independent bucket-line loads
      auto issue = [&](int jb, ProbeGroup<kUnroll>& G) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const int src = g * 16 + jb + u;
          G.k[u] = shfl_i64(key, src);
          G.b[u] = (uint32_t)__shfl((int)my_bucket, src, 64);
          if (kUniform) {
            G.tt[u] = 0;
            G.bk[u] = du.bucket_keys[(uint64_t)G.b[u] * kBucketSlots + lig];
          } else {
            G.tt[u] = __shfl(t, src, 64);
            G.bk[u] = sh_tab[G.tt[u]].bucket_keys[(uint64_t)G.b[u] * kBucketSlots + lig];
          }
        }
      };
      // phases 2-4: compare + group ballot -> slot; row loads; streaming stores (hit rows only)
      auto finish = [&](int jb, const ProbeGroup<kUnroll>& G) {
        int32_t s[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const bool match = (G.bk[u] == G.k[u]) && (G.k[u] != HPS_EMPTY_KEY);
          const uint64_t m = __ballot(match);
          const uint32_t m16 = (uint32_t)(m >> (g * 16)) & 0xFFFFu;
          s[u] = m16 ? (int32_t)(G.b[u] * kBucketSlots + (uint32_t)__builtin_ctz(m16)) : kSlotMiss;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const int src = g * 16 + jb + u;
          if (s[u] >= 0) {
            const TableLds& d = kUniform ? du : sh_tab[G.tt[u]];
            const uint32_t D = d.dim;
            const uint64_t gi = chunk * 64 + (uint64_t)src;
            const float* row = d.rows + (uint64_t)(uint32_t)s[u] * D;
            float* dst = d.out + (gi - d.key_start) * D;
            // out == nullptr: probe-only call (the fused lookup+interaction path reads the rows from the slots itself)
            if (d.out != nullptr) copy_row<true>(row, dst, D, lig, (d.flags & 2u) != 0);
            if (lig == 0 && !(d.flags & 1u)) {
              bool touch = true;
              if (kStampShift > 0)
                touch = ((((uint32_t)s[u] * 0x9E3779B1u + epoch * 0x85EBCA6Bu) >> 13) & ((1u << kStampShift) - 1u)) == 0u;
              if (touch) d.stamps[(uint32_t)s[u]] = epoch;
            }
          }
          if (lane == src) my_slot = s[u];
        }
      };
      if (kOuter == 0) {
        // rolled + software-pipelined: the bucket loads of group j+1 are in flight while group j's rows move
        ProbeGroup<kUnroll> cur, nxt;
        issue(0, cur);
#pragma unroll 1
        for (int jb = 0; jb < 16; jb += kUnroll) {
          if (jb + kUnroll < 16) issue(jb + kUnroll, nxt);
          finish(jb, cur);
          cur = nxt;
        }
      } else {
#pragma unroll(kOuter > 0 ? kOuter : 1)
        for (int jb = 0; jb < 16; jb += kUnroll) {
          ProbeGroup<kUnroll> G;
          issue(jb, G);
          finish(jb, G);
        }
      }
    };
    const uint64_t chunk_last = (chunk * 64 + 63 < N ? chunk * 64 + 63 : N - 1);
    if (chunk_last < sh_ks[t0 + 1]) {
      TableLds du;  // wave-uniform copy in scalar registers
      du.bucket_keys = reinterpret_cast<const int64_t*>(uniform_u64((uint64_t)sh_tab[t0].bucket_keys));
      du.stamps = reinterpret_cast<uint32_t*>(uniform_u64((uint64_t)sh_tab[t0].stamps));
      du.rows = reinterpret_cast<const float*>(uniform_u64((uint64_t)sh_tab[t0].rows));
      du.out = reinterpret_cast<float*>(uniform_u64((uint64_t)sh_tab[t0].out));
      du.key_start = uniform_u64(sh_tab[t0].key_start);
      du.num_buckets = uniform_u32(sh_tab[t0].num_buckets);
      du.dim = uniform_u32(sh_tab[t0].dim);
      du.flags = uniform_u32(sh_tab[t0].flags);
      du.pad = 0;
      run(std::true_type{}, du);
    } else {
      run(std::false_type{}, sh_tab[t0]);
    }
    if (valid) {
      slot_out[i] = my_slot;
      my_misses += (my_slot < 0) ? 1u : 0u;
    }
  }

  // per-block miss count (plain store; K_B0 sums them — no contended atomics on the hot path)
  uint32_t w = my_misses;
  for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off, 64);
  if (lane == 0) sh_cnt[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int q = 0; q < kProbeBlockThreads / 64; ++q) tot += sh_cnt[q];
    block_miss[blockIdx.x] = tot;
  }
}

// ------------------------------------------------------------------------------------------------
// K_B0: every block sums the per-block miss counts of K_A (a few KB, L2-resident), block 0 publishes
// the total; when there are misses the dedup hash set is cleared (grid-stride) and the per-table
// unique counters are zeroed.
// counts layout: [0] = total misses of the call, [1 .. T] = unique misses per table.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t* sh) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  uint32_t tot = 0;
  for (unsigned q = 0; q < (blockDim.x + 63) / 64; ++q) tot += sh[q];
  __syncthreads();
  return tot;
}

__global__ __launch_bounds__(256) void hps_miss_begin_kernel(const uint32_t* __restrict__ block_miss,
                                                              uint32_t probe_blocks, int32_t* __restrict__ set,
                                                              uint64_t set_cap, uint32_t* __restrict__ counts,
                                                              uint32_t T) {
  __shared__ uint32_t sh[4];
  uint32_t v = 0;
  for (uint32_t b = threadIdx.x; b < probe_blocks; b += blockDim.x) v += block_miss[b];
  const uint32_t total = block_sum_u32(v, sh);
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) counts[0] = total;
    for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) { counts[1 + t] = 0; counts[kTableMissBase + t] = 0; }
  }
  if (total == 0) return;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < set_cap;
       e += (uint64_t)gridDim.x * blockDim.x)
    set[e] = -1;
}

// ------------------------------------------------------------------------------------------------
// K_B1: unique missed keys per table.  Blocks are aligned to tables (a block never straddles two
// tables), one key per thread.  A missed key claims a slot of the open-addressing set with CAS on
// the key's global index; the winner is the representative.  Winners are ranked inside the block
// (ballot + LDS prefix) and the block takes its range of the table's unique segment with ONE atomic.
// Representative i gets slot_out[i] = -2 - uidx and publishes its key at uniq_keys[key_start[t]+uidx]
// (device copy) and in the host-mapped pinned mirror the parameter-server threads read.
// ------------------------------------------------------------------------------------------------
constexpr int kDedupBlock = 1024;

__device__ __forceinline__ uint64_t set_hash(int64_t key, uint32_t t) {
  return hps_mix64((uint64_t)key ^ ((uint64_t)(t + 1) * 0xD6E8FEB86659FD93ull));
}

// block -> (table, first key) for table-aligned 1-D grids of `per_block` keys
__device__ __forceinline__ bool block_to_table(const CallDesc* call, uint32_t per_block, uint32_t* t_out,
                                               uint64_t* begin_out, uint64_t* end_out) {
  uint64_t b = blockIdx.x;
  const uint32_t T = call->num_tables;
  for (uint32_t t = 0; t < T; ++t) {
    const uint64_t n = call->key_start[t + 1] - call->key_start[t];
    const uint64_t nb = (n + per_block - 1) / per_block;
    if (b < nb) {
      *t_out = t;
      *begin_out = call->key_start[t] + b * per_block;
      const uint64_t e = *begin_out + per_block;
      *end_out = e < call->key_start[t + 1] ? e : call->key_start[t + 1];
      return true;
    }
    b -= nb;
  }
  return false;
}

__global__ __launch_bounds__(kDedupBlock) void hps_miss_dedup_kernel(
    const CallDesc* __restrict__ call, int32_t* __restrict__ slot_io, int32_t* __restrict__ set, uint64_t set_cap,
    uint32_t* __restrict__ counts, int64_t* __restrict__ uniq_keys_dev, int64_t* __restrict__ uniq_keys_host) {
  if (counts[0] == 0) return;
  uint32_t t;
  uint64_t begin, end;
  if (!block_to_table(call, kDedupBlock, &t, &begin, &end)) return;
  const uint64_t i = begin + threadIdx.x;
  const int64_t* __restrict__ keys = call->keys;
  const uint64_t mask = set_cap - 1;

  bool winner = false;
  int64_t key = 0;
  const bool missed = i < end && slot_io[i] == kSlotMiss;
  if (missed) {
    key = keys[i];
    uint64_t h = set_hash(key, t) & mask;
    for (;;) {
      const int32_t prev = atomicCAS(&set[h], -1, (int32_t)i);
      if (prev == -1) { winner = true; break; }
      // same table is implied: entries of other tables hash with another salt but may still collide,
      // so compare the owning range too
      const uint64_t pi = (uint64_t)(uint32_t)prev;
      if (pi >= call->key_start[t] && pi < call->key_start[t + 1] && keys[pi] == key) break;  // duplicate
      h = (h + 1) & mask;
    }
  }
  // rank winners inside the block
  __shared__ uint32_t sh_wave[kDedupBlock / 64];
  __shared__ uint32_t sh_miss[kDedupBlock / 64];
  __shared__ uint32_t sh_base;
  const uint64_t bal = __ballot(winner);
  const uint64_t bal_miss = __ballot(missed);
  const int lane = lane_id();
  const uint32_t rank_in_wave = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) { sh_wave[threadIdx.x >> 6] = (uint32_t)__popcll(bal); sh_miss[threadIdx.x >> 6] = (uint32_t)__popcll(bal_miss); }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0, miss = 0;
    for (int w = 0; w < kDedupBlock / 64; ++w) { const uint32_t c = sh_wave[w]; sh_wave[w] = run; run += c; miss += sh_miss[w]; }
    sh_base = run ? atomicAdd(&counts[1 + t], run) : 0u;
    // missed keys of the table, duplicates included: the per-table hit rate behind the insertion policy
    if (miss) atomicAdd(&counts[kTableMissBase + t], miss);
  }
  __syncthreads();
  if (winner) {
    const uint32_t uidx = sh_base + sh_wave[threadIdx.x >> 6] + rank_in_wave;
    slot_io[i] = -2 - (int32_t)uidx;
    const uint64_t pos = call->key_start[t] + uidx;
    uniq_keys_dev[pos] = key;
    uniq_keys_host[pos] = key;  // zero-copy store into pinned host memory
  }
}

// K_B2: a duplicate finds its representative through the set and copies its encoded index.
__global__ __launch_bounds__(kDedupBlock) void hps_miss_resolve_kernel(const CallDesc* __restrict__ call,
                                                                        int32_t* __restrict__ slot_io,
                                                                        const int32_t* __restrict__ set,
                                                                        uint64_t set_cap,
                                                                        const uint32_t* __restrict__ counts) {
  if (counts[0] == 0) return;
  uint32_t t;
  uint64_t begin, end;
  if (!block_to_table(call, kDedupBlock, &t, &begin, &end)) return;
  const uint64_t i = begin + threadIdx.x;
  if (i >= end) return;
  const int32_t s = slot_io[i];
  if (s != kSlotMiss) return;  // hit, or a representative (<= -2)
  const int64_t* __restrict__ keys = call->keys;
  const int64_t key = keys[i];
  const uint64_t mask = set_cap - 1;
  uint64_t h = set_hash(key, t) & mask;
  for (;;) {
    const int32_t e = set[h];
    if (e < 0) return;  // cannot happen: every missed key has a representative
    const uint64_t pi = (uint64_t)(uint32_t)e;
    if (pi >= call->key_start[t] && pi < call->key_start[t + 1] && keys[pi] == key) {
      // representative's slot was finalised by the previous kernel
      slot_io[i] = slot_io[pi];
      return;
    }
    h = (h + 1) & mask;
  }
}

// ------------------------------------------------------------------------------------------------
// K_C1: missed rows staging -> output.  Scans the slot array 64 keys per wave; missed keys of the wave
// are handed to the four 16-lane groups round-robin.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hps_miss_scatter_kernel(const CallDesc* __restrict__ call,
                                                                const TableCacheDev* __restrict__ tables,
                                                                const MissDesc* __restrict__ md,
                                                                const int32_t* __restrict__ slot_in,
                                                                const float* __restrict__ staging) {
  const uint64_t N = call->total_keys;
  const int T = (int)call->num_tables;
  const int lane = lane_id();
  const int g = lane >> 4, lig = lane & 15;
  const uint64_t waves_total = (uint64_t)gridDim.x * 4;
  const uint64_t chunks = (N + 63) / 64;
  for (uint64_t chunk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < chunks; chunk += waves_total) {
    const uint64_t i = chunk * 64 + (uint64_t)lane;
    const int32_t s = i < N ? slot_in[i] : 0;
    uint64_t todo = __ballot(s <= -2);
    while (todo) {
      // group g takes the g-th set bit (if any)
      uint64_t m = todo;
      int src = -1;
      for (int q = 0; q <= g && m; ++q) { src = __builtin_ctzll(m); m &= m - 1; }
      const bool have = __popcll(todo) > g;
      // drop up to 4 bits
      for (int q = 0; q < 4 && todo; ++q) todo &= todo - 1;
      const int32_t ss = __shfl(s, have ? src : 0, 64);
      if (!have) continue;
      const uint64_t gi = chunk * 64 + (uint64_t)src;
      const int t = find_table(call->key_start, T, gi);
      const uint32_t uidx = (uint32_t)(-2 - ss);
      if (uidx < md->chunk_lo[t] || uidx >= md->chunk_hi[t]) continue;  // other chunk of this call
      const uint32_t D = tables[t].dim;
      const float* row = staging + md->stage_off[t] + (uint64_t)(uidx - md->chunk_lo[t]) * D;
      float* dst = call->out[t] + (gi - call->key_start[t]) * D;
      // staging rows are packed (offset multiple of D): vector path needs D%4==0 and aligned out
      copy_row<true>(row, dst, D, lig, call->vec_ok[t] != 0 && (md->stage_off[t] & 3) == 0);
    }
  }
}

// K_D: async-insert mode — missed rows return the table's default vector
// (docs/architecture.md:32, docs/hierarchical_parameter_server.md:244-246).
__global__ __launch_bounds__(256) void hps_miss_fill_default_kernel(const CallDesc* __restrict__ call,
                                                                     const TableCacheDev* __restrict__ tables,
                                                                     const int32_t* __restrict__ slot_in,
                                                                     const uint32_t* __restrict__ table_mode) {
  // table_mode (optional): per table 1 = async insert (fill its misses), 0 = synchronous (leave them to K_C1)
  const uint64_t N = call->total_keys;
  const int T = (int)call->num_tables;
  const int lane = lane_id();
  const int g = lane >> 4, lig = lane & 15;
  const uint64_t waves_total = (uint64_t)gridDim.x * 4;
  const uint64_t chunks = (N + 63) / 64;
  for (uint64_t chunk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < chunks; chunk += waves_total) {
    const uint64_t i = chunk * 64 + (uint64_t)lane;
    const int32_t s = i < N ? slot_in[i] : 0;
    uint64_t todo = __ballot(s < 0);
    while (todo) {
      uint64_t m = todo;
      int src = -1;
      for (int q = 0; q <= g && m; ++q) { src = __builtin_ctzll(m); m &= m - 1; }
      const bool have = __popcll(todo) > g;
      for (int q = 0; q < 4 && todo; ++q) todo &= todo - 1;
      if (!have) continue;
      const uint64_t gi = chunk * 64 + (uint64_t)src;
      const int t = find_table(call->key_start, T, gi);
      if (table_mode && table_mode[t] == 0) continue;
      const uint32_t D = tables[t].dim;
      const float dv = tables[t].default_value;
      float* dst = call->out[t] + (gi - call->key_start[t]) * D;
      for (uint32_t c = (uint32_t)lig; c < D; c += 16) dst[c] = dv;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K_G: hit rows cache -> output from the slot indices of a probe-only K_A ("split probe": the miss path of the call —
// PCIe-bound — starts right after the 75-us probe and runs while this HBM-bound kernel moves the hits).
// Same decomposition as K_A: a wave takes 64 keys, each 16-lane group walks its 16 keys kU at a time; with the slots
// known up front every row load is independent (no bucket -> row dependency).
// Algorithmic bytes per key: 4 (slot) + 4D (row read) + 4D (row write) for hits, 4 for misses.
// ------------------------------------------------------------------------------------------------
// kFast: every table is 128 wide with 16-B aligned output (checked on the host): rows are staged in registers, 2*kU
// independent 16-B loads per lane in flight.  Otherwise: one row at a time with copy_row.
template <int kU, bool kFast>
__global__ __launch_bounds__(kProbeBlockThreads) void hps_gather_hits_kernel(const CallDesc* __restrict__ call,
                                                                             const TableCacheDev* __restrict__ tables,
                                                                             const int32_t* __restrict__ slot_in) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = (int)call->num_tables;
  TableLds* sh_tab = reinterpret_cast<TableLds*>(smem);
  uint64_t* sh_ks = reinterpret_cast<uint64_t*>(smem + sizeof(TableLds) * (size_t)T);
  load_tables_to_lds(sh_tab, sh_ks, call, tables, T);
  __syncthreads();
  const uint64_t N = call->total_keys;
  const int lane = lane_id();
  const int g = lane >> 4, lig = lane & 15;
  const uint64_t waves_total = (uint64_t)gridDim.x * (kProbeBlockThreads / 64);
  const uint64_t wave_global = (uint64_t)blockIdx.x * (kProbeBlockThreads / 64) + (threadIdx.x >> 6);
  const uint64_t chunks = (N + 63) / 64;
  for (uint64_t chunk = wave_global; chunk < chunks; chunk += waves_total) {
    const uint64_t i = chunk * 64 + (uint64_t)lane;
    const int32_t s = i < N ? slot_in[i] : -1;
    int t = (int)uniform_u32((uint32_t)find_table(sh_ks, T, chunk * 64));
    if (i < N) { while (i >= sh_ks[t + 1]) ++t; }
#pragma unroll 1
    for (int j0 = 0; j0 < 16; j0 += kU) {
      if (kFast) {
        f4 v[kU][2];
        float* dst[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int src = g * 16 + j0 + u;
          const int32_t ss = __shfl(s, src, 64);
          const int tt = __shfl(t, src, 64);
          dst[u] = nullptr;
          if (ss >= 0) {
            const float* row = sh_tab[tt].rows + (uint64_t)(uint32_t)ss * 128u;
            dst[u] = sh_tab[tt].out + (chunk * 64 + (uint64_t)src - sh_tab[tt].key_start) * 128u;
            v[u][0] = *reinterpret_cast<const f4*>(row + lig * 4);
            v[u][1] = *reinterpret_cast<const f4*>(row + 64 + lig * 4);
          }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          if (dst[u]) {
            __builtin_nontemporal_store(v[u][0], reinterpret_cast<f4*>(dst[u] + lig * 4));
            __builtin_nontemporal_store(v[u][1], reinterpret_cast<f4*>(dst[u] + 64 + lig * 4));
          }
        }
      } else {
#pragma unroll 1
        for (int u = 0; u < kU; ++u) {
          const int src = g * 16 + j0 + u;
          const int32_t ss = __shfl(s, src, 64);
          const int tt = __shfl(t, src, 64);
          if (ss < 0) continue;
          const TableLds& tb = sh_tab[tt];
          copy_row<true>(tb.rows + (uint64_t)(uint32_t)ss * tb.dim, tb.out + (chunk * 64 + (uint64_t)src - tb.key_start) * tb.dim,
                         tb.dim, lig, (tb.flags & 2u) != 0);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K_C2: insert unique missed (key,row) pairs.  One 16-lane group per key.  The group loads the
// bucket's keys and LRU stamps; if the key is already resident the row is refreshed in place;
// otherwise the victim is the slot with the smallest stamp that was not used in this epoch (empty
// slots carry stamp 0 and therefore go first).  Two groups of the same launch may want the same
// victim: the slot is claimed by atomicCAS(stamp: old -> epoch); the loser learns the new stamp from
// the CAS return value and moves to its next candidate.  Inserts never run concurrently with another
// kernel on the same cache (EmbeddingCache orders them with events), so plain loads of keys/stamps
// at kernel entry are coherent; only slots claimed inside this launch change under us, and those
// changes are observed through the CAS.
// `found[f]`==0 (key unknown to every parameter-server tier) -> not cached.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hps_cache_insert_kernel(const TableCacheDev* __restrict__ tables, uint32_t T,
                                                                const MissDesc* __restrict__ md,
                                                                const uint64_t* __restrict__ key_start,
                                                                const int64_t* __restrict__ uniq_keys,
                                                                const float* __restrict__ staging,
                                                                const uint8_t* __restrict__ found, uint32_t epoch,
                                                                uint32_t* __restrict__ stats) {
  const uint64_t total = md->useg_start[T];
  const int lane = lane_id();
  const int g = lane >> 4, lig = lane & 15;
  const uint64_t groups_total = (uint64_t)gridDim.x * 16;
  uint32_t n_dropped = 0, n_inserted = 0, n_refreshed = 0;  // counted by each group's lane 0
  for (uint64_t f = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4); f < total; f += groups_total) {
    const int t = find_table(md->useg_start, (int)T, f);
    const TableCacheDev tb = tables[t];
    if (tb.flags & 1u) continue;  // static cache: never insert
    const uint32_t u = md->chunk_lo[t] + (uint32_t)(f - md->useg_start[t]);
    const int64_t key = uniq_keys[key_start[t] + u];
    if (key == HPS_EMPTY_KEY) continue;
    const uint32_t D = tb.dim;
    const float* row = staging + md->stage_off[t] + (uint64_t)(u - md->chunk_lo[t]) * D;
    const uint32_t b = hps_bucket_of(key, tb.num_buckets);
    const uint64_t base = (uint64_t)b * kBucketSlots;
    const int64_t bk = tb.bucket_keys[base + lig];
    uint32_t st = tb.stamps[base + lig];

    const uint32_t present = (uint32_t)(__ballot(bk == key) >> (g * 16)) & 0xFFFFu;
    if (found && !found[f]) {
      // the key exists in no parameter-server tier (any more): never cache it, and if a refresh finds it still
      // resident, drop it so that later lookups fall through to the default value instead of a stale row
      if (present) {
        const int v = __builtin_ctz(present);
        uint32_t old = epoch;
        if (lig == v && st != epoch) old = atomicCAS(&tb.stamps[base + v], st, epoch);
        if (lig == v && st != epoch && old == st) { tb.bucket_keys[base + v] = HPS_EMPTY_KEY; tb.stamps[base + v] = 0; }
      }
      continue;
    }
    int victim = -1;
    if (present) {
      // Already resident (another session inserted it after our probe): refresh the row in place, but
      // only after claiming the slot like any other writer — a second group of this launch may be
      // about to evict exactly this slot, and an unclaimed refresh would interleave its row with the
      // evictor's key (key/row mismatch = poisoned slot).  Losing the claim just skips the refresh.
      const int v = __builtin_ctz(present);
      uint32_t old = epoch;
      if (lig == v && st != epoch) old = atomicCAS(&tb.stamps[base + v], st, epoch);
      old = __shfl(old, g * 16 + v, 64);
      const uint32_t expect = __shfl(st, g * 16 + v, 64);
      if (expect != epoch && old == expect) victim = v;
    } else {
      for (int tries = 0; tries < kBucketSlots; ++tries) {
        // min over the group of (stamp, lane) among slots not used in this epoch
        uint64_t cand = (st != epoch) ? (((uint64_t)st << 4) | (uint64_t)lig) : ~0ull;
        for (int off = 8; off > 0; off >>= 1) {
          const uint32_t lo = __shfl_xor((uint32_t)cand, off, 16);
          const uint32_t hi = __shfl_xor((uint32_t)(cand >> 32), off, 16);
          const uint64_t o = ((uint64_t)hi << 32) | lo;
          cand = o < cand ? o : cand;
        }
        if (cand == ~0ull) break;  // whole bucket is in use by this epoch
        const int v = (int)(cand & 15);
        uint32_t old = 0;
        if (lig == v) old = atomicCAS(&tb.stamps[base + v], st, epoch);
        old = __shfl(old, g * 16 + v, 64);
        const uint32_t expect = __shfl(st, g * 16 + v, 64);
        if (lig == v) st = (old == expect) ? epoch : old;
        if (old == expect) { victim = v; break; }
      }
      if (victim >= 0 && lig == victim) tb.bucket_keys[base + victim] = key;
    }
    if (victim < 0) {
      n_dropped += (lig == 0);  // bucket full of this epoch's keys (or lost the claim)
      continue;
    }
    float* dst = tb.rows + (base + (uint64_t)victim) * D;
    copy_row<false>(row, dst, D, lig, (D & 3u) == 0 && (md->stage_off[t] & 3) == 0);
    if (lig == 0) { if (present) ++n_refreshed; else ++n_inserted; }
  }
  // one atomic per block and counter (a per-key atomic on one word serialises the whole launch)
  __shared__ uint32_t sh_stat[3][4];
  uint32_t c0 = n_dropped, c1 = n_inserted, c2 = n_refreshed;
  for (int off = 32; off > 0; off >>= 1) {
    c0 += __shfl_down(c0, off, 64);
    c1 += __shfl_down(c1, off, 64);
    c2 += __shfl_down(c2, off, 64);
  }
  if (lane == 0) { sh_stat[0][threadIdx.x >> 6] = c0; sh_stat[1][threadIdx.x >> 6] = c1; sh_stat[2][threadIdx.x >> 6] = c2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const uint32_t v = sh_stat[threadIdx.x][0] + sh_stat[threadIdx.x][1] + sh_stat[threadIdx.x][2] + sh_stat[threadIdx.x][3];
    if (v) atomicAdd(&stats[threadIdx.x], v);
  }
}

// LRU epochs are 32-bit and advance once per lookup call; long before they wrap, every stamp is folded back:
// stamps younger than `keep_from` keep their order in [1, span], everything older becomes 1, never-used stays 0.
__global__ void hps_cache_renorm_kernel(uint32_t* stamps, uint64_t slots, uint32_t keep_from) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t s = stamps[i];
    if (s != 0) stamps[i] = s > keep_from ? s - keep_from + 1u : 1u;
  }
}

// Utility: fill bucket keys with EMPTY and stamps with 0.
__global__ void hps_cache_clear_kernel(int64_t* keys, uint32_t* stamps, uint64_t slots) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (uint64_t)gridDim.x * blockDim.x) {
    keys[i] = HPS_EMPTY_KEY;
    stamps[i] = 0;
  }
}

// Utility for tests / refresh: per-key residency (slot index or -1), no side effects.
__global__ void hps_cache_query_kernel(TableCacheDev tb, const int64_t* __restrict__ keys, uint64_t n,
                                       int32_t* __restrict__ slot) {
  const int lane = lane_id();
  const int g = lane >> 4, lig = lane & 15;
  const uint64_t groups_total = (uint64_t)gridDim.x * (blockDim.x / 16);
  for (uint64_t i = (uint64_t)blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4); i < n; i += groups_total) {
    const int64_t key = keys[i];
    const uint32_t b = hps_bucket_of(key, tb.num_buckets);
    const int64_t bk = tb.bucket_keys[(uint64_t)b * kBucketSlots + lig];
    const uint32_t m16 = (uint32_t)(__ballot(bk == key && key != HPS_EMPTY_KEY) >> (g * 16)) & 0xFFFFu;
    if (lig == 0) slot[i] = m16 ? (int32_t)(b * kBucketSlots + (uint32_t)__builtin_ctz(m16)) : -1;
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline uint32_t probe_grid(uint64_t N, int cu_count) {
  const uint64_t chunks = (N + 63) / 64;
  const uint64_t want = (chunks + 3) / 4;
  const uint64_t cap = (uint64_t)cu_count * 8;  // 8 blocks of 256 threads per CU
  return (uint32_t)(want < cap ? (want ? want : 1) : cap);
}

// Grid of the probe/gather kernel.  Every wave walks 64-key chunks with stride = number of waves, and all
// blocks are resident at once (<= 8 blocks of 4 waves per CU), so the launch ends when the waves with the
// most chunks end.  balanced: pick the wave count so that every wave gets the same number of chunks
// (config 2: 26,624 chunks -> 6,656 waves x 4 chunks) instead of filling the machine (8,192 waves, a quarter
// of which run a 4th chunk while the rest of the chip idles).
uint32_t ProbeGridBlocks(uint64_t N, int cu_count, bool balanced) {
  if (!balanced) return probe_grid(N, cu_count);
  const uint64_t chunks = (N + 63) / 64;
  const uint64_t cap_waves = (uint64_t)cu_count * 8 * (kProbeBlockThreads / 64);
  if (chunks <= cap_waves) return probe_grid(N, cu_count);
  const uint64_t k = (chunks + cap_waves - 1) / cap_waves;  // chunks per wave
  const uint64_t waves = (chunks + k - 1) / k;
  return (uint32_t)((waves + 3) / 4);
}

hipError_t LaunchProbeGather(const CallDesc* d_call, const TableCacheDev* d_tables, uint32_t num_tables, uint64_t N,
                             int32_t* d_slot, uint32_t* d_block_miss, uint32_t grid, int unroll, hipStream_t stream) {
  const size_t smem = sizeof(TableLds) * (size_t)num_tables + sizeof(uint64_t) * ((size_t)num_tables + 1) +
                      sizeof(uint32_t) * (kProbeBlockThreads / 64);
  // `unroll` encodes the variant: U + 100*mode + 1000*stamp_mode  (U in {1,2,4,8};
  //  mode 0 full unroll, 1 rolled, 2 rolled + pipelined; stamp_mode 0 every hit, 1 = 1/4 of hits, 2 = 1/16)
  const int U = unroll % 100, mode = (unroll / 100) % 10, smode = unroll / 1000;
#define HPS_PG(UU, OO, SS)                                                                                          \
  hipLaunchKernelGGL((hps_probe_gather_kernel<UU, OO, SS>), dim3(grid), dim3(kProbeBlockThreads), smem, stream, d_call, \
                     d_tables, d_slot, d_block_miss)
#define HPS_PG_S(UU, OO) \
  do { if (smode == 0) HPS_PG(UU, OO, 0); else if (smode == 1) HPS_PG(UU, OO, 2); else HPS_PG(UU, OO, 4); } while (0)
#define HPS_PG_U(UU)                                 \
  do {                                               \
    if (mode == 1) HPS_PG_S(UU, 1);                  \
    else if (mode == 2) HPS_PG_S(UU, 0);             \
    else HPS_PG_S(UU, 16 / UU);                      \
  } while (0)
  switch (U) {
    case 1: HPS_PG_U(1); break;
    case 2: HPS_PG_U(2); break;
    case 8: HPS_PG_U(8); break;
    default: HPS_PG_U(4); break;
  }
#undef HPS_PG_U
#undef HPS_PG_S
#undef HPS_PG
  return hipGetLastError();
}

static inline uint32_t table_aligned_blocks(const uint64_t* key_start, uint32_t T, uint32_t per_block) {
  uint64_t nb = 0;
  for (uint32_t t = 0; t < T; ++t) nb += (key_start[t + 1] - key_start[t] + per_block - 1) / per_block;
  return (uint32_t)(nb ? nb : 1);
}

hipError_t LaunchMissDedup(const CallDesc* d_call, const uint64_t* h_key_start, uint32_t T, uint32_t probe_blocks,
                           int32_t* d_slot, const uint32_t* d_block_miss, int32_t* d_set, uint64_t set_cap,
                           uint32_t* d_counts, int64_t* d_uniq_keys, int64_t* uniq_keys_host_mapped, int cu_count,
                           hipStream_t stream) {
  hipLaunchKernelGGL(hps_miss_begin_kernel, dim3((uint32_t)cu_count * 4), dim3(256), 0, stream, d_block_miss,
                     probe_blocks, d_set, set_cap, d_counts, T);
  const uint32_t nb = table_aligned_blocks(h_key_start, T, kDedupBlock);
  hipLaunchKernelGGL(hps_miss_dedup_kernel, dim3(nb), dim3(kDedupBlock), 0, stream, d_call, d_slot, d_set, set_cap,
                     d_counts, d_uniq_keys, uniq_keys_host_mapped);
  hipLaunchKernelGGL(hps_miss_resolve_kernel, dim3(nb), dim3(kDedupBlock), 0, stream, d_call, d_slot,
                     (const int32_t*)d_set, set_cap, (const uint32_t*)d_counts);
  return hipGetLastError();
}

hipError_t LaunchGatherHits(const CallDesc* d_call, const TableCacheDev* d_tables, uint32_t num_tables, uint64_t N,
                            const int32_t* d_slot, uint32_t grid, bool all_128_aligned, hipStream_t stream) {
  if (N == 0) return hipSuccess;
  const size_t lds = sizeof(TableLds) * num_tables + sizeof(uint64_t) * (num_tables + 1);
  if (all_128_aligned)
    hipLaunchKernelGGL((hps_gather_hits_kernel<4, true>), dim3(grid), dim3(kProbeBlockThreads), lds, stream, d_call, d_tables, d_slot);
  else
    hipLaunchKernelGGL((hps_gather_hits_kernel<4, false>), dim3(grid), dim3(kProbeBlockThreads), lds, stream, d_call, d_tables, d_slot);
  return hipGetLastError();
}

hipError_t LaunchMissScatter(const CallDesc* d_call, const TableCacheDev* d_tables, const MissDesc* d_md, uint64_t N,
                             const int32_t* d_slot, const float* d_staging, int cu_count, hipStream_t stream) {
  const uint32_t grid = probe_grid(N, cu_count);
  hipLaunchKernelGGL(hps_miss_scatter_kernel, dim3(grid), dim3(256), 0, stream, d_call, d_tables, d_md, d_slot,
                     d_staging);
  return hipGetLastError();
}

hipError_t LaunchMissFillDefault(const CallDesc* d_call, const TableCacheDev* d_tables, uint64_t N,
                                 const int32_t* d_slot, const uint32_t* d_table_mode, int cu_count, hipStream_t stream) {
  const uint32_t grid = probe_grid(N, cu_count);
  hipLaunchKernelGGL(hps_miss_fill_default_kernel, dim3(grid), dim3(256), 0, stream, d_call, d_tables, d_slot, d_table_mode);
  return hipGetLastError();
}

hipError_t LaunchCacheInsert(const TableCacheDev* d_tables, uint32_t T, const MissDesc* d_md, uint64_t total_unique,
                             const uint64_t* d_key_start, const int64_t* d_uniq_keys, const float* d_staging,
                             const uint8_t* d_found, uint32_t epoch, uint32_t* d_stats, int cu_count,
                             hipStream_t stream) {
  if (total_unique == 0) return hipSuccess;
  uint64_t want = (total_unique + 15) / 16;
  const uint64_t cap = (uint64_t)cu_count * 8;
  if (want > cap) want = cap;
  hipLaunchKernelGGL(hps_cache_insert_kernel, dim3((uint32_t)want), dim3(256), 0, stream, d_tables, T, d_md,
                     d_key_start, d_uniq_keys, d_staging, d_found, epoch, d_stats);
  return hipGetLastError();
}

hipError_t LaunchCacheClear(int64_t* d_keys, uint32_t* d_stamps, uint64_t slots, hipStream_t stream) {
  uint64_t want = (slots + 255) / 256;
  if (want > 4096) want = 4096;
  if (want == 0) want = 1;
  hipLaunchKernelGGL(hps_cache_clear_kernel, dim3((uint32_t)want), dim3(256), 0, stream, d_keys, d_stamps, slots);
  return hipGetLastError();
}

hipError_t LaunchCacheRenorm(uint32_t* d_stamps, uint64_t slots, uint32_t keep_from, hipStream_t stream) {
  uint64_t want = (slots + 255) / 256;
  if (want > 4096) want = 4096;
  if (want == 0) want = 1;
  hipLaunchKernelGGL(hps_cache_renorm_kernel, dim3((uint32_t)want), dim3(256), 0, stream, d_stamps, slots, keep_from);
  return hipGetLastError();
}

hipError_t LaunchCacheQuery(const TableCacheDev& tb, const int64_t* d_keys, uint64_t n, int32_t* d_slot,
                            hipStream_t stream) {
  if (n == 0) return hipSuccess;
  uint64_t want = (n + 15) / 16;
  if (want > 8192) want = 8192;
  hipLaunchKernelGGL(hps_cache_query_kernel, dim3((uint32_t)want), dim3(256), 0, stream, tb, d_keys, n, d_slot);
  return hipGetLastError();
}

}  // namespace hps
