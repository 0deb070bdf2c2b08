// HIP kernels of the GPU embedding cache, written for gfx950 (MI355X, wave64, 256 CUs / 8 XCDs).
//
// There is nothing to port: /root/reference contains no device code; the kernels are
// defined by function, from docs/hierarchical_parameter_server.md:65-78 (unique keys -> cache query -> miss ->
// parameter server -> insert) and the north star in BASELINE.json.
//
//   K_P  hps_probe_tile        per tile of 1,024 keys of one table: input dedup in LDS, one bucket probe per
//                              tile-unique key, slot index for every key, the tile's miss lists      (HBM, latency)
//   K_M  hps_miss_unique       call-wide unique missed keys per table from the tiles' short miss lists
//   K_H  hps_unique_hits       the call's unique hit keys per table = distinct slots hit, counted in LDS bitmaps
//                              (only when the insertion policy needs the hit rate)
//   K_G  hps_gather_hits       hit rows cache -> output from the slot indices   HBM-bound, the roofline kernel
//   K_C1 hps_miss_scatter      missed rows: staging -> output (walks the tiles' miss lists, not the slot array)
//   K_C2 hps_cache_insert      unique missed (key,row) -> bucket, LRU victim claimed by CAS
//   K_D  hps_miss_fill_default async-insert mode: missed rows = default vector
//
// Work decomposition of every row mover: one 16-lane group per key (4 keys per wave at a time), a D=128 row as
// 2 x 16 B per lane (two fully coalesced 256-B segments).  A bucket line (14 keys + 14 one-byte recency stamps =
// 128 B, device_types.h) is read by an 8-lane group with a single 16-B load per lane: tools/micro/probe_width.hip
// measures 30.7 us for 1.1 M random lines that way against 39.0 us with sixteen 8-B loads.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "device_types.h"
#include "kernels.h"

namespace hps {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// recency stamp of slot v from the two stamp words of a bucket line
__device__ __forceinline__ uint32_t stamp_of(uint64_t a, uint64_t b, uint32_t v) {
  return (uint32_t)((v < 8 ? a : b) >> (8u * (v & 7u))) & 0xFFu;
}

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

__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  return ((uint64_t)uniform_u32((uint32_t)(v >> 32)) << 32) | uniform_u32((uint32_t)v);
}

// Per-table values the gather kernel needs; kept in LDS once per block.
struct __attribute__((aligned(16))) TableLds {
  const float* rows;
  float* out;          // output slice of this table for this call
  uint64_t key_start;  // first global key index of this table
  uint32_t dim;
  uint32_t flags;      // bit1 vec_ok
};

__device__ __forceinline__ void load_tables_to_lds(TableLds* sh, uint64_t* sh_ks, const CallDesc* call,
                                                   const TableCacheDev* tables, int T) {
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    TableLds e;
    e.rows = tables[t].rows;
    e.out = call->out[t];
    e.key_start = call->key_start[t];
    e.dim = tables[t].dim;
    e.flags = call->vec_ok[t] ? 2u : 0u;
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

// Wave-aggregated append to a list whose fill count lives in LDS: every lane with `take` gets a distinct position.
// Must be called by all lanes of the wave (take = false for lanes with nothing to append).
__device__ __forceinline__ uint32_t lds_append(uint32_t* sh_count, bool take) {
  const uint64_t bal = __ballot(take);
  if (bal == 0) return 0;
  const int lane = lane_id();
  const int leader = __builtin_ctzll(bal);
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(sh_count, (uint32_t)__popcll(bal));
  base = (uint32_t)__shfl((int)base, leader, 64);
  return base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
}

// position of a (table, key) in a session's call-wide miss set (K_P's tail / K_M)
__device__ __forceinline__ uint64_t set_hash(int64_t key, uint32_t t) {
  return hps_mix64((uint64_t)key ^ ((uint64_t)(t + 1) * 0xD6E8FEB86659FD93ull));
}

// ------------------------------------------------------------------------------------------------
// K_P: probe, one workgroup per tile (<= kTileKeys consecutive keys of one table).
//   1. keys -> LDS (coalesced), one hash per key: high half = cache bucket, low bits = LDS set position
//   2. kDedup: tile-local input dedup — 32-bit LDS CAS claims a set entry with the key's tile-local index;
//      a key that finds an equal key there takes that key as its representative.  Under the Zipf-like
//      distributions of recommender traffic a third to a half of a tile's keys are duplicates.
//   3. representatives only: 8-lane group per key, kU independent 128-B bucket lines in flight per group (the probe
//      is bound by the rate of random line requests, not by their bytes: 64-B granules are no faster,
//      tools/micro/probe_width.hip); a hit whose slot's recency stamp is not the current unit's rewrites that byte
//   4. every key takes its representative's result: slot[i] >= 0, or -2 - m with m the representative's position in
//      the tile's miss list; the lists the later kernels walk (missed representatives' keys, missed keys as sent)
//      are compacted in the tile's own region — no global atomic in this kernel.
// Algorithmic bytes per key: 8 (key) + 4 (slot); overhead: one 128-B bucket line per tile-unique key.
// ------------------------------------------------------------------------------------------------
template <bool kDedup, int kU, int kThreads, bool kTail>
__global__ __launch_bounds__(kThreads) void hps_probe_tile_kernel(const CallDesc* __restrict__ call,
                                                                            const TableCacheDev* __restrict__ tables,
                                                                            const CallWork w) {
  __shared__ int64_t sh_key[kTileKeys];
  __shared__ uint32_t sh_bkt[kTileKeys];
  __shared__ __attribute__((aligned(8))) uint32_t sh_set[kDedup ? kTileSet : 1];   // (kTail: reused as the tile's missed keys, int64)
  __shared__ uint16_t sh_rep[kTileKeys];
  __shared__ int32_t sh_slot[kTileKeys];
  __shared__ uint16_t sh_list[kTileKeys];
  __shared__ uint32_t sh_cnt[4];  // representatives, missed representatives, missed keys as sent

  // XCD-aware tile order (w.xcd_tiles, speed only): workgroup b runs on XCD b % 8, and the tiles are table-major — with tile = b
  // every XCD sees every table's tiles, and the bucket lines of the keys a batch repeats (the hot head of a Zipf-like
  // distribution turns up in most of a table's 64 tiles) are fetched from HBM by all eight L2s.  Giving XCD x the x-th eighth of
  // the tiles keeps a table's tiles, and with them its hot bucket lines, in ONE 4-MB L2 (the gather kernel walks its chunks
  // the same way).  One-to-one without holes: XCD x owns q + (x < r) tiles from x*q + min(x, r), q = tiles / 8, r = tiles % 8.
  const uint32_t tq = w.num_tiles >> 3, tr = w.num_tiles & 7u, bx = blockIdx.x & 7u;
  const uint32_t tile = w.xcd_tiles ? bx * tq + (bx < tr ? bx : tr) + (blockIdx.x >> 3) : blockIdx.x;
  const TileDesc td = w.tiles[tile];
  const TableCacheDev tb = tables[td.table];
  const uint32_t n = td.count;
  const uint32_t tid = threadIdx.x;
  const uint32_t* __restrict__ keys32 = call->keys32;   // wave-uniform: one of the three loads below
  const uint8_t* __restrict__ keys24 = call->keys24;
  const int64_t* __restrict__ keys = call->keys + td.begin;
  const uint32_t stamp8 = call->stamp8;
  const bool skip_empty = call->skip_empty_keys != 0;
  const int64_t kbase = call->key_base[td.table];   // narrowed keys are offsets from their table's base (key_pack.h)
  constexpr int kPerThread = kTileKeys / kThreads;

  if (tid < 4) sh_cnt[tid] = 0;
  if (kDedup) {
    for (uint32_t e = tid; e < (uint32_t)kTileSet; e += kThreads) sh_set[e] = 0xFFFFFFFFu;
  }
  int64_t k[kPerThread];
  uint32_t hlo[kPerThread];
#pragma unroll
  for (int q = 0; q < kPerThread; ++q) {
    const uint32_t j = tid + (uint32_t)q * kThreads;
    if (keys24) {
      const uint8_t* p = keys24 + 3ull * (td.begin + j);   // consecutive lanes read consecutive 3-byte keys
      k[q] = j < n ? kbase + (int64_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16)) : HPS_EMPTY_KEY;
    } else if (keys32) k[q] = j < n ? kbase + (int64_t)(uint64_t)keys32[td.begin + j] : HPS_EMPTY_KEY;
    else k[q] = j < n ? keys[j] : HPS_EMPTY_KEY;
  }
#pragma unroll
  for (int q = 0; q < kPerThread; ++q) {
    const uint32_t j = tid + (uint32_t)q * kThreads;
    const uint64_t h = hps_mix64((uint64_t)k[q]);
    hlo[q] = (uint32_t)h;
    if (j < n) {
      sh_key[j] = k[q];
      sh_bkt[j] = (uint32_t)(((h >> 32) * (uint64_t)tb.num_buckets) >> 32);  // == hps_bucket_of(key, num_buckets)
    }
  }
  __syncthreads();

  // ---- 2. representatives ----
#pragma unroll
  for (int q = 0; q < kPerThread; ++q) {
    const uint32_t j = tid + (uint32_t)q * kThreads;
    uint32_t rep = j;
    const bool pad = skip_empty && k[q] == HPS_EMPTY_KEY;   // padding of the sharded exchange: represents nothing
    if (pad && j < n) sh_slot[j] = kSlotMiss;
    if (kDedup && j < n && !pad) {
      uint32_t e = hlo[q] & (uint32_t)(kTileSet - 1);
      for (;;) {
        const uint32_t prev = atomicCAS(&sh_set[e], 0xFFFFFFFFu, j);
        if (prev == 0xFFFFFFFFu) break;
        if (sh_key[prev] == k[q]) { rep = prev; break; }
        e = (e + 1) & (uint32_t)(kTileSet - 1);
      }
    }
    const bool is_rep = j < n && rep == j && !pad;
    if (j < n) sh_rep[j] = (uint16_t)rep;
    const uint32_t pos = lds_append(&sh_cnt[0], is_rep);
    if (is_rep) sh_list[pos] = (uint16_t)j;
  }
  __syncthreads();

  // ---- 3. one bucket probe per representative ----
  const uint32_t nrep = sh_cnt[0];
  const int lane = lane_id();
  const int g8 = (int)(tid >> 3), gw = lane >> 3, lig = lane & 7;
  for (uint32_t r0 = (uint32_t)g8 * kU; r0 < nrep; r0 += (kThreads / kProbeLanes) * kU) {
    uint32_t jj[kU], bb[kU];
    u64x2 ln[kU];   // lanes 0..6: two keys each; lane 7: the two stamp words
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const uint32_t r = r0 + u < nrep ? r0 + u : r0;
      jj[u] = sh_list[r];
      bb[u] = sh_bkt[jj[u]];
      ln[u] = *reinterpret_cast<const u64x2*>(tb.lines + (uint64_t)bb[u] * kLineWords + lig * 2);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t key = sh_key[jj[u]];
      const bool live = lig < 7 && key != HPS_EMPTY_KEY;
      const uint32_t m0 = (uint32_t)(__ballot(live && (int64_t)ln[u].x == key) >> (gw * 8)) & 0xFFu;
      const uint32_t m1 = (uint32_t)(__ballot(live && (int64_t)ln[u].y == key) >> (gw * 8)) & 0xFFu;
      if (lig == 7 && r0 + u < nrep) {   // the lane that holds the stamps finishes the probe
        int32_t s = kSlotMiss;
        if (m0 | m1) {
          const uint32_t v = m0 ? 2u * (uint32_t)__builtin_ctz(m0) : 2u * (uint32_t)__builtin_ctz(m1) + 1u;
          s = (int32_t)(bb[u] * kBucketSlots + v);
          // recency: one byte of the line just read, rewritten only when it is not already this unit's stamp
          if (!(tb.flags & 1u) && stamp_of(ln[u].x, ln[u].y, v) != stamp8)
            reinterpret_cast<uint8_t*>(tb.lines + (uint64_t)bb[u] * kLineWords + kBucketSlots)[v] = (uint8_t)stamp8;
        }
        sh_slot[jj[u]] = s;
      }
    }
  }
  __syncthreads();

  // ---- 4a. missed representatives take their place in the tile's miss list ----
  const uint32_t region = tile * (uint32_t)kTileKeys;
#pragma unroll
  for (int q = 0; q < kPerThread; ++q) {
    const uint32_t j = tid + (uint32_t)q * kThreads;
    const bool is_rep = j < n && sh_rep[j] == (uint16_t)j && !(skip_empty && k[q] == HPS_EMPTY_KEY);
    const int32_t s = is_rep ? sh_slot[j] : 0;
    const bool miss = is_rep && s < 0;
    const uint32_t pos = lds_append(&sh_cnt[1], miss);
    if (miss) {
      if (kTail) {
        reinterpret_cast<int64_t*>(sh_set)[pos] = k[q];   // the set is dead since step 2; the tail writes miss_key itself
      } else {
        w.miss_key[region + pos] = k[q];
      }
      sh_slot[j] = -2 - (int32_t)(region + pos);
    }
  }
  __syncthreads();
  // ---- 4b. every key takes its representative's result ----
#pragma unroll
  for (int q = 0; q < kPerThread; ++q) {
    const uint32_t j = tid + (uint32_t)q * kThreads;
    const int32_t s = j < n ? sh_slot[sh_rep[j]] : 0;
    if (j < n) w.slot[td.begin + j] = s;   // (padding: its own kSlotMiss)
    const uint32_t pos = lds_append(&sh_cnt[2], s <= -2);
    if (s <= -2) { w.sent_i[region + pos] = (int32_t)(td.begin + j); w.sent_m[region + pos] = -2 - s; }
  }
  __syncthreads();
  if (tid == 0) {
    w.tile_cnt[tile * 4 + kTileCntRepMiss] = sh_cnt[1];
    w.tile_cnt[tile * 4 + kTileCntSentMiss] = sh_cnt[2];
  }
  // ---- 5. (kTail) call-wide unique misses: what hps_miss_unique_kernel does in a launch of its own, done here by the tile's
  // first wave while the other waves retire.  Round 3: the separate kernel cost 15-17 us per call in the timed region for a few
  // microseconds of dependent accesses per tile (launch gap, tile_cnt / miss_key round trips); the price of folding it in is
  // the hand-off inside a launch: a loser reads the WINNER's key, which another workgroup wrote moments ago.  The lane that
  // publishes a set entry therefore writes its key itself, as a device-scope store (write-through past the XCD's L2), and
  // waits for that store before the compare-and-swap; the loser's read is a device-scope load whose address comes out of the
  // entry.  No fence: a device-scope release / acquire fence on gfx950 writes back / invalidates the whole L2 of the XCD —
  // tried first: 500 us instead of 45 for this kernel.
  // What this rests on, at the ISA level (gfx950 = gfx9 family, LLVM AMDGPU memory model for gfx942/gfx950): an atomic store
  // of agent scope is a global_store with sc1 — it writes through the XCD's L2 and is counted by vmcnt until the write is
  // acknowledged at the device's point of coherence; an atomic load of agent scope is a global_load with sc1, which an L2
  // does not serve from a line it cannot vouch for.  The compiler's own release sequence is `buffer_wbl2 sc1; s_waitcnt
  // vmcnt(0)`: the write-back is there for EARLIER PLAIN stores; the only store a loser depends on is the sc1 store itself,
  // so the wait alone orders it before the compare-and-swap, and the loser's load carries an address dependency on the entry
  // it read.  tests/test_gpu_lookup.py::test_fused_unique_tail_against_the_separate_kernel_on_full_size_duplicate_heavy_batches
  // hammers exactly this hand-off (1,664 tiles on all XCDs, every tile of a table missing the same keys).
  if (kTail) {
    if (tid >= 64) return;
    const uint32_t M = sh_cnt[1], S = sh_cnt[2];
    if ((M | S) == 0) return;
    const uint32_t t = td.table;
    const unsigned long long tag = (unsigned long long)w.call_tag << 32;
    const uint64_t ks = call->key_start[t];
    const int64_t* sh_mk = reinterpret_cast<const int64_t*>(sh_set);
    for (uint32_t r0 = 0; r0 < M; r0 += 64) {
      const uint32_t r = r0 + (uint32_t)lane;
      const bool active = r < M;
      const uint32_t m = region + r;
      bool winner = false;
      uint32_t rep = m;
      int64_t key = 0;
      if (active) {
        key = sh_mk[r];
        __hip_atomic_store(&w.miss_key[m], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the keys are out before any of this wave's entries can be seen
      if (active) {
        uint64_t h = set_hash(key, t) & w.set_mask;
        unsigned long long cur = __hip_atomic_load(&w.set[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
          if ((cur & 0xFFFFFFFF00000000ull) != tag) {  // free: left by an earlier call
            const unsigned long long prev = atomicCAS(&w.set[h], cur, tag | m);
            if (prev == cur) { winner = true; break; }
            cur = prev;
            continue;
          }
          const uint32_t pm = (uint32_t)cur;
          const int64_t other = __hip_atomic_load(&w.miss_key[pm], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (other == key && w.tiles[pm / (uint32_t)kTileKeys].table == t) { rep = pm; break; }
          h = (h + 1) & w.set_mask;
          cur = __hip_atomic_load(&w.set[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      const uint64_t bal = __ballot(winner);
      uint32_t base = 0;
      if (bal) {
        if (lane == 0) base = atomicAdd(&w.acc[AccTableWord(t, kAccUniqMiss)], (uint32_t)__popcll(bal));
        base = uniform_u32(base);
      }
      if (active) {
        w.rep_of[m] = (int32_t)rep;
        if (winner) {
          const uint32_t u = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
          w.uidx_of[m] = (int32_t)u;
          w.uniq_keys[ks + u] = key;
          if (w.uniq_keys_host32) w.uniq_keys_host32[ks + u] = (uint32_t)(key - kbase);
          else if (w.uniq_keys_host) w.uniq_keys_host[ks + u] = key;
        }
      }
    }
    if (lane == 0 && S) atomicAdd(&w.acc[AccTableWord(t, kAccSentMiss)], S);
  }
}

// ------------------------------------------------------------------------------------------------
// K_M: call-wide unique missed keys per table.  One wave per tile walks the tile's missed representatives
// (a few dozen at 95 % hit): each claims an entry of the session's open-addressing set with a 64-bit CAS on
// (call tag, m); an entry carrying another call's tag is free, so the set is never cleared.  The winner is the
// representative of its (table, key) for the whole call: winners are ranked inside the wave (ballot) and the wave takes
// its range of the table's unique segment with ONE atomic on the table's own accumulator line.  A loser learns
// the winner's m from the CAS and records it (rep_of).  Unique keys go to HBM and, zero-copy, to the pinned host
// array the parameter-server threads read.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hps_miss_unique_kernel(const CallDesc* __restrict__ call,
                                                               const TableCacheDev* __restrict__ tables, const CallWork w) {
  // ONE WAVE per tile (four tiles per workgroup): a tile's miss list is a few dozen entries, and with a wave as the unit
  // the ranking needs no LDS and no barrier — every dependent global access removed from this kernel is a microsecond
  // on the critical path between the probe and the miss counts reaching the host.
  const uint32_t tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= w.num_tiles) return;
  const int lane = lane_id();
  const uint32_t region = tile * (uint32_t)kTileKeys;
  // the first list entries are loaded before the count is known (a slot past the end holds an old key: harmless)
  const int64_t first_key = w.miss_key[region + (uint32_t)lane];
  const uint32_t M = uniform_u32(w.tile_cnt[tile * 4 + kTileCntRepMiss]);
  const uint32_t S = uniform_u32(w.tile_cnt[tile * 4 + kTileCntSentMiss]);
  if ((M | S) == 0) return;
  const uint32_t t = uniform_u32(w.tiles[tile].table);
  const unsigned long long tag = (unsigned long long)w.call_tag << 32;
  const uint64_t ks = call->key_start[t];

  for (uint32_t r0 = 0; r0 < M; r0 += 64) {
    const uint32_t r = r0 + (uint32_t)lane;
    const bool active = r < M;
    const uint32_t m = region + r;
    bool winner = false;
    uint32_t rep = m;
    int64_t key = 0;
    if (active) {
      key = r0 == 0 ? first_key : w.miss_key[m];
      uint64_t h = set_hash(key, t) & w.set_mask;
      unsigned long long cur = __hip_atomic_load(&w.set[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        if ((cur & 0xFFFFFFFF00000000ull) != tag) {  // free: left by an earlier call
          const unsigned long long prev = atomicCAS(&w.set[h], cur, tag | m);
          if (prev == cur) { winner = true; break; }
          cur = prev;  // somebody took it meanwhile: look at what is there now
          continue;
        }
        const uint32_t pm = (uint32_t)cur;
        if (w.miss_key[pm] == key && w.tiles[pm / (uint32_t)kTileKeys].table == t) { rep = pm; break; }
        h = (h + 1) & w.set_mask;
        cur = __hip_atomic_load(&w.set[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    const uint64_t bal = __ballot(winner);
    uint32_t base = 0;
    if (bal) {
      if (lane == 0) base = atomicAdd(&w.acc[AccTableWord(t, kAccUniqMiss)], (uint32_t)__popcll(bal));
      base = uniform_u32(base);
    }
    if (active) {
      w.rep_of[m] = (int32_t)rep;
      if (winner) {
        const uint32_t u = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        w.uidx_of[m] = (int32_t)u;
        w.uniq_keys[ks + u] = key;
        // zero-copy store into pinned host memory (host-gather tier), at 4 bytes when the request's keys were narrowed
        if (w.uniq_keys_host32) w.uniq_keys_host32[ks + u] = (uint32_t)(key - call->key_base[t]);
        else if (w.uniq_keys_host) w.uniq_keys_host[ks + u] = key;
      }
    }
  }
  if (lane == 0 && S) atomicAdd(&w.acc[AccTableWord(t, kAccSentMiss)], S);
}

// ------------------------------------------------------------------------------------------------
// K_H: the call's unique HIT keys per table, for the insertion policy (hit rate over the call's unique keys,
// docs/hierarchical_parameter_server.md:69) — taken only for thresholds inside (0,1).  A key that hit owns exactly one
// cache slot, so unique hit keys = distinct non-negative values among the table's slot words.  One workgroup per
// (table, range of kHitPartBits = 1 M slots): it walks the table's slot words (coalesced, L2-resident: K_P has just written
// them), marks the slots of its range in an LDS bitmap (duplicates set the same bit) and counts the bits: no sort, no
// global set, one global atomic per workgroup.  Rounds 2-3 marked a 4-byte claim word per slot from the probe kernel
// (random stores into a 277-MB array), listed the hit representatives per tile and re-read the claim words in a K_M
// launch of its own — which also kept the call-wide unique misses out of the probe kernel's tail: probe pair 87 us
// instead of 44 on the headline workload.
// Cost: parts(t) = ceil(slots_t / kHitPartBits) workgroups per table, each reading the table's n_t slot words.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kHitPartBits = 1u << 20;   // 128 KB of LDS per workgroup (one workgroup per CU)
constexpr int kHitThreads = 1024;

__global__ __launch_bounds__(kHitThreads) void hps_unique_hits_kernel(const CallDesc* __restrict__ call,
                                                                      const TableCacheDev* __restrict__ tables,
                                                                      const int32_t* __restrict__ slot, uint32_t* __restrict__ acc) {
  extern __shared__ __attribute__((aligned(16))) uint32_t hit_bits[];   // [kHitPartBits / 32]
  __shared__ uint32_t sh_sum[kHitThreads / 64];
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  typedef int32_t i4 __attribute__((ext_vector_type(4)));
  // which (table, part) is this workgroup?  Every table's part count is fetched by its own thread (a scan that loaded the
  // descriptors one after the other was 26 dependent round trips = 15 of this kernel's first 20 us), the scan runs over LDS.
  __shared__ uint32_t sh_parts[kMaxTables];
  const uint32_t T = call->num_tables;
  for (uint32_t tt = threadIdx.x; tt < T; tt += kHitThreads) {
    const uint64_t slots = (uint64_t)tables[tt].num_buckets * kBucketSlots;
    sh_parts[tt] = call->key_start[tt + 1] > call->key_start[tt] ? (uint32_t)((slots + kHitPartBits - 1) / kHitPartBits) : 0u;
  }
  for (uint32_t e = threadIdx.x; e < kHitPartBits / 128; e += kHitThreads) reinterpret_cast<u4*>(hit_bits)[e] = u4{0u, 0u, 0u, 0u};
  __syncthreads();
  uint32_t b = blockIdx.x, t = 0;
  for (; t < T; ++t) {
    if (b < sh_parts[t]) break;
    b -= sh_parts[t];
  }
  if (t >= T) return;
  const uint64_t lo = (uint64_t)b * kHitPartBits;
  const uint64_t i0 = call->key_start[t], i1 = call->key_start[t + 1];
  auto mark = [&](int32_t sl) {
    const uint64_t r = (uint64_t)(uint32_t)sl - lo;   // (a negative slot word — miss or padding — is out of every range)
    // (a hot key's slot comes by thousands of times: once its bit is set the atomic — same-address LDS atomics serialise — is skipped)
    if (sl >= 0 && r < kHitPartBits && !((hit_bits[r >> 5] >> (r & 31u)) & 1u)) atomicOr(&hit_bits[r >> 5], 1u << (r & 31u));
  };
  // the table's slot words: 16 B per lane, sixteen loads in flight per lane; the unaligned head and the tail one word at a time
  const uint64_t a0 = (i0 + 3) & ~(uint64_t)3, a1 = i1 & ~(uint64_t)3;
  if (a0 <= a1) {
    for (uint64_t i = i0 + threadIdx.x; i < a0; i += kHitThreads) mark(slot[i]);
    for (uint64_t i = a1 + threadIdx.x; i < i1; i += kHitThreads) mark(slot[i]);
    const i4* __restrict__ s4 = reinterpret_cast<const i4*>(slot + a0);
    const uint64_t n4 = (a1 - a0) >> 2;
    constexpr int kLoads = 16;   // (65,536 keys per table = one batch of 16 loads per lane: one L2 round trip, not four)
    for (uint64_t q = threadIdx.x; q < n4; q += (uint64_t)kLoads * kHitThreads) {
      i4 v[kLoads];
#pragma unroll
      for (int u = 0; u < kLoads; ++u) {
        const uint64_t qq = q + (uint64_t)u * kHitThreads;
        v[u] = qq < n4 ? s4[qq] : i4{-1, -1, -1, -1};
      }
#pragma unroll
      for (int u = 0; u < kLoads; ++u) { mark(v[u].x); mark(v[u].y); mark(v[u].z); mark(v[u].w); }
    }
  } else {
    for (uint64_t i = i0 + threadIdx.x; i < i1; i += kHitThreads) mark(slot[i]);
  }
  __syncthreads();
  uint32_t mine = 0;
  for (uint32_t e = threadIdx.x; e < kHitPartBits / 128; e += kHitThreads) {
    const u4 w4 = reinterpret_cast<const u4*>(hit_bits)[e];
    mine += (uint32_t)(__popc(w4.x) + __popc(w4.y) + __popc(w4.z) + __popc(w4.w));
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
  if ((threadIdx.x & 63) == 0) sh_sum[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t v = 0;
    for (int k = 0; k < kHitThreads / 64; ++k) v += sh_sum[k];
    if (v) atomicAdd(&acc[AccTableWord(t, kAccUniqHit)], v);
  }
}

// ------------------------------------------------------------------------------------------------
// K_C1: missed rows staging -> output.  One workgroup per tile walks the tile's list of missed keys (as sent); tiles
// without misses leave at once.  Two phases per pass of up to 256 list entries:
//   resolve  one THREAD per entry: (sent_i, sent_m) -> rep_of[m] -> uidx_of[rep] — the dependent loads of all entries of
//            the pass are in flight together (round 2 walked the chain once per 16-lane group and row: a tile's ~50 misses
//            cost four rounds of four dependent loads each, 31 us for 85 MB)
//   copy     one 16-lane group per row, kRows rows in flight per group, all addresses known
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hps_miss_scatter_kernel(const CallDesc* __restrict__ call,
                                                                const TableCacheDev* __restrict__ tables,
                                                                const MissDesc* __restrict__ md, const CallWork w,
                                                                const float* __restrict__ staging) {
  __shared__ int32_t sh_i[256];
  __shared__ uint32_t sh_u[256];
  const uint32_t tile = blockIdx.x;
  const uint32_t S = w.tile_cnt[tile * 4 + kTileCntSentMiss];
  if (S == 0 || blockIdx.y * 256u >= S) return;
  const uint32_t t = w.tiles[tile].table;
  const uint32_t lo = md->chunk_lo[t], hi = md->chunk_hi[t];
  if (hi == lo) return;  // table served in async mode, or nothing of it in this chunk
  const uint32_t D = tables[t].dim;
  const uint64_t stage_off = md->stage_off[t];
  const bool vec = call->vec_ok[t] != 0 && (stage_off & 3) == 0;  // staging rows are packed (offset multiple of D)
  const bool fast = vec && D == 128;
  float* __restrict__ out = call->out[t];
  const uint64_t ks = call->key_start[t];
  const uint32_t* __restrict__ dst_index = call->dst_index;   // table-sharded lookup: where each key's row goes (device_types.h)
  const uint32_t region = tile * (uint32_t)kTileKeys;
  const int lig = (int)(threadIdx.x & 15), g = (int)(threadIdx.x >> 4);
  constexpr int kRows = 4;
  // gridDim.y workgroups share a tile's list (few tiles = a small request: one workgroup per tile would copy its rows
  // one after the other)
  for (uint32_t base = blockIdx.y * 256u; base < S; base += 256u * gridDim.y) {
    const uint32_t cnt = S - base < 256u ? S - base : 256u;
    if (threadIdx.x < cnt) {
      const int32_t i = w.sent_i[region + base + threadIdx.x];
      const uint32_t m = (uint32_t)w.sent_m[region + base + threadIdx.x];
      const uint32_t u = (uint32_t)w.uidx_of[(uint32_t)w.rep_of[m]];
      sh_i[threadIdx.x] = dst_index ? (int32_t)dst_index[i] : (int32_t)((uint64_t)i - ks);   // row position in the table's output slice
      sh_u[threadIdx.x] = (u >= lo && u < hi) ? u - lo : 0xFFFFFFFFu;   // other chunk of this call
    }
    __syncthreads();
    for (uint32_t q0 = (uint32_t)g * kRows; q0 < cnt; q0 += 16 * kRows) {
      if (fast) {
        f4 v[kRows][2];
        float* dst[kRows];
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
          const uint32_t q = q0 + r;
          dst[r] = nullptr;
          if (q < cnt && sh_u[q] != 0xFFFFFFFFu) {
            const float* src = staging + stage_off + (uint64_t)sh_u[q] * 128u;
            dst[r] = out + (uint64_t)(uint32_t)sh_i[q] * 128u;
            v[r][0] = *reinterpret_cast<const f4*>(src + lig * 4);
            v[r][1] = *reinterpret_cast<const f4*>(src + 64 + lig * 4);
          }
        }
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
          if (dst[r]) {
            __builtin_nontemporal_store(v[r][0], reinterpret_cast<f4*>(dst[r] + lig * 4));
            __builtin_nontemporal_store(v[r][1], reinterpret_cast<f4*>(dst[r] + 64 + lig * 4));
          }
        }
      } else {
        for (int r = 0; r < kRows; ++r) {
          const uint32_t q = q0 + r;
          if (q >= cnt || sh_u[q] == 0xFFFFFFFFu) continue;
          copy_row<true>(staging + stage_off + (uint64_t)sh_u[q] * D, out + (uint64_t)(uint32_t)sh_i[q] * D, D, lig, vec);
        }
      }
    }
    __syncthreads();
  }
}

// K_D: async-insert mode — missed rows return the table's default vector
// (docs/architecture.md:32, docs/hierarchical_parameter_server.md:244-246).
__global__ __launch_bounds__(256) void hps_miss_fill_default_kernel(const CallDesc* __restrict__ call,
                                                                     const TableCacheDev* __restrict__ tables,
                                                                     const CallWork w,
                                                                     const uint32_t* __restrict__ table_mode) {
  // table_mode (optional): per table 1 = async insert (fill its misses), 0 = synchronous (leave them to K_C1)
  const uint32_t tile = blockIdx.x;
  const uint32_t S = w.tile_cnt[tile * 4 + kTileCntSentMiss];
  if (S == 0) return;
  const uint32_t t = w.tiles[tile].table;
  if (table_mode && table_mode[t] == 0) return;
  const uint32_t D = tables[t].dim;
  const float dv = tables[t].default_value;
  float* __restrict__ out = call->out[t];
  const uint64_t ks = call->key_start[t];
  const uint32_t region = tile * (uint32_t)kTileKeys;
  const int lig = (int)(threadIdx.x & 15);
  for (uint32_t r = blockIdx.y * 16 + (threadIdx.x >> 4); r < S; r += 16 * gridDim.y) {
    const int32_t i = w.sent_i[region + r];
    float* dst = out + (call->dst_index ? (uint64_t)call->dst_index[i] : (uint64_t)i - ks) * D;
    for (uint32_t c = (uint32_t)lig; c < D; c += 16) dst[c] = dv;
  }
}

// ------------------------------------------------------------------------------------------------
// K_G: hit rows cache -> output from the slot indices K_P left (the miss path of the call — PCIe-bound — starts
// right after the probe and runs while this HBM-bound kernel moves the hits).
// A wave takes 64 keys, each 16-lane group walks its 16 keys kU at a time; with the slots
// known up front every row load is independent (no bucket -> row dependency).
// Algorithmic bytes per key: 4 (slot) + 4D (row read) + 4D (row write) for hits, 4 for misses.
// ------------------------------------------------------------------------------------------------
// kFast: every table is 128 wide with 16-B aligned output (checked on the host): rows are staged in registers, 2*kU
// independent 16-B loads per lane in flight.  Otherwise: one row at a time with copy_row.
// kIndexed: the row of key i goes to out[t] + dst_index[i] * D (CallDesc::dst_index, table-sharded lookup) — one more
// coalesced 4-B load per key next to the slot word.
template <int kU, bool kFast, bool kIndexed>
__global__ __launch_bounds__(kProbeBlockThreads) void hps_gather_hits_kernel(const CallDesc* __restrict__ call,
                                                                             const TableCacheDev* __restrict__ tables,
                                                                             const int32_t* __restrict__ slot_in, uint32_t xcd_walk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = (int)call->num_tables;
  TableLds* sh_tab = reinterpret_cast<TableLds*>(smem);
  uint64_t* sh_ks = reinterpret_cast<uint64_t*>(smem + sizeof(TableLds) * (size_t)T);
  load_tables_to_lds(sh_tab, sh_ks, call, tables, T);
  __syncthreads();
  const uint64_t N = call->total_keys;
  const uint32_t* __restrict__ dst_index = kIndexed ? call->dst_index : nullptr;
  const int lane = lane_id();
  const int g = lane >> 4, lig = lane & 15;
  const uint64_t chunks = (N + 63) / 64;
  // XCD-aware walk (speed only; nothing depends on it): workgroup b is dispatched to XCD b % 8, so each XCD gets one
  // contiguous eighth of the key range and its workgroups sweep it front to back together.  KEYS is table-major:
  // at any moment an XCD's CUs then work inside one or two tables, and the rows the batch repeats (the hot head of
  // a Zipf-like key distribution) are re-read from that XCD's own 4-MB L2 instead of thrashing all eight L2s with
  // the hot sets of every table at once.
  const uint32_t nx = (xcd_walk && gridDim.x >= 8) ? 8u : 1u;
  const uint32_t xcd = blockIdx.x % nx, xb = blockIdx.x / nx;
  const uint32_t blocks_x = (gridDim.x - xcd + nx - 1) / nx;
  const uint64_t c_lo = chunks * xcd / nx, c_hi = chunks * (xcd + 1) / nx;
  const uint64_t waves_x = (uint64_t)blocks_x * (kProbeBlockThreads / 64);
  for (uint64_t chunk = c_lo + (uint64_t)xb * (kProbeBlockThreads / 64) + (threadIdx.x >> 6); chunk < c_hi; chunk += waves_x) {
    const uint64_t i = chunk * 64 + (uint64_t)lane;
    const int32_t s = i < N ? slot_in[i] : -1;
    int t = (int)uniform_u32((uint32_t)find_table(sh_ks, T, chunk * 64));
    if (i < N) { while (i >= sh_ks[t + 1]) ++t; }
    // kIndexed: the key's row position inside its table's output slice comes from the caller
    uint32_t di = 0;
    if (kIndexed) di = i < N ? dst_index[i] : 0u;
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
          uint64_t pos;
          if (kIndexed) pos = (uint64_t)(uint32_t)__shfl((int)di, src, 64);
          else pos = chunk * 64 + (uint64_t)src - sh_tab[tt].key_start;
          dst[u] = nullptr;
          if (ss >= 0) {
            const float* row = sh_tab[tt].rows + (uint64_t)(uint32_t)ss * 128u;
            dst[u] = sh_tab[tt].out + pos * 128u;
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
          const uint32_t dd = kIndexed ? (uint32_t)__shfl((int)di, src, 64) : 0u;
          if (ss < 0) continue;
          const TableLds& tb = sh_tab[tt];
          const uint64_t pos = kIndexed ? (uint64_t)dd : chunk * 64 + (uint64_t)src - tb.key_start;
          copy_row<true>(tb.rows + (uint64_t)(uint32_t)ss * tb.dim, tb.out + pos * tb.dim, tb.dim, lig, (tb.flags & 2u) != 0);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K_C2: insert unique missed (key,row) pairs.  One 16-lane group per key.  The group reads the key's bucket line (lanes
// 0..7 and their mirrors 8..15: 16 B each — 14 keys and the two stamp words).  Key already resident: the row is refreshed
// in place.  Otherwise the victim is a free slot, else the slot of greatest age; a slot whose stamp is the current
// unit's (hit or written since the clock last advanced) is never taken.
// Ownership inside one launch: a writer claims slot v by a 64-bit compare-and-swap on the stamp WORD that holds v's
// byte: old word -> same word with byte v = kStampClaimed.  Only slots whose stamp is neither the current unit's nor
// kStampClaimed are claimed, so a claim always changes the word; a second group that read the same line loses its CAS,
// learns the new word from the return value and picks again.  The owner writes key and row and then turns the byte into
// the current stamp.  Every update of a bucket line in this kernel is a device-scope atomic (CAS, atomic store of the
// key, atomicAnd of the stamp): no line is left dirty in one XCD's L2 while another XCD updates a neighbouring word of
// it.  Inserts never run concurrently with another kernel on the same cache (EmbeddingCache orders them with events);
// the plain loads at a group's start may be older than the launch's latest claims, and every decision taken on them is
// validated by the CAS.  While it rewrites a stamp word the claim also pulls stamps older than kAgeSaturate units
// back to exactly that age (they would wrap around and look young).
// `found[f]`==0 (key unknown to every parameter-server tier) -> not cached, and dropped if resident.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t age_of(uint32_t now8, uint32_t st) {   // st in [0, kStampMod)
  const uint32_t d = now8 + kStampMod - st;
  return d >= kStampMod ? d - kStampMod : d;
}

__device__ __forceinline__ uint64_t saturate_stamps(uint64_t word, uint32_t now8) {
  uint64_t r = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    uint32_t st = (uint32_t)(word >> (8 * k)) & 0xFFu;
    if (st < kStampMod && age_of(now8, st) > kAgeSaturate) st = (now8 + kStampMod - kAgeSaturate) % kStampMod;   // (claimed / never-used: left alone)
    r |= (uint64_t)st << (8 * k);
  }
  return r;
}

__device__ __forceinline__ uint64_t group_bcast64(uint64_t v, int src_lane) {
  return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src_lane, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src_lane, 64);
}

// (register budget: 6 waves per SIMD = 80 VGPRs, no scratch; left alone the compiler takes 83 = 5 waves: 37.0 -> 35.8 us;
//  8 waves = 64 VGPRs + 60 B of scratch: 42 us — profiles/round3/ab_probe_tile_sizes_and_insert_registers.txt)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void hps_cache_insert_kernel(const TableCacheDev* __restrict__ tables, uint32_t T,
                                                                const MissDesc* __restrict__ md,
                                                                const uint64_t* __restrict__ key_start,
                                                                const int64_t* __restrict__ uniq_keys,
                                                                const float* __restrict__ staging,
                                                                const uint8_t* __restrict__ found, uint32_t stamps,
                                                                uint32_t* __restrict__ stats /* kStatLines lines of kAccStride words */) {
  // stamps: byte 0 = the current recency unit (what a hit of this call writes), byte 1 = the stamp a NEWLY inserted key
  // gets — the current unit minus the cache's insert age (EmbeddingCache::InsertStamps): a key that was asked for once
  // enters the bucket older than the keys that have been hit, and is the first to go unless it is asked for again
  const uint32_t now8 = stamps & 0xFFu, ins8 = (stamps >> 8) & 0xFFu;
  // Admission (bits 24..27 = k > 0, insert age > 0): the new key's nominal age is the insert age, and plain LRU logic says a
  // key that would be the OLDEST of its bucket is its own victim — so a slot that was hit more recently than the insert age
  // is not given up for a key seen once (the key's row has been served; it just stays out of the cache).  One newcomer in
  // 2^k is let through regardless, chosen by (key hash ^ call counter, bits 16..23): a key that keeps being asked for gets in
  // within 2^k calls even when all 14 keys of its bucket are hit all the time.
  const uint32_t ins_age = age_of(now8, ins8), adm_k = (stamps >> 24) & 15u, call8 = (stamps >> 16) & 0xFFu;
  // per-table words of the call in LDS: the table of a flat index is a binary search over the unique-segment starts, and from
  // global memory that search alone was five dependent round trips in front of every key's bucket line (the kernel is a chain
  // of dependent accesses per key: 45 us for 84 K keys)
  extern __shared__ __attribute__((aligned(16))) char ins_smem[];
  uint64_t* sh_us = reinterpret_cast<uint64_t*>(ins_smem);            // [T + 1] unique-segment starts
  uint64_t* sh_ks = sh_us + (T + 1);                                   // [T] key_start
  uint64_t* sh_so = sh_ks + T;                                         // [T] stage_off
  uint32_t* sh_lo = reinterpret_cast<uint32_t*>(sh_so + T);            // [T] chunk_lo
  for (uint32_t t = threadIdx.x; t <= T; t += blockDim.x) sh_us[t] = md->useg_start[t];
  for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) { sh_ks[t] = key_start[t]; sh_so[t] = md->stage_off[t]; sh_lo[t] = md->chunk_lo[t]; }
  __syncthreads();
  const uint64_t total = sh_us[T];
  const int lane = lane_id();
  const int g = lane >> 4, lig = lane & 15;
  const uint64_t groups_total = (uint64_t)gridDim.x * 16;
  uint32_t n_dropped = 0, n_inserted = 0, n_refreshed = 0;  // counted by each group's lane 0
  for (uint64_t f = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4); f < total; f += groups_total) {
    const int t = find_table(sh_us, (int)T, f);
    const uint32_t u = sh_lo[t] + (uint32_t)(f - sh_us[t]);
    const int64_t key = uniq_keys[sh_ks[t] + u];          // (independent of the table descriptor: both loads go out together)
    const TableCacheDev tb = tables[t];
    if (tb.flags & 1u) continue;  // static cache: never insert
    if (key == HPS_EMPTY_KEY) continue;
    const uint32_t D = tb.dim;
    const float* row = staging + sh_so[t] + (uint64_t)(u - sh_lo[t]) * D;
    const uint32_t b = hps_bucket_of(key, tb.num_buckets);
    unsigned long long* line = reinterpret_cast<unsigned long long*>(tb.lines) + (uint64_t)b * kLineWords;
    const u64x2 ln = *reinterpret_cast<const u64x2*>(line + (lig & 7) * 2);
    // (fetching the staged row here, together with the bucket line, instead of after the claim was measured: 40 us either way)
    const bool klane = lig < 7;
    const uint32_t p0 = (uint32_t)(__ballot(klane && (int64_t)ln.x == key) >> (g * 16)) & 0x7Fu;
    const uint32_t p1 = (uint32_t)(__ballot(klane && (int64_t)ln.y == key) >> (g * 16)) & 0x7Fu;
    const uint32_t e0 = (uint32_t)(__ballot(klane && (int64_t)ln.x == HPS_EMPTY_KEY) >> (g * 16)) & 0x7Fu;
    const uint32_t e1 = (uint32_t)(__ballot(klane && (int64_t)ln.y == HPS_EMPTY_KEY) >> (g * 16)) & 0x7Fu;
    uint64_t sw[2];   // the two stamp words, from the group's lane 7, in every lane
    sw[0] = group_bcast64(ln.x, g * 16 + 7);
    sw[1] = group_bcast64(ln.y, g * 16 + 7);
    uint32_t empty = 0;   // bit v: slot v holds no key
    for (int k = 0; k < 7; ++k) empty |= (((e0 >> k) & 1u) << (2 * k)) | (((e1 >> k) & 1u) << (2 * k + 1));
    const int present = (p0 | p1) ? (p0 ? 2 * __builtin_ctz(p0) : 2 * __builtin_ctz(p1) + 1) : -1;
    const bool unknown = found && !found[f];   // the key exists in no parameter-server tier (any more)
    if (unknown && present < 0) continue;

    // claim(v): group-uniform; true when this group now owns slot v (its stamp byte reads kStampClaimed)
    // (a CAS also fails when another group claimed a DIFFERENT slot of the same word meanwhile: try again as long as
    //  slot v's own byte is what it was)
    auto claim = [&](int v) -> bool {
      const int wi = v >> 3, sh = 8 * (v & 7);
      for (;;) {
        const unsigned long long old = sw[wi];
        const unsigned long long want = (saturate_stamps(old, now8) & ~(0xFFull << sh)) | ((unsigned long long)kStampClaimed << sh);
        unsigned long long got = 0;
        if (lig == 0) got = atomicCAS(line + kBucketSlots + wi, old, want);
        got = group_bcast64(got, g * 16);
        if (got == old) {
          sw[wi] = want;
          // the bucket's other stamp word is looked after too (one more CAS, only when something in it is about to wrap)
          const unsigned long long other = sw[wi ^ 1], sat = saturate_stamps(other, now8);
          if (sat != other) {
            if (lig == 0) atomicCAS(line + kBucketSlots + (wi ^ 1), other, sat);   // losing it to a concurrent claim is fine
            sw[wi ^ 1] = sat;   // (a stale view at worst: every later decision on it is validated by its own CAS)
          }
          return true;
        }
        sw[wi] = got;
        const uint32_t nb = (uint32_t)(got >> sh) & 0xFFu, ob = (uint32_t)(old >> sh) & 0xFFu;
        if (nb != ob) {
          // Slot v's own byte changed.  Either somebody took the slot (the byte reads kStampClaimed, or the stamp its new owner
          // left: the current unit's or the insert stamp) — or a neighbour's claim on this word merely pulled v's old stamp back to
          // the saturation age (saturate_stamps): same occupant, go on with the new word.  (Found by the turnover clock on a
          // 3,500-slot cache, where stamps older than kAgeSaturate are common: a refresh of all resident keys skipped the rows
          // whose stamp a neighbour had just re-aged — tests/test_gpu_bounded_host_tier.py.)
          const uint32_t sat = (now8 + kStampMod - kAgeSaturate) % kStampMod;
          if (!(nb == sat && ob < kStampMod && age_of(now8, ob) > kAgeSaturate)) return false;
        }
      }
    };
    const bool guarded = adm_k != 0 && ins_age != 0 && ((((uint32_t)(hps_mix64((uint64_t)key) >> 8)) ^ call8) & ((1u << adm_k) - 1u)) != 0;
    int victim = -1;
    bool owned = false;   // victim's stamp byte reads kStampClaimed and has to be turned into now8 at the end
    if (present >= 0) {
      // Already resident (another session inserted it after our probe, or a refresh): rewrite the row in place — or drop
      // the slot when the key is unknown now — but only as the slot's owner: a second group of this launch may be about
      // to evict exactly this slot, and an unowned write would interleave with the evictor's (key/row mismatch = poisoned
      // slot).  A slot stamped in the current unit is taken by nobody, so it needs no claim; losing the claim skips the write.
      const uint32_t st = stamp_of(sw[0], sw[1], (uint32_t)present);
      if (st == now8 || st == ins8) victim = present;   // nobody takes such a slot in this launch (see the victim search)
      else if (st != kStampClaimed && claim(present)) { victim = present; owned = true; }
    } else {
      for (int tries = 0; tries < kBucketSlots + 2 && victim < 0; ++tries) {
        // lane v rates slot v; 16-lane maximum of (age << 4 | 15 - v): the greatest age, the lowest slot among equals
        // (a free slot counts as age 256; a slot used in the current unit or owned by another group of this launch as none)
        uint32_t cand = 0;
        if (lig < kBucketSlots) {
          const uint32_t st = stamp_of(sw[0], sw[1], (uint32_t)lig);
          // (ins8: what this launch's own inserts leave behind.  Such a slot must not change hands again inside the launch —
          //  its first owner's row stores may still be in flight when the second owner's arrive — so the stamp of new keys
          //  is off limits like the current unit's; a key that was last hit exactly insert-age units ago shares the privilege)
          if (st != now8 && st != ins8 && st != kStampClaimed) {
            // (a never-used slot carries kStampFree, which is no clock value: it passes the test above whatever the clock reads)
            const uint32_t a = (((empty >> lig) & 1u) || st == kStampFree) ? 256u : age_of(now8, st);
            if (!(guarded && a < ins_age)) cand = (a << 4) | (15u - (uint32_t)lig);
          }
        }
        for (int off = 8; off > 0; off >>= 1) {
          const uint32_t o = (uint32_t)__shfl_xor((int)cand, off, 16);
          cand = o > cand ? o : cand;
        }
        const int best = cand ? 15 - (int)(cand & 15u) : -1;
        if (best < 0) break;  // whole bucket is in use by the current unit (or hit more recently than the newcomer's nominal age)
        if (claim(best)) { victim = best; owned = true; }
      }
    }
    if (victim < 0) {
      n_dropped += (lig == 0);  // bucket full of the current unit's keys (or lost the claim)
      continue;
    }
    if (unknown) {
      // drop it, so that later lookups fall through to the default value instead of a stale row
      if (lig == 0) __hip_atomic_store(line + victim, (unsigned long long)HPS_EMPTY_KEY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (present < 0 && lig == 0) __hip_atomic_store(line + victim, (unsigned long long)key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float* dst = tb.rows + ((uint64_t)b * kBucketSlots + (uint64_t)victim) * D;
      copy_row<false>(row, dst, D, lig, (D & 3u) == 0 && (sh_so[t] & 3) == 0);
      if (lig == 0) { if (present >= 0) ++n_refreshed; else ++n_inserted; }
    }
    // kStampClaimed (all ones) AND now8 = now8; the other bytes of the word keep whatever they hold by now
    if (owned && lig == 0) {
      const int sh = 8 * (victim & 7);
      const unsigned long long st = (present >= 0 || unknown) ? now8 : ins8;
      atomicAnd(line + kBucketSlots + (victim >> 3), ~(0xFFull << sh) | (st << sh));
    }
  }
  // one atomic per block and counter, spread over kStatLines lines of the accumulator block: atomics on one
  // 128-B line serialise at ~90 per microsecond, and 2,048 blocks x 3 counters on one line cost 70 us
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
    if (v) atomicAdd(&stats[(blockIdx.x % (uint32_t)kStatLines) * (uint32_t)kAccStride + threadIdx.x], v);
  }
}

// Utility: every key slot EMPTY, every stamp `stamp8` (kStampFree: no clock value, always claimable).
__global__ void hps_cache_clear_kernel(int64_t* lines, uint64_t num_buckets, uint32_t stamp8) {
  const uint64_t words = num_buckets * kLineWords;
  const uint64_t sw = 0x0101010101010101ull * (uint64_t)(stamp8 & 0xFFu);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x)
    lines[i] = (i % kLineWords) < (uint64_t)kBucketSlots ? HPS_EMPTY_KEY : (int64_t)sw;
}

// Utility for tests / refresh: per-key residency (slot index or -1), no side effects.
__global__ void hps_cache_query_kernel(TableCacheDev tb, const int64_t* __restrict__ keys, uint64_t n,
                                       int32_t* __restrict__ slot) {
  const int lane = lane_id();
  const int gw = lane >> 3, lig = lane & 7;
  const uint64_t groups_total = (uint64_t)gridDim.x * (blockDim.x / kProbeLanes);
  for (uint64_t i = (uint64_t)blockIdx.x * (blockDim.x / kProbeLanes) + (threadIdx.x >> 3); i < n; i += groups_total) {
    const int64_t key = keys[i];
    const uint32_t b = hps_bucket_of(key, tb.num_buckets);
    const u64x2 ln = *reinterpret_cast<const u64x2*>(tb.lines + (uint64_t)b * kLineWords + lig * 2);
    const bool live = lig < 7 && key != HPS_EMPTY_KEY;
    const uint32_t m0 = (uint32_t)(__ballot(live && (int64_t)ln.x == key) >> (gw * 8)) & 0xFFu;
    const uint32_t m1 = (uint32_t)(__ballot(live && (int64_t)ln.y == key) >> (gw * 8)) & 0xFFu;
    if (lig == 0)
      slot[i] = (m0 | m1) ? (int32_t)(b * kBucketSlots + (m0 ? 2u * (uint32_t)__builtin_ctz(m0) : 2u * (uint32_t)__builtin_ctz(m1) + 1u)) : -1;
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// Grid of the gather kernel: one 64-key chunk per wave up to 32 workgroups per CU (config 2: 6,656 workgroups, each wave one
// chunk), a grid-stride walk beyond that.  Round 2 first ran it persistent (8 resident workgroups per CU walking all chunks):
// the same speed (243-246 against 246-249 us), but with every wave slot held until the kernel's end another session's scatter
// kernel that arrived meanwhile could take the whole rest of the gather (max 194-245 us against 55-70 us with this grid;
// profiles/round2/kernel_time_distribution_within_runs.txt).
uint32_t GatherGridBlocks(uint64_t N, int cu_count) {
  const uint64_t chunks = (N + 63) / 64;
  const uint64_t want = (chunks + 3) / 4;
  const uint64_t cap = (uint64_t)cu_count * 32;
  return (uint32_t)(want < cap ? (want ? want : 1) : cap);
}

// workgroups per tile for the kernels that walk a tile's miss list: enough to put ~2,048 workgroups on the chip when the
// request has few tiles, one or two per tile when it has many
static inline uint32_t ListSubBlocks(uint32_t tiles) {
  const uint32_t k = (2048 + tiles - 1) / tiles;
  return k < 1 ? 1u : (k > 16 ? 16u : k);
}

bool ProbeTailAvailable(int variant) { return (variant / 100) % 10 == 0; }

hipError_t LaunchProbeTiles(const CallDesc* d_call, const TableCacheDev* d_tables, const CallWork& w, int variant,
                            bool tail, hipStream_t stream, KTimer kt) {
  if (w.num_tiles == 0) return hipSuccess;
  if (tail && !ProbeTailAvailable(variant)) return hipErrorInvalidValue;
  // variant: 1002 = the default (tile-local input dedup, two bucket lines in flight per 8-lane group, 512 threads per tile);
  // 1102 = the same without the tile-local dedup (the fallback a deployment can select, session option "probe_variant": every
  // key probes its bucket itself; 3-5 us slower on Zipf traffic).  Rounds 2-4 carried 24 instantiations (U in {1,2,4,8} x
  // 256/512 threads x dedup x tail) for the A/B runs recorded in profiles/round2..3/kbench_*.txt; they went with round 5.
  const bool dedup = (variant / 100) % 10 == 0;
  const uint32_t probe_grid = w.num_tiles;
  if (dedup && tail)
    hipExtLaunchKernelGGL((hps_probe_tile_kernel<true, 2, 512, true>), dim3(probe_grid), dim3(512), 0, stream, kt.start, kt.stop, 0, d_call, d_tables, w);
  else if (dedup)
    hipExtLaunchKernelGGL((hps_probe_tile_kernel<true, 2, 512, false>), dim3(probe_grid), dim3(512), 0, stream, kt.start, kt.stop, 0, d_call, d_tables, w);
  else
    hipExtLaunchKernelGGL((hps_probe_tile_kernel<false, 2, 512, false>), dim3(probe_grid), dim3(512), 0, stream, kt.start, kt.stop, 0, d_call, d_tables, w);
  return hipGetLastError();
}

hipError_t LaunchMissUnique(const CallDesc* d_call, const TableCacheDev* d_tables, const CallWork& w, hipStream_t stream,
                            KTimer kt) {
  if (w.num_tiles == 0) return hipSuccess;
  const uint32_t blocks = (w.num_tiles + 3) / 4;   // one wave per tile
  hipExtLaunchKernelGGL(hps_miss_unique_kernel, dim3(blocks), dim3(256), 0, stream, kt.start, kt.stop, 0, d_call, d_tables, w);
  return hipGetLastError();
}

uint32_t UniqueHitsParts(uint64_t slots) { return (uint32_t)((slots + kHitPartBits - 1) / kHitPartBits); }

hipError_t LaunchUniqueHits(const CallDesc* d_call, const TableCacheDev* d_tables, uint32_t total_parts, const int32_t* d_slot,
                            uint32_t* d_acc, hipStream_t stream, KTimer kt) {
  if (total_parts == 0) return hipSuccess;
  constexpr uint32_t lds = kHitPartBits / 8;
  // (per call, not once: the attribute belongs to the function on the CURRENT device, and a process serves several)
  const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(hps_unique_hits_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr != hipSuccess) return attr;
  hipExtLaunchKernelGGL(hps_unique_hits_kernel, dim3(total_parts), dim3(kHitThreads), lds, stream, kt.start, kt.stop, 0, d_call, d_tables, d_slot,
                        d_acc);
  return hipGetLastError();
}

hipError_t LaunchGatherHits(const CallDesc* d_call, const TableCacheDev* d_tables, uint32_t num_tables, uint64_t N,
                            const int32_t* d_slot, uint32_t grid, bool all_128_aligned, bool xcd_walk, hipStream_t stream, KTimer kt,
                            bool indexed) {
  if (N == 0) return hipSuccess;
  const size_t lds = sizeof(TableLds) * num_tables + sizeof(uint64_t) * (num_tables + 1);
  // rows in flight per 16-lane group (two 16-B loads per lane and row): 8 (106 VGPRs, 4 waves per SIMD).  Against 4 (64 VGPRs,
  // 8 waves per SIMD): the same 210 us on the boxes whose copies run at 5.5+ TB/s, 228-234 against 235-246 us on the slow-gather
  // boxes (profiles/round3/ab_gather_rows_in_flight.txt); 16 (214 VGPRs) no better than 8.  (The A/B switch went with round 4.)
#define HPS_KG(UU, FF, II)                                                                                                          \
  hipExtLaunchKernelGGL((hps_gather_hits_kernel<UU, FF, II>), dim3(grid), dim3(kProbeBlockThreads), (uint32_t)lds, stream, kt.start, kt.stop, 0, \
                        d_call, d_tables, d_slot, xcd_walk ? 1u : 0u)
  if (all_128_aligned) { if (indexed) HPS_KG(8, true, true); else HPS_KG(8, true, false); }
  else { if (indexed) HPS_KG(4, false, true); else HPS_KG(4, false, false); }
#undef HPS_KG
  return hipGetLastError();
}

hipError_t LaunchMissScatter(const CallDesc* d_call, const TableCacheDev* d_tables, const MissDesc* d_md, const CallWork& w,
                             const float* d_staging, hipStream_t stream, KTimer kt) {
  if (w.num_tiles == 0) return hipSuccess;
  hipExtLaunchKernelGGL(hps_miss_scatter_kernel, dim3(w.num_tiles, ListSubBlocks(w.num_tiles)), dim3(256), 0, stream, kt.start, kt.stop, 0,
                        d_call, d_tables, d_md, w, d_staging);
  return hipGetLastError();
}

hipError_t LaunchMissFillDefault(const CallDesc* d_call, const TableCacheDev* d_tables, const CallWork& w,
                                 const uint32_t* d_table_mode, hipStream_t stream) {
  if (w.num_tiles == 0) return hipSuccess;
  hipLaunchKernelGGL(hps_miss_fill_default_kernel, dim3(w.num_tiles, ListSubBlocks(w.num_tiles)), dim3(256), 0, stream, d_call, d_tables,
                     w, d_table_mode);
  return hipGetLastError();
}

hipError_t LaunchCacheInsert(const TableCacheDev* d_tables, uint32_t T, const MissDesc* d_md, uint64_t total_unique,
                             const uint64_t* d_key_start, const int64_t* d_uniq_keys, const float* d_staging,
                             const uint8_t* d_found, uint32_t stamps, uint32_t* d_stats, int cu_count,
                             hipStream_t stream, KTimer kt) {
  if (total_unique == 0) return hipSuccess;
  uint64_t want = (total_unique + 15) / 16;
  const uint64_t cap = (uint64_t)cu_count * 8;
  if (want > cap) want = cap;
  const size_t lds = (size_t)(T + 1) * 8 + (size_t)T * (8 + 8 + 4) + 16;
  hipExtLaunchKernelGGL(hps_cache_insert_kernel, dim3((uint32_t)want), dim3(256), (uint32_t)lds, stream, kt.start, kt.stop, 0, d_tables, T, d_md,
                        d_key_start, d_uniq_keys, d_staging, d_found, stamps, d_stats);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Control words of a call over the compute queue instead of the SDMA engines.  A copy of a few KB placed between two
// kernels of one stream costs two hand-offs between the compute queue and an SDMA queue (signal + barrier packet each way,
// 10-20 us together); a small kernel that reads or writes page-locked host memory directly stays on the queue.
//   pull: call block (descriptor | zeroed accumulators | tile descriptors) host -> HBM, 16 B per lane
//   push: accumulator words HBM -> host, then a sequence word the waiting host thread polls
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hps_pull16_kernel(const uint4* __restrict__ src_host, uint4* __restrict__ dst, uint32_t n16) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n16) dst[i] = src_host[i];
}

// Staged keys, page-locked host memory -> HBM, read by the compute units instead of a copy engine: a request's keys then share
// the link with the other session's row uploads packet by packet instead of queueing behind its 4-MB copies (engine.cpp, stage()).
// Grid-stride, four 16-B loads in flight per lane; head and tail of a range that does not start / end on 16 B go byte by byte.
__global__ __launch_bounds__(256) void hps_pull_bytes_kernel(const uint8_t* __restrict__ src_host, uint8_t* __restrict__ dst, uint64_t bytes) {
  const uint64_t head = min<uint64_t>(bytes, (16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u);
  const uint64_t n16 = (bytes - head) / 16;
  const uint64_t tail0 = head + n16 * 16;
  const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x, nthr = (uint64_t)gridDim.x * 256u;
  if (tid < head) dst[tid] = src_host[tid];
  if (tid < bytes - tail0) dst[tail0 + tid] = src_host[tail0 + tid];
  const u64x2* s = reinterpret_cast<const u64x2*>(src_host + head);
  u64x2* d = reinterpret_cast<u64x2*>(dst + head);
  uint64_t i = tid;
  for (; i + 3 * nthr < n16; i += 4 * nthr) {
    const u64x2 a = __builtin_nontemporal_load(s + i), b = __builtin_nontemporal_load(s + i + nthr),
                c = __builtin_nontemporal_load(s + i + 2 * nthr), e = __builtin_nontemporal_load(s + i + 3 * nthr);
    d[i] = a; d[i + nthr] = b; d[i + 2 * nthr] = c; d[i + 3 * nthr] = e;
  }
  for (; i < n16; i += nthr) d[i] = __builtin_nontemporal_load(s + i);
}

__global__ __launch_bounds__(256) void hps_push_words_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst_host,
                                                             uint32_t words, uint32_t* __restrict__ seq_host, uint32_t seq) {
  for (uint32_t i = threadIdx.x; i < words; i += 256u) dst_host[i] = src[i];
  __threadfence_system();   // every lane's words are on their way before the barrier lets lane 0 publish
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(seq_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t LaunchPull16(const void* src_host_devptr, void* dst, size_t bytes, hipStream_t stream) {
  const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
  if (n16 == 0) return hipSuccess;
  hipLaunchKernelGGL(hps_pull16_kernel, dim3((n16 + 255) / 256), dim3(256), 0, stream, (const uint4*)src_host_devptr, (uint4*)dst, n16);
  return hipGetLastError();
}

hipError_t LaunchPullBytes(const void* src_host_devptr, void* dst, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return hipSuccess;
  if ((reinterpret_cast<uintptr_t>(src_host_devptr) & 15u) != (reinterpret_cast<uintptr_t>(dst) & 15u)) return hipErrorInvalidValue;
  uint64_t blocks = (bytes / 16 + 1023) / 1024;   // up to four 16-B units per lane
  if (blocks > 96) blocks = 96;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(hps_pull_bytes_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, (const uint8_t*)src_host_devptr, (uint8_t*)dst, (uint64_t)bytes);
  return hipGetLastError();
}

hipError_t LaunchPushWords(const uint32_t* src, uint32_t* dst_host_devptr, uint32_t words, uint32_t* seq_host_devptr, uint32_t seq,
                           hipStream_t stream) {
  hipLaunchKernelGGL(hps_push_words_kernel, dim3(1), dim3(256), 0, stream, src, dst_host_devptr, words, seq_host_devptr, seq);
  return hipGetLastError();
}

hipError_t LaunchCacheClear(int64_t* d_lines, uint64_t num_buckets, uint32_t stamp8, hipStream_t stream) {
  uint64_t want = (num_buckets * kLineWords + 255) / 256;
  if (want > 4096) want = 4096;
  if (want == 0) want = 1;
  hipLaunchKernelGGL(hps_cache_clear_kernel, dim3((uint32_t)want), dim3(256), 0, stream, d_lines, num_buckets, stamp8);
  return hipGetLastError();
}

hipError_t LaunchCacheQuery(const TableCacheDev& tb, const int64_t* d_keys, uint64_t n, int32_t* d_slot,
                            hipStream_t stream) {
  if (n == 0) return hipSuccess;
  uint64_t want = (n + 31) / 32;
  if (want > 8192) want = 8192;
  hipLaunchKernelGGL(hps_cache_query_kernel, dim3((uint32_t)want), dim3(256), 0, stream, tb, d_keys, n, d_slot);
  return hipGetLastError();
}

}  // namespace hps
