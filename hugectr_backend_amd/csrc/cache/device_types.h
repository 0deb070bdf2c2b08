// Plain-old-data shared between host code and the HIP kernels of the GPU embedding cache.
//
// Data layout in HBM:
//   per table t:  bucket_keys[num_buckets][16] int64   one bucket = 128 B = one L2 line
//                 stamps     [num_buckets][16] uint32  LRU epoch of last use (0 = never used)
//                 rows       [num_buckets*16][D] fp32  slot s owns rows[s*D .. s*D+D)
// A key lives in exactly one bucket: hps_bucket_of(key, num_buckets) (common/hps_hash.h).
#pragma once
#include <stdint.h>

#include "../common/hps_hash.h"

namespace hps {

constexpr int kBucketSlots = HPS_BUCKET_SLOTS;  // 16
constexpr int kMaxTables = 256;                  // per model
// Per-call counter block of a lookup session (uint32 words, device + pinned host mirror):
//   [0] missed keys of the call   [1 + t] unique missed keys of table t
//   [kMaxTables + 1 .. + 4] insert statistics (dropped, inserted, refreshed, spare)
//   [kTableMissBase + t] missed keys of table t, duplicates included (per-table hit rate); after the host has read
//                        them the same words carry the per-table insertion mode to the kernels (1 = async, 0 = sync)
constexpr int kTableMissBase = kMaxTables + 8;
constexpr int kCountWords = 2 * kMaxTables + 8;
constexpr int kProbeBlockThreads = 256;

// slot_out[] encoding produced by the probe kernel and refined by the miss kernels
constexpr int32_t kSlotMiss = -1;  // not in cache (before dedup)
// after dedup: slot = -2 - uidx   (uidx = index of the key in its table's unique-miss segment)

struct TableCacheDev {
  int64_t* bucket_keys;
  uint32_t* stamps;
  float* rows;
  uint32_t num_buckets;
  uint32_t dim;
  float default_value;
  uint32_t flags;  // bit0: static cache (no stamp writes, no inserts)
};

// One lookup call.  Filled on the host (pinned), copied to the session's device buffer, then read by
// every kernel of the call.  The reference's ProcessRequest builds the same per-table pointer slices
// (model_instance_state.cpp:180-193).
struct CallDesc {
  uint32_t num_tables;
  uint32_t epoch;           // LRU epoch of this call (monotonic per cache)
  uint64_t total_keys;      // N = sum n_t
  const int64_t* keys;      // flat, table-major, device
  uint64_t key_start[kMaxTables + 1];  // prefix sums of n_t
  float* out[kMaxTables];              // device pointer of table t's output slice
  uint8_t vec_ok[kMaxTables];          // 1: D%4==0 and out[t] 16-B aligned -> float4 path
};

// Second descriptor, valid after the host has sized the miss staging (per call, per chunk).
struct MissDesc {
  uint64_t useg_start[kMaxTables + 1];  // prefix sums of unique-miss counts per table (flat index space)
  uint64_t stage_off[kMaxTables];       // float offset of table t's first staged row
  uint32_t chunk_lo[kMaxTables];        // this chunk covers uidx in [chunk_lo[t], chunk_hi[t]) of table t
  uint32_t chunk_hi[kMaxTables];
};

}  // namespace hps
