// Launchers of entry_kernels.hip: the entry side of a table-sharded lookup (shard_entry.h).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "device_types.h"

namespace hps {

// One request as the entry instance sees it.  Filled on the host (pinned), copied to the entry GPU, read by the kernels.
struct EntryDesc {
  uint32_t num_tables;
  uint32_t num_shards;
  uint32_t num_tiles;
  uint32_t pad_;
  uint64_t total_keys;
  uint64_t key_start[kMaxTables + 1];    // prefix sums of the request's keys per table
  uint32_t first_tile[kMaxTables + 1];   // first tile of table t (tiles never straddle tables); first_tile[T] = num_tiles
  float* out[kMaxTables];                // the request's output slice of table t (entry GPU)
  uint32_t dim[kMaxTables];
  int64_t key_base[kMaxTables];          // narrowed host keys (hps_entry_widen): key = key_base[t] + what crossed PCIe
};

// Host keys that crossed PCIe narrowed (3 bytes packed, or uint32, as offsets from their table's smallest key: key_pack.h) ->
// int64 keys for the bucket kernels and the owners.  key_bytes 3 or 4.
hipError_t LaunchEntryWiden(const EntryDesc* d_desc, const TileDesc* d_tiles, uint32_t num_tiles, const void* d_narrow, uint32_t key_bytes,
                            int64_t* d_keys, hipStream_t stream);
// (d_set == nullptr: the tile level only — a key that several tiles hold travels once per tile)
// d_rep[i] = index of the representative of (table of i, key i): i itself for one key of every distinct pair.
// d_set: set_mask + 1 (a power of two >= 2 n) words, zeroed once; tag != 0, different from the tags still in the set.
hipError_t LaunchEntryDedup(const EntryDesc* d_desc, const TileDesc* d_tiles, uint32_t num_tiles, const int64_t* d_keys, uint64_t n,
                            unsigned long long* d_set, uint64_t set_mask, uint32_t tag, uint32_t* d_rep, hipStream_t stream);
// Stable bucket of the request's representatives (all keys when d_rep is null) by owner = mix64(key) mod num_shards:
//   d_bkeys / d_bidx   owner-major, table-major inside an owner, input order inside a table; d_bidx = row position of the key
//                      in its table's output slice
//   d_base[P + 1]      first bucket position per owner; d_base[P] = keys bucketed
//   d_counts[P][T]     keys per (owner, table)
//   d_hist, d_within   workspace, num_tiles * num_shards words each
hipError_t LaunchEntryBucket(const EntryDesc* d_desc, const TileDesc* d_tiles, uint32_t num_tiles, uint32_t num_shards,
                             const int64_t* d_keys, const uint32_t* d_rep, uint32_t* d_hist, uint32_t* d_within, uint32_t* d_base,
                             uint32_t* d_counts, int64_t* d_bkeys, uint32_t* d_bidx, hipStream_t stream);
// "staged_copy" transport (shard_entry.h): an owner gathered the rows of one PIECE of its bucket into a local block — table-major,
// every table's rows starting on a 16-byte boundary — and a copy engine shipped the block into the entry GPU's receive buffer.
// This kernel puts every row where it belongs: out[table][d_bidx[j]] = block row of bucket key j.  One launch covers up to
// kPlaceMaxSegments non-empty tables of the piece (by-value argument block; a piece with more takes several launches).
constexpr int kPlaceMaxSegments = 64;
struct PlaceArgs {
  uint32_t num_segments;
  uint32_t num_keys;                            // keys of this launch
  uint32_t start[kPlaceMaxSegments + 1];        // segment g covers launch keys [start[g], start[g + 1])
  uint32_t table[kPlaceMaxSegments];            // its table
  uint32_t src_off[kPlaceMaxSegments];          // float offset of its first row in the block (a multiple of 4)
};
// d_bidx: the launch's slice of the bucket index array; d_block: the piece's block in the receive buffer
hipError_t LaunchEntryPlace(const EntryDesc* d_desc, const PlaceArgs& args, const uint32_t* d_bidx, const float* d_block, hipStream_t stream);
// out[i] = out[d_rep[i]] where d_rep[i] != i
hipError_t LaunchEntryExpand(const EntryDesc* d_desc, const uint32_t* d_rep, uint64_t n, hipStream_t stream);

}  // namespace hps
