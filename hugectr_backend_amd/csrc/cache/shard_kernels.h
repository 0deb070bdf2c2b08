// Launchers of shard_kernels.hip (table-sharded lookup: bucket keys by owner rank, restore order).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace hps {

uint32_t ShardOwnerHost(int64_t key, uint32_t num_shards);  // mix64(key) mod P — same function on host and device
size_t ShardBucketWorkspaceBytes(uint64_t n, uint32_t num_shards);
// keys grouped by owner (shard 0 first, input order kept inside a shard), perm[j] = input index of sorted key j,
// totals[s] = keys owned by shard s
hipError_t LaunchShardBucket(const int64_t* d_keys, uint64_t n, uint32_t num_shards, int64_t* d_keys_sorted, int32_t* d_perm,
                             uint64_t* d_totals, void* d_workspace, hipStream_t stream);
// Fixed-capacity exchange (shard_session.cpp): d_send = P blocks of (2 + cap) int64 — [0] keys in the block (<= cap),
// [1] the largest block this rank needed (> cap: it overflowed), then the keys in input order; d_keys: int64 (key_bytes 8) or
// uint32 (key_bytes 4); d_pos[i] = owner(i) * cap + rank of key i in its block, 0xFFFFFFFF for the cache's reserved key
// (HPS_EMPTY_KEY: never sent, answered with the default vector); d_totals[s] = keys owned by shard s (uint64, may exceed cap)
// d_rep (optional, from LaunchShardDedup): only keys with d_rep[i] == i travel; the others' rows are read through their
// representative's position by LaunchShardGatherBack (same d_rep)
hipError_t LaunchShardBucketPadded(const void* d_keys, uint32_t key_bytes, uint64_t n, uint32_t num_shards, uint64_t cap, int64_t* d_send,
                                   uint32_t* d_pos, uint64_t* d_totals, void* d_workspace, hipStream_t stream, const uint32_t* d_rep = nullptr);
// call-wide input dedup in front of the exchange: d_rep[i] = index of key i's representative (i itself for one key of every
// distinct value and for the reserved key).  d_set: set_mask + 1 (a power of two >= 2 n) words, zeroed once; tag != 0 and
// different from the tags of the calls whose entries are still in the set (a counter).
hipError_t LaunchShardDedup(const void* d_keys, uint32_t key_bytes, uint64_t n, unsigned long long* d_set, uint64_t set_mask, uint32_t tag,
                            uint32_t* d_rep, hipStream_t stream);
// received blocks -> contiguous [P][cap] keys (unused slots = HPS_EMPTY_KEY, skipped by the probe);
// d_flags[0] = max(d_flags[0], largest block any peer needed), d_flags[1] += keys received
hipError_t LaunchShardPrepare(const int64_t* d_recv, uint32_t num_shards, uint64_t cap, int64_t* d_keys_pad, uint32_t* d_flags,
                              hipStream_t stream);
// d_out[i] = d_rows[d_pos[i]], or the default vector where d_pos[i] == 0xFFFFFFFF
hipError_t LaunchShardGatherBack(const float* d_rows, const uint32_t* d_pos, uint64_t n, uint32_t dim, float* d_out, float default_value,
                                 hipStream_t stream, const uint32_t* d_rep = nullptr);
hipError_t LaunchShardUnpermute(const float* d_rows, const int32_t* d_perm, uint64_t n, uint32_t dim, float* d_out,
                                hipStream_t stream);

}  // namespace hps
