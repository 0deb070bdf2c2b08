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
hipError_t LaunchShardUnpermute(const float* d_rows, const int32_t* d_perm, uint64_t n, uint32_t dim, float* d_out,
                                hipStream_t stream);

}  // namespace hps
