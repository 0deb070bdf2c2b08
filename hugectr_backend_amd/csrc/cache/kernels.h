// Host-callable launchers of the kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "device_types.h"

namespace hps {

uint32_t ProbeGridBlocks(uint64_t N, int cu_count, bool balanced = false);

hipError_t LaunchProbeGather(const CallDesc* d_call, const TableCacheDev* d_tables, uint32_t num_tables, uint64_t N,
                             int32_t* d_slot, uint32_t* d_block_miss, uint32_t grid, int unroll, hipStream_t stream);

hipError_t LaunchMissDedup(const CallDesc* d_call, const uint64_t* h_key_start, uint32_t T, uint32_t probe_blocks,
                           int32_t* d_slot, const uint32_t* d_block_miss, int32_t* d_set, uint64_t set_cap,
                           uint32_t* d_counts, int64_t* d_uniq_keys, int64_t* uniq_keys_host_mapped, int cu_count,
                           hipStream_t stream);

// K_G: hit rows cache -> output from the slot indices a probe-only K_A left (d_call carries the output pointers).
hipError_t LaunchGatherHits(const CallDesc* d_call, const TableCacheDev* d_tables, uint32_t num_tables, uint64_t N,
                            const int32_t* d_slot, uint32_t grid, bool all_128_aligned, hipStream_t stream);

hipError_t LaunchMissScatter(const CallDesc* d_call, const TableCacheDev* d_tables, const MissDesc* d_md, uint64_t N,
                             const int32_t* d_slot, const float* d_staging, int cu_count, hipStream_t stream);

hipError_t LaunchMissFillDefault(const CallDesc* d_call, const TableCacheDev* d_tables, uint64_t N,
                                 const int32_t* d_slot, const uint32_t* d_table_mode, int cu_count, hipStream_t stream);

hipError_t LaunchCacheInsert(const TableCacheDev* d_tables, uint32_t T, const MissDesc* d_md, uint64_t total_unique,
                             const uint64_t* d_key_start, const int64_t* d_uniq_keys, const float* d_staging,
                             const uint8_t* d_found, uint32_t epoch, uint32_t* d_stats, int cu_count,
                             hipStream_t stream);

hipError_t LaunchCacheClear(int64_t* d_keys, uint32_t* d_stamps, uint64_t slots, hipStream_t stream);

// stamps > keep_from -> stamp - keep_from + 1 ; other used stamps -> 1 ; 0 stays 0
hipError_t LaunchCacheRenorm(uint32_t* d_stamps, uint64_t slots, uint32_t keep_from, hipStream_t stream);

hipError_t LaunchCacheQuery(const TableCacheDev& tb, const int64_t* d_keys, uint64_t n, int32_t* d_slot,
                            hipStream_t stream);

}  // namespace hps
