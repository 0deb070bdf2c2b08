// Host-callable launchers of the kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "device_types.h"

namespace hps {

uint32_t GatherGridBlocks(uint64_t N, int cu_count);

// K_P: one workgroup per tile of w.tiles.  variant = U + 100 * no_dedup + 1000 * wide (U in {1,2,4,8}).
// tail: the tile's first wave also does K_M's work (call-wide unique misses) — then LaunchMissUnique must NOT follow.  Only for
// variants with tile dedup (ProbeTailAvailable).
bool ProbeTailAvailable(int variant);
// KTimer: events that take the KERNEL's own start / stop timestamps (hipExtLaunchKernel): what rocprofv3 reports as the launch's
// duration.  A pair of hipEventRecord around a launch also measures two packet hand-offs on the queue (6-8 us per kernel on
// the MI355X box: probe 57 against 51.6 us, gather 249 against 241, scatter 30 against 23 in the same run).
struct KTimer { hipEvent_t start = nullptr, stop = nullptr; };
hipError_t LaunchProbeTiles(const CallDesc* d_call, const TableCacheDev* d_tables, const CallWork& w, int variant, bool tail,
                            hipStream_t stream, KTimer kt = {});
// K_M: call-wide unique missed keys per table into w.acc / w.uniq_keys / w.rep_of / w.uidx_of
hipError_t LaunchMissUnique(const CallDesc* d_call, const TableCacheDev* d_tables, const CallWork& w, hipStream_t stream,
                            KTimer kt = {});
// K_H: the call's unique hit keys per table (distinct slots among d_slot's non-negative words) added to
// d_acc[AccTableWord(t, kAccUniqHit)].  total_parts = sum of UniqueHitsParts(slots of table t) over the tables with keys in the call.
uint32_t UniqueHitsParts(uint64_t slots);
hipError_t LaunchUniqueHits(const CallDesc* d_call, const TableCacheDev* d_tables, uint32_t total_parts, const int32_t* d_slot,
                            uint32_t* d_acc, hipStream_t stream, KTimer kt = {});

// K_G: hit rows cache -> output from the slot indices K_P left (d_call carries the output pointers).
hipError_t LaunchGatherHits(const CallDesc* d_call, const TableCacheDev* d_tables, uint32_t num_tables, uint64_t N,
                            const int32_t* d_slot, uint32_t grid, bool all_128_aligned, bool xcd_walk, hipStream_t stream, KTimer kt = {},
                            bool indexed = false /* rows go to out[t] + d_call->dst_index[i] * D */);

hipError_t LaunchMissScatter(const CallDesc* d_call, const TableCacheDev* d_tables, const MissDesc* d_md, const CallWork& w,
                             const float* d_staging, hipStream_t stream, KTimer kt = {});

hipError_t LaunchMissFillDefault(const CallDesc* d_call, const TableCacheDev* d_tables, const CallWork& w,
                                 const uint32_t* d_table_mode, hipStream_t stream);

// d_stats: kStatLines lines of kAccStride words (insert statistics, see device_types.h); stamps: the current recency unit in byte 0, the stamp of newly inserted keys in byte 1, the call counter's low byte in byte 2, the admission parameter in bits 24..27 (EmbeddingCache::InsertStamps)
// ((epoch >> age_shift) & 255, EmbeddingCache::Stamp8)
hipError_t LaunchCacheInsert(const TableCacheDev* d_tables, uint32_t T, const MissDesc* d_md, uint64_t total_unique,
                             const uint64_t* d_key_start, const int64_t* d_uniq_keys, const float* d_staging,
                             const uint8_t* d_found, uint32_t stamps, uint32_t* d_stats, int cu_count,
                             hipStream_t stream, KTimer kt = {});

// control words over the compute queue (kernels.hip): call block host -> HBM (bytes rounded up to 16), accumulator words
// HBM -> host followed by a sequence word
hipError_t LaunchPull16(const void* src_host_devptr, void* dst, size_t bytes, hipStream_t stream);
// a range of any size and alignment out of page-locked host memory (src and dst equally misaligned to 16 B): staged KEYS, see engine.cpp
hipError_t LaunchPullBytes(const void* src_host_devptr, void* dst, size_t bytes, hipStream_t stream);
hipError_t LaunchPushWords(const uint32_t* src, uint32_t* dst_host_devptr, uint32_t words, uint32_t* seq_host_devptr, uint32_t seq,
                           hipStream_t stream);
// every key slot of the bucket lines EMPTY, every recency stamp = stamp8
hipError_t LaunchCacheClear(int64_t* d_lines, uint64_t num_buckets, uint32_t stamp8, hipStream_t stream);

hipError_t LaunchCacheQuery(const TableCacheDev& tb, const int64_t* d_keys, uint64_t n, int32_t* d_slot,
                            hipStream_t stream);

}  // namespace hps
