// Launchers of direct_kernels.hip (device-driven parameter-server tier, "ps_direct_access").
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "device_types.h"

namespace hps {

// Device-resident index of one host table + where its rows are (pinned host memory mapped into the device).
struct PsIndexDev {
  int64_t* keys;           // [mask+1] open addressing, HPS_EMPTY_KEY = free; nullptr for an empty table
  uint32_t* rows;          // [mask+1] row number in the host slab
  uint64_t mask;           // capacity - 1 (power of two, >= 16)
  const float* host_rows;  // device view of the pinned R x D slab
  uint32_t dim;
  uint32_t has_sentinel;   // HPS_EMPTY_KEY itself is a key of the table
  uint32_t sentinel_row;
  float default_value;
};

hipError_t LaunchPsIndexBuild(const int64_t* table_keys_devptr, uint64_t R, int64_t* d_keys, uint32_t* d_rows, uint64_t cap,
                              uint32_t* d_sentinel /*[2]: flag, row*/, hipStream_t stream);
// d_acc: a call's accumulator block (device_types.h); clear_stats: also zero its insert-statistics lines
hipError_t LaunchMissDescBuild(const TableCacheDev* d_tables, uint32_t T, uint32_t* d_acc, MissDesc* d_md, bool clear_stats,
                               const uint32_t* d_table_mode /*optional: 1 = skip table*/, hipStream_t stream);
hipError_t LaunchPsFetchDirect(const PsIndexDev* d_index, uint32_t T, const MissDesc* d_md, const uint64_t* d_key_start,
                               const int64_t* d_uniq_keys, float* d_staging, uint8_t* d_found, uint64_t max_unique,
                               int grid_blocks /*0: default (big request); > 0: at most this many workgroups; < 0: small request, one key per group, up to 128 workgroups*/, hipStream_t stream);

}  // namespace hps
