// Device-driven parameter-server tier ("ps_direct_access").
//
// The prescribed miss path gathers missed rows with host threads into pinned staging and ships them with
// hipMemcpyAsync.  On the MI355X boxes of this project the host side is the scarce resource (a 16-CPU cgroup
// quota next to a GPU that finishes its share in 0.3 ms), and a GPU kernel reading 512-B rows straight out of
// pinned host memory reaches the same PCIe rate as the DMA engine (55 GB/s measured, tools/micro/pcie_gather.hip).
// With 288 GB of HBM the *index* of the whole host tier fits on the device (12 B per slot, load factor <= 0.5:
// 10.5 GB for the 260 M rows of BASELINE config 2), so the GPU can resolve key -> host row itself:
//
//   hps_psindex_build      open-addressing index (key -> row number) built on the device from the pinned key array
//   hps_missdesc_build     per-table staging offsets of this call's unique misses (what the host computes otherwise)
//   hps_ps_fetch_direct    unique missed key -> index probe (HBM) -> 512-B row read over PCIe -> staging in HBM
//
// Staging, scatter and insert are shared with the host-gather path (kernels.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "device_types.h"
#include "direct_kernels.h"

namespace hps {

typedef float f4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t idx_hash(int64_t key) { return hps_mix64((uint64_t)key ^ 0xA24BAED4963EE407ull); }

__global__ void hps_psindex_clear_kernel(int64_t* keys, uint32_t* rows, uint64_t cap) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
    keys[i] = HPS_EMPTY_KEY;
    rows[i] = 0;
  }
}

// one thread per table row; duplicate keys: the highest row number wins
__global__ void hps_psindex_build_kernel(const int64_t* __restrict__ table_keys, uint64_t R, int64_t* __restrict__ keys,
                                         uint32_t* __restrict__ rows, uint64_t mask, uint32_t* __restrict__ sentinel) {
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (uint64_t)gridDim.x * blockDim.x) {
    const int64_t key = table_keys[r];
    if (key == HPS_EMPTY_KEY) {  // legal key that collides with the empty marker: kept on the side
      atomicMax(&sentinel[1], (uint32_t)r);
      sentinel[0] = 1;
      continue;
    }
    uint64_t s = idx_hash(key) & mask;
    for (;;) {
      const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[s]),
                                                (unsigned long long)HPS_EMPTY_KEY, (unsigned long long)key);
      if (prev == (unsigned long long)HPS_EMPTY_KEY || prev == (unsigned long long)key) {
        atomicMax(&rows[s], (uint32_t)r);
        break;
      }
      s = (s + 1) & mask;
    }
  }
}

// single block: staging layout of the call's unique misses, from the per-table unique counts K_M left in the call's
// accumulator block (device_types.h)
__global__ void hps_missdesc_build_kernel(const TableCacheDev* __restrict__ tables, uint32_t T, uint32_t* __restrict__ acc,
                                          MissDesc* __restrict__ md, uint32_t clear_stats,
                                          const uint32_t* __restrict__ table_mode) {
  if (blockIdx.x != 0) return;
  if (clear_stats)  // a job that re-uses its accumulator block (the background inserter): saves a memset launch
    for (uint32_t i = threadIdx.x; i < (uint32_t)kStatLines * kAccStride; i += blockDim.x) acc[i] = 0;
  if (threadIdx.x != 0) return;
  uint64_t fl = 0, uq = 0;
  for (uint32_t t = 0; t < T; ++t) {
    // table_mode (optional): tables in async-insert mode (1) take no part in the synchronous miss path
    uint32_t c = acc[AccTableWord(t, kAccUniqMiss)];
    if (table_mode && table_mode[t] != 0) c = 0;
    fl = (fl + 3) & ~(uint64_t)3;
    md->useg_start[t] = uq;
    md->stage_off[t] = fl;
    md->chunk_lo[t] = 0;
    md->chunk_hi[t] = c;
    fl += (uint64_t)c * tables[t].dim;
    uq += c;
  }
  md->useg_start[T] = uq;
}

__device__ __forceinline__ int find_table_d(const uint64_t* ks, int T, uint64_t i) {
  int lo = 0, hi = T;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ks[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// One 16-lane group per unique missed key: the group reads 16 consecutive index slots (128 B) per probe step,
// then moves the row host -> staging with 16-B loads per lane (two coalesced 256-B PCIe read bursts for D=128).
__global__ __launch_bounds__(256) void hps_ps_fetch_direct_kernel(const PsIndexDev* __restrict__ index, uint32_t T,
                                                                  const MissDesc* __restrict__ md,
                                                                  const uint64_t* __restrict__ key_start,
                                                                  const int64_t* __restrict__ uniq_keys,
                                                                  float* __restrict__ staging, uint8_t* __restrict__ found) {
  const uint64_t total = md->useg_start[T];
  const int lane = threadIdx.x & 63, g = lane >> 4, lig = lane & 15;
  const uint64_t groups_total = (uint64_t)gridDim.x * 16;
  for (uint64_t f = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4); f < total; f += groups_total) {
    const int t = find_table_d(md->useg_start, (int)T, f);
    const PsIndexDev ix = index[t];
    const uint32_t u = (uint32_t)(f - md->useg_start[t]);
    const int64_t key = uniq_keys[key_start[t] + u];
    const uint32_t D = ix.dim;
    float* dst = staging + md->stage_off[t] + (uint64_t)u * D;
    int64_t row = -1;
    if (key == HPS_EMPTY_KEY) {
      row = ix.has_sentinel ? (int64_t)ix.sentinel_row : -1;
    } else if (ix.keys != nullptr) {
      const uint64_t h = idx_hash(key) & ix.mask;
      uint64_t blk = h >> 4;
      const uint64_t nblk = (ix.mask + 1) >> 4;
      uint32_t first_lane = (uint32_t)(h & 15);
      for (uint64_t step = 0; step < nblk; ++step) {
        const uint64_t s = (blk << 4) + lig;
        const int64_t k = ix.keys[s];
        const uint32_t hit = (uint32_t)(__ballot(k == key) >> (g * 16)) & 0xFFFFu;
        if (hit) { row = (int64_t)ix.rows[(blk << 4) + (uint32_t)__builtin_ctz(hit)]; break; }
        const uint32_t empty = (uint32_t)(__ballot(k == HPS_EMPTY_KEY && (uint32_t)lig >= first_lane) >> (g * 16)) & 0xFFFFu;
        if (empty) break;  // an empty slot ends the probe sequence: the key is not in the table
        blk = blk + 1 == nblk ? 0 : blk + 1;
        first_lane = 0;
      }
    }
    if (row >= 0) {
      const float* src = ix.host_rows + (uint64_t)row * D;  // pinned host memory, read over PCIe
      if ((D & 3u) == 0) {
        for (uint32_t c = (uint32_t)lig * 4; c < D; c += 64) *reinterpret_cast<f4d*>(dst + c) = *reinterpret_cast<const f4d*>(src + c);
      } else {
        for (uint32_t c = (uint32_t)lig; c < D; c += 16) dst[c] = src[c];
      }
    } else {
      for (uint32_t c = (uint32_t)lig; c < D; c += 16) dst[c] = ix.default_value;
    }
    if (lig == 0) found[f] = row >= 0 ? 1 : 0;
  }
}

hipError_t LaunchPsIndexBuild(const int64_t* table_keys_devptr, uint64_t R, int64_t* d_keys, uint32_t* d_rows, uint64_t cap,
                              uint32_t* d_sentinel, hipStream_t stream) {
  uint64_t want = (cap + 255) / 256;
  if (want > 8192) want = 8192;
  hipLaunchKernelGGL(hps_psindex_clear_kernel, dim3((uint32_t)want), dim3(256), 0, stream, d_keys, d_rows, cap);
  if (R == 0) return hipGetLastError();
  want = (R + 255) / 256;
  if (want > 8192) want = 8192;
  hipLaunchKernelGGL(hps_psindex_build_kernel, dim3((uint32_t)want), dim3(256), 0, stream, table_keys_devptr, R, d_keys, d_rows,
                     cap - 1, d_sentinel);
  return hipGetLastError();
}

hipError_t LaunchMissDescBuild(const TableCacheDev* d_tables, uint32_t T, uint32_t* d_acc, MissDesc* d_md, bool clear_stats,
                               const uint32_t* d_table_mode, hipStream_t stream) {
  hipLaunchKernelGGL(hps_missdesc_build_kernel, dim3(1), dim3(64), 0, stream, d_tables, T, d_acc, d_md, clear_stats ? 1u : 0u,
                     d_table_mode);
  return hipGetLastError();
}

hipError_t LaunchPsFetchDirect(const PsIndexDev* d_index, uint32_t T, const MissDesc* d_md, const uint64_t* d_key_start,
                               const int64_t* d_uniq_keys, float* d_staging, uint8_t* d_found, uint64_t max_unique,
                               int grid_blocks, hipStream_t stream) {
  if (max_unique == 0) return hipSuccess;
  uint64_t want = (max_unique + 15) / 16;
  // Grid = rows in flight over PCIe (16 per block), not chip occupancy: ~512 rows in flight saturate the link
  // (tools/micro/pcie_contention.hip), each group also spends part of its time in the index probe, and every
  // extra wave only costs the HBM-bound kernels of the other sessions running underneath.
  static const int max_blocks = [] {
    const char* e = getenv("HPS_DIRECT_FETCH_BLOCKS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 128;
  }();
  // grid_blocks > 0: the caller's own bound (the background inserter runs a small grid: it is in no hurry, and
  // fewer PCIe reads in flight disturb the probe kernels of the foreground lookups less)
  const uint64_t cap = (uint64_t)(grid_blocks > 0 ? grid_blocks : max_blocks);
  if (want > cap) want = cap;
  hipLaunchKernelGGL(hps_ps_fetch_direct_kernel, dim3((uint32_t)want), dim3(256), 0, stream, d_index, T, d_md, d_key_start,
                     d_uniq_keys, d_staging, d_found);
  return hipGetLastError();
}

}  // namespace hps
