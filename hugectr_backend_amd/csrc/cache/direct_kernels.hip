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

// One 16-lane group per unique missed key, kRows keys per group and step: the group reads 16 consecutive index slots
// (128 B) per key and probe step, then moves the rows host -> staging with 16-B loads per lane (two coalesced 256-B PCIe
// read bursts per D=128 row).  Round 3 (last session): a group used to walk ONE key at a time through a chain of dependent
// accesses — table search over md in global memory, key, index line (HBM), row (PCIe, a few microseconds) — so a group had
// a row on the link for about half of its time, and the kernel moved 41-43 GB/s of the 53-55 the link delivers (fetch
// 0.9 ms for 37.5 MB, with or without other kernels next to it).  Now the per-table words sit in LDS, the kRows index
// lines of a step are in flight together, and so are its kRows rows (2 x kRows 16-B loads per lane).
template <int kRows>
__global__ __launch_bounds__(256) void hps_ps_fetch_direct_kernel(const PsIndexDev* __restrict__ index, uint32_t T,
                                                                  const MissDesc* __restrict__ md,
                                                                  const uint64_t* __restrict__ key_start,
                                                                  const int64_t* __restrict__ uniq_keys,
                                                                  float* __restrict__ staging, uint8_t* __restrict__ found) {
  extern __shared__ __attribute__((aligned(16))) char fd_smem[];
  uint64_t* sh_us = reinterpret_cast<uint64_t*>(fd_smem);   // [T + 1] unique-segment starts
  uint64_t* sh_ks = sh_us + (T + 1);                         // [T] key_start
  uint64_t* sh_so = sh_ks + T;                               // [T] stage_off
  PsIndexDev* sh_ix = reinterpret_cast<PsIndexDev*>(sh_so + T);   // [T]
  for (uint32_t t = threadIdx.x; t <= T; t += blockDim.x) sh_us[t] = md->useg_start[t];
  for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) { sh_ks[t] = key_start[t]; sh_so[t] = md->stage_off[t]; sh_ix[t] = index[t]; }
  __syncthreads();
  const uint64_t total = sh_us[T];
  const int lane = threadIdx.x & 63, g = lane >> 4, lig = lane & 15;
  const uint64_t groups_total = (uint64_t)gridDim.x * 16;
  for (uint64_t f0 = ((uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4)) * kRows; f0 < total; f0 += groups_total * kRows) {
    int tt[kRows];
    int64_t key[kRows], row[kRows], k0[kRows];
    uint64_t blk[kRows];
    float* dst[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const uint64_t f = f0 + r < total ? f0 + r : f0;   // (a group's last step may be short: its spare rows repeat the first)
      tt[r] = find_table_d(sh_us, (int)T, f);
      const uint32_t u = (uint32_t)(f - sh_us[tt[r]]);
      key[r] = uniq_keys[sh_ks[tt[r]] + u];
      dst[r] = staging + sh_so[tt[r]] + (uint64_t)u * sh_ix[tt[r]].dim;
    }
    // first index block of every key: the kRows 128-B lines are in flight together
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const PsIndexDev& ix = sh_ix[tt[r]];
      blk[r] = (idx_hash(key[r]) & ix.mask) >> 4;
      k0[r] = (ix.keys != nullptr && key[r] != HPS_EMPTY_KEY) ? ix.keys[(blk[r] << 4) + lig] : HPS_EMPTY_KEY;
    }
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const PsIndexDev& ix = sh_ix[tt[r]];
      row[r] = -1;
      if (key[r] == HPS_EMPTY_KEY) {
        row[r] = ix.has_sentinel ? (int64_t)ix.sentinel_row : -1;
      } else if (ix.keys != nullptr) {
        const uint64_t nblk = (ix.mask + 1) >> 4;
        uint32_t first_lane = (uint32_t)((idx_hash(key[r]) & ix.mask) & 15);
        int64_t k = k0[r];
        for (uint64_t step = 0; step < nblk; ++step) {
          const uint32_t hit = (uint32_t)(__ballot(k == key[r]) >> (g * 16)) & 0xFFFFu;
          if (hit) { row[r] = (int64_t)ix.rows[(blk[r] << 4) + (uint32_t)__builtin_ctz(hit)]; break; }
          const uint32_t empty = (uint32_t)(__ballot(k == HPS_EMPTY_KEY && (uint32_t)lig >= first_lane) >> (g * 16)) & 0xFFFFu;
          if (empty) break;  // an empty slot ends the probe sequence: the key is not in the table
          blk[r] = blk[r] + 1 == nblk ? 0 : blk[r] + 1;
          first_lane = 0;
          k = ix.keys[(blk[r] << 4) + lig];
        }
      }
    }
    // rows: pinned host memory, read over PCIe
    bool fast = true;
#pragma unroll
    for (int r = 0; r < kRows; ++r) fast = fast && sh_ix[tt[r]].dim == 128u;
    if (fast) {
      f4d v[kRows][2];
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        if (row[r] >= 0) {
          const float* src = sh_ix[tt[r]].host_rows + (uint64_t)row[r] * 128u;
          v[r][0] = *reinterpret_cast<const f4d*>(src + lig * 4);
          v[r][1] = *reinterpret_cast<const f4d*>(src + 64 + lig * 4);
        } else {
          const float dv = sh_ix[tt[r]].default_value;
          v[r][0] = f4d{dv, dv, dv, dv};
          v[r][1] = v[r][0];
        }
      }
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        if (f0 + r < total) {
          *reinterpret_cast<f4d*>(dst[r] + lig * 4) = v[r][0];
          *reinterpret_cast<f4d*>(dst[r] + 64 + lig * 4) = v[r][1];
        }
      }
    } else {
      // any width: the first chunk of every row (16 B per lane where the width allows, else 4 B) is in flight together; what a
      // row has beyond 64 (16) floats follows row by row
      f4d v4[kRows];
      float v1[kRows];
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        const PsIndexDev& ix = sh_ix[tt[r]];
        const uint32_t D = ix.dim;
        v4[r] = f4d{ix.default_value, ix.default_value, ix.default_value, ix.default_value};
        v1[r] = ix.default_value;
        if (row[r] >= 0) {
          const float* src = ix.host_rows + (uint64_t)row[r] * D;
          if ((D & 3u) == 0) { if ((uint32_t)lig * 4 < D) v4[r] = *reinterpret_cast<const f4d*>(src + lig * 4); }
          else if ((uint32_t)lig < D) v1[r] = src[lig];
        }
      }
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        if (f0 + r >= total) continue;
        const PsIndexDev& ix = sh_ix[tt[r]];
        const uint32_t D = ix.dim;
        if ((D & 3u) == 0) {
          if ((uint32_t)lig * 4 < D) *reinterpret_cast<f4d*>(dst[r] + lig * 4) = v4[r];
          for (uint32_t c = (uint32_t)lig * 4 + 64; c < D; c += 64)
            *reinterpret_cast<f4d*>(dst[r] + c) = row[r] >= 0 ? *reinterpret_cast<const f4d*>(ix.host_rows + (uint64_t)row[r] * D + c) : v4[r];
        } else {
          if ((uint32_t)lig < D) dst[r][lig] = v1[r];
          for (uint32_t c = (uint32_t)lig + 16; c < D; c += 16) dst[r][c] = row[r] >= 0 ? ix.host_rows[(uint64_t)row[r] * D + c] : ix.default_value;
        }
      }
    }
    if (lig == 0) {
#pragma unroll
      for (int r = 0; r < kRows; ++r)
        if (f0 + r < total) found[f0 + r] = row[r] >= 0 ? 1 : 0;
    }
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
  // Grid = rows in flight over PCIe, not chip occupancy.  Round 3 (last session), 37.5 MB of missed rows per call out of a
  // 133-GB page-locked host tier, two sessions (profiles/round3/ab_direct_fetch_grid.txt): 32 blocks x 16 groups x 4 rows in
  // flight move 43 GB/s and leave the other session's probe and gather alone (59 / 214 us); 64 blocks saturate the link
  // (0.70 ms = 53.6 GB/s) but the probe next to them takes 151 us and the gather 241, and the step is no shorter; one
  // row per group needs 128 blocks for the same rate and disturbs more (probe 190 us).
  constexpr int max_blocks = 32;
  constexpr int rows_in_flight = 4;   // keys (index lines, then rows) in flight per 16-lane group
  // grid_blocks < 0: a small request (its whole miss path is tens of microseconds and nothing runs next to it long enough to
  // be disturbed): one key per group, up to 128 workgroups — every row of a few thousand misses on the link at once.  With the
  // big-request shape a 28,672-key W&D request took 0.233 / 0.136 / 0.112 ms at 50 / 90 / 99 % hit instead of 0.166 / 0.120 /
  // 0.100 (tests/tools/bench_configs.py c4 direct).
  const int rows = grid_blocks < 0 ? 1 : rows_in_flight;
  const uint64_t per_block = 16ull * (uint64_t)rows;
  uint64_t want = (max_unique + per_block - 1) / per_block;
  // grid_blocks > 0: the caller's own bound (the background inserter runs a small grid: it is in no hurry, and
  // fewer PCIe reads in flight disturb the probe kernels of the foreground lookups less)
  const uint64_t cap = (uint64_t)(grid_blocks > 0 ? grid_blocks : grid_blocks < 0 ? 128 : max_blocks);
  if (want > cap) want = cap;
  const size_t lds = (size_t)(T + 1) * 8 + (size_t)T * (8 + 8 + sizeof(PsIndexDev)) + 16;
  if (rows >= 4)
    hipLaunchKernelGGL(hps_ps_fetch_direct_kernel<4>, dim3((uint32_t)want), dim3(256), (uint32_t)lds, stream, d_index, T, d_md, d_key_start,
                       d_uniq_keys, d_staging, d_found);
  else
    hipLaunchKernelGGL(hps_ps_fetch_direct_kernel<1>, dim3((uint32_t)want), dim3(256), (uint32_t)lds, stream, d_index, T, d_md, d_key_start,
                       d_uniq_keys, d_staging, d_found);
  return hipGetLastError();
}

}  // namespace hps
