// Lookup fused into the dot interaction (BASELINE config 5, second arrangement): the interaction kernel reads the
// embedding rows straight from where the probe found them — cache slot or miss staging — so OUTPUT0 (4*D bytes
// written and read again per lookup, 872 MB + 872 MB per 64 K batch at T = 26, D = 128) never exists.
// Same arithmetic and output layout as hps_dense_interact_kernel (dense_kernels.hip); only the row source differs:
//   slot >= 0   row = tables[t].rows + slot * D            (hit: the slot the probe recorded)
//   slot <= -2  row = staging + stage_off[t] + uidx_of[rep_of[-2 - slot]] * D   (miss: the row the fetch staged, or the
//               default vector; -2 - slot is the key's entry in the probe tiles' miss lists, cache/device_types.h)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../cache/device_types.h"
#include "dense.h"

namespace hps {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr uint32_t kFTriElems = 32 * 31 / 2 + 16;
constexpr int kFZPad = 8;

// NR float4 chunks per lane (NR*64 >= T*D/4), one wave per sample, 4 samples per block.
template <int NR>
__global__ __launch_bounds__(256) void hps_lookup_interact_kernel(const TableCacheDev* __restrict__ tables, const MissDesc* __restrict__ md,
                                                                  const int32_t* __restrict__ slot_in, const int32_t* __restrict__ rep_of,
                                                                  const int32_t* __restrict__ uidx_of, const float* __restrict__ staging,
                                                                  const _Float16* __restrict__ bottom, uint64_t batch, uint32_t T, uint32_t D,
                                                                  uint32_t out_stride, _Float16* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) _Float16 zlds[];
  __shared__ const float* sh_rows[32];
  __shared__ uint64_t sh_stage[32];
  if (threadIdx.x < T) {
    sh_rows[threadIdx.x] = tables[threadIdx.x].rows;
    sh_stage[threadIdx.x] = md->stage_off[threadIdx.x];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const uint32_t V = T + 1;
  const uint32_t npairs = V * (V - 1) / 2;
  const uint32_t zstride = D + kFZPad;
  _Float16* z = zlds + (size_t)wave * (32 * zstride + kFTriElems);
  _Float16* tri = z + 32 * zstride;
  const uint32_t d4 = D >> 2;
  const uint32_t nchunks = T * d4;
  uint32_t tq[NR];  // chunk u of this lane: table << 16 | first column (out-of-range chunks repeat the last one)
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const uint32_t c = u * 64 + lane;
    const uint32_t ce = c < nchunks ? c : nchunks - 1;
    const uint32_t t = ce / d4, q = ce - t * d4;
    tq[u] = (t << 16) | (q * 4);
  }
  const uint64_t waves_total = (uint64_t)gridDim.x * 4;
  for (uint64_t i = (uint64_t)blockIdx.x * 4 + wave; i < batch; i += waves_total) {
    for (uint32_t c = lane * 8; c < D; c += 512) *reinterpret_cast<h8*>(z + c) = *reinterpret_cast<const h8*>(bottom + i * D + c);
    // straight-line: all slot loads, then all row loads (a branch per chunk makes the compiler serialise them)
    int32_t s[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) s[u] = slot_in[(uint64_t)(tq[u] >> 16) * batch + i];
    // missed keys (few): entry of the tile miss lists -> call-wide representative -> row of the unique-miss segment
    uint32_t mu[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) mu[u] = s[u] <= -2 ? (uint32_t)uidx_of[(uint32_t)rep_of[(uint32_t)(-2 - s[u])]] : 0u;
    f4v v[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const uint32_t t = tq[u] >> 16, col = tq[u] & 0xFFFFu;
      const float* hit = sh_rows[t] + (uint64_t)(uint32_t)(s[u] >= 0 ? s[u] : 0) * D;
      const float* mis = staging + sh_stage[t] + (uint64_t)mu[u] * D;
      const float* src = s[u] >= 0 ? hit : mis;
      v[u] = *reinterpret_cast<const f4v*>(src + col);
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      h4 w;
      w[0] = (_Float16)v[u][0]; w[1] = (_Float16)v[u][1]; w[2] = (_Float16)v[u][2]; w[3] = (_Float16)v[u][3];
      if ((uint32_t)(u * 64 + lane) < nchunks) *reinterpret_cast<h4*>(z + (1 + (tq[u] >> 16)) * zstride + (tq[u] & 0xFFFFu)) = w;
    }
    __builtin_amdgcn_wave_barrier();
    f16x acc = {0};
    const bool live = (uint32_t)r < V;
    const _Float16* zr = z + (size_t)(live ? r : 0) * zstride + 8 * h;
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
    for (uint32_t k0 = 0; k0 < D; k0 += 16) {
      const h8 ld = *reinterpret_cast<const h8*>(zr + k0);
      const h8 f = live ? ld : zero8;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, f, acc, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const uint32_t a = (uint32_t)((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)), b = (uint32_t)r;
      const uint32_t idx = (a < V && b < a) ? a * (a - 1) / 2 + b : 496u + (lane & 15);
      tri[idx] = (_Float16)acc[reg];
    }
    __builtin_amdgcn_wave_barrier();
    _Float16* o = out + i * out_stride;
    for (uint32_t c = lane * 8; c < D; c += 512) *reinterpret_cast<h8*>(o + c) = *reinterpret_cast<const h8*>(z + c);
    const uint32_t tail = out_stride - D;
    for (uint32_t c = lane * 2; c < tail; c += 128) {
      h2 w;
      w[0] = c < npairs ? tri[c] : (_Float16)0.f;
      w[1] = c + 1 < npairs ? tri[c + 1] : (_Float16)0.f;
      *reinterpret_cast<h2*>(o + D + c) = w;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

hipError_t LaunchLookupInteract(const TableCacheDev* d_tables, const MissDesc* d_md, const int32_t* d_slot, const int32_t* d_rep_of,
                                const int32_t* d_uidx_of, const float* d_staging,
                                const void* d_bottom_f16, uint64_t batch, uint32_t T, uint32_t D, uint32_t out_stride, void* d_out_f16,
                                int cu_count, hipStream_t stream) {
  if (batch == 0) return hipSuccess;
  const uint32_t per_lane = (T * (D / 4) + 63) / 64;
  if (per_lane > 16 || T > 31 || D >= 65536) return hipErrorInvalidValue;
  const size_t lds_bytes = 4 * (32 * (size_t)(D + kFZPad) + kFTriElems) * sizeof(_Float16);
  uint64_t want = (batch + 3) / 4;
  uint64_t per_cu = (150u << 10) / lds_bytes;
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  const uint64_t cap = (uint64_t)cu_count * per_cu;
  if (want > cap) want = cap;
  auto go = [&](auto kernel) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3((uint32_t)want), dim3(256), lds_bytes, stream, d_tables, d_md, d_slot, d_rep_of, d_uidx_of, d_staging,
                       reinterpret_cast<const _Float16*>(d_bottom_f16), batch, T, D, out_stride, reinterpret_cast<_Float16*>(d_out_f16));
    return hipGetLastError();
  };
  if (per_lane <= 4) return go(hps_lookup_interact_kernel<4>);
  if (per_lane <= 8) return go(hps_lookup_interact_kernel<8>);
  if (per_lane <= 13) return go(hps_lookup_interact_kernel<13>);   // T = 26, D = 128: 13 chunks exactly (fewer registers than <16>)
  return go(hps_lookup_interact_kernel<16>);
}

}  // namespace hps
