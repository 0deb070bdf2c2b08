// BASELINE config 5: the dense step that consumes the lookup's OUTPUT0 in place on the same GPU —
// DLRM bottom MLP over the numeric features and the pairwise dot interaction of {bottom output, T embedding rows}.
// Not part of the reference backend (there the dense model is a second Triton backend behind an ensemble hand-off,
// SURVEY.md §2 K8 / §8f rank 2).  fp16 operands, fp32 accumulation on the matrix cores
// (v_mfma_f32_32x32x16_f16, gfx950).  Both kernels are HBM-bound: the interaction reads every embedding row once
// (4*D bytes per lookup), the MLP's 22 GFLOP per 64 K batch are noise next to that.
//
// Fragment convention (32x32x16): lane l supplies row (A) / column (B) `l & 31` and the eight k values
// `16*s + 8*(l >> 5) .. +7` of K-step s.  A and B use the same k subset per lane, which is all the instruction
// needs for a correct sum over k.  Accumulator: col = l & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(l >> 5).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "dense.h"

namespace hps {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int acc_row(int lane, int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ------------------------------------------------------------------------------------------------------------
// Bottom MLP: up to kDenseMaxLayers fully connected layers with ReLU, 64 samples (two MFMA M tiles) per 512-thread
// block, activations alternate between two LDS buffers, weights (f16, pre-arranged on the host in MFMA fragment
// order: [column tile][K-step][lane][8], so a wave's fragment load is one contiguous 1-KB read instead of 32 rows
// at a K*2-byte stride, which lands on a quarter of the L2 channels) streamed from L2.  Every weight fragment feeds both M tiles: the weights are re-read once per block (340 KB x batch/64 =
// 350 MB of L2 traffic per 64 K batch, the cost that sizes the block).  The K loop fetches the fragments of 8
// K-steps with straight-line loads before the MFMAs that use them; a one-load-one-MFMA loop keeps a single L2
// request in flight per wave.
// ------------------------------------------------------------------------------------------------------------
constexpr int kMlpRows = 64;      // samples per block
constexpr int kMlpThreads = 512;  // 8 waves
constexpr int kLdsPad = 8;        // f16 elements of padding per activation row (keeps ds_read_b128 conflict-free)

__global__ __launch_bounds__(kMlpThreads) void hps_dense_mlp_kernel(DenseMlpDesc d, const float* __restrict__ x, uint64_t batch,
                                                            _Float16* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const uint32_t stride0 = d.buf_dim[0] + kLdsPad, stride1 = d.buf_dim[1] + kLdsPad;
  _Float16* buf0 = lds;
  _Float16* buf1 = lds + (size_t)kMlpRows * stride0;
  for (uint64_t m_base = (uint64_t)blockIdx.x * kMlpRows; m_base < batch; m_base += (uint64_t)gridDim.x * kMlpRows) {
    // stage the numeric features as f16, zero-padded to in_pad columns and to 64 rows
    for (uint32_t i = threadIdx.x; i < kMlpRows * d.in_pad; i += kMlpThreads) {
      const uint32_t m = i / d.in_pad, k = i % d.in_pad;
      float v = 0.f;
      if (m_base + m < batch && k < d.in_dim) v = x[(m_base + m) * d.in_dim + k];
      buf0[m * stride0 + k] = (_Float16)v;
    }
    __syncthreads();
    uint32_t K = d.in_pad;
    for (uint32_t l = 0; l < d.num_layers; ++l) {
      const uint32_t Nout = d.dims[l];
      const _Float16* __restrict__ W = reinterpret_cast<const _Float16*>(d.weights[l]);  // [Nout][K] f16
      const float* __restrict__ bias = d.biases[l];
      const _Float16* src = (l & 1) ? buf1 : buf0;
      _Float16* dst = (l & 1) ? buf0 : buf1;
      const uint32_t sstride = (l & 1) ? stride1 : stride0, dstride = (l & 1) ? stride0 : stride1;
      // N tiles of 32 columns are dealt round-robin to the 8 waves; a wave does both M tiles of its column tile
      for (uint32_t nt = wave; nt < Nout / 32; nt += kMlpThreads / 64) {
        f16x acc0 = {0}, acc1 = {0};
        // weights are stored in fragment order: one contiguous 1-KB block per (column tile, K-step)
        const _Float16* wrow = W + ((size_t)nt * (K / 16) * 64 + lane) * 8;
        const _Float16* a0 = src + (size_t)r * sstride + 8 * h;
        const _Float16* a1 = src + (size_t)(32 + r) * sstride + 8 * h;
        uint32_t k0 = 0;
        for (; k0 + 128 <= K; k0 += 128) {
          h8 bw[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) bw[u] = *reinterpret_cast<const h8*>(wrow + (size_t)(k0 / 16 + u) * 512);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const h8 fa0 = *reinterpret_cast<const h8*>(a0 + k0 + 16 * u);
            const h8 fa1 = *reinterpret_cast<const h8*>(a1 + k0 + 16 * u);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, bw[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1, bw[u], acc1, 0, 0, 0);
          }
        }
        for (; k0 < K; k0 += 16) {
          const h8 b = *reinterpret_cast<const h8*>(wrow + (size_t)(k0 / 16) * 512);
          const h8 fa0 = *reinterpret_cast<const h8*>(a0 + k0);
          const h8 fa1 = *reinterpret_cast<const h8*>(a1 + k0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1, b, acc1, 0, 0, 0);
        }
        const uint32_t n = nt * 32 + r;
        const float bn = bias[n];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          float v0 = acc0[reg] + bn, v1 = acc1[reg] + bn;
          v0 = v0 > 0.f ? v0 : 0.f;
          v1 = v1 > 0.f ? v1 : 0.f;
          const size_t m0 = (size_t)acc_row(lane, reg);
          dst[m0 * dstride + n] = (_Float16)v0;
          dst[(32 + m0) * dstride + n] = (_Float16)v1;
        }
      }
      __syncthreads();
      K = Nout;
    }
    // the last layer's rows leave LDS with 16-byte coalesced stores
    {
      const uint32_t L = d.num_layers, Nout = d.dims[L - 1];
      const _Float16* fin = ((L - 1) & 1) ? buf0 : buf1;
      const uint32_t fstride = ((L - 1) & 1) ? stride0 : stride1;
      const uint32_t per_row = Nout / 8;
      for (uint32_t i = threadIdx.x; i < kMlpRows * per_row; i += kMlpThreads) {
        const uint32_t m = i / per_row, c = (i % per_row) * 8;
        if (m_base + m < batch)
          *reinterpret_cast<h8*>(out + (m_base + m) * Nout + c) = *reinterpret_cast<const h8*>(fin + (size_t)m * fstride + c);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------
// Bottom MLP, 128 samples per block (the default when every layer is at most 512 wide).  The 64-row kernel above re-reads
// the weights from L2 once per 64 samples (340 KB x 1,024 blocks = 350 MB per 64 K batch: 58 us, L2-bound).  Here a wave
// owns whole COLUMN tiles and feeds all four 32-row M tiles from each weight fragment it loads, so the weights are read once
// per 128 samples where the layer has at least 8 column tiles (narrower layers split the M tiles over the waves instead of
// leaving waves idle: a 128-wide layer is read twice).  One LDS buffer holds the activations of all 128 rows; a layer's
// outputs stay in the accumulators until every wave has finished reading its inputs (barrier), then overwrite them in
// place — 128 x (512 + 8) f16 = 133 KB, one block per CU.
// ------------------------------------------------------------------------------------------------------------
constexpr int kMlp2Rows = 128;
constexpr int kMlp2Threads = 512;   // 8 waves, 256 VGPRs each
constexpr int kMlp2MaxUnits = 2;    // (column tile, group of M tiles) units a wave may own in one layer: layers up to 512 wide

__global__ __launch_bounds__(kMlp2Threads) void hps_dense_mlp128_kernel(DenseMlpDesc d, const float* __restrict__ x, uint64_t batch,
                                                                       _Float16* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const uint32_t stride = d.max_dim + kLdsPad;
  _Float16* act = lds;
  for (uint64_t m_base = (uint64_t)blockIdx.x * kMlp2Rows; m_base < batch; m_base += (uint64_t)gridDim.x * kMlp2Rows) {
    // numeric features as f16, zero-padded to in_pad columns and to 128 rows
    for (uint32_t i = threadIdx.x; i < kMlp2Rows * d.in_pad; i += kMlp2Threads) {
      const uint32_t m = i / d.in_pad, k = i % d.in_pad;
      float v = 0.f;
      if (m_base + m < batch && k < d.in_dim) v = x[(m_base + m) * d.in_dim + k];
      act[m * stride + k] = (_Float16)v;
    }
    __syncthreads();
    uint32_t K = d.in_pad;
    for (uint32_t l = 0; l < d.num_layers; ++l) {
      const uint32_t Nout = d.dims[l];
      const uint32_t ctiles = Nout / 32;
      // units: (column tile, group of `mper` consecutive M tiles); at least 8 of them so that every wave has work
      const uint32_t msplit = ctiles >= 8 ? 1u : (ctiles >= 4 ? 2u : 4u);
      const uint32_t mper = 4u / msplit;
      const uint32_t units = ctiles * msplit;
      const _Float16* __restrict__ W = reinterpret_cast<const _Float16*>(d.weights[l]);
      const float* __restrict__ bias = d.biases[l];
      f16x acc[kMlp2MaxUnits][4];
#pragma unroll
      for (int u = 0; u < kMlp2MaxUnits; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[u][t] = f16x{0};
#pragma unroll
      for (int u = 0; u < kMlp2MaxUnits; ++u) {
        const uint32_t unit = (uint32_t)wave + 8u * (uint32_t)u;
        if (unit < units) {
          const uint32_t nt = unit / msplit, m0 = (unit % msplit) * mper;   // column tile, first M tile of the group
          const _Float16* wrow = W + ((size_t)nt * (K / 16) * 64 + lane) * 8;
          const _Float16* a = act + (size_t)(32 * m0 + r) * stride + 8 * h;
          uint32_t k0 = 0;
          for (; k0 + 128 <= K; k0 += 128) {
            h8 bw[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) bw[q] = *reinterpret_cast<const h8*>(wrow + (size_t)(k0 / 16 + q) * 512);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                if ((uint32_t)t < mper) {
                  const h8 fa = *reinterpret_cast<const h8*>(a + (size_t)(32 * t) * stride + k0 + 16 * q);
                  acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, bw[q], acc[u][t], 0, 0, 0);
                }
              }
            }
          }
          for (; k0 < K; k0 += 16) {
            const h8 b = *reinterpret_cast<const h8*>(wrow + (size_t)(k0 / 16) * 512);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              if ((uint32_t)t < mper) {
                const h8 fa = *reinterpret_cast<const h8*>(a + (size_t)(32 * t) * stride + k0);
                acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, b, acc[u][t], 0, 0, 0);
              }
            }
          }
        }
      }
      __syncthreads();   // every wave has read what it needs of this layer's input: the outputs may take its place
#pragma unroll
      for (int u = 0; u < kMlp2MaxUnits; ++u) {
        const uint32_t unit = (uint32_t)wave + 8u * (uint32_t)u;
        if (unit < units) {
          const uint32_t nt = unit / msplit, m0 = (unit % msplit) * mper;
          const uint32_t n = nt * 32 + r;
          const float bn = bias[n];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if ((uint32_t)t < mper) {
#pragma unroll
              for (int reg = 0; reg < 16; ++reg) {
                float v = acc[u][t][reg] + bn;
                v = v > 0.f ? v : 0.f;
                act[(size_t)(32 * (m0 + t) + acc_row(lane, reg)) * stride + n] = (_Float16)v;
              }
            }
          }
        }
      }
      __syncthreads();
      K = Nout;
    }
    {
      const uint32_t Nout = d.dims[d.num_layers - 1];
      const uint32_t per_row = Nout / 8;
      for (uint32_t i = threadIdx.x; i < kMlp2Rows * per_row; i += kMlp2Threads) {
        const uint32_t m = i / per_row, c = (i % per_row) * 8;
        if (m_base + m < batch)
          *reinterpret_cast<h8*>(out + (m_base + m) * Nout + c) = *reinterpret_cast<const h8*>(act + (size_t)m * stride + c);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------
// Dot interaction, one wave per sample.  Z = [bottom(i); emb(0,i); ... emb(T-1,i)] (V = T+1 <= 32 rows, D columns),
// G = Z Z^T on the matrix cores with the SAME register as A and B fragment, output row =
// [bottom (D f16) | G[a][b] for a in 1..V-1, b in 0..a-1 | zero padding to out_stride].
// emb is OUTPUT0 of the lookup: table-major fp32, row of (table t, sample i) at (t*batch + i)*D.
//
// The rows are read fully coalesced (the wave walks the T*D/4 float4 chunks of the sample, 8 loads in flight per
// lane), rounded to f16 into a per-wave LDS tile, and the MFMA fragments come from LDS: the fragment shape (32 rows
// x 8 k per half-wave) read straight from global memory touches every 64-byte segment twice and keeps one load in
// flight per step (250 us per 64 K batch against 115 us of HBM time).
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t kTriElems = 32 * 31 / 2 + 16;  // 496 pair slots + 16 spare slots for lanes without an element
constexpr int kZPad = 8;  // f16 of padding per LDS row: 16-byte fragment reads of 16 consecutive rows hit distinct banks

// NT: the embedding rows are read exactly once, by exactly one wave: non-temporal loads (round 4: 190 -> 178-186 us under
// rocprofv3; the read pattern alone streams at 6.0 TB/s with plain loads and 6.75 TB/s with non-temporal ones,
// tools/micro/read_stream.hip).  Where the rest of the time goes (tools/micro/interact_stages.hip, same box): the loads alone
// 132 us; + conversion and LDS tile, + the 8 MFMAs, + the triangle through LDS: still 132-134 us — all of it hides under the
// read stream; + the 63 MB of output rows: 172-177 us.  The 7 % of the bytes that are WRITES cost 42 us, whatever their shape:
// 4-B or 16-B stores per lane, one store instruction per row or four, rows 960 B or 1,024 B apart (whole 128-B lines),
// non-temporal or not (170-178 us in every combination).  A second register set that issues sample i+1's loads before sample
// i is touched changes nothing either (189.5 us, at 2 waves per SIMD instead of 3) and was withdrawn.
template <bool NT>
__device__ __forceinline__ f4v ld_row(const float* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
  return *reinterpret_cast<const f4v*>(p);
}

// NCH = float4 chunks per lane held in registers (NCH*64 >= T*D/4): the loads of the wave's NEXT sample are issued
// before the LDS/MFMA/store phase of the current one, so a wave always has a full sample (13 KB at T=26, D=128) in
// flight.  NCH = 0: any size, two-phase (8 loads in flight, none during the compute phase).
template <int NCH, bool NT>
__global__ __launch_bounds__(256) void hps_dense_interact_kernel(const float* __restrict__ emb, const _Float16* __restrict__ bottom,
                                                                 uint64_t batch, uint32_t T, uint32_t D, uint32_t out_stride,
                                                                 _Float16* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) _Float16 zlds[];
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const uint32_t V = T + 1;
  const uint32_t npairs = V * (V - 1) / 2;
  const uint32_t zstride = D + kZPad;
  _Float16* z = zlds + (size_t)wave * (32 * zstride + kTriElems);  // [V][zstride] rows, then the triangle
  _Float16* tri = z + 32 * zstride;
  const uint32_t d4 = D >> 2;                 // float4 chunks per row
  const uint32_t nchunks = T * d4;
  const uint64_t waves_total = (uint64_t)gridDim.x * 4;
  constexpr int NR = NCH > 0 ? NCH : 1;
  f4v pre[NR];
  uint64_t goff[NR];   // chunk u of this lane: float offset of (table, column) inside a sample's rows
  uint32_t zoff[NR];   // ... and its f16 offset in the LDS tile (0xFFFFFFFF: no such chunk)
  uint64_t i = (uint64_t)blockIdx.x * 4 + wave;
  if (NCH > 0) {
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const uint32_t c = u * 64 + lane;
      const uint32_t ce = c < nchunks ? c : nchunks - 1;   // out-of-range chunks re-read the last one (no branch)
      const uint32_t t = ce / d4, q = ce - t * d4;
      goff[u] = (uint64_t)t * batch * D + q * 4;
      zoff[u] = c < nchunks ? (1 + t) * zstride + q * 4 : 0xFFFFFFFFu;
    }
    if (i < batch) {
#pragma unroll
      for (int u = 0; u < NR; ++u) pre[u] = ld_row<NT>(emb + goff[u] + i * D);
    }
  }
  for (; i < batch; i += waves_total) {
    // ---- stage: bottom row (already f16) + T embedding rows (fp32 -> f16) ----
    for (uint32_t c = lane * 8; c < D; c += 512) *reinterpret_cast<h8*>(z + c) = *reinterpret_cast<const h8*>(bottom + i * D + c);
    if (NCH > 0) {
#pragma unroll
      for (int u = 0; u < NR; ++u) {
        h4 w;
        w[0] = (_Float16)pre[u][0]; w[1] = (_Float16)pre[u][1]; w[2] = (_Float16)pre[u][2]; w[3] = (_Float16)pre[u][3];
        if (zoff[u] != 0xFFFFFFFFu) *reinterpret_cast<h4*>(z + zoff[u]) = w;
      }
      const uint64_t nxt = i + waves_total < batch ? i + waves_total : i;   // the last round re-reads itself
#pragma unroll
      for (int u = 0; u < NR; ++u) pre[u] = ld_row<NT>(emb + goff[u] + nxt * D);
    } else {
      // (straight-line on purpose: with the loads under per-chunk branches the compiler drains vmcnt before each
      //  one and a wave keeps a single load in flight)
      for (uint32_t c0 = 0; c0 < nchunks; c0 += 64 * 8) {
        f4v v[8];
        uint32_t zo[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const uint32_t c = c0 + u * 64 + lane;
          const uint32_t ce = c < nchunks ? c : nchunks - 1;
          const uint32_t t = ce / d4, q = ce - t * d4;
          v[u] = ld_row<NT>(emb + ((uint64_t)t * batch + i) * D + q * 4);
          zo[u] = c < nchunks ? (1 + t) * zstride + q * 4 : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          h4 w;
          w[0] = (_Float16)v[u][0]; w[1] = (_Float16)v[u][1]; w[2] = (_Float16)v[u][2]; w[3] = (_Float16)v[u][3];
          if (zo[u] != 0xFFFFFFFFu) *reinterpret_cast<h4*>(z + zo[u]) = w;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- G = Z Z^T ----
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
    // ---- strict lower triangle -> LDS in output order (lanes without an element hit a spare slot), coalesced stores ----
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const uint32_t a = (uint32_t)acc_row(lane, reg), b = (uint32_t)r;
      const uint32_t idx = (a < V && b < a) ? a * (a - 1) / 2 + b : 496u + (lane & 15);
      tri[idx] = (_Float16)acc[reg];
    }
    __builtin_amdgcn_wave_barrier();
    _Float16* o = out + i * out_stride;
    for (uint32_t c = lane * 8; c < D; c += 512) *reinterpret_cast<h8*>(o + c) = *reinterpret_cast<const h8*>(z + c);
    // D is a multiple of 16 and out_stride of 8: the tail [D, out_stride) is written as aligned pairs
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

hipError_t LaunchDenseMlp(const DenseMlpDesc& d, const float* d_x, uint64_t batch, void* d_out_f16, int cu_count, hipStream_t stream) {
  if (batch == 0) return hipSuccess;
  // (A 256-row variant that fused the first two layers — the hidden layer produced 128 columns at a time into LDS and consumed
  //  at once as a K-chunk of layer 2, weights read once per 256 samples: 106 MB of L2 reads per batch instead of 205 — was
  //  measured at the same 235-240 us per forward as this one and withdrawn: below ~200 MB the kernel is no longer bound by the
  //  weight reads but by the serial chain of barriers and LDS round trips of one resident block per CU.)
  bool fits128 = batch > (uint64_t)kMlpRows;
  for (uint32_t l = 0; l < d.num_layers; ++l) fits128 = fits128 && d.dims[l] / 32 <= 8u * kMlp2MaxUnits;
  const size_t lds128 = (size_t)kMlp2Rows * (d.max_dim + kLdsPad) * sizeof(_Float16);
  if (fits128 && lds128 <= (160u << 10)) {
    uint64_t want = (batch + kMlp2Rows - 1) / kMlp2Rows;
    if (want > (uint64_t)cu_count) want = (uint64_t)cu_count;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hps_dense_mlp128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds128);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(hps_dense_mlp128_kernel, dim3((uint32_t)want), dim3(kMlp2Threads), lds128, stream, d, d_x, batch,
                       reinterpret_cast<_Float16*>(d_out_f16));
    return hipGetLastError();
  }
  uint64_t want = (batch + kMlpRows - 1) / kMlpRows;
  const size_t lds_bytes = (size_t)kMlpRows * (d.buf_dim[0] + d.buf_dim[1] + 2 * kLdsPad) * sizeof(_Float16);
  uint64_t per_cu = (160u << 10) / lds_bytes;  // resident blocks per CU (160 KB of LDS)
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  const uint64_t cap = (uint64_t)cu_count * per_cu;
  if (want > cap) want = cap;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hps_dense_mlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(hps_dense_mlp_kernel, dim3((uint32_t)want), dim3(kMlpThreads), lds_bytes, stream, d, d_x, batch,
                     reinterpret_cast<_Float16*>(d_out_f16));
  return hipGetLastError();
}

hipError_t LaunchDenseInteract(const float* d_emb, const void* d_bottom_f16, uint64_t batch, uint32_t T, uint32_t D,
                               uint32_t out_stride, void* d_out_f16, int cu_count, hipStream_t stream) {
  if (batch == 0) return hipSuccess;
  const size_t lds_bytes = 4 * (32 * (size_t)(D + kZPad) + kTriElems) * sizeof(_Float16);
  uint64_t want = (batch + 3) / 4;
  uint64_t per_cu = (160u << 10) / lds_bytes;  // resident blocks per CU (160 KB of LDS)
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  const uint64_t cap = (uint64_t)cu_count * per_cu;
  if (want > cap) want = cap;
  const uint32_t per_lane = (T * (D / 4) + 63) / 64;
  auto go = [&](auto kernel) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3((uint32_t)want), dim3(256), lds_bytes, stream, d_emb, reinterpret_cast<const _Float16*>(d_bottom_f16),
                       batch, T, D, out_stride, reinterpret_cast<_Float16*>(d_out_f16));
    return hipGetLastError();
  };
  // (non-temporal row loads: 190 -> 178-186 us under rocprofv3, profiles/round4/ab_dense_interaction_nontemporal_loads.txt)
  if (per_lane <= 4) return go(hps_dense_interact_kernel<4, true>);
  if (per_lane <= 8) return go(hps_dense_interact_kernel<8, true>);
  if (per_lane <= 13) return go(hps_dense_interact_kernel<13, true>);   // T = 26, D = 128: 13 chunks exactly (fewer registers than <16>)
  if (per_lane <= 16) return go(hps_dense_interact_kernel<16, true>);
  return go(hps_dense_interact_kernel<0, true>);
}

}  // namespace hps
