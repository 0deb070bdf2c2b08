// Where do the config-5 interaction kernel's 180-190 us go?  Its read pattern alone streams at 6.0-6.75 TB/s (read_stream.hip:
// 129 us for the 872 MB).  This adds the kernel's stages one at a time to that read loop (one wave per sample, T = 26 rows of
// D = 128, two register sets, loads of sample i+1 issued before sample i is touched):
//   stage 0  loads only (sum into a register)
//   stage 1  + fp32 -> f16 conversion and the LDS tile
//   stage 2  + G = Z Z^T on the matrix cores (8 MFMA 32x32x16 f16)
//   stage 3  + strict lower triangle through LDS
//   stage 4  + the output row (960 B per sample) — the whole kernel without the bottom-MLP row
//   stage 5  the same, the output row composed contiguously in LDS and written by ONE 16-B-per-lane store (60 lanes)
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/interact_stages tools/micro/interact_stages.hip && /tmp/interact_stages
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16x __attribute__((ext_vector_type(16)));
constexpr int NR = 13, kZPad = 8;
constexpr unsigned kTri = 32 * 31 / 2 + 16 + 512;   // triangle scratch + (stage 5) a contiguous 480-f16 output row + spare
__device__ __forceinline__ int acc_row(int lane, int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
template <int STAGE, bool NTS>
__global__ __launch_bounds__(256) void k(const float* __restrict__ emb, size_t batch, unsigned T, unsigned D, unsigned out_stride,
                                         _Float16* __restrict__ out, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) _Float16 zlds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const unsigned V = T + 1, npairs = V * (V - 1) / 2, zstride = D + kZPad, d4 = D >> 2, nchunks = T * d4;
  _Float16* z = zlds + (size_t)wave * (32 * zstride + kTri);
  _Float16* tri = z + 32 * zstride;
  size_t goff[NR]; unsigned zoff[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const unsigned c = u * 64 + lane, ce = c < nchunks ? c : nchunks - 1, t = ce / d4, q = ce - t * d4;
    goff[u] = (size_t)t * batch * D + q * 4;
    zoff[u] = c < nchunks ? (1 + t) * zstride + q * 4 : 0xFFFFFFFFu;
  }
  const size_t W = (size_t)gridDim.x * 4;
  f4 pa[NR], pb[NR];
  f4 s0 = {0, 0, 0, 0};
  auto issue = [&](f4 (&d)[NR], size_t s) {
#pragma unroll
    for (int u = 0; u < NR; ++u) d[u] = __builtin_nontemporal_load((const f4*)(emb + goff[u] + s * D));
  };
  auto work = [&](const f4 (&src)[NR], size_t i) {
    if (STAGE == 0) {
#pragma unroll
      for (int u = 0; u < NR; ++u) s0 += src[u];
      return;
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      h4 w; w[0] = (_Float16)src[u][0]; w[1] = (_Float16)src[u][1]; w[2] = (_Float16)src[u][2]; w[3] = (_Float16)src[u][3];
      if (zoff[u] != 0xFFFFFFFFu) *(h4*)(z + zoff[u]) = w;
    }
    __builtin_amdgcn_wave_barrier();
    if (STAGE == 1) { s0[0] += (float)z[zstride + lane]; return; }
    f16x acc = {0};
    const bool live = (unsigned)r < V;
    const _Float16* zr = z + (size_t)(live ? r : 0) * zstride + 8 * h;
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
    for (unsigned k0 = 0; k0 < D; k0 += 16) {
      const h8 ld = *(const h8*)(zr + k0);
      const h8 f = live ? ld : zero8;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, f, acc, 0, 0, 0);
    }
    if (STAGE == 2) { s0[0] += acc[0] + acc[7] + acc[15]; __builtin_amdgcn_wave_barrier(); return; }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const unsigned a = (unsigned)acc_row(lane, reg), b = (unsigned)r;
      const unsigned idx = (a < V && b < a) ? a * (a - 1) / 2 + b : 496u + (lane & 15);
      tri[idx] = (_Float16)acc[reg];
    }
    __builtin_amdgcn_wave_barrier();
    if (STAGE == 3) { s0[0] += (float)tri[lane]; __builtin_amdgcn_wave_barrier(); return; }
    if (STAGE == 5) {
      _Float16* orow = tri + 512;
      // (the real kernel writes the triangle straight into orow + D; here it is copied: a few LDS ops more than needed)
      for (unsigned c = lane; c < 352; c += 64) orow[128 + c] = c < npairs ? tri[c] : (_Float16)0.f;
      for (unsigned c = lane * 8; c < D; c += 512) *(h8*)(orow + c) = *(const h8*)(z + zstride + c);
      __builtin_amdgcn_wave_barrier();
      _Float16* o = out + i * out_stride;
      if (lane * 8 < out_stride) {   // 480: 60 lanes, rows 960 B apart (not a multiple of the 128-B line); 512: all 64 lanes, whole lines
        if (NTS) __builtin_nontemporal_store(*(const h8*)(orow + lane * 8), (h8*)(o + lane * 8)); else *(h8*)(o + lane * 8) = *(const h8*)(orow + lane * 8);
      }
      __builtin_amdgcn_wave_barrier();
      return;
    }
    _Float16* o = out + i * out_stride;
    for (unsigned c = lane * 8; c < D; c += 512) {
      if (NTS) __builtin_nontemporal_store(*(const h8*)(z + zstride + c), (h8*)(o + c)); else *(h8*)(o + c) = *(const h8*)(z + zstride + c);
    }
    const unsigned tail = out_stride - D;
    for (unsigned c = lane * 2; c < tail; c += 128) {
      h2 w; w[0] = c < npairs ? tri[c] : (_Float16)0.f; w[1] = c + 1 < npairs ? tri[c + 1] : (_Float16)0.f;
      if (NTS) __builtin_nontemporal_store(w, (h2*)(o + D + c)); else *(h2*)(o + D + c) = w;
    }
    __builtin_amdgcn_wave_barrier();
  };
  size_t i = (size_t)blockIdx.x * 4 + wave;
  if (i < batch) issue(pa, i);
  while (i < batch) {
    size_t nxt = i + W;
    issue(pb, nxt < batch ? nxt : i);
    work(pa, i);
    i = nxt;
    if (i >= batch) break;
    nxt = i + W;
    issue(pa, nxt < batch ? nxt : i);
    work(pb, i);
    i = nxt;
  }
  if (lane == 0) sink[blockIdx.x * 4 + wave] = s0[0] + s0[1] + s0[2] + s0[3];
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a, 0);
  for (int i = 0; i < 20; ++i) f();
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / 20;
}
template <int STAGE, bool NTS> float run(const float* emb, _Float16* out, float* sink, int grid, unsigned out_stride = 480u) {
  const size_t lds = 4 * (32 * (size_t)(128 + kZPad) + kTri) * sizeof(_Float16);
  hipFuncSetAttribute((const void*)k<STAGE, NTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  return timeit([&] { hipLaunchKernelGGL((k<STAGE, NTS>), dim3(grid), dim3(256), lds, 0, emb, (size_t)65536, 26u, 128u, out_stride, out, sink); });
}
int main() {
  const size_t batch = 65536, bytes = batch * 26 * 128 * 4;
  float *emb, *sink; _Float16* out;
  hipMalloc(&emb, bytes); hipMalloc(&out, batch * 512 * 2); hipMalloc(&sink, 1 << 20);
  hipMemset(emb, 0, bytes);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  for (int per : {2, 3}) {
    const int g = pr.multiProcessorCount * per;
    printf("%d blocks/CU: one 16-B store per lane for the whole output row: %.1f us (nt %.1f)\n", per, run<5, false>(emb, out, sink, g) * 1e3, run<5, true>(emb, out, sink, g) * 1e3);
    printf("%d blocks/CU: output rows padded to 1,024 B (whole 128-B lines, one store): %.1f us (nt %.1f)\n", per, run<5, false>(emb, out, sink, g, 512u) * 1e3, run<5, true>(emb, out, sink, g, 512u) * 1e3);
    printf("%d blocks/CU: loads %.1f us | + LDS tile %.1f | + MFMA %.1f | + triangle %.1f | + output rows %.1f (nt stores %.1f) us\n", per,
           run<0, false>(emb, out, sink, g) * 1e3, run<1, false>(emb, out, sink, g) * 1e3, run<2, false>(emb, out, sink, g) * 1e3,
           run<3, false>(emb, out, sink, g) * 1e3, run<4, false>(emb, out, sink, g) * 1e3, run<4, true>(emb, out, sink, g) * 1e3);
  }
  return 0;
}
