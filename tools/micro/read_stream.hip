// What does a READ-dominated streaming kernel reach on this box?  (The config-5 interaction kernel reads 872 MB and writes 63 MB
// per 64 K batch; its 5.1 TB/s equal the box's device-to-device copy rate — is a copy the right ceiling for it?)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/read_stream tools/micro/read_stream.hip && /tmp/read_stream
// Prints GB/s of a float4 read stream (sum into a register, one 4-B store per wave) for plain and non-temporal loads, several
// loads in flight per lane, persistent grids of 2 / 4 / 8 blocks per CU; and of a float4 copy for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void rd(const f4* __restrict__ p, size_t n, float* __restrict__ out) {
  f4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < n; i += stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <bool NT>
__global__ __launch_bounds__(256) void cp(const f4* __restrict__ p, f4* __restrict__ q, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const f4 v = NT ? __builtin_nontemporal_load(p + i) : p[i];
    if (NT) __builtin_nontemporal_store(v, q + i); else q[i] = v;
  }
}
// the interaction kernel's read pattern without its compute: one wave per sample, T rows of D floats at a table stride of
// batch x D (table-major OUTPUT0), 13 float4 chunks per lane (T = 26, D = 128), the next sample's loads issued before the
// current sample's values are consumed; NW waves per block
template <bool NT, int NCH>
__global__ __launch_bounds__(256) void rd_rows(const float* __restrict__ emb, size_t batch, unsigned T, unsigned D, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned d4 = D >> 2, nchunks = T * d4;
  size_t goff[NCH];
#pragma unroll
  for (int u = 0; u < NCH; ++u) {
    const unsigned c = u * 64 + lane, ce = c < nchunks ? c : nchunks - 1;
    const unsigned t = ce / d4, q = ce - t * d4;
    goff[u] = (size_t)t * batch * D + q * 4;
  }
  const size_t waves_total = (size_t)gridDim.x * 4;
  f4 acc = {0, 0, 0, 0};
  f4 pre[NCH];
  size_t i = (size_t)blockIdx.x * 4 + wave;
  if (i < batch) {
#pragma unroll
    for (int u = 0; u < NCH; ++u) pre[u] = NT ? __builtin_nontemporal_load((const f4*)(emb + goff[u] + i * D)) : *(const f4*)(emb + goff[u] + i * D);
  }
  for (; i < batch; i += waves_total) {
#pragma unroll
    for (int u = 0; u < NCH; ++u) acc += pre[u];
    const size_t nxt = i + waves_total < batch ? i + waves_total : i;
#pragma unroll
    for (int u = 0; u < NCH; ++u) pre[u] = NT ? __builtin_nontemporal_load((const f4*)(emb + goff[u] + nxt * D)) : *(const f4*)(emb + goff[u] + nxt * D);
  }
  if (lane == 0) out[blockIdx.x * 4 + wave] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a, 0);
  for (int i = 0; i < 20; ++i) f();
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / 20;
}
int main() {
  const size_t bytes = 872ull << 20, n = bytes / 16;
  f4 *p, *q; float* out;
  hipMalloc(&p, bytes); hipMalloc(&q, bytes); hipMalloc(&out, 1 << 20);
  hipMemset(p, 1, bytes); hipMemset(q, 0, bytes);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cu = pr.multiProcessorCount;
  for (int per : {2, 4, 8}) {
    const int g = cu * per;
    printf("blocks/CU %d: read U=4 plain %.0f  nt %.0f | U=8 plain %.0f  nt %.0f | U=13 nt %.0f GB/s\n", per,
           bytes / timeit([&] { hipLaunchKernelGGL((rd<4, false>), dim3(g), dim3(256), 0, 0, p, n, out); }) / 1e6,
           bytes / timeit([&] { hipLaunchKernelGGL((rd<4, true>), dim3(g), dim3(256), 0, 0, p, n, out); }) / 1e6,
           bytes / timeit([&] { hipLaunchKernelGGL((rd<8, false>), dim3(g), dim3(256), 0, 0, p, n, out); }) / 1e6,
           bytes / timeit([&] { hipLaunchKernelGGL((rd<8, true>), dim3(g), dim3(256), 0, 0, p, n, out); }) / 1e6,
           bytes / timeit([&] { hipLaunchKernelGGL((rd<13, true>), dim3(g), dim3(256), 0, 0, p, n, out); }) / 1e6);
  }
  {
    const size_t batch = 65536; const unsigned T = 26, D = 128;
    const double rb = (double)batch * T * D * 4;
    for (int per : {2, 3, 4, 8}) {
      const int g = cu * per;
      printf("interaction pattern (wave per sample, 26 rows of 512 B at table stride), %d blocks/CU: plain %.0f  nt %.0f GB/s\n", per,
             rb / timeit([&] { hipLaunchKernelGGL((rd_rows<false, 13>), dim3(g), dim3(256), 0, 0, (const float*)p, batch, T, D, out); }) / 1e6,
             rb / timeit([&] { hipLaunchKernelGGL((rd_rows<true, 13>), dim3(g), dim3(256), 0, 0, (const float*)p, batch, T, D, out); }) / 1e6);
    }
  }
  printf("copy (read + write bytes): plain %.0f  nt %.0f GB/s\n",
         2.0 * bytes / timeit([&] { hipLaunchKernelGGL((cp<false>), dim3(cu * 8), dim3(256), 0, 0, p, q, n); }) / 1e6,
         2.0 * bytes / timeit([&] { hipLaunchKernelGGL((cp<true>), dim3(cu * 8), dim3(256), 0, 0, p, q, n); }) / 1e6);
  return 0;
}
