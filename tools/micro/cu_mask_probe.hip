// Which XCDs does a CU-masked stream run on?  hipExtStreamCreateWithCUMask takes one bit per CU; this prints, for a few mask
// patterns, how many workgroups of a 4,096-block launch ran on each XCD (s_getreg HW_REG_XCC_ID) and how long the launch took.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

__global__ void where(unsigned* hist, unsigned* spin_out) {
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID, 4 bits
  if (threadIdx.x == 0) atomicAdd(&hist[xcc], 1u);
  unsigned v = threadIdx.x;
  for (int i = 0; i < 2000; ++i) v = v * 1664525u + 1013904223u;
  if (v == 1) *spin_out = v;
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
  unsigned *hist, *out;
  hipMalloc(&hist, 16 * sizeof(unsigned));
  hipMalloc(&out, sizeof(unsigned));
  hipMemsetAsync(hist, 0, 16 * sizeof(unsigned), s);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(where, dim3(4096), dim3(256), 0, s, hist, out);   // warm
  hipMemsetAsync(hist, 0, 16 * sizeof(unsigned), s);
  hipEventRecord(a, s);
  hipLaunchKernelGGL(where, dim3(4096), dim3(256), 0, s, hist, out);
  hipEventRecord(b, s);
  hipStreamSynchronize(s);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  unsigned h[16];
  hipMemcpy(h, hist, sizeof(h), hipMemcpyDeviceToHost);
  int bits = 0;
  for (uint32_t w : mask) bits += __builtin_popcount(w);
  printf("%-34s %3d bits: %.3f ms, blocks per XCD:", name, bits, ms);
  for (int i = 0; i < 8; ++i) printf(" %u", h[i]);
  printf("\n");
  hipFree(hist); hipFree(out); hipStreamDestroy(s);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("%s: %d CUs\n", p.name, cus);
  const int words = (cus + 31) / 32;
  std::vector<uint32_t> m(words, 0xFFFFFFFFu);
  run("all CUs", m);
  std::fill(m.begin(), m.end(), 0u);
  for (int k = 0; k < 32; ++k) m[k / 32] |= 1u << (k % 32);
  run("bits 0..31", m);
  std::fill(m.begin(), m.end(), 0u);
  for (int k = 0; k < cus; k += 8) m[k / 32] |= 1u << (k % 32);
  run("bits k % 8 == 0", m);
  std::fill(m.begin(), m.end(), 0u);
  for (int k = 7; k < cus; k += 8) m[k / 32] |= 1u << (k % 32);
  run("bits k % 8 == 7", m);
  std::fill(m.begin(), m.end(), 0u);
  for (int k = 0; k < cus; ++k) if ((k % 8) != 7) m[k / 32] |= 1u << (k % 32);
  run("bits k % 8 != 7", m);
  std::fill(m.begin(), m.end(), 0u);
  for (int k = 0; k < 16; ++k) m[k / 32] |= 1u << (k % 32);
  run("bits 0..15", m);
  std::fill(m.begin(), m.end(), 0u);
  for (int k = 0; k < cus; ++k) if (k >= 16) m[k / 32] |= 1u << (k % 32);
  run("bits 16..", m);
  return 0;
}
