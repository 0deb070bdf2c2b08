// Does the stopEvent of hipExtLaunchKernelGGL order OTHER streams?  (The engine records an event behind its probe and gather
// kernels with hipEventRecord — one marker packet each, 5-9 us on the queue.  If hipStreamWaitEvent honours the event a kernel
// launch was given as its stop event, those markers can go.)
//   hipcc --offload-arch=gfx950 -O2 -w -o /tmp/ext_stop_event tools/micro/ext_stop_event.hip && /tmp/ext_stop_event
// Stream 1: a kernel that spins ~1.5 ms, then writes `iter` — launched with stop event E.  Stream 2: waits for E, then copies
// the word.  Stale copies = the wait did not wait.  Also timed: gap between two dependent kernels of one stream with a
// hipEventRecord between them / with the first kernel's stop event instead / with nothing.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void slow_write(volatile unsigned* p, unsigned v, long long cycles) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  *p = v;
}
__global__ void copy_word(const volatile unsigned* p, unsigned* q) { *q = *p; }
__global__ void tiny(unsigned* p) { if (threadIdx.x == 0) p[0] += 1; }
int main() {
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  unsigned *a, *b, *h;
  hipMalloc(&a, 4); hipMalloc(&b, 4); hipHostMalloc(&h, 4);
  hipMemset(a, 0, 4);
  hipEvent_t e, t0, t1;
  hipEventCreate(&e); hipEventCreate(&t0); hipEventCreate(&t1);
  int stale = 0;
  for (unsigned it = 1; it <= 200; ++it) {
    hipExtLaunchKernelGGL(slow_write, dim3(1), dim3(1), 0, s1, nullptr, e, 0, a, it, 150000LL);   // 100 MHz wall clock: 1.5 ms
    hipStreamWaitEvent(s2, e, 0);
    hipLaunchKernelGGL(copy_word, dim3(1), dim3(1), 0, s2, a, b);
    hipMemcpyAsync(h, b, 4, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s2);
    if (*h != it) ++stale;
    hipStreamSynchronize(s1);
  }
  printf("cross-stream wait on a launch's stop event: %d stale reads of 200 (0 = the wait waits)\n", stale);
  // the same with a recorded event, as a control
  stale = 0;
  for (unsigned it = 1001; it <= 1100; ++it) {
    hipLaunchKernelGGL(slow_write, dim3(1), dim3(1), 0, s1, a, it, 150000LL);
    hipEventRecord(e, s1);
    hipStreamWaitEvent(s2, e, 0);
    hipLaunchKernelGGL(copy_word, dim3(1), dim3(1), 0, s2, a, b);
    hipMemcpyAsync(h, b, 4, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s2);
    if (*h != it) ++stale;
    hipStreamSynchronize(s1);
  }
  printf("control (hipEventRecord): %d stale of 100\n", stale);
  // chain of 200 dependent tiny kernels on one stream: plain / an event record after each / each launched with a stop event
  auto chain = [&](int mode) {
    hipStreamSynchronize(s1);
    hipEventRecord(t0, s1);
    for (int i = 0; i < 200; ++i) {
      if (mode == 2) hipExtLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, nullptr, e, 0, a);
      else hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, a);
      if (mode == 1) hipEventRecord(e, s1);
    }
    hipEventRecord(t1, s1);
    hipStreamSynchronize(s1);
    float ms = 0; hipEventElapsedTime(&ms, t0, t1);
    return ms * 1000.f / 200.f;
  };
  chain(0);
  printf("per dependent tiny kernel: plain %.2f us | + hipEventRecord after each %.2f us | launched with a stop event %.2f us\n", chain(0), chain(1), chain(2));
  return 0;
}
