#include <atomic>
#include <unistd.h>
#include <cstdio>
#include <thread>
#include <vector>
#include <chrono>
#include "ps/thread_pool.h"
using namespace hps;
int main(int argc, char** argv) {
  const int callers = argc > 1 ? atoi(argv[1]) : 12;
  const int secs = argc > 2 ? atoi(argv[2]) : 20;
  ThreadPool pool(13, 100);
  std::atomic<bool> stop{false};
  std::vector<std::atomic<long>> prog(callers);
  for (auto& p : prog) p = 0;
  std::vector<std::thread> th;
  for (int c = 0; c < callers; ++c)
    th.emplace_back([&, c] {
      unsigned x = 1234567u * (c + 1);
      while (!stop.load()) {
        x = x * 1664525u + 1013904223u;
        const size_t n = 2 + (x >> 8) % 60;
        std::atomic<size_t> sum{0};
        pool.ParallelFor(n, [&](size_t i) { volatile int s = 0; for (int k = 0; k < 200 + (int)(i & 63) * 20; ++k) s += k; sum.fetch_add(i + 1); });
        if (sum.load() != n * (n + 1) / 2) { fprintf(stderr, "WRONG SUM\n"); _exit(3); }
        prog[c].fetch_add(1);
        if ((x >> 28) == 0) std::this_thread::sleep_for(std::chrono::microseconds(300 + (x >> 20) % 2000));   // let the workers park
      }
    });
  long last = 0;
  for (int s = 0; s < secs; ++s) {
    std::this_thread::sleep_for(std::chrono::seconds(1));
    std::vector<long> now(callers);
    long tot = 0;
    for (int c = 0; c < callers; ++c) { now[c] = prog[c].load(); tot += now[c]; }
    static std::vector<long> prev(callers, -1);
    int stuck = 0;
    for (int c = 0; c < callers; ++c) { if (now[c] == prev[c]) ++stuck; prev[c] = now[c]; }
    printf("t=%d loops=%ld (+%ld) stuck callers=%d\n", s + 1, tot, tot - last, stuck);
    fflush(stdout);
    if (stuck) { printf("HANG DETECTED\n"); _exit(2); }
    last = tot;
  }
  stop.store(true);
  for (auto& t : th) t.join();
  if (hps::ThreadPool::FastOverruns()) { printf("OVERRUN %llu\n", (unsigned long long)hps::ThreadPool::FastOverruns()); _exit(4); }
  printf("ok\n");
  return 0;
}
