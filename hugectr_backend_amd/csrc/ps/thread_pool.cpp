#include "thread_pool.h"

#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace hps {

// CPUs this process may actually use: hardware threads, clipped by the scheduler affinity mask and by the
// cgroup CPU bandwidth quota (cpu.max "quota period", cgroup v2; cfs_quota_us/cfs_period_us, v1).  A container
// with 256 visible hardware threads and a 16-CPU quota gets 16: running 255 workers there only burns the quota
// and gets the whole process throttled for the rest of the 100 ms period.
static size_t EffectiveCpus() {
  size_t n = std::thread::hardware_concurrency();
  if (n == 0) n = 4;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) {
    const int c = CPU_COUNT(&set);
    if (c > 0 && (size_t)c < n) n = (size_t)c;
  }
  auto clip = [&](double cpus) {
    if (cpus > 0) {
      const size_t q = (size_t)std::ceil(cpus);
      if (q >= 1 && q < n) n = q;
    }
  };
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64] = {0};
    long long period = 0;
    if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
      clip((double)strtoll(q, nullptr, 10) / (double)period);
    fclose(f);
  } else {
    long long quota = -1, period = 0;
    if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
    if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
    if (quota > 0 && period > 0) clip((double)quota / (double)period);
  }
  return n;
}

size_t ThreadPool::DefaultConcurrency() {
  // thread_pool.cpp:25-41 of the reference: env override, else hardware_concurrency (here: the CPUs the
  // process is really allowed to use).
  if (const char* e = std::getenv("HCTR_DEFAULT_CONCURRENCY")) {
    const long v = std::strtol(e, nullptr, 10);
    if (v > 0) return (size_t)v;
  }
  static const size_t n = EffectiveCpus();
  return n;
}

ThreadPool& ThreadPool::Global() {
  static ThreadPool pool(DefaultConcurrency() > 1 ? DefaultConcurrency() - 1 : 1, 0);
  return pool;
}

ThreadPool& ThreadPool::Serving() {
  // A 512-B row is 8 cache lines and a core keeps only ~a dozen line fills in flight, so the gather is bound
  // by line-fill buffers per core, not by DRAM: it scales with the number of physical cores working on it.
  // HPS_SERVING_THREADS overrides the default: the CPUs this process may really use (affinity mask clipped by the
  // cgroup quota) minus a few for the callers, at most 64.
  static ThreadPool pool([] {
    if (const char* e = std::getenv("HPS_SERVING_THREADS")) {
      const long v = std::strtol(e, nullptr, 10);
      if (v > 0) return (size_t)v;
    }
    // leave two CPUs of the budget to the callers (session threads, HIP runtime threads)
    const size_t c = DefaultConcurrency();
    return std::max<size_t>(1, std::min<size_t>(64, c > 4 ? c - 3 : c / 2));
  }(), 100);
  return pool;
}

ThreadPool::ThreadPool(size_t num_workers, unsigned spin_us) : spin_us_(spin_us) {
  workers_.reserve(num_workers);
  for (size_t i = 0; i < num_workers; ++i) workers_.emplace_back([this] { WorkerMain(); });
}

ThreadPool::~ThreadPool() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = true;
    seq_.fetch_add(1, std::memory_order_release);
  }
  cv_.notify_all();
  for (auto& t : workers_) t.join();
}

void ThreadPool::RunLoop(Loop* l) {
  size_t mine = 0;
  for (;;) {
    const size_t i = l->next.fetch_add(1, std::memory_order_relaxed);
    if (i >= l->n) break;
    (*l->fn)(i);
    ++mine;
  }
  if (mine) {
    const size_t d = l->done.fetch_add(mine, std::memory_order_acq_rel) + mine;
    if (d == l->n) {
      std::lock_guard<std::mutex> lk(l->mu);
      l->cv.notify_all();
    }
  }
}

static inline void CpuRelax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}

void ThreadPool::WorkerMain() {
  uint64_t seen = 0;
  for (;;) {
    std::shared_ptr<Loop> loop;
    std::function<void()> task;
    {
      std::unique_lock<std::mutex> lk(mu_);
      seen = seq_.load(std::memory_order_acquire);
      // drop loops whose tasks are all claimed
      while (!loops_.empty() && loops_.front()->next.load(std::memory_order_relaxed) >= loops_.front()->n)
        loops_.pop_front();
      for (auto& l : loops_) {
        if (l->next.load(std::memory_order_relaxed) < l->n &&
            l->helpers.load(std::memory_order_relaxed) < l->max_helpers) {
          l->helpers.fetch_add(1, std::memory_order_relaxed);
          loop = l;
          break;
        }
      }
      if (!loop && !tasks_.empty()) { task = std::move(tasks_.front()); tasks_.pop_front(); }
      if (!loop && !task && stop_) return;
    }
    if (loop) { RunLoop(loop.get()); continue; }
    if (task) { task(); continue; }
    // idle: spin on the enqueue counter for a while (serving pool), then sleep
    if (spin_us_) {
      const auto t0 = std::chrono::steady_clock::now();
      bool woke = false;
      for (unsigned it = 0;; ++it) {
        if (seq_.load(std::memory_order_acquire) != seen) { woke = true; break; }
        CpuRelax();
        if ((it & 63) == 63 &&
            std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us_)) break;
      }
      if (woke) continue;
    }
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return stop_ || seq_.load(std::memory_order_acquire) != seen; });
  }
}

void ThreadPool::ParallelFor(size_t num_tasks, const std::function<void(size_t)>& fn, size_t max_parallel) {
  if (num_tasks == 0) return;
  if (num_tasks == 1 || workers_.empty() || max_parallel == 1) {
    for (size_t i = 0; i < num_tasks; ++i) fn(i);
    return;
  }
  auto loop = std::make_shared<Loop>();
  loop->fn = &fn;
  loop->n = num_tasks;
  size_t helpers = workers_.size();
  if (max_parallel && max_parallel - 1 < helpers) helpers = max_parallel - 1;
  if (num_tasks - 1 < helpers) helpers = num_tasks - 1;
  loop->max_helpers = helpers;
  {
    std::lock_guard<std::mutex> lk(mu_);
    loops_.push_back(loop);
    seq_.fetch_add(1, std::memory_order_release);
  }
  if (helpers >= workers_.size() / 2) cv_.notify_all();
  else for (size_t i = 0; i < helpers; ++i) cv_.notify_one();
  RunLoop(loop.get());
  // the stragglers usually finish within microseconds: poll before paying for a futex sleep + wake
  for (int it = 0; it < 20000; ++it) {
    if (loop->done.load(std::memory_order_acquire) == loop->n) return;
    CpuRelax();
  }
  std::unique_lock<std::mutex> lk(loop->mu);
  loop->cv.wait(lk, [&] { return loop->done.load(std::memory_order_acquire) == loop->n; });
}

void ThreadPool::Submit(std::function<void()> fn) {
  {
    std::lock_guard<std::mutex> lk(mu_);
    tasks_.push_back(std::move(fn));
    seq_.fetch_add(1, std::memory_order_release);
  }
  cv_.notify_one();
}

}  // namespace hps
