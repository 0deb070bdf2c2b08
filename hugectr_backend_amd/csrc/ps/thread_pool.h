// Fork-join worker pool for the host parameter-server tier.
//
// Plays the role of the reference's ThreadPool (/root/reference/hps_backend/include/thread_pool.hpp,
// src/thread_pool.cpp:25-41: sized by HCTR_DEFAULT_CONCURRENCY or hardware_concurrency) but is built for
// data-parallel loops: ParallelFor splits [0,n) into tasks claimed with one atomic each, the calling
// thread works too, and several callers (lookup sessions) can have loops in flight at once.
// Submit() runs a detached task (async cache insertion / refresh).
//
// Two instances exist: Global() (all cores, sleeps when idle: table loading, index builds) and
// Serving() (a few dozen workers that spin briefly after each job, so that the per-request
// parameter-server gather does not pay a futex wake-up per worker per request).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace hps {

class ThreadPool {
 public:
  explicit ThreadPool(size_t num_workers, unsigned spin_us = 0);
  ~ThreadPool();
  ThreadPool(const ThreadPool&) = delete;
  ThreadPool& operator=(const ThreadPool&) = delete;

  static ThreadPool& Global();          // lazily built, DefaultConcurrency()-1 workers, no spinning
  static ThreadPool& Serving();         // lazily built, min(32, cores/2) workers, spin 200 us
  static size_t DefaultConcurrency();   // HCTR_DEFAULT_CONCURRENCY env, else hardware_concurrency

  size_t size() const { return workers_.size(); }

  // fn(task_index) for task_index in [0, num_tasks); returns when all are done.  max_parallel caps the
  // number of threads (including the caller) that work on this loop; 0 = no cap.
  void ParallelFor(size_t num_tasks, const std::function<void(size_t)>& fn, size_t max_parallel = 0);

  // fire-and-forget
  void Submit(std::function<void()> fn);

 private:
  struct Loop {
    const std::function<void(size_t)>* fn;
    size_t n;
    size_t max_helpers;
    std::atomic<size_t> next{0};
    std::atomic<size_t> done{0};
    std::atomic<size_t> helpers{0};
    std::mutex mu;
    std::condition_variable cv;
  };
  void WorkerMain();
  static void RunLoop(Loop* l);

  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::shared_ptr<Loop>> loops_;
  std::deque<std::function<void()>> tasks_;
  std::atomic<uint64_t> seq_{0};  // bumped on every enqueue; spinning workers poll it
  unsigned spin_us_ = 0;
  bool stop_ = false;
};

}  // namespace hps
