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

  // NUMA placement of the two pools' workers (and, through first touch, of the host tables they load).  On a two-socket box the
  // scheduler spreads the workers over both sockets and the tables end up wherever their loaders happened to run: round 5's
  // headline was 1.59-2.01 G lookups/s from run to run on one box (key staging 0.19-0.50 ms, host gather 0.43-0.85 ms per
  // call) and 2.03-2.06 G in every run confined to ONE node — either one (profiles/round5/numa_one_node.txt).
  // BindToNumaNode is called once, before the pools start (HierParameterServer: the node of the deployed GPUs when they share
  // one, none for a CPU-only deployment or when the GPUs span nodes; bound pools are sized for that node's CPUs); HPS_NUMA_NODE=<n> names the node,
  // HPS_NUMA_NODE=off leaves the workers where the scheduler puts them.  The callers' threads are never touched.
  static void BindToNumaNode(int node);   // -1: none.  Ignored once a pool exists or a node has been chosen
  static int NumaNode();                  // the node the workers are bound to, -1 when they are not
  // The calling thread joins the workers' node — for the threads that drive lookups (the plugin does it for every Triton
  // instance thread at its first request: through the plugin boundary 1.57-2.04 G lookups/s from process to process with the
  // instance threads left alone, 1.99-2.04 G with them bound, profiles/round5/triton_abi_numa_bound_vs_free.txt).  A thread
  // whose affinity mask already lies inside ONE node (numactl, Triton's host policy, taskset) is left as it is.
  // false: nothing changed (workers not bound, or the thread already placed).
  static bool BindCallingThread();

  size_t size() const { return workers_.size(); }
  // fork-joins (lock-free path, all pools of the process) in which a task ran more than once: 0 unless the slot-reuse race of
  // round 5 is back (tools/micro/forkjoin_stress.cpp checks it; hps_pool_fast_overruns in the C ABI)
  static uint64_t FastOverruns();

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

  // Fast path of ParallelFor in a spinning (serving) pool: a few loop slots that live as long as the pool, found by
  // the workers with plain loads (no lock) and claimed in grains with ONE compare-and-swap on a word that carries the
  // slot's generation — a worker that read the parameters of an older generation fails its CAS and runs nothing.
  // Measured on the MI355X box (14 threads spread over two sockets): 14 empty tasks through the locked queue 13.7 us,
  // the hardware floor of a flag + a counter 2.9 us (tools/micro/forkjoin_min.bin).
  struct alignas(64) FastLoop {
    std::atomic<uint64_t> next{0};                                     // (generation << 32) | next task index
    alignas(64) std::atomic<uint32_t> done{0};                         // tasks finished
    alignas(64) std::atomic<uint32_t> state{0};                        // 0 free, 1 published, 2 being set up / torn down
    std::atomic<const std::function<void(size_t)>*> fn{nullptr};
    std::atomic<uint32_t> n{0}, grain{1};
    uint32_t gen = 0;                                                  // owner only
  };
  static constexpr int kFastSlots = 16;   // two sessions per GPU on an 8-GPU node fork-join side by side
  FastLoop fast_[kFastSlots];
  // claims and runs grains of slot `L` while its generation is `gen`; returns the number of tasks this thread ran
  static uint32_t RunFast(FastLoop& L, uint32_t gen);
  bool HelpFastLoops();   // worker side: one pass over the published slots

  // The queues are guarded by a spin lock, not by mu_: when a loop is published, every spinning worker comes for it at
  // once, and a dozen threads handing a std::mutex to each other through the futex cost ~20 us per request (the whole
  // gather of a 4,096-key request is 7 us on 14 threads).  mu_/cv_ only park and wake idle workers.
  struct SpinLock {
    std::atomic_flag f = ATOMIC_FLAG_INIT;
    void lock() {
      while (f.test_and_set(std::memory_order_acquire)) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
      }
    }
    void unlock() { f.clear(std::memory_order_release); }
  };
  std::vector<std::thread> workers_;
  SpinLock qlock_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::shared_ptr<Loop>> loops_;
  std::deque<std::function<void()>> tasks_;
  std::atomic<uint64_t> seq_{0};  // bumped on every enqueue; spinning workers poll it
  std::atomic<int> sleepers_{0};   // workers parked on cv_ (or about to be)
  unsigned spin_us_ = 0;
  std::atomic<bool> stop_{false};
};

}  // namespace hps
