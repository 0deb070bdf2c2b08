#include "thread_pool.h"

#include <pthread.h>
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

// ---- NUMA node of the pools' workers ----
namespace {
std::atomic<int> g_numa_node{-2};        // -2: not decided yet, -1: none
std::atomic<bool> g_pool_started{false};
cpu_set_t g_numa_cpus;                   // valid when g_numa_node >= 0
std::atomic<uint64_t> g_fast_overruns{0};   // fork-joins of the lock-free path whose tasks ran more often than once each

// the CPUs of `node` this process may run on (sysfs cpulist: "0-63,128-191"); false when there are fewer than two
bool NodeCpus(int node, cpu_set_t* out) {
  char path[96];
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = fopen(path, "r");
  if (!f) return false;
  char buf[4096] = {0};
  const bool ok = fgets(buf, sizeof buf, f) != nullptr;
  fclose(f);
  if (!ok) return false;
  cpu_set_t allowed;
  if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
  CPU_ZERO(out);
  for (char* p = buf; *p;) {
    char* e = nullptr;
    const long a = strtol(p, &e, 10);
    if (e == p) break;
    long b = a;
    if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
      if (c >= 0 && CPU_ISSET((int)c, &allowed)) CPU_SET((int)c, out);
    p = (*e == ',') ? e + 1 : e;
    if (*e != ',' ) break;
  }
  return CPU_COUNT(out) >= 2;
}
}  // namespace

void ThreadPool::BindToNumaNode(int node) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (g_pool_started.load(std::memory_order_acquire) || g_numa_node.load(std::memory_order_acquire) != -2) return;
  int expect = -2;
  if (const char* e = std::getenv("HPS_NUMA_NODE")) {
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    node = (end != e && v >= 0) ? (int)v : -1;      // "off", "-1", anything that is not a node number: no binding
  }
  cpu_set_t set;
  if (node >= 0 && !NodeCpus(node, &set)) node = -1;
  if (node >= 0) g_numa_cpus = set;                 // (published by the store below; read only after g_numa_node >= 0 is seen)
  g_numa_node.compare_exchange_strong(expect, node, std::memory_order_acq_rel);
}

bool ThreadPool::BindCallingThread() {
  if (g_numa_node.load(std::memory_order_acquire) < 0) return false;
  cpu_set_t cur;
  if (pthread_getaffinity_np(pthread_self(), sizeof cur, &cur) != 0) return false;
  // already inside one node (somebody placed this thread on purpose)?
  for (int n = 0; n < 64; ++n) {
    cpu_set_t node;
    if (!NodeCpus(n, &node)) { if (n > 0) break; else continue; }
    cpu_set_t outside;
    CPU_XOR(&outside, &cur, &node);          // bits in exactly one of the two
    CPU_AND(&outside, &outside, &cur);       // ... that are in cur: cur \ node
    if (CPU_COUNT(&outside) == 0) return false;
  }
  cpu_set_t want;
  CPU_AND(&want, &cur, &g_numa_cpus);
  if (CPU_COUNT(&want) == 0) return false;
  return pthread_setaffinity_np(pthread_self(), sizeof want, &want) == 0;
}

uint64_t ThreadPool::FastOverruns() { return g_fast_overruns.load(std::memory_order_relaxed); }

int ThreadPool::NumaNode() {
  const int n = g_numa_node.load(std::memory_order_acquire);
  return n < 0 ? -1 : n;
}

size_t ThreadPool::DefaultConcurrency() {
  // thread_pool.cpp:25-41 of the reference: env override, else hardware_concurrency (here: the CPUs the
  // process is really allowed to use).
  if (const char* e = std::getenv("HCTR_DEFAULT_CONCURRENCY")) {
    const long v = std::strtol(e, nullptr, 10);
    if (v > 0) return (size_t)v;
  }
  static const size_t n = EffectiveCpus();
  // workers bound to one NUMA node (BindToNumaNode) share that node's CPUs: on a two-socket box without a tight quota the
  // machine-wide count would put twice as many spinning workers on them as there are CPUs
  if (g_numa_node.load(std::memory_order_acquire) >= 0) {
    const int c = CPU_COUNT(&g_numa_cpus);
    if (c > 0 && (size_t)c < n) return (size_t)c;
  }
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
  }(), /*idle spin of the workers after a job, microseconds*/ 100u);
  return pool;
}

ThreadPool::ThreadPool(size_t num_workers, unsigned spin_us) : spin_us_(spin_us) {
  g_pool_started.store(true, std::memory_order_release);
  workers_.reserve(num_workers);
  for (size_t i = 0; i < num_workers; ++i) workers_.emplace_back([this] { WorkerMain(); });
}

ThreadPool::~ThreadPool() {
  stop_.store(true, std::memory_order_release);
  seq_.fetch_add(1, std::memory_order_release);
  { std::lock_guard<std::mutex> lk(mu_); }   // a worker between its predicate check and its wait sees the new seq_
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

// waiters and idle spinners yield the CPU now and then (round 4: profiles/round4/ab_pool_yield_tail_latency.txt)
static constexpr bool kPoolYield = true;

static inline void CpuRelax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}

uint32_t ThreadPool::RunFast(FastLoop& L, uint32_t gen) {
  uint32_t ran = 0;
  for (;;) {
    uint64_t cur = L.next.load(std::memory_order_acquire);
    if ((uint32_t)(cur >> 32) != gen) break;                 // the slot moved on to another loop
    const uint32_t idx = (uint32_t)cur;
    // (acquire, paired with the owner's release stores: a thread that sees a NEW loop's n / grain / fn has the new claim word —
    //  stored before them — ordered before its compare-and-swap below, which therefore fails against an old `cur`.  On x86 this
    //  costs nothing; round 5's fix relied on TSO for it.)
    const uint32_t n = L.n.load(std::memory_order_acquire);
    if (idx >= n) break;                                     // everything claimed
    const uint32_t g = L.grain.load(std::memory_order_acquire);
    const uint32_t end = idx + g < n ? idx + g : n;
    // the generation in the word makes the claim fail if the parameters just read belong to an older loop
    if (!L.next.compare_exchange_weak(cur, ((uint64_t)gen << 32) | end, std::memory_order_acq_rel, std::memory_order_acquire)) continue;
    const std::function<void(size_t)>* fn = L.fn.load(std::memory_order_acquire);
    for (uint32_t i = idx; i < end; ++i) (*fn)(i);
    ran += end - idx;
    L.done.fetch_add(end - idx, std::memory_order_acq_rel);   // the owner returns (and fn dies) only after this
  }
  return ran;
}

bool ThreadPool::HelpFastLoops() {
  bool any = false;
  for (FastLoop& L : fast_) {
    if (L.state.load(std::memory_order_acquire) != 1) continue;
    const uint32_t gen = (uint32_t)(L.next.load(std::memory_order_acquire) >> 32);
    any |= RunFast(L, gen) > 0;
  }
  return any;
}

void ThreadPool::WorkerMain() {
  (void)pthread_setname_np(pthread_self(), spin_us_ ? "hps-serving" : "hps-pool");
  if (g_numa_node.load(std::memory_order_acquire) >= 0) (void)pthread_setaffinity_np(pthread_self(), sizeof g_numa_cpus, &g_numa_cpus);
  uint64_t seen = 0;
  for (;;) {
    std::shared_ptr<Loop> loop;
    std::function<void()> task;
    // read the enqueue counter BEFORE looking for work: anything published after this read changes it, so the idle
    // wait below cannot sleep through a loop this pass did not see
    seen = seq_.load(std::memory_order_acquire);
    if (spin_us_ && HelpFastLoops()) continue;
    {
      std::lock_guard<SpinLock> lk(qlock_);
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
      if (!loop && !task && stop_.load(std::memory_order_acquire)) return;
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
        // an idle spinner must not keep a CPU from a thread that has work (see ParallelFor): offer it now and then —
        // a yield with nobody waiting returns at once
        if ((it & 255) == 255 && kPoolYield) sched_yield();
        if ((it & 63) == 63 &&
            std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us_)) break;
      }
      if (woke) continue;
    }
    // park.  sleepers_ is raised BEFORE the predicate is evaluated and producers read it AFTER bumping seq_ (both
    // sequentially consistent): either the producer sees a sleeper and wakes it, or the sleeper sees the new seq_.
    sleepers_.fetch_add(1);
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return stop_.load() || seq_.load() != seen; });
    }
    sleepers_.fetch_sub(1);
  }
}

void ThreadPool::ParallelFor(size_t num_tasks, const std::function<void(size_t)>& fn, size_t max_parallel) {
  if (num_tasks == 0) return;
  if (num_tasks == 1 || workers_.empty() || max_parallel == 1) {
    for (size_t i = 0; i < num_tasks; ++i) fn(i);
    return;
  }
  // ---- fast path (spinning pools): a lock-free slot, see FastLoop ----
  if (spin_us_ && num_tasks < (1u << 31)) {
    FastLoop* L = nullptr;
    for (FastLoop& s : fast_) {
      uint32_t expect = 0;
      if (s.state.compare_exchange_strong(expect, 2, std::memory_order_acq_rel)) { L = &s; break; }
    }
    if (L) {
      size_t threads = workers_.size() + 1;
      if (max_parallel && max_parallel < threads) threads = max_parallel;
      // grains: one claim per thread when the caller asked for a cap, else task by task (callers size their tasks)
      const uint32_t grain = max_parallel ? (uint32_t)((num_tasks + threads - 1) / threads) : 1u;
      const uint32_t gen = ++L->gen;
      // The slot's claim word changes generation BEFORE any parameter of the new loop is stored, with "everything claimed" as
      // its index.  Round 5 found the window this closes (tools/micro/forkjoin_stress.cpp reproduces it in seconds, 14
      // callers on 13 workers): a worker still inside RunFast for the slot's PREVIOUS loop — claim word (old generation, n_old)
      // — read the NEW n before the new claim word was there, took n_old < n_new for unclaimed work, won its compare-and-swap
      // (the word had not changed yet) and ran tasks of the new loop; the owner then reset the index and every task ran again:
      // tasks executed twice (rows were still right: staging and gather are idempotent) and `done` overshot n, which the
      // owner's wait for done == n never survives — the one hang in ~60 bench runs, one in two runs of the sharded stress
      // driver (9 sessions + 3 entry sessions fork-joining side by side).
      L->next.store(((uint64_t)gen << 32) | 0xFFFFFFFFu, std::memory_order_seq_cst);
      L->fn.store(&fn, std::memory_order_release);
      L->n.store((uint32_t)num_tasks, std::memory_order_release);
      L->grain.store(grain ? grain : 1u, std::memory_order_release);
      L->done.store(0, std::memory_order_relaxed);
      L->next.store((uint64_t)gen << 32, std::memory_order_release);
      L->state.store(1, std::memory_order_release);
      seq_.fetch_add(1);
      if (sleepers_.load() > 0) {
        { std::lock_guard<std::mutex> lk(mu_); }
        cv_.notify_all();
      }
      RunFast(*L, gen);
      // Stragglers.  Usually microseconds.  But a worker that claimed a grain and then lost its CPU (more runnable threads
      // than CPUs in the container: 13 spinning workers, the session threads, the HIP runtime's own) comes back a
      // scheduler slice later — 4 to 6 ms, the 5-to-12-ms requests of rounds 2-3 (HPS_TRACE_TAIL: "pool 4.12 ms" of a
      // 0.15-ms staging loop, "ps fetch 5.56 ms").  The grain cannot be taken over (fn dies with this call), but the CPU
      // can be handed over: after a short spin the waiter yields, and the runnable worker gets it at once.
      for (uint32_t it = 0; L->done.load(std::memory_order_acquire) < (uint32_t)num_tasks; ++it) {
        if (it < 2048 || !kPoolYield) CpuRelax();
        else sched_yield();
      }
      // a task that ran twice shows here (the wait above is `<`, so it would not hang any more): counted, never silent
      if (L->done.load(std::memory_order_acquire) > (uint32_t)num_tasks) g_fast_overruns.fetch_add(1, std::memory_order_relaxed);
      L->state.store(0, std::memory_order_release);
      return;
    }
  }
  auto loop = std::make_shared<Loop>();
  loop->fn = &fn;
  loop->n = num_tasks;
  size_t helpers = workers_.size();
  if (max_parallel && max_parallel - 1 < helpers) helpers = max_parallel - 1;
  if (num_tasks - 1 < helpers) helpers = num_tasks - 1;
  loop->max_helpers = helpers;
  {
    std::lock_guard<SpinLock> lk(qlock_);
    loops_.push_back(loop);
    seq_.fetch_add(1);
  }
  // parked workers: an empty critical section on mu_ orders the seq_ bump against a worker that has checked its
  // predicate but not yet gone to sleep, then the wake-up (spinning workers have seen seq_ already)
  if (sleepers_.load() > 0) {
    { std::lock_guard<std::mutex> lk(mu_); }
    if (helpers >= workers_.size() / 2) cv_.notify_all();
    else for (size_t i = 0; i < helpers; ++i) cv_.notify_one();
  }
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
    std::lock_guard<SpinLock> lk(qlock_);
    tasks_.push_back(std::move(fn));
    seq_.fetch_add(1);
  }
  { std::lock_guard<std::mutex> lk(mu_); }
  cv_.notify_one();
}

}  // namespace hps
