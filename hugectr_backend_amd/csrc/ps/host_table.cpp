#include "host_table.h"

#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../common/hps_hash.h"

namespace hps {

namespace {

constexpr size_t kHuge = 2ull << 20;

void* SlabAlloc(size_t bytes) {
  if (bytes == 0) bytes = 64;
  void* p = nullptr;
  const size_t align = bytes >= kHuge ? kHuge : 64;
  const size_t rounded = (bytes + align - 1) / align * align;
  if (posix_memalign(&p, align, rounded) != 0) return nullptr;
#ifdef MADV_HUGEPAGE
  if (rounded >= kHuge) madvise(p, rounded, MADV_HUGEPAGE);
#endif
  return p;
}

Status ReadFileParallel(const std::string& path, void* dst, size_t bytes, ThreadPool* pool) {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) return Error(Code::kNotFound, "cannot open '", path, "': ", strerror(errno));
  const size_t chunk = 64ull << 20;
  const size_t ntasks = (bytes + chunk - 1) / chunk;
  std::atomic<int> bad{0};
  auto body = [&](size_t ti) {
    size_t off = ti * chunk;
    const size_t end = std::min(bytes, off + chunk);
    while (off < end) {
      const ssize_t r = pread(fd, (char*)dst + off, end - off, (off_t)off);
      if (r <= 0) { bad.store(1); return; }
      off += (size_t)r;
    }
  };
  if (pool) pool->ParallelFor(ntasks, body);
  else for (size_t i = 0; i < ntasks; ++i) body(i);
  close(fd);
  if (bad.load()) return Error(Code::kInternal, "short read on '", path, "'");
  return Status::Ok();
}

Status FileSize(const std::string& path, size_t* out) {
  struct stat st;
  if (stat(path.c_str(), &st) != 0) return Error(Code::kNotFound, "cannot stat '", path, "': ", strerror(errno));
  *out = (size_t)st.st_size;
  return Status::Ok();
}

}  // namespace

// Several GPUs read a page-locked host tier at once (ps_direct_access, one process, one copy of the tables: DESIGN.md §5).
// By default HIP places page-locked memory on the NUMA node next to the CURRENT device: on a two-socket node every GPU of the
// other socket would read all of its missed rows across the socket link, and one socket's DRAM channels would serve all of
// them.  With more than one GPU and more than one NUMA node the tables are interleaved over the nodes instead
// (hipHostMallocNumaUser + MPOL_INTERLEAVE around the allocation; HPS_HOST_NUMA_INTERLEAVE=0/1 overrides).  One GPU: unchanged.
static bool InterleavePinnedTables() {
  static const bool on = [] {
    if (const char* e = std::getenv("HPS_HOST_NUMA_INTERLEAVE")) return std::strtol(e, nullptr, 10) != 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); return false; }
    struct stat st;
    return ndev > 1 && stat("/sys/devices/system/node/node1", &st) == 0;
  }();
  return on;
}

void* HostTable::DataAlloc(size_t bytes) {
  if (!pinned_) return SlabAlloc(bytes);
  void* p = nullptr;
  unsigned flags = hipHostMallocMapped | hipHostMallocPortable;
  bool policy_set = false;
  if (InterleavePinnedTables()) {
    unsigned long mask[16];   // every node the kernel lists
    unsigned long maxnode = 0;
    for (int n = 0; n < 1024; ++n) {
      struct stat st;
      char path[64];
      snprintf(path, sizeof path, "/sys/devices/system/node/node%d", n);
      if (stat(path, &st) != 0) break;
      maxnode = (unsigned long)n + 1;
    }
    memset(mask, 0, sizeof mask);
    for (unsigned long n = 0; n < maxnode && n < sizeof(mask) * 8; ++n) mask[n / (8 * sizeof(long))] |= 1ul << (n % (8 * sizeof(long)));
    policy_set = maxnode > 1 && syscall(SYS_set_mempolicy, 3 /*MPOL_INTERLEAVE*/, mask, maxnode + 1) == 0;
    if (policy_set) flags |= hipHostMallocNumaUser;
  }
  const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 64, flags);
  if (policy_set) (void)syscall(SYS_set_mempolicy, 0 /*MPOL_DEFAULT*/, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void HostTable::DataFree(void* p) {
  if (!p) return;
  if (pinned_) (void)hipHostFree(p);
  else free(p);
}

HostTable::HostTable(std::string name, uint32_t dim, size_t num_partitions, bool pinned)
    : name_(std::move(name)), dim_(dim), pinned_(pinned) {
  if (num_partitions == 0) num_partitions = 1;
  for (size_t p = 0; p < num_partitions; ++p) parts_.emplace_back(new Partition());
  pow2_parts_ = (num_partitions & (num_partitions - 1)) == 0;
}

HostTable::~HostTable() { FreeAll(); }

void HostTable::FreeAll() {
  for (auto& p : parts_) { free(p->slots); p->slots = nullptr; p->mask = 0; p->used.store(0); }
  if (owns_keys_) DataFree(keys_);
  if (owns_rows_) DataFree(rows_);
  if (map_bytes_) munmap(rows_, map_bytes_);
  map_bytes_ = 0; map_dir_.clear(); rows_writable_ = true;
  vt_.reset();
  keys_ = nullptr; rows_ = nullptr; num_rows_ = cap_rows_ = 0;
  owns_keys_ = owns_rows_ = false;
  has_sentinel_ = false; sentinel_row_ = -1; has_dups_ = false;
}

uint64_t HostTable::SlotOf(int64_t key, uint64_t mask) { return (hps_mix64((uint64_t)key) >> 20) & mask; }

size_t HostTable::PartitionOf(int64_t key) const {
  const uint64_t k = (uint64_t)key;
  return pow2_parts_ ? (size_t)(k & (parts_.size() - 1)) : (size_t)(k % parts_.size());
}

Status HostTable::AllocPartitions(const std::vector<size_t>& counts) {
  for (size_t p = 0; p < parts_.size(); ++p) {
    uint64_t cap = 16;
    while (cap < (uint64_t)((double)counts[p] * 2.0 * index_headroom_)) cap <<= 1;
    free(parts_[p]->slots);
    parts_[p]->slots = (Entry*)SlabAlloc(cap * sizeof(Entry));
    if (!parts_[p]->slots) return Error(Code::kInternal, "host table '", name_, "': out of memory for index");
    parts_[p]->mask = cap - 1;
    parts_[p]->used.store(0);
  }
  return Status::Ok();
}

void HostTable::InsertConcurrent(int64_t key, int64_t row) {
  if (key == HPS_EMPTY_KEY) {  // legal key that collides with the empty marker: kept on the side
    has_sentinel_ = true;
    int64_t cur = __atomic_load_n(&sentinel_row_, __ATOMIC_RELAXED);
    while (cur < row && !__atomic_compare_exchange_n(&sentinel_row_, &cur, row, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return;
  }
  Partition& P = *parts_[PartitionOf(key)];
  uint64_t s = SlotOf(key, P.mask);
  for (;;) {
    int64_t k = __atomic_load_n(&P.slots[s].key, __ATOMIC_ACQUIRE);
    if (k == HPS_EMPTY_KEY) {
      int64_t expected = HPS_EMPTY_KEY;
      if (__atomic_compare_exchange_n(&P.slots[s].key, &expected, key, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
        P.used.fetch_add(1, std::memory_order_relaxed);
        k = key;
      } else {
        k = expected;
      }
    }
    if (k == key) {
      // duplicate keys in one load: the later row wins -> atomic max of the row number
      int64_t cur = __atomic_load_n(&P.slots[s].row, __ATOMIC_RELAXED);
      while (cur < row && !__atomic_compare_exchange_n(&P.slots[s].row, &cur, row, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
      return;
    }
    s = (s + 1) & P.mask;
  }
}

Status HostTable::BuildIndex(ThreadPool* pool) {
  const size_t R = num_rows_;
  const size_t P = parts_.size();
  const size_t chunk = 1 << 16;
  const size_t ntasks = (R + chunk - 1) / chunk;
  // 1) rows per partition
  std::vector<size_t> counts(P, 0);
  {
    std::mutex mu;
    int64_t lowest = INT64_MAX;
    auto body = [&](size_t ti) {
      std::vector<size_t> local(P, 0);
      int64_t lo = INT64_MAX;
      const size_t b = ti * chunk, e = std::min(R, b + chunk);
      for (size_t r = b; r < e; ++r)
        if (keys_[r] != HPS_EMPTY_KEY) { ++local[PartitionOf(keys_[r])]; lo = std::min(lo, keys_[r]); }
      std::lock_guard<std::mutex> lk(mu);
      for (size_t p = 0; p < P; ++p) counts[p] += local[p];
      lowest = std::min(lowest, lo);
    };
    if (pool) pool->ParallelFor(ntasks, body); else for (size_t i = 0; i < ntasks; ++i) body(i);
    min_key_.store(lowest == INT64_MAX ? 0 : lowest, std::memory_order_relaxed);
  }
  HPS_RETURN_IF_ERROR(AllocPartitions(counts));
  // 2) clear
  {
    struct Span { Entry* p; size_t n; };
    std::vector<Span> spans;
    for (auto& part : parts_) {
      const size_t cap = part->mask + 1;
      for (size_t o = 0; o < cap; o += chunk) spans.push_back({part->slots + o, std::min(chunk, cap - o)});
    }
    auto body = [&](size_t ti) {
      Entry* p = spans[ti].p;
      for (size_t i = 0; i < spans[ti].n; ++i) { p[i].key = HPS_EMPTY_KEY; p[i].row = -1; }
    };
    if (pool) pool->ParallelFor(spans.size(), body); else for (size_t i = 0; i < spans.size(); ++i) body(i);
  }
  // 3) concurrent insert
  has_sentinel_ = false; sentinel_row_ = -1;
  {
    auto body = [&](size_t ti) {
      const size_t b = ti * chunk, e = std::min(R, b + chunk);
      for (size_t r = b; r < e; ++r) InsertConcurrent(keys_[r], (int64_t)r);
    };
    if (pool) pool->ParallelFor(ntasks, body); else for (size_t i = 0; i < ntasks; ++i) body(i);
  }
  size_t uniq = has_sentinel_ ? 1 : 0;
  for (auto& part : parts_) uniq += part->used.load();
  has_dups_ = uniq != R;
  generation_.fetch_add(1, std::memory_order_acq_rel);
  return Status::Ok();
}

Status HostTable::LoadFromDir(const std::string& dir, ThreadPool* pool) {
  size_t kb = 0, vb = 0;
  HPS_RETURN_IF_ERROR(FileSize(dir + "/key", &kb));
  HPS_RETURN_IF_ERROR(FileSize(dir + "/emb_vector", &vb));
  if (kb % sizeof(int64_t) != 0)
    return Error(Code::kInvalidArg, "'", dir, "/key': size ", kb, " is not a multiple of 8 (int64 keys)");
  const size_t R = kb / sizeof(int64_t);
  if (vb != R * (size_t)dim_ * sizeof(float))
    return Error(Code::kInvalidArg, "'", dir, "/emb_vector': size ", vb, " != rows(", R, ") x embedding_vecsize(",
                 dim_, ") x 4; check embedding_vecsize_per_table");
  WriteLock lk(*this);
  FreeAll();
  if (tier_opt_.tiered) {
    if (pinned_) return Error(Code::kUnsupported, "host table '", name_, "': ps_direct_access needs the whole table in RAM");
    // rows stay on disk: map the row store, keep only the keys (and their index) in RAM
    keys_ = (int64_t*)DataAlloc(kb);
    owns_keys_ = true;
    if (!keys_) return Error(Code::kInternal, "host table '", name_, "': out of memory (", kb, " bytes)");
    HPS_RETURN_IF_ERROR(ReadFileParallel(dir + "/key", keys_, kb, pool));
    const bool writable = tier_opt_.persistent && tier_opt_.store_writable;
    const int fd = open((dir + "/emb_vector").c_str(), writable ? O_RDWR : O_RDONLY);
    if (fd < 0) return Error(Code::kNotFound, "cannot open '", dir, "/emb_vector': ", strerror(errno));
    void* m = vb ? mmap(nullptr, vb, writable ? PROT_READ | PROT_WRITE : PROT_READ, MAP_SHARED, fd, 0) : nullptr;
    close(fd);
    if (vb && m == MAP_FAILED) return Error(Code::kInternal, "cannot map '", dir, "/emb_vector': ", strerror(errno));
    if (vb) madvise(m, vb, MADV_RANDOM);
    rows_ = (float*)m;
    map_bytes_ = vb;
    map_dir_ = dir;
    rows_writable_ = writable;
    num_rows_ = cap_rows_ = R;
    return FinishLoad(pool);
  }
  keys_ = (int64_t*)DataAlloc(kb);
  rows_ = (float*)DataAlloc(vb);
  owns_keys_ = owns_rows_ = true;
  if (!keys_ || !rows_) return Error(Code::kInternal, "host table '", name_, "': out of memory (", kb + vb, " bytes)");
  num_rows_ = cap_rows_ = R;
  HPS_RETURN_IF_ERROR(ReadFileParallel(dir + "/key", keys_, kb, pool));
  HPS_RETURN_IF_ERROR(ReadFileParallel(dir + "/emb_vector", rows_, vb, pool));
  return FinishLoad(pool);
}

Status HostTable::FinishLoad(ThreadPool* pool) {
  HPS_RETURN_IF_ERROR(BuildIndex(pool));
  SetupTier();
  {
    // a new load epoch: whoever holds copies of rows of the previous contents has to take them all again
    std::lock_guard<std::mutex> lk(log_mu_);
    ++load_epoch_;
    log_base_ += change_log_.size();
    change_log_.clear();
  }
  return Status::Ok();
}

void HostTable::LogChanges(const int64_t* keys, size_t n) {
  std::lock_guard<std::mutex> lk(log_mu_);
  if (n >= kChangeLogMax) {   // more than the log holds in one go: nothing older is worth keeping
    log_base_ += change_log_.size() + n;
    change_log_.clear();
    return;
  }
  if (change_log_.size() + n > kChangeLogMax) {
    const size_t drop = std::max(change_log_.size() / 2, change_log_.size() + n - kChangeLogMax);
    change_log_.erase(change_log_.begin(), change_log_.begin() + drop);
    log_base_ += drop;
  }
  change_log_.insert(change_log_.end(), keys, keys + n);
}

void HostTable::ChangeMark(uint64_t* load_epoch, uint64_t* log_seq) const {
  std::lock_guard<std::mutex> lk(log_mu_);
  *load_epoch = load_epoch_;
  *log_seq = log_base_ + change_log_.size();
}

bool HostTable::ChangesSince(uint64_t load_epoch, uint64_t log_seq, std::vector<int64_t>* keys, uint64_t* new_epoch, uint64_t* new_seq) const {
  std::lock_guard<std::mutex> lk(log_mu_);
  *new_epoch = load_epoch_;
  *new_seq = log_base_ + change_log_.size();
  keys->clear();
  if (load_epoch != load_epoch_ || log_seq < log_base_ || log_seq > *new_seq) return false;
  keys->assign(change_log_.begin() + (log_seq - log_base_), change_log_.end());
  return true;
}

// Builds the volatile tier over the freshly indexed row store and caches the first initial_cache_rate * R rows in
// file order (docs/hierarchical_parameter_server.md:491-495; the same convention as the GPU cache's warm-up).
void HostTable::SetupTier() {
  vt_.reset();
  if (!tier_opt_.tiered) return;
  std::vector<size_t> per_part(parts_.size());
  for (size_t p = 0; p < parts_.size(); ++p) per_part[p] = parts_[p]->used.load();
  if (has_sentinel_) ++per_part[PartitionOf(HPS_EMPTY_KEY)];
  vt_.reset(new VolatileTier(dim_, per_part, tier_opt_.vdb));
  double rate = tier_opt_.vdb.initial_cache_rate;
  if (!(rate >= 0.0)) rate = 0.0;
  if (rate > 1.0) rate = 1.0;
  const size_t first = (size_t)std::ceil(rate * (double)num_rows_);
  for (size_t r = 0; r < first && r < num_rows_; ++r) {
    if (FindUnlocked(keys_[r]) != (int64_t)r) continue;   // an older duplicate of a key: not the live row
    vt_->Insert(PartitionOf(keys_[r]), keys_[r], rows_ + r * dim_, 0);
  }
  clock_.store(0);
  persistent_hits_.store(0);
  not_found_.store(0);
}

HostTierStats HostTable::tier_stats() const {
  ReadLock lk(*this);
  HostTierStats s;
  s.persistent_rows = num_rows_;
  if (vt_) s.vdb = vt_->stats();
  s.persistent_hits = persistent_hits_.load();
  s.not_found = not_found_.load();
  return s;
}

void HostTable::DumpVolatileKeys(std::vector<int64_t>* out) const {
  ReadLock lk(*this);
  out->clear();
  if (!vt_) return;
  vt_->DumpKeys(out);
}

size_t HostTable::FetchTiered(const int64_t* keys, size_t n, float* out, size_t stride, float default_value,
                              uint8_t* found) const {
  const uint32_t D = dim_;
  const uint64_t now = clock_.fetch_add(1, std::memory_order_relaxed) + 1;
  const bool cache_missed = tier_opt_.vdb.cache_missed_embeddings;
  // Bucket the request by partition (stable): one shared lock per partition for its lookups, then one exclusive lock
  // for the rows that had to come from behind the tier.  Partitions are independent, so this is the same as "all
  // lookups of the call, then all inserts in request order" (oracle: VolatileDbModel.fetch).
  const size_t P = parts_.size();
  std::vector<uint32_t> start(P + 1, 0), order(n);
  for (size_t i = 0; i < n; ++i) ++start[PartitionOf(keys[i]) + 1];
  for (size_t p = 0; p < P; ++p) start[p + 1] += start[p];
  {
    std::vector<uint32_t> fill(start.begin(), start.end() - 1);
    for (size_t i = 0; i < n; ++i) order[fill[PartitionOf(keys[i])]++] = (uint32_t)i;
  }
  std::vector<std::pair<int64_t, int64_t>> missed;   // (key, row in the store) served from behind the volatile tier
  size_t nfound = 0, from_store = 0, absent = 0;
  for (size_t p = 0; p < P; ++p) {
    if (start[p] == start[p + 1]) continue;
    {
      std::shared_lock<std::shared_mutex> tl(vt_->mutex(p));
      for (uint32_t o = start[p]; o < start[p + 1]; ++o) {
        const size_t i = order[o];
        const int64_t key = keys[i];
        float* dst = out + i * stride;
        bool ok = vt_->Lookup(p, key, dst, now);
        if (!ok && tier_opt_.persistent) {
          const int64_t r = FindUnlocked(key);
          if (r >= 0) {
            memcpy(dst, rows_ + (size_t)r * D, (size_t)D * sizeof(float));
            ok = true;
            ++from_store;
            if (cache_missed) missed.emplace_back(key, r);
          }
        }
        if (!ok) {
          for (uint32_t c = 0; c < D; ++c) dst[c] = default_value;
          ++absent;
        }
        nfound += ok;
        if (found) found[i] = ok ? 1 : 0;
      }
    }
    if (!missed.empty()) {
      std::unique_lock<std::shared_mutex> tl(vt_->mutex(p));
      for (const auto& kr : missed) vt_->Insert(p, kr.first, rows_ + (size_t)kr.second * D, now);
      missed.clear();
    }
  }
  if (from_store) persistent_hits_.fetch_add(from_store, std::memory_order_relaxed);
  if (absent) not_found_.fetch_add(absent, std::memory_order_relaxed);
  return nfound;
}

Status HostTable::LoadFromArrays(const int64_t* keys, const float* rows, size_t R, bool borrow, ThreadPool* pool) {
  WriteLock lk(*this);
  FreeAll();
  if (borrow && !pinned_) {  // a pinned table always owns its (device-mapped) storage
    keys_ = const_cast<int64_t*>(keys);
    rows_ = const_cast<float*>(rows);
  } else {
    keys_ = (int64_t*)DataAlloc(R * sizeof(int64_t));
    rows_ = (float*)DataAlloc(R * (size_t)dim_ * sizeof(float));
    owns_keys_ = owns_rows_ = true;
    if (!keys_ || !rows_) return Error(Code::kInternal, "host table '", name_, "': out of memory");
    memcpy(keys_, keys, R * sizeof(int64_t));
    memcpy(rows_, rows, R * (size_t)dim_ * sizeof(float));
  }
  num_rows_ = cap_rows_ = R;
  return FinishLoad(pool);
}

// Synthetic rows, 8 splitmix64 lanes at a time (the generator is the setup cost of the benchmark: 33 G
// elements for BASELINE config 2).  Same arithmetic as hps_synth_elem_bits: word j/2 of the row yields
// element j (low 32 bits) and j+1 (high 32 bits), each masked to 23 mantissa bits under exponent 0x3F0.
// Returns the number of elements written (a multiple of 16).
__attribute__((target("avx512f,avx512dq"))) static uint32_t SynthRowAvx512(uint64_t rb, uint32_t D, uint32_t* dst) {
  const __m512i c0 = _mm512_set1_epi64((long long)0x9E3779B97F4A7C15ull);
  const __m512i c1 = _mm512_set1_epi64((long long)0xBF58476D1CE4E5B9ull);
  const __m512i c2 = _mm512_set1_epi64((long long)0x94D049BB133111EBull);
  const __m512i mant = _mm512_set1_epi64((long long)0x007FFFFF007FFFFFull);
  const __m512i expo = _mm512_set1_epi64((long long)0x3F0000003F000000ull);
  const __m512i lane = _mm512_set_epi64(7, 6, 5, 4, 3, 2, 1, 0);
  uint32_t j = 0;
  for (; j + 16 <= D; j += 16) {
    __m512i x = _mm512_add_epi64(_mm512_add_epi64(_mm512_set1_epi64((long long)(rb + (j >> 1))), lane), c0);
    x = _mm512_mullo_epi64(_mm512_xor_si512(x, _mm512_srli_epi64(x, 30)), c1);
    x = _mm512_mullo_epi64(_mm512_xor_si512(x, _mm512_srli_epi64(x, 27)), c2);
    x = _mm512_xor_si512(x, _mm512_srli_epi64(x, 31));
    x = _mm512_or_si512(_mm512_and_si512(x, mant), expo);
    _mm512_storeu_si512((void*)(dst + j), x);
  }
  return j;
}

Status HostTable::LoadSynthetic(uint64_t seed, uint32_t table_id, int64_t key0, size_t R, ThreadPool* pool,
                                uint32_t shard, uint32_t num_shards) {
  if (num_shards > 1) {
    // One shard of the table key0..key0+R-1: the keys whose owner (mix64(key) mod num_shards, the function the
    // sharded lookup routes with) is `shard`, in key order.  Pass 1 counts per chunk, pass 2 fills.
    if (shard >= num_shards) return Error(Code::kInvalidArg, "shard ", shard, " of ", num_shards);
    WriteLock lk(*this);
    FreeAll();
    const size_t chunk = 1u << 16;
    const size_t ntasks = (R + chunk - 1) / chunk;
    std::vector<size_t> start(ntasks + 1, 0);
    auto owned = [&](int64_t key) { return hps_mix64((uint64_t)key) % num_shards == shard; };
    auto count = [&](size_t ti) {
      const size_t b = ti * chunk, e = std::min(R, b + chunk);
      size_t c = 0;
      for (size_t r = b; r < e; ++r) c += owned(key0 + (int64_t)r);
      start[ti + 1] = c;
    };
    if (pool) pool->ParallelFor(ntasks, count); else for (size_t i = 0; i < ntasks; ++i) count(i);
    for (size_t i = 0; i < ntasks; ++i) start[i + 1] += start[i];
    const size_t Rs = start[ntasks];
    keys_ = (int64_t*)DataAlloc(std::max<size_t>(Rs, 1) * sizeof(int64_t));
    rows_ = (float*)DataAlloc(std::max<size_t>(Rs, 1) * (size_t)dim_ * sizeof(float));
    owns_keys_ = owns_rows_ = true;
    if (!keys_ || !rows_) return Error(Code::kInternal, "host table '", name_, "': out of memory");
    num_rows_ = cap_rows_ = Rs;
    const uint64_t tb = hps_synth_table_base(seed, table_id);
    const uint32_t D = dim_;
    const bool use_avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
    auto fill = [&](size_t ti) {
      const size_t b = ti * chunk, e = std::min(R, b + chunk);
      size_t w = start[ti];
      for (size_t r = b; r < e; ++r) {
        const int64_t key = key0 + (int64_t)r;
        if (!owned(key)) continue;
        keys_[w] = key;
        const uint64_t rb = hps_synth_row_base(tb, key);
        uint32_t* dst = reinterpret_cast<uint32_t*>(rows_ + w * D);
        const uint32_t j0 = use_avx512 ? SynthRowAvx512(rb, D, dst) : 0u;
        for (uint32_t j = j0; j < D; ++j) dst[j] = hps_synth_elem_bits(rb, j);
        ++w;
      }
    };
    if (pool) pool->ParallelFor(ntasks, fill); else for (size_t i = 0; i < ntasks; ++i) fill(i);
    return FinishLoad(pool);
  }
  WriteLock lk(*this);
  FreeAll();
  keys_ = (int64_t*)DataAlloc(R * sizeof(int64_t));
  rows_ = (float*)DataAlloc(R * (size_t)dim_ * sizeof(float));
  owns_keys_ = owns_rows_ = true;
  if (!keys_ || !rows_) return Error(Code::kInternal, "host table '", name_, "': out of memory");
  num_rows_ = cap_rows_ = R;
  const uint64_t tb = hps_synth_table_base(seed, table_id);
  const size_t chunk = 4096;
  const size_t ntasks = (R + chunk - 1) / chunk;
  const uint32_t D = dim_;
  const bool use_avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
  auto body = [&](size_t ti) {
    const size_t b = ti * chunk, e = std::min(R, b + chunk);
    for (size_t r = b; r < e; ++r) {
      const int64_t key = key0 + (int64_t)r;
      keys_[r] = key;
      const uint64_t rb = hps_synth_row_base(tb, key);
      uint32_t* dst = reinterpret_cast<uint32_t*>(rows_ + r * D);
      uint32_t j0 = 0;
      if (use_avx512) j0 = SynthRowAvx512(rb, D, dst);  // whole groups of 16 elements; the tail below
      for (uint32_t j = j0; j + 1 < D; j += 2) {
        const uint64_t w = hps_mix64(rb + (uint64_t)(j >> 1));
        dst[j] = 0x3F000000u | ((uint32_t)w & 0x007FFFFFu);
        dst[j + 1] = 0x3F000000u | ((uint32_t)(w >> 32) & 0x007FFFFFu);
      }
      if (D & 1u) dst[D - 1] = hps_synth_elem_bits(rb, D - 1);
    }
  };
  if (pool) pool->ParallelFor(ntasks, body); else for (size_t i = 0; i < ntasks; ++i) body(i);
  return FinishLoad(pool);
}


int64_t HostTable::FindUnlocked(int64_t key) const {
  if (key == HPS_EMPTY_KEY) return has_sentinel_ ? sentinel_row_ : -1;
  const Partition& P = *parts_[PartitionOf(key)];
  if (!P.slots) return -1;
  uint64_t s = SlotOf(key, P.mask);
  for (;;) {
    const int64_t k = P.slots[s].key;
    if (k == key) return P.slots[s].row;
    if (k == HPS_EMPTY_KEY) return -1;
    s = (s + 1) & P.mask;
  }
}

int64_t HostTable::Find(int64_t key) const {
  ReadLock lk(*this);
  return FindUnlocked(key);
}

size_t HostTable::Fetch(const int64_t* keys, size_t n, float* out, size_t stride, float default_value,
                        uint8_t* found) const {
  ReadLock lk(*this);
  if (vt_) return FetchTiered(keys, n, out, stride, default_value, found);
  const uint32_t D = dim_;
  const size_t row_bytes = (size_t)D * sizeof(float);
  size_t nfound = 0;
  // Three-stage software pipeline over blocks of B keys.  The lookup of one key is two dependent DRAM
  // round trips (index slot, then the row = row_bytes/64 cache lines); a thread that waits for them one
  // key at a time spends ~all of its CPU time stalled.  While block j's rows are copied, block j+1's
  // index slots have been read and its row lines are being prefetched, and block j+2's index slots are
  // being prefetched.  Rows are written with non-temporal stores when 16-B aligned: the destination
  // (pinned staging / the response buffer) is not read back by this thread, and a regular store would first
  // fetch every destination line (read-for-ownership), doubling the memory traffic of the gather.
  // Blocks of 8 rows, non-temporal prefetch hint, 16-B streaming stores: measured on the box's EPYC 9575F with tools/host_gather_bench.py
  // (rounds 2-3, then behind environment switches that went with round 5): 6-16 rows per block and any hint are inside the
  // run-to-run spread (80-105 GB/s), no row prefetch 53-58 GB/s, blocks of 2 69 GB/s, 64-B AVX-512 copies no gain.
  constexpr size_t B = 8;
  const size_t nb = (n + B - 1) / B;
  int64_t row[3][B];
  const bool nt_ok = (row_bytes % 16 == 0) && (stride % 4 == 0) && (((uintptr_t)out & 15u) == 0);
  for (size_t j = 0; j < nb + 2; ++j) {
    if (j < nb) {  // stage A: prefetch index slots of block j
      const size_t base = j * B, m = std::min(B, n - base);
      for (size_t i = 0; i < m; ++i) {
        const int64_t key = keys[base + i];
        if (key == HPS_EMPTY_KEY) continue;
        const Partition& P = *parts_[PartitionOf(key)];
        if (P.slots) __builtin_prefetch(&P.slots[SlotOf(key, P.mask)], 0, 0);
      }
    }
    if (j >= 1 && j - 1 < nb) {  // stage B: resolve block j-1, prefetch its rows
      const size_t blk = j - 1, base = blk * B, m = std::min(B, n - base);
      int64_t* r = row[blk % 3];
      for (size_t i = 0; i < m; ++i) {
        r[i] = FindUnlocked(keys[base + i]);
        if (r[i] >= 0) {
          const char* p = reinterpret_cast<const char*>(rows_ + (size_t)r[i] * D);
          for (size_t o = 0; o < row_bytes; o += 64) __builtin_prefetch(p + o, 0, 0);
        }
      }
    }
    if (j >= 2) {  // stage C: copy block j-2
      const size_t blk = j - 2, base = blk * B, m = std::min(B, n - base);
      const int64_t* r = row[blk % 3];
      for (size_t i = 0; i < m; ++i) {
        float* dst = out + (base + i) * stride;
        if (r[i] >= 0) {
          const float* src = rows_ + (size_t)r[i] * D;
          if (nt_ok) {
            for (uint32_t c = 0; c < D; c += 4) _mm_stream_ps(dst + c, _mm_loadu_ps(src + c));
          } else {
            memcpy(dst, src, row_bytes);
          }
          ++nfound;
        } else {
          for (uint32_t c = 0; c < D; ++c) dst[c] = default_value;
        }
        if (found) found[base + i] = r[i] >= 0 ? 1 : 0;
      }
    }
  }
  if (nt_ok) _mm_sfence();  // make the streamed rows globally visible before the caller hands them to the DMA engine
  return nfound;
}

Status HostTable::Upsert(const int64_t* keys, const float* rows, size_t n, unsigned layers) {
  if ((layers & (kLayerVolatile | kLayerPersistent)) == 0) return Status::Ok();
  WriteLock lk(*this);
  // logged before the rows change (and whatever the outcome): a holder that reads a half-applied update finds its keys in the log
  LogChanges(keys, n);
  if (vt_) return UpsertTiered(keys, rows, n, layers);
  const uint32_t D = dim_;
  // overwrite existing, collect new
  std::vector<size_t> fresh;
  for (size_t i = 0; i < n; ++i) {
    const int64_t r = FindUnlocked(keys[i]);
    if (r >= 0) memcpy(rows_ + (size_t)r * D, rows + i * D, (size_t)D * sizeof(float));
    else fresh.push_back(i);
  }
  if (fresh.empty()) return Status::Ok();
  return AppendRows(keys, rows, fresh);
}

// Appends the rows keys[i], i in fresh, to the row store and re-indexes (updates are rare relative to lookups).
Status HostTable::AppendRows(const int64_t* keys, const float* rows, const std::vector<size_t>& fresh) {
  const uint32_t D = dim_;
  {
    int64_t lo = num_rows_ ? min_key_.load(std::memory_order_relaxed) : INT64_MAX;
    for (size_t i : fresh) if (keys[i] != HPS_EMPTY_KEY) lo = std::min(lo, keys[i]);
    if (lo != INT64_MAX) min_key_.store(lo, std::memory_order_relaxed);
  }
  const size_t newR = num_rows_ + fresh.size();
  if (map_bytes_ || !map_dir_.empty()) {
    // mapped row store: append to its two files, map the longer file
    const size_t row_bytes = (size_t)D * sizeof(float);
    const int kfd = open((map_dir_ + "/key").c_str(), O_WRONLY);
    const int vfd = open((map_dir_ + "/emb_vector").c_str(), O_RDWR);
    if (kfd < 0 || vfd < 0) {
      if (kfd >= 0) close(kfd);
      if (vfd >= 0) close(vfd);
      return Error(Code::kInternal, "cannot open the row store under '", map_dir_, "' for appending: ", strerror(errno));
    }
    bool ok = true;
    size_t w = num_rows_;
    for (size_t i : fresh) {
      ok &= pwrite(kfd, keys + i, sizeof(int64_t), (off_t)(w * sizeof(int64_t))) == (ssize_t)sizeof(int64_t);
      ok &= pwrite(vfd, rows + i * D, row_bytes, (off_t)(w * row_bytes)) == (ssize_t)row_bytes;
      ++w;
    }
    close(kfd);
    if (!ok) { close(vfd); return Error(Code::kInternal, "short write to the row store under '", map_dir_, "'"); }
    if (map_bytes_) munmap(rows_, map_bytes_);
    void* m = mmap(nullptr, newR * row_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, vfd, 0);
    close(vfd);
    if (m == MAP_FAILED) { rows_ = nullptr; map_bytes_ = 0; return Error(Code::kInternal, "cannot re-map the row store: ", strerror(errno)); }
    madvise(m, newR * row_bytes, MADV_RANDOM);
    rows_ = (float*)m;
    map_bytes_ = newR * row_bytes;
    int64_t* nk = (int64_t*)DataAlloc(newR * sizeof(int64_t));
    if (!nk) return Error(Code::kInternal, "host table '", name_, "': out of memory");
    memcpy(nk, keys_, num_rows_ * sizeof(int64_t));
    if (owns_keys_) DataFree(keys_);
    keys_ = nk; owns_keys_ = true;
    const size_t first_new_mapped = num_rows_;
    for (size_t i : fresh) keys_[num_rows_++] = keys[i];
    cap_rows_ = num_rows_;
    return IndexAppended(first_new_mapped);
  }
  // grow slab (always into owned memory) and rebuild the index
  if (newR > cap_rows_ || !owns_keys_ || !owns_rows_) {
    const size_t cap = std::max(newR, cap_rows_ + cap_rows_ / 2);
    int64_t* nk = (int64_t*)DataAlloc(cap * sizeof(int64_t));
    float* nr = (float*)DataAlloc(cap * (size_t)D * sizeof(float));
    if (!nk || !nr) { DataFree(nk); DataFree(nr); return Error(Code::kInternal, "host table '", name_, "': out of memory"); }
    if (num_rows_) {
      memcpy(nk, keys_, num_rows_ * sizeof(int64_t));
      memcpy(nr, rows_, num_rows_ * (size_t)D * sizeof(float));
    }
    if (owns_keys_) DataFree(keys_);
    if (owns_rows_) DataFree(rows_);
    keys_ = nk; rows_ = nr; owns_keys_ = owns_rows_ = true; cap_rows_ = cap;
  }
  const size_t first_new = num_rows_;
  for (size_t i : fresh) {
    // a key may repeat inside `fresh`; the index resolves to the last row
    keys_[num_rows_] = keys[i];
    memcpy(rows_ + num_rows_ * D, rows + i * D, (size_t)D * sizeof(float));
    ++num_rows_;
  }
  return IndexAppended(first_new);
}

// Rows [first_new, num_rows_) have just been appended (writer lock held): they enter the index one by one while every
// partition stays at load <= 0.5; only when one would not is the whole index rebuilt — with room for as many keys again,
// so that a stream of online updates that keeps adding keys (incremental training does) costs O(1) per key, not one
// rebuild per message chunk (round 3: 2,900 messages of 64 new keys in chunks of 8 took over a minute).
Status HostTable::IndexAppended(size_t first_new) {
  std::vector<size_t> add(parts_.size(), 0);
  for (size_t r = first_new; r < num_rows_; ++r)
    if (keys_[r] != HPS_EMPTY_KEY) ++add[PartitionOf(keys_[r])];
  bool fits = true;
  for (size_t p = 0; p < parts_.size() && fits; ++p)
    fits = parts_[p]->slots != nullptr && (parts_[p]->used.load(std::memory_order_relaxed) + add[p]) * 2 <= parts_[p]->mask + 1;
  if (!fits) {
    index_headroom_ = 2.0;
    const Status st = BuildIndex(nullptr);
    index_headroom_ = 1.0;
    return st;
  }
  for (size_t r = first_new; r < num_rows_; ++r) InsertConcurrent(keys_[r], (int64_t)r);
  size_t uniq = has_sentinel_ ? 1 : 0;
  for (auto& part : parts_) uniq += part->used.load();
  has_dups_ = uniq != num_rows_;
  generation_.fetch_add(1, std::memory_order_acq_rel);
  return Status::Ok();
}

// Online update with a bounded volatile tier.  With a persistent database behind it the row store is the database of
// record (written through unless read_only) and a cached copy is refreshed in place; without one the volatile tier IS
// the database and takes the rows, pruning by its overflow policy like any other insert.
Status HostTable::UpsertTiered(const int64_t* keys, const float* rows, size_t n, unsigned layers) {
  const uint32_t D = dim_;
  const uint64_t now = clock_.fetch_add(1, std::memory_order_relaxed) + 1;
  // without a persistent database behind it the volatile tier is the only layer there is
  const bool vol = (layers & kLayerVolatile) != 0 || !tier_opt_.persistent;
  const bool per = (layers & kLayerPersistent) != 0 && tier_opt_.persistent;
  std::vector<size_t> fresh;
  // (the table's writer lock is held: no lookup is inside the tier, its partition locks are not needed)
  for (size_t i = 0; i < n; ++i) {
    const int64_t r = FindUnlocked(keys[i]);
    if (r >= 0 && rows_writable_ && per) memcpy(rows_ + (size_t)r * D, rows + i * D, (size_t)D * sizeof(float));
    if (r < 0) fresh.push_back(i);
    if (!vol) continue;
    const size_t p = PartitionOf(keys[i]);
    if (tier_opt_.persistent) vt_->Overwrite(p, keys[i], rows + i * D);
    else vt_->Insert(p, keys[i], rows + i * D, now);
  }
  if (fresh.empty() || !per) return Status::Ok();
  if (!rows_writable_)
    return Error(Code::kUnsupported, "host table '", name_, "': the persistent database is read_only, new keys cannot be added");
  return AppendRows(keys, rows, fresh);
}

}  // namespace hps
