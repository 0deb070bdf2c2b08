#include "engine.h"
#include "copy_engines.h"
#include "key_pack.h"

#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kernels.h"
#include "shard_kernels.h"

namespace hps {

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return ::hps::Error(::hps::Code::kInternal, #expr, " failed: ", hipGetErrorString(_e), " (", \
                          __FILE__, ":", __LINE__, ")");                                           \
  } while (0)

namespace {

constexpr size_t kStagingCapBytes = 256ull << 20;  // per-session staging chunk for missed rows
constexpr uint64_t kSmallRequestKeys = 1u << 17;   // requests up to this many keys are probed in tiles of ...
constexpr uint64_t kSmallTileKeys = 256;           // ... this many keys

Status RequireDevice(int device) {
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return Error(Code::kUnavailable,
                 "the GPU embedding cache needs a HIP device and none is visible (hipGetDeviceCount: ",
                 e == hipSuccess ? "0 devices" : hipGetErrorString(e),
                 "); there is no CPU fallback for gpucache=true models");
  if (device < 0 || device >= n)
    return Error(Code::kInvalidArg, "device ", device, " is not visible (", n, " HIP devices)");
  return Status::Ok();
}

template <typename T>
Status DevAlloc(T** p, size_t count) {
  void* v = nullptr;
  HIP_TRY(hipMalloc(&v, std::max<size_t>(count, 1) * sizeof(T)));
  *p = (T*)v;
  return Status::Ok();
}
template <typename T>
Status PinAlloc(T** p, size_t count, unsigned flags = hipHostMallocDefault) {
  void* v = nullptr;
  HIP_TRY(hipHostMalloc(&v, std::max<size_t>(count, 1) * sizeof(T), flags));
  *p = (T*)v;
  return Status::Ok();
}

}  // namespace

// =================================================================================================
// EmbeddingCache
// =================================================================================================
EmbeddingCache::~EmbeddingCache() { Release(); }

void EmbeddingCache::Release() {
  if (allocations_.empty() && !d_tables_) return;
  (void)hipSetDevice(cfg_.device_id_);
  (void)hipDeviceSynchronize();
  FreeInserter();
  FreeDirectInserter();
  for (auto& m : index_mem_) { if (m.first) (void)hipFree(m.first); if (m.second) (void)hipFree(m.second); }
  index_mem_.clear();
  if (d_index_) (void)hipFree(d_index_);
  d_index_ = nullptr;
  for (void* p : allocations_) (void)hipFree(p);
  allocations_.clear();
  if (d_tables_) (void)hipFree(d_tables_);
  d_tables_ = nullptr;
  if (last_write_) (void)hipEventDestroy(last_write_);
  last_write_ = nullptr;
}

CacheCounters EmbeddingCache::counters() const {
  {
    // inserts that sessions left running behind their last call: their statistics are part of the picture
    // (CollectDeferred selects the cache's device when there is something to wait for: the caller's current device is put back)
    // The wait for the GPU happens OUTSIDE sess_mu_ (a monitoring call must not hold up RegisterSession, or a session's next
    // call behind it): the list is copied under the lock, and a session that goes away meanwhile waits in UnregisterSession
    // until the collectors that may still hold its pointer are done.
    int dev = -1;
    (void)hipGetDevice(&dev);
    std::vector<LookupSession*> list;
    {
      std::lock_guard<std::mutex> lk(sess_mu_);
      list = sessions_;
      ++collectors_;
    }
    for (LookupSession* s : list) (void)s->CollectDeferred();
    {
      std::lock_guard<std::mutex> lk(sess_mu_);
      --collectors_;
    }
    sess_cv_.notify_all();
    if (dev >= 0) (void)hipSetDevice(dev);
  }
  std::lock_guard<std::mutex> lk(stat_mu_);
  return counters_;
}

void EmbeddingCache::RegisterSession(LookupSession* s) {
  std::lock_guard<std::mutex> lk(sess_mu_);
  sessions_.push_back(s);
}
void EmbeddingCache::UnregisterSession(LookupSession* s) {
  std::unique_lock<std::mutex> lk(sess_mu_);
  sessions_.erase(std::remove(sessions_.begin(), sessions_.end(), s), sessions_.end());
  sess_cv_.wait(lk, [&] { return collectors_ == 0; });   // a counters() call may still be collecting from the copy it took
}

// Insert statistics come back as kStatLines lines of the accumulator block (device_types.h): sum and add.
void EmbeddingCache::AddStatLines(const uint32_t* lines) {
  uint64_t v[3] = {0, 0, 0};
  for (int l = 0; l < kStatLines; ++l)
    for (int k = 0; k < 3; ++k) v[k] += lines[(size_t)l * kAccStride + k];
  std::lock_guard<std::mutex> lk(stat_mu_);
  counters_.dropped += v[0];
  counters_.inserted += v[1];
  counters_.refreshed += v[2];
}

// The call counter of the cache: one tick per lookup call (and per background insert), 32 bits, wraps freely.  What the
// kernels see of it is Stamp8(): the counter in units of 2^age_shift calls modulo kStampMod (device_types.h).  At the
// 32-bit wrap the stamps jump once (2^32 is not a multiple of 255 units): a blip in the eviction order, nothing else.
uint32_t EmbeddingCache::NextEpoch() {
  const uint64_t c = calls_.fetch_add(1, std::memory_order_relaxed) + 1;
  const uint64_t unit = call_clock_ ? (c + call_start_) >> age_shift_
                                    : 1 + clock_rows_.load(std::memory_order_relaxed) / rows_per_unit_;   // (lookups start in unit 1)
  return (uint32_t)((unit << 8) | (c & 0xFFu));
}

void EmbeddingCache::BeginRead(hipStream_t stream) {
  order_mu_.lock();
  if (has_write_) (void)hipStreamWaitEvent(stream, last_write_, 0);
  // One probe/gather at a time per cache: every launch already fills all CUs and saturates HBM, so two of
  // them side by side only interleave (each takes twice as long, no throughput gained).  Chaining them keeps
  // the first caller's latency at one kernel time.
  if (last_reader_ != nullptr && last_reader_stream_ != stream) (void)hipStreamWaitEvent(stream, last_reader_, 0);
}
void EmbeddingCache::EndRead(hipStream_t stream, hipEvent_t reader_done) {
  (void)hipEventRecord(reader_done, stream);
  if (std::find(readers_.begin(), readers_.end(), reader_done) == readers_.end()) readers_.push_back(reader_done);
  last_reader_ = reader_done;
  last_reader_stream_ = stream;
  order_mu_.unlock();
}
// Fused lookup+interaction: the probe was followed, under the same lock, by kernels that still read the slots it
// found.  Writers must wait for the last of them (reader_done, recorded here); other sessions' probes only chain
// behind the probe itself (probe_done, recorded by the caller right after it).
void EmbeddingCache::EndReadFused(hipStream_t stream, hipEvent_t probe_done, hipEvent_t reader_done) {
  (void)hipEventRecord(reader_done, stream);
  if (std::find(readers_.begin(), readers_.end(), reader_done) == readers_.end()) readers_.push_back(reader_done);
  last_reader_ = probe_done;
  last_reader_stream_ = stream;
  order_mu_.unlock();
}
void EmbeddingCache::ForgetReader(hipEvent_t reader_done) {
  std::lock_guard<std::mutex> lk(order_mu_);
  readers_.erase(std::remove(readers_.begin(), readers_.end(), reader_done), readers_.end());
  if (last_reader_ == reader_done) { last_reader_ = nullptr; last_reader_stream_ = nullptr; }
}
void EmbeddingCache::BeginFetch(hipStream_t stream) {
  fetch_mu_.lock();
  if (last_fetch_ != nullptr && last_fetch_stream_ != stream) (void)hipStreamWaitEvent(stream, last_fetch_, 0);
}
void EmbeddingCache::EndFetch(hipStream_t stream, hipEvent_t fetch_done) {
  (void)hipEventRecord(fetch_done, stream);
  last_fetch_ = fetch_done;
  last_fetch_stream_ = stream;
  fetch_mu_.unlock();
}
void EmbeddingCache::ForgetFetch(hipEvent_t fetch_done) {
  std::lock_guard<std::mutex> lk(fetch_mu_);
  if (last_fetch_ == fetch_done) { last_fetch_ = nullptr; last_fetch_stream_ = nullptr; }
}
void EmbeddingCache::LaneEnter(hipStream_t stream) {
  lane_mu_.lock();
  if (last_lane_ != nullptr && last_lane_stream_ != stream) (void)hipStreamWaitEvent(stream, last_lane_, 0);
}
void EmbeddingCache::LaneLeave(hipStream_t stream, hipEvent_t done) {
  (void)hipEventRecord(done, stream);
  last_lane_ = done;
  last_lane_stream_ = stream;
  lane_mu_.unlock();
}
void EmbeddingCache::ForgetLane(hipEvent_t done) {
  std::lock_guard<std::mutex> lk(lane_mu_);
  if (last_lane_ == done) { last_lane_ = nullptr; last_lane_stream_ = nullptr; }
}
void EmbeddingCache::BeginWrite(hipStream_t stream) {
  order_mu_.lock();
  if (has_write_) (void)hipStreamWaitEvent(stream, last_write_, 0);
  for (hipEvent_t e : readers_) (void)hipStreamWaitEvent(stream, e, 0);
}
void EmbeddingCache::EndWrite(hipStream_t stream) {
  (void)hipEventRecord(last_write_, stream);
  has_write_ = true;
  readers_.clear();
  order_mu_.unlock();
}

Status EmbeddingCache::Init(const std::string& model, const InferenceParams& p,
                            const std::vector<std::shared_ptr<HostTable>>& tables, int device, int shard, uint32_t num_shards) {
  HPS_RETURN_IF_ERROR(RequireDevice(device));
  if (shard >= 0 && (num_shards == 0 || (uint32_t)shard >= num_shards)) return Error(Code::kInvalidArg, "shard ", shard, " of ", num_shards);
  shard_ = shard;
  num_shards_ = shard >= 0 ? num_shards : 1;
  const bool sharded = shard >= 0 && num_shards > 1;
  auto owned = [&](int64_t key) { return !sharded || ShardOwnerHost(key, num_shards) == (uint32_t)shard; };
  HIP_TRY(hipSetDevice(device));
  {
    // every SDMA engine takes one tiny copy now, while the model loads (copy_engines.h); HPS_WAKE_COPY_ENGINES=0: A/B
    const char* e = std::getenv("HPS_WAKE_COPY_ENGINES");
    if (!(e && e[0] == '0')) {
      const std::string report = WakeCopyEngines(device);
      if (std::getenv("HPS_TRACE_TAIL")) fprintf(stderr, "[hps] copy engines of device %d: %s\n", device, report.c_str());
    }
  }
  model_ = model;
  const size_t T = tables.size();
  if (T == 0 || T > (size_t)kMaxTables)
    return Error(Code::kInvalidArg, "model '", model, "': ", T, " tables (supported: 1..", kMaxTables, ")");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  cu_count_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  static_ = p.embedding_cache_type == EmbeddingCacheType::Static;
  if (const char* e = std::getenv("HPS_LRU_AGE_SHIFT")) {   // recency unit = 2^shift CALLS instead of a share of the cache's turnover
    const long v = std::strtol(e, nullptr, 10);
    if (v >= 0 && v <= 8) { age_shift_ = (uint32_t)v; call_clock_ = true; insert_age_ = 32; }
  }
  if (!p.cache_admission) admit_log2_ = 0;   // ps.json "gpucache_admission": false
  small_interval_ = (uint32_t)std::max(1, p.small_miss_insert_interval);
  if (const char* e = std::getenv("HPS_LRU_ADMIT")) {
    const long v = std::strtol(e, nullptr, 10);
    if (v >= 0 && v <= 15) admit_log2_ = (uint32_t)v;
  }
  if (const char* e = std::getenv("HPS_LRU_INSERT_AGE")) {
    const long v = std::strtol(e, nullptr, 10);
    if (v >= 0 && v < (long)kAgeSaturate) insert_age_ = (uint32_t)v;
  }

  cfg_.num_emb_table_ = T;
  cfg_.use_gpu_embedding_cache_ = true;
  cfg_.device_id_ = device;
  h_tables_.resize(T);
  {
    // where the host tables stand BEFORE the warm-up reads them: the refresh replays what changed since (HostTable::ChangeMark)
    std::lock_guard<std::mutex> rlk(refresh_mu_);
    seen_epoch_.assign(T, 0);
    seen_prev_.assign(T, 0);
    for (size_t t = 0; t < T; ++t) tables[t]->ChangeMark(&seen_epoch_[t], &seen_prev_[t]);
    seen_last_ = seen_prev_;
  }
  HIP_TRY(hipEventCreateWithFlags(&last_write_, hipEventDisableTiming));

  for (size_t t = 0; t < T; ++t) {
    const uint32_t D = tables[t]->dim();
    size_t R = tables[t]->size();
    if (sharded) {
      // a shard is sized by the rows it owns (counted: hashed owners split a table evenly, a small table they may not)
      const int64_t* kk = tables[t]->keys();
      const size_t chunk = 1u << 18, tasks = (R + chunk - 1) / chunk;
      std::atomic<size_t> mine{0};
      ThreadPool::Global().ParallelFor(tasks, [&](size_t i) {
        size_t c = 0;
        for (size_t r = i * chunk, e = std::min(R, r + chunk); r < e; ++r) c += owned(kk[r]) ? 1 : 0;
        mine.fetch_add(c, std::memory_order_relaxed);
      });
      R = mine.load();
    }
    // capacity = ceil(gpucacheper * rows) (docs/architecture.md:50), at least one bucket
    size_t cap = (size_t)std::ceil((double)p.cache_size_percentage * (double)R);
    if (cap < 1) cap = 1;
    size_t slots = (size_t)std::ceil((double)cap / p.cache_load_factor);
    slots = (slots + kBucketSlots - 1) / kBucketSlots * kBucketSlots;
    const size_t buckets = slots / kBucketSlots;
    if (buckets > (1ull << 27))
      return Error(Code::kUnsupported, "model '", model, "' table ", t, ": cache of ", slots,
                   " slots exceeds the 2^31-slot limit of one table");
    TableCacheDev& tb = h_tables_[t];
    int64_t* dl = nullptr; float* dr = nullptr;
    HPS_RETURN_IF_ERROR(DevAlloc(&dl, buckets * (size_t)kLineWords)); allocations_.push_back(dl);
    HPS_RETURN_IF_ERROR(DevAlloc(&dr, slots * (size_t)D)); allocations_.push_back(dr);
    HIP_TRY(LaunchCacheClear(dl, buckets, kStampFree, nullptr));
    tb.lines = dl; tb.rows = dr;
    tb.num_buckets = (uint32_t)buckets;
    tb.dim = D;
    tb.default_value = p.default_value_for_each_table[t];
    tb.flags = static_ ? 1u : 0u;
    cfg_.embedding_vec_size_.push_back(D);
    cfg_.num_set_in_cache_.push_back(buckets);
    cfg_.capacity_rows_.push_back(cap);
    cfg_.default_value_.push_back(tb.default_value);
  }
  HPS_RETURN_IF_ERROR(DevAlloc(&d_tables_, T));
  HIP_TRY(hipMemcpy(d_tables_, h_tables_.data(), T * sizeof(TableCacheDev), hipMemcpyHostToDevice));
  HIP_TRY(hipDeviceSynchronize());
  if (p.ps_direct_access) {
    for (size_t t = 0; t < T; ++t)
      if (!tables[t]->pinned())
        return Error(Code::kInternal, "ps_direct_access: table ", t, " of model '", model, "' is not in pinned host memory");
    direct_ = true;
    HPS_RETURN_IF_ERROR(SyncDirectIndex(tables));
  }

  // the clock: lookups start in unit 1 (the warm-up below runs in unit 0)
  total_slots_ = 0;
  for (const TableCacheDev& tb : h_tables_) total_slots_ += (uint64_t)tb.num_buckets * kBucketSlots;
  if (total_slots_ == 0) total_slots_ = 1;
  rows_per_unit_ = std::max<uint64_t>(1, total_slots_ / units_per_turnover_);
  call_start_ = (1ull << age_shift_) - 1;
  if (const char* e = std::getenv("HPS_TEST_EPOCH_START")) call_start_ = std::strtoull(e, nullptr, 0);  // test hook: counter wrap (call clock)
  if (!p.init_ec) return Status::Ok();

  // ---- warm-up: the first `capacity` rows of each table in file order ----
  // Static caches are filled here too (flag cleared for the duration of the warm-up).
  std::vector<TableCacheDev> warm = h_tables_;
  for (auto& w : warm) w.flags = 0;
  TableCacheDev* d_warm = nullptr;
  HPS_RETURN_IF_ERROR(DevAlloc(&d_warm, T));
  HIP_TRY(hipMemcpy(d_warm, warm.data(), T * sizeof(TableCacheDev), hipMemcpyHostToDevice));

  const size_t chunk_rows_max = 1u << 20;
  MissDesc* d_md = nullptr; uint64_t* d_zero_ks = nullptr; uint32_t* d_stats = nullptr;
  HPS_RETURN_IF_ERROR(DevAlloc(&d_md, 1));
  HPS_RETURN_IF_ERROR(DevAlloc(&d_zero_ks, (size_t)kMaxTables + 1));
  HPS_RETURN_IF_ERROR(DevAlloc(&d_stats, (size_t)kStatLines * kAccStride));
  HIP_TRY(hipMemset(d_zero_ks, 0, sizeof(uint64_t) * ((size_t)kMaxTables + 1)));
  HIP_TRY(hipMemset(d_stats, 0, (size_t)kStatLines * kAccStride * sizeof(uint32_t)));
  int64_t* d_keys = nullptr; float* d_rows = nullptr;
  size_t maxD = 1;
  for (size_t t = 0; t < T; ++t) maxD = std::max<size_t>(maxD, tables[t]->dim());
  size_t chunk_rows = std::min(chunk_rows_max, std::max<size_t>(1, kStagingCapBytes / (maxD * sizeof(float))));
  HPS_RETURN_IF_ERROR(DevAlloc(&d_keys, chunk_rows));
  HPS_RETURN_IF_ERROR(DevAlloc(&d_rows, chunk_rows * maxD));
  std::vector<int64_t> hk;
  std::vector<float> hr;
  const uint32_t epoch = 0;   // unit 0; lookups start in unit 1 (below), so that the warm rows are evictable from the first call
  Status st = Status::Ok();
  for (size_t t = 0; t < T && st.ok(); ++t) {
    const HostTable& ht = *tables[t];
    const uint32_t D = ht.dim();
    // a replica takes the first `capacity` rows of the file; a shard the first `capacity` rows it OWNS, wherever they are
    const size_t want = sharded ? cfg_.capacity_rows_[t] : std::min(cfg_.capacity_rows_[t], ht.size());
    const size_t scan_end = sharded ? ht.size() : want;
    size_t taken = 0;
    std::vector<size_t> pick;
    for (size_t r0 = 0; r0 < scan_end && taken < want && st.ok(); r0 += chunk_rows) {
      const size_t n = std::min(chunk_rows, scan_end - r0);
      // canonical rows only: a key repeated in the file is represented by its last row
      hk.clear(); hr.clear();
      const int64_t* src_keys = ht.keys() + r0;
      const float* src_rows = ht.row_at(r0);
      const bool contiguous = !ht.has_duplicate_keys() && !sharded;
      size_t m = n;
      if (!contiguous) {
        const bool dups = ht.has_duplicate_keys();
        pick.clear();
        for (size_t i = 0; i < n && taken + pick.size() < want; ++i) {
          if (!owned(src_keys[i])) continue;
          if (dups && ht.Find(src_keys[i]) != (int64_t)(r0 + i)) continue;
          pick.push_back(i);
        }
        m = pick.size();
        hk.resize(m); hr.resize(m * (size_t)D);
        const size_t grain = 4096, tasks = (m + grain - 1) / grain;
        ThreadPool::Global().ParallelFor(tasks, [&](size_t ti) {
          for (size_t j = ti * grain, e = std::min(m, j + grain); j < e; ++j) {
            hk[j] = src_keys[pick[j]];
            memcpy(hr.data() + j * (size_t)D, ht.row_at(r0 + pick[j]), (size_t)D * sizeof(float));
          }
        });
        src_keys = hk.data(); src_rows = hr.data();
      }
      taken += m;
      if (m == 0) continue;
      if (hipMemcpy(d_keys, src_keys, m * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(d_rows, src_rows, m * (size_t)D * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        st = Error(Code::kInternal, "cache warm-up: H2D copy failed");
        break;
      }
      MissDesc md;
      memset(&md, 0, sizeof md);
      for (size_t u = 0; u <= T; ++u) md.useg_start[u] = u > t ? m : 0;
      md.chunk_lo[t] = 0; md.chunk_hi[t] = (uint32_t)m; md.stage_off[t] = 0;
      if (hipMemcpy(d_md, &md, sizeof md, hipMemcpyHostToDevice) != hipSuccess) { st = Error(Code::kInternal, "cache warm-up: H2D copy failed"); break; }
      const hipError_t e = LaunchCacheInsert(d_warm, (uint32_t)T, d_md, m, d_zero_ks, d_keys, d_rows, nullptr, Stamp8(epoch) * 0x101u,   // warm-up rows enter as if they had just been hit
                                             
                                             d_stats, cu_count_, nullptr);
      if (e != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        st = Error(Code::kInternal, "cache warm-up: insert kernel failed: ", hipGetErrorString(e));
        break;
      }
    }
  }
  std::vector<uint32_t> lines((size_t)kStatLines * kAccStride, 0u);
  (void)hipMemcpy(lines.data(), d_stats, lines.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
  AddStatLines(lines.data());
  (void)hipFree(d_keys); (void)hipFree(d_rows); (void)hipFree(d_md); (void)hipFree(d_zero_ks); (void)hipFree(d_stats);
  (void)hipFree(d_warm);
  return st;
}

Status EmbeddingCache::SyncDirectIndex(const std::vector<std::shared_ptr<HostTable>>& tables) {
  if (!direct_) return Status::Ok();
  const size_t T = tables.size();
  if (T != num_tables()) return Error(Code::kInvalidArg, "SyncDirectIndex: table count mismatch");
  HIP_TRY(hipSetDevice(cfg_.device_id_));
  if (h_index_.size() != T) {
    h_index_.assign(T, PsIndexDev{});
    index_generation_.assign(T, ~0ull);
    index_mem_.assign(T, {nullptr, nullptr});
  }
  uint32_t* d_sent = nullptr;
  HPS_RETURN_IF_ERROR(DevAlloc(&d_sent, 2));
  bool changed = false;
  for (size_t t = 0; t < T; ++t) {
    const HostTable& ht = *tables[t];
    if (!ht.pinned()) { (void)hipFree(d_sent); return Error(Code::kInternal, "ps_direct_access: table ", t, " is not pinned"); }
    if (index_generation_[t] == ht.generation()) continue;
    changed = true;
    HIP_TRY(hipDeviceSynchronize());
    if (index_mem_[t].first) (void)hipFree(index_mem_[t].first);
    if (index_mem_[t].second) (void)hipFree(index_mem_[t].second);
    index_mem_[t] = {nullptr, nullptr};
    const uint64_t R = ht.size();
    if (R >= (1ull << 32)) {
      (void)hipFree(d_sent);
      return Error(Code::kUnsupported, "ps_direct_access: table ", t, " has ", R, " rows; the device index holds 32-bit row numbers");
    }
    uint64_t cap = 16;
    while (cap < 2 * R) cap <<= 1;
    int64_t* dk = nullptr; uint32_t* dr = nullptr;
    HPS_RETURN_IF_ERROR(DevAlloc(&dk, cap));
    HPS_RETURN_IF_ERROR(DevAlloc(&dr, cap));
    index_mem_[t] = {dk, dr};
    void* keys_dev = nullptr; void* rows_dev = nullptr;
    if (R) {
      HIP_TRY(hipHostGetDevicePointer(&keys_dev, (void*)ht.keys(), 0));
      HIP_TRY(hipHostGetDevicePointer(&rows_dev, (void*)ht.row_at(0), 0));
    }
    HIP_TRY(hipMemset(d_sent, 0, 2 * sizeof(uint32_t)));
    HIP_TRY(LaunchPsIndexBuild((const int64_t*)keys_dev, R, dk, dr, cap, d_sent, nullptr));
    uint32_t sent[2] = {0, 0};
    HIP_TRY(hipMemcpy(sent, d_sent, sizeof sent, hipMemcpyDeviceToHost));
    PsIndexDev& ix = h_index_[t];
    ix.keys = dk; ix.rows = dr; ix.mask = cap - 1;
    ix.host_rows = (const float*)rows_dev;
    ix.dim = ht.dim();
    ix.has_sentinel = sent[0];
    ix.sentinel_row = sent[1];
    ix.default_value = cfg_.default_value_[t];
    index_generation_[t] = ht.generation();
  }
  (void)hipFree(d_sent);
  if (!d_index_) HPS_RETURN_IF_ERROR(DevAlloc(&d_index_, T));
  if (changed) HIP_TRY(hipMemcpy(d_index_, h_index_.data(), T * sizeof(PsIndexDev), hipMemcpyHostToDevice));
  HIP_TRY(hipDeviceSynchronize());
  return Status::Ok();
}

// ---- background inserter of the device-driven tier (async-insert mode) --------------------------------------
struct EmbeddingCache::DirectInserter {
  hipStream_t stream = nullptr;
  hipEvent_t ev_copied = nullptr, ev_done = nullptr, ev_fetch = nullptr;
  int64_t* d_keys = nullptr;      // snapshot of the session's unique-key array (table-major, key_start offsets)
  uint64_t* d_key_start = nullptr;
  uint32_t* d_acc = nullptr;      // accumulator block of the job (device_types.h): stat lines, then the T table lines
  MissDesc* d_md = nullptr;
  float* d_staging = nullptr;
  uint8_t* d_found = nullptr;
  size_t cap_keys = 0, cap_floats = 0, cap_found = 0;
  uint64_t unique_total = 0;
  bool in_flight = false;   // one job at a time: set by SubmitDirectInsert, cleared when FinishDirectInsert returns
};

void EmbeddingCache::FreeDirectInserter() {
  if (!dins_) return;
  DirectInserter& I = *dins_;
  if (I.stream) { (void)hipStreamSynchronize(I.stream); (void)hipStreamDestroy(I.stream); }
  for (hipEvent_t e : {I.ev_copied, I.ev_done, I.ev_fetch}) if (e) (void)hipEventDestroy(e);
  for (void* p : {(void*)I.d_keys, (void*)I.d_key_start, (void*)I.d_acc, (void*)I.d_md, (void*)I.d_staging, (void*)I.d_found})
    if (p) (void)hipFree(p);
  delete dins_;
  dins_ = nullptr;
}

// Part 1, on the calling lookup's thread: claim the (single) job slot and snapshot the session's unique missed keys
// with copies enqueued on the session's stream (ordered after its dedup kernels, before its next call reuses them).
Status EmbeddingCache::SubmitDirectInsert(hipStream_t session_stream, const uint64_t* d_key_start, const int64_t* d_uniq_keys,
                                          const uint32_t* d_acc_tables, const uint32_t* h_acc_tables_override, uint64_t N,
                                          uint64_t unique_total, uint64_t staging_floats, bool* accepted) {
  *accepted = false;
  if (!direct_ || static_ || unique_total == 0) return Status::Ok();
  std::unique_lock<std::mutex> lk(dins_mu_, std::try_to_lock);
  if (!lk.owns_lock()) return Status::Ok();  // another session is submitting: this batch's misses stay uncached
  HIP_TRY(hipSetDevice(cfg_.device_id_));
  const size_t T = num_tables();
  if (!dins_) {
    dins_ = new DirectInserter();
    HIP_TRY(hipStreamCreateWithFlags(&dins_->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&dins_->ev_copied, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&dins_->ev_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&dins_->ev_fetch, hipEventDisableTiming));
    HPS_RETURN_IF_ERROR(DevAlloc(&dins_->d_key_start, (size_t)kMaxTables + 1));
    HPS_RETURN_IF_ERROR(DevAlloc(&dins_->d_acc, (size_t)kAccWordsMax));
    HPS_RETURN_IF_ERROR(DevAlloc(&dins_->d_md, 1));
  }
  DirectInserter& I = *dins_;
  if (I.in_flight) return Status::Ok();  // saturated: drop (best effort, like the bounded host inserter)
  auto grow = [](auto** p, size_t* cap, size_t want) -> Status {
    if (want <= *cap) return Status::Ok();
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t n = want + want / 4;
    HPS_RETURN_IF_ERROR(DevAlloc(p, n));
    *cap = n;
    return Status::Ok();
  };
  HPS_RETURN_IF_ERROR(grow(&I.d_keys, &I.cap_keys, (size_t)N));
  HPS_RETURN_IF_ERROR(grow(&I.d_staging, &I.cap_floats, (size_t)staging_floats + 4 * T));
  HPS_RETURN_IF_ERROR(grow(&I.d_found, &I.cap_found, (size_t)unique_total));
  HIP_TRY(hipMemcpyAsync(I.d_keys, d_uniq_keys, N * sizeof(int64_t), hipMemcpyDeviceToDevice, session_stream));
  HIP_TRY(hipMemcpyAsync(I.d_key_start, d_key_start, (T + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, session_stream));
  uint32_t* dst_lines = I.d_acc + (size_t)kStatLines * kAccStride;
  const size_t line_bytes = T * (size_t)kAccStride * sizeof(uint32_t);
  if (h_acc_tables_override)  // mixed call: only the async tables' counts (pinned host words, stable until the call's final sync)
    HIP_TRY(hipMemcpyAsync(dst_lines, h_acc_tables_override, line_bytes, hipMemcpyHostToDevice, session_stream));
  else
    HIP_TRY(hipMemcpyAsync(dst_lines, d_acc_tables, line_bytes, hipMemcpyDeviceToDevice, session_stream));
  HIP_TRY(hipEventRecord(I.ev_copied, session_stream));
  I.unique_total = unique_total;
  I.in_flight = true;
  *accepted = true;
  return Status::Ok();
}

// Part 2, on a pool thread (no CPU work, only enqueueing): fetch on the inserter's stream, and only when that has
// drained the insert kernel, so that the writer event other sessions' probes wait for covers the insert alone.
Status EmbeddingCache::FinishDirectInsert() {
  DirectInserter& I = *dins_;
  struct Done {
    EmbeddingCache* c;
    ~Done() { std::lock_guard<std::mutex> lk(c->dins_mu_); c->dins_->in_flight = false; }
  } done{this};
  // the fetch kernel reads the pinned host tables: same fence against table reloads as a lookup
  std::shared_lock<std::shared_mutex> tables_lock(direct_mu_);
  HIP_TRY(hipSetDevice(cfg_.device_id_));
  const size_t T = num_tables();
  HIP_TRY(hipStreamWaitEvent(I.stream, I.ev_copied, 0));
  const uint32_t epoch = NextEpoch();
  hipError_t e = LaunchMissDescBuild(d_tables_, (uint32_t)T, I.d_acc, I.d_md, /*clear_stats=*/true, nullptr, I.stream);
  // not part of the foreground fetch chain: a small grid that takes its time must not hold up a lookup's fetch
  if (e == hipSuccess)
    e = LaunchPsFetchDirect(d_index_, (uint32_t)T, I.d_md, I.d_key_start, I.d_keys, I.d_staging, I.d_found, I.unique_total,
                            /*grid_blocks=*/8, I.stream);
  if (e != hipSuccess) return Error(Code::kInternal, "direct background fetch launch failed: ", hipGetErrorString(e));
  HIP_TRY(hipStreamSynchronize(I.stream));
  BeginWrite(I.stream);
  e = LaunchCacheInsert(d_tables_, (uint32_t)T, I.d_md, I.unique_total, I.d_key_start, I.d_keys, I.d_staging, I.d_found, InsertStamps(epoch),
                        I.d_acc, cu_count_, I.stream);
  EndWrite(I.stream);
  if (e != hipSuccess) return Error(Code::kInternal, "direct background insert launch failed: ", hipGetErrorString(e));
  std::vector<uint32_t> lines((size_t)kStatLines * kAccStride, 0u);
  HIP_TRY(hipMemcpyAsync(lines.data(), I.d_acc, lines.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, I.stream));
  HIP_TRY(hipStreamSynchronize(I.stream));
  AddStatLines(lines.data());
  return Status::Ok();
}

Status EmbeddingCache::Query(uint32_t table, const int64_t* h_keys, size_t n, int32_t* h_slots) {
  if (table >= num_tables()) return Error(Code::kInvalidArg, "table index out of range");
  if (n == 0) return Status::Ok();
  HIP_TRY(hipSetDevice(cfg_.device_id_));
  int64_t* dk = nullptr; int32_t* ds = nullptr;
  HPS_RETURN_IF_ERROR(DevAlloc(&dk, n));
  HPS_RETURN_IF_ERROR(DevAlloc(&ds, n));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(dk, h_keys, n * sizeof(int64_t), hipMemcpyHostToDevice));
  HIP_TRY(LaunchCacheQuery(h_tables_[table], dk, n, ds, nullptr));
  HIP_TRY(hipMemcpy(h_slots, ds, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  (void)hipFree(dk); (void)hipFree(ds);
  return Status::Ok();
}

Status EmbeddingCache::DumpKeys(uint32_t table, std::vector<int64_t>* keys) {
  if (table >= num_tables()) return Error(Code::kInvalidArg, "table index out of range");
  HIP_TRY(hipSetDevice(cfg_.device_id_));
  const size_t words = (size_t)h_tables_[table].num_buckets * kLineWords;
  std::vector<int64_t> all(words);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(all.data(), h_tables_[table].lines, words * sizeof(int64_t), hipMemcpyDeviceToHost));
  keys->clear();
  for (size_t i = 0; i < words; ++i)
    if (i % kLineWords < (size_t)kBucketSlots && all[i] != HPS_EMPTY_KEY) keys->push_back(all[i]);
  return Status::Ok();
}

// =================================================================================================
// LookupSession
// =================================================================================================
namespace {
constexpr int kMaxSideDevices = 64;
std::atomic<int> g_side_hi[kMaxSideDevices];   // sessions with a high-priority side stream, per device (at most two)
}  // namespace

LookupSession::~LookupSession() { Release(); }

void LookupSession::Release() {
  if (d_interact_emb_) { (void)hipSetDevice(device_); (void)hipFree(d_interact_emb_); d_interact_emb_ = nullptr; interact_emb_floats_ = 0; }
  if (!cache_) return;
  (void)hipSetDevice(device_);
  cache_->UnregisterSession(this);
  (void)CollectDeferred();
  if (stream_) (void)hipStreamSynchronize(stream_);
  if (copy_stream_) (void)hipStreamSynchronize(copy_stream_);
  cache_->ForgetReader(ev_read_);  // our reader event may still be registered with the cache
  cache_->ForgetReader(ev_probe_);
  cache_->ForgetFetch(ev_fetch_);
  for (hipEvent_t e : ev_lane_) if (e) { cache_->ForgetLane(e); (void)hipEventDestroy(e); }
  auto hfree = [](void* p) { if (p) (void)hipHostFree(p); };
  auto dfree = [](void* p) { if (p) (void)hipFree(p); };
  hfree(h_keys_pinned_); dfree(d_keys_); hfree(h_block_); dfree(d_block_); hfree(h_acc_); dfree(d_mode_);
  hfree(h_md_); dfree(d_md_);
  dfree(work_.slot); dfree(work_.tile_cnt); dfree(work_.miss_key); dfree(work_.sent_i); dfree(work_.sent_m);
  dfree(work_.rep_of); dfree(work_.uidx_of); dfree(work_.set); dfree(work_.uniq_keys);
  hfree(h_mode_);
  hfree(h_uniq_keys_); hfree(h_staging_); dfree(d_staging_); hfree(h_found_); dfree(d_found_);
  for (hipEvent_t e : {ev_keys_, ev_done_, ev_done2_, ev_read_, ev_fetch_, ev_t0_, ev_t1_, ev_f0_, ev_f1_, ev_c1_, ev_probe_, ev_copy_,
                       ev_g0_, ev_g1_, ev_s0_, ev_s1_, ev_i0_, ev_i1_})
    if (e) (void)hipEventDestroy(e);
  if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
  if (side_hi_ && device_ >= 0 && device_ < kMaxSideDevices) g_side_hi[device_].fetch_sub(1, std::memory_order_relaxed);
  side_hi_ = false;
  if (stream_) (void)hipStreamDestroy(stream_);
  stream_ = nullptr;
  cache_.reset();
}

Status LookupSession::Init(HierParameterServer* ps, const InferenceParams& p, std::shared_ptr<EmbeddingCache> cache, size_t max_keys_override) {
  ps_ = ps;
  params_ = p;
  tables_ = ps->tables_of(p.model_name);
  const size_t T = tables_.size();
  if (T == 0) return Error(Code::kNotFound, "model '", p.model_name, "' has no tables loaded in the parameter server");
  if (T > (size_t)kMaxTables)
    return Error(Code::kInvalidArg, "model '", p.model_name, "': ", T, " tables (supported: 1..", kMaxTables, ")");
  size_t per_sample = 0;
  for (size_t c : p.maxnum_catfeature_query_per_table_per_sample) per_sample += c;
  max_keys_ = p.max_batchsize * per_sample;  // model_instance_state.cpp:98-99
  if (max_keys_override) max_keys_ = max_keys_override;   // a shard session of a table-sharded model's entry instance (shard_entry.h)
  if (max_keys_ == 0) return Error(Code::kInvalidArg, "model '", p.model_name, "': max_batch_size * sum(maxnum_catfeature...) is 0");
  max_tiles_ = std::max(max_keys_ / kTileKeys, std::min<size_t>(max_keys_, kSmallRequestKeys) / kSmallTileKeys) + T;
  if (max_tiles_ * (size_t)kTileKeys >= (1ull << 31) - 2)
    return Error(Code::kUnsupported, "more than 2^31 keys per request are not supported");
  if (!p.use_gpu_embedding_cache) return Status::Ok();  // host-tier session: no device state at all

  if (!cache) return Error(Code::kInvalidArg, "model '", p.model_name, "' uses the GPU cache but no EmbeddingCache was given");
  cache_ = std::move(cache);
  device_ = cache_->device();
  HIP_TRY(hipSetDevice(device_));
  HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  {
    // The second stream carries the small things a call's host side waits for while the first stream's 230-us hit gather
    // owns every CU: the miss counts' push, the descriptor pull, the side scatter.  It is a HIGH-PRIORITY queue: its workgroups
    // are placed before the gather's remaining ones — the counts reach the host 50 us earlier (0.23-0.25 against 0.28-0.31 ms),
    // headline 2.03 / 2.05 / 2.08 against 2.01 / 1.69 / 1.92 G lookups/s in three interleaved pairs (the two low ones look like
    // the box's host noise: 2.01 is the fair comparison), p50 1.55-1.58 against 1.61 ms
    // (profiles/round4/ab_side_stream_priority.txt).
    // Round 5: NOT for more than two sessions per device, and never for the shard sessions of a table-sharded model's entry
    // instances.  A high-priority queue that is merely WAITING (its head is a barrier on another queue's event) makes the
    // hardware scheduler preempt the waves of the normal-priority queue it waits for: with 16 sessions on one GPU (4 entry
    // instances x 4 shards) kernels that had started stood still for 60-160 ms with the GPU idle until the scheduler's quantum
    // expired — 35 ms per round of requests against 4 ms with normal-priority side streams
    // (profiles/round5/ab_side_priority_many_sessions.txt; rocprofv3: hps_pull16_kernel "running" for 283 ms).
    int least = 0, greatest = 0;
    side_hi_ = false;
    if (!max_keys_override && device_ >= 0 && device_ < kMaxSideDevices &&
        hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least) {
      if (g_side_hi[device_].fetch_add(1, std::memory_order_relaxed) < 2) side_hi_ = true;
      else g_side_hi[device_].fetch_sub(1, std::memory_order_relaxed);
    }
    if (side_hi_) HIP_TRY(hipStreamCreateWithPriority(&copy_stream_, hipStreamNonBlocking, greatest));
    else HIP_TRY(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
  }
  for (hipEvent_t* e : {&ev_copy_, &ev_keys_, &ev_done_, &ev_done2_, &ev_read_, &ev_fetch_, &ev_probe_})
    HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
  for (hipEvent_t* e : {&ev_t0_, &ev_t1_, &ev_f0_, &ev_f1_, &ev_c1_, &ev_g0_, &ev_g1_, &ev_s0_, &ev_s1_, &ev_i0_, &ev_i1_}) HIP_TRY(hipEventCreate(e));
  for (hipEvent_t& e : ev_lane_) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));

  HPS_RETURN_IF_ERROR(PinAlloc(&h_keys_pinned_, max_keys_, hipHostMallocMapped));
  {
    void* dv = nullptr;
    if (hipHostGetDevicePointer(&dv, h_keys_pinned_, 0) == hipSuccess) h_keys_dev_ = (const int64_t*)dv;
    else (void)hipGetLastError();
  }
  HPS_RETURN_IF_ERROR(DevAlloc(&d_keys_, max_keys_));
  // call block: descriptor | accumulator block | tile descriptors
  acc_words_ = (size_t)(kStatLines + T) * kAccStride;
  block_acc_off_ = (sizeof(CallDesc) + 127) & ~(size_t)127;
  block_tiles_off_ = block_acc_off_ + acc_words_ * sizeof(uint32_t);
  const size_t block_bytes = block_tiles_off_ + max_tiles_ * sizeof(TileDesc);
  HPS_RETURN_IF_ERROR(PinAlloc(&h_block_, block_bytes, hipHostMallocMapped));
  HPS_RETURN_IF_ERROR(DevAlloc(&d_block_, block_bytes));
  memset(h_block_, 0, block_bytes);
  h_call_ = reinterpret_cast<CallDesc*>(h_block_);
  d_call_ = reinterpret_cast<CallDesc*>(d_block_);
  h_tiles_ = reinterpret_cast<TileDesc*>(h_block_ + block_tiles_off_);
  d_acc_ = reinterpret_cast<uint32_t*>(d_block_ + block_acc_off_);
  HPS_RETURN_IF_ERROR(PinAlloc(&h_acc_, acc_words_ + kAccStride, hipHostMallocMapped));
  memset(h_acc_, 0, (acc_words_ + kAccStride) * sizeof(uint32_t));
  h_seq_ = h_acc_ + acc_words_;
  {
    void* dv = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dv, h_block_, 0));
    h_block_dev_ = dv;
    HIP_TRY(hipHostGetDevicePointer(&dv, h_acc_, 0));
    h_acc_dev_ = (uint32_t*)dv;
    h_seq_dev_ = h_acc_dev_ + acc_words_;
    if (const char* e = std::getenv("HPS_ZC_CONTROL")) zc_control_ = e[0] != '0';
  }
  HPS_RETURN_IF_ERROR(DevAlloc(&d_mode_, (size_t)kMaxTables));
  HPS_RETURN_IF_ERROR(PinAlloc(&h_mode_, (size_t)kMaxTables + (size_t)kMaxTables * kAccStride));
  HPS_RETURN_IF_ERROR(PinAlloc(&h_md_, 2, hipHostMallocMapped));   // (2: the pull kernel moves whole 16-B units)
  HPS_RETURN_IF_ERROR(DevAlloc(&d_md_, 2));
  {
    void* dv = nullptr;
    h_md_dev_ = hipHostGetDevicePointer(&dv, h_md_, 0) == hipSuccess ? (const MissDesc*)dv : nullptr;
  }

  const size_t regions = max_tiles_ * (size_t)kTileKeys;
  CallWork& w = work_;
  w.tiles = reinterpret_cast<const TileDesc*>(d_block_ + block_tiles_off_);
  w.acc = d_acc_;
  HPS_RETURN_IF_ERROR(DevAlloc(&w.slot, max_keys_));
  HPS_RETURN_IF_ERROR(DevAlloc(&w.tile_cnt, max_tiles_ * 4));
  HPS_RETURN_IF_ERROR(DevAlloc(&w.miss_key, regions));
  HPS_RETURN_IF_ERROR(DevAlloc(&w.sent_i, regions));
  HPS_RETURN_IF_ERROR(DevAlloc(&w.sent_m, regions));
  HPS_RETURN_IF_ERROR(DevAlloc(&w.rep_of, regions));
  HPS_RETURN_IF_ERROR(DevAlloc(&w.uidx_of, regions));
  uint64_t set_cap = 1024;
  while (set_cap < 2 * (uint64_t)max_keys_) set_cap <<= 1;
  HPS_RETURN_IF_ERROR(DevAlloc(&w.set, set_cap));
  HIP_TRY(hipMemset(w.set, 0, set_cap * sizeof(unsigned long long)));  // tag 0 is never used by a call
  w.set_mask = set_cap - 1;
  HPS_RETURN_IF_ERROR(DevAlloc(&w.uniq_keys, max_keys_));
  HPS_RETURN_IF_ERROR(PinAlloc(&h_uniq_keys_, max_keys_, hipHostMallocMapped));
  void* dv = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&dv, h_uniq_keys_, 0));
  w.uniq_keys_host = (int64_t*)dv;
  w.uniq_keys_host32 = nullptr;   // set per call (LookupDevice)
  uniq_miss_.assign(T, 0);
  if (cache_->direct()) {
    // the device sizes the staging layout itself (hps_missdesc_build), so the buffer must hold the worst case:
    // every key of a full batch missing.  Device memory only — no pinned host staging in this mode.
    size_t worst = 4 * T;
    for (size_t t = 0; t < T; ++t)
      worst += p.max_batchsize * p.maxnum_catfeature_query_per_table_per_sample[t] * (size_t)tables_[t]->dim();
    if (max_keys_override) {   // any mix of tables up to max_keys_ keys
      size_t maxD = 1;
      for (size_t t = 0; t < T; ++t) maxD = std::max<size_t>(maxD, tables_[t]->dim());
      worst = 4 * T + max_keys_ * maxD;
    }
    HPS_RETURN_IF_ERROR(DevAlloc(&d_staging_, worst));
    HPS_RETURN_IF_ERROR(DevAlloc(&d_found_, max_keys_));
    staging_floats_ = worst;
    staging_uniq_ = max_keys_;
  } else {
    // host-gather tier: page-locked staging for an eighth of a full request up front (a request that misses more grows
    // it): the first calls of a fresh session then do not stop for a pinned allocation in the middle of a lookup
    size_t floats = 0;
    for (size_t t = 0; t < T; ++t)
      floats += p.max_batchsize * p.maxnum_catfeature_query_per_table_per_sample[t] * (size_t)tables_[t]->dim();
    if (max_keys_override) {
      size_t maxD = 1;
      for (size_t t = 0; t < T; ++t) maxD = std::max<size_t>(maxD, tables_[t]->dim());
      floats = max_keys_ * maxD;
    }
    floats = std::min(floats / 8 + 4 * T, kStagingCapBytes / sizeof(float));
    HPS_RETURN_IF_ERROR(EnsureStaging(floats, max_keys_ / 8 + 1));
  }
  HIP_TRY(hipDeviceSynchronize());
  cache_->RegisterSession(this);
  return Status::Ok();
}

Status LookupSession::EnsureStaging(size_t floats, size_t uniq) {
  if (floats > staging_floats_ || (!h_staging_ && floats > 0)) {
    HIP_TRY(hipStreamSynchronize(stream_));
    if (h_staging_) (void)hipHostFree(h_staging_);
    h_staging_ = nullptr;
    size_t want = std::max(floats, staging_floats_);
    if (floats > staging_floats_) {
      if (d_staging_) (void)hipFree(d_staging_);
      d_staging_ = nullptr;
      // half as much again as asked for: the miss count of a steady workload wobbles by a few per cent from call to
      // call, and every growth is a page-locked allocation (tens of milliseconds in the middle of a lookup)
      want = std::max(floats + floats / 2, staging_floats_ * 2);
      want = std::min(want, std::max(floats, kStagingCapBytes / sizeof(float)));
      want = std::max<size_t>(want, 1u << 16);
      HPS_RETURN_IF_ERROR(DevAlloc(&d_staging_, want));
      staging_floats_ = want;
    }
    HPS_RETURN_IF_ERROR(PinAlloc(&h_staging_, staging_floats_, hipHostMallocMapped));
    void* dv = nullptr;
    h_staging_dev_ = hipHostGetDevicePointer(&dv, h_staging_, 0) == hipSuccess ? (const float*)dv : nullptr;
  }
  if (uniq > staging_uniq_ || (!h_found_ && uniq > 0)) {
    HIP_TRY(hipStreamSynchronize(stream_));
    if (h_found_) (void)hipHostFree(h_found_);
    h_found_ = nullptr;
    if (uniq > staging_uniq_) {
      if (d_found_) (void)hipFree(d_found_);
      d_found_ = nullptr;
      size_t want = std::max(uniq + uniq / 2, staging_uniq_ * 2);
      want = std::max<size_t>(want, 1u << 12);
      HPS_RETURN_IF_ERROR(DevAlloc(&d_found_, want));
      staging_uniq_ = want;
    }
    HPS_RETURN_IF_ERROR(PinAlloc(&h_found_, staging_uniq_, hipHostMallocMapped));
    void* dv = nullptr;
    h_found_dev_ = hipHostGetDevicePointer(&dv, h_found_, 0) == hipSuccess ? (const uint8_t*)dv : nullptr;
  }
  return Status::Ok();
}

// The reference's contract: host key pointers in (docs/architecture.md:308-323).  The reference shell memcpy's the
// request's keys into its (unpinned) key buffer on one thread and the engine copies that to the device
// (hps.cc:586-597, hps_buffer.hpp:114-123).  Here:
//   * keys that already sit in page-locked host memory as one flat table-major array (what ProcessRequest slices:
//     model_instance_state.cpp:180-193; Triton hands GPU-instance backends pinned input buffers) are DMA'd from where
//     they are — no host copy at all;
//   * pageable keys are staged into the session's pinned buffer in pieces of 1 MB by the serving pool, and each
//     piece's H2D copy is enqueued by the thread that staged it, so staging and DMA overlap (13.6 MB on one thread
//     is 1.2 ms of memcpy — longer than the whole lookup).
Status LookupSession::lookup(const void* const* h_keys_per_table, float* const* vectors_per_table,
                             const size_t* num_keys_per_table, size_t num_tables) {
  if (num_tables != tables_.size())
    return Error(Code::kInvalidArg, "lookup: got ", num_tables, " tables, model '", params_.model_name, "' has ",
                 tables_.size());
  if (!cache_) return LookupHostTier(h_keys_per_table, vectors_per_table, num_keys_per_table, num_tables);
  size_t N = 0;
  for (size_t t = 0; t < num_tables; ++t) N += num_keys_per_table[t];
  if (N > max_keys_)
    return Error(Code::kInvalidArg, "lookup: ", N, " keys exceed the session capacity of ", max_keys_,
                 " (max_batch_size x sum(maxnum_catfeature_query_per_table_per_sample))");
  if (N == 0) return Status::Ok();
  HIP_TRY(hipSetDevice(device_));
  const auto tk0 = std::chrono::steady_clock::now();
  bool flat = true;
  const int64_t* base = nullptr;
  {
    const int64_t* expect = nullptr;
    for (size_t t = 0; t < num_tables; ++t) {
      const size_t n = num_keys_per_table[t];
      if (n == 0) continue;
      const int64_t* p = (const int64_t*)h_keys_per_table[t];
      if (!p) return Error(Code::kInvalidArg, "lookup: null key pointer for table ", t);
      if (!base) base = p;
      else if (p != expect) flat = false;
      expect = p + n;
    }
  }
  bool direct_dma = false;
  if (flat && keys_pinned_hint_ != 0) {
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof attr);
    if (hipPointerGetAttributes(&attr, base) == hipSuccess && attr.type == hipMemoryTypeHost) direct_dma = true;
    else (void)hipGetLastError();  // an unregistered pointer is not an error of ours
  }
  keys_narrow_ = false;
  key_bytes_ = 8;
  stage_pool_ms_ = stage_enqueue_ms_ = 0.f;
  if (zc_control_ && N <= kSmallRequestKeys && h_keys_dev_) {
    // Small request (at most 1 MB of keys): the probe kernel reads the keys out of page-locked host memory itself — the
    // caller's buffer when that is page-locked, else the session's staging buffer after one memcpy — instead of waiting for
    // an SDMA copy and the hand-off between the queues (tools/ab_zc_control.sh).
    const int64_t* dev_view = nullptr;
    if (direct_dma) {
      void* dv = nullptr;
      if (hipHostGetDevicePointer(&dv, const_cast<int64_t*>(base), 0) == hipSuccess) dev_view = (const int64_t*)dv;
      else (void)hipGetLastError();
    }
    if (!dev_view) {
      size_t off = 0;
      for (size_t t = 0; t < num_tables; ++t) {
        const size_t n = num_keys_per_table[t];
        if (n) memcpy(h_keys_pinned_ + off, h_keys_per_table[t], n * sizeof(int64_t));
        off += n;
      }
      dev_view = h_keys_dev_;
    }
    key_stage_ms_ = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tk0).count();
    return TimedLookupDevice(dev_view, vectors_per_table, num_keys_per_table, num_tables);
  }
  {
    constexpr size_t kTaskKeys = 32768, kGroupKeys = (4u << 20) / sizeof(int64_t);
    struct Task { const int64_t* src; size_t off, n; uint64_t base; };
    std::vector<Task> tasks;
    size_t off = 0;
    // frame of reference of a narrowed request: every table's keys as offsets from the table's smallest key (key_pack.h)
    key_base_.assign(num_tables, 0);
    for (size_t t = 0; t < num_tables; ++t) {
      const size_t n = num_keys_per_table[t];
      const int64_t* p = (const int64_t*)h_keys_per_table[t];
      if (frame_of_reference_) key_base_[t] = tables_[t]->min_key();
      for (size_t b = 0; b < n; b += kTaskKeys) tasks.push_back({p + b, off + b, std::min(kTaskKeys, n - b), (uint64_t)key_base_[t]});
      off += n;
    }
    // (page-locked keys are DMA'd in place, never narrowed: host threads read that memory an order of magnitude slower
    //  than ordinary memory on the MI355X boxes — 0.41 against 2.66 G lookups/s in bench.py's pinned-keys legs)
    bool try_narrow = !direct_dma && narrow_keys_ && narrow_backoff_ == 0 && N >= 4 * kTaskKeys;
    if (narrow_backoff_ > 0) --narrow_backoff_;
    bool try_pack24 = try_narrow && pack24_keys_ && narrow24_backoff_ == 0;
    if (narrow24_backoff_ > 0) --narrow24_backoff_;
    if (try_narrow) {
      // A look at a few keys of every task before any copying: traffic whose keys are wide (hashed 64-bit ids, negative
      // keys) shows it in the first sample, and the call goes straight to the width that can work instead of paying for a
      // failed 4-MB group, a stream synchronisation and a restage (the optimistic check while copying still catches the
      // request with one wide key among narrow ones).
      uint64_t sample = 0;
      for (const Task& tk : tasks)
        sample |= ((uint64_t)tk.src[0] - tk.base) | ((uint64_t)tk.src[tk.n / 2] - tk.base) | ((uint64_t)tk.src[tk.n - 1] - tk.base);
      if (sample >> 32) { try_narrow = try_pack24 = false; NarrowFailed(false); }
      else if (sample >> 24) { if (try_pack24) NarrowFailed(true); try_pack24 = false; }
    }
    // stage(width): the keys leave at `width` bytes each — 8 as they are, 4 as uint32, 3 packed little-endian.  A narrower
    // width is optimistic: every task ORs its keys together while it copies, and the first group that saw a key too wide
    // ends the attempt (the caller restages at the next width; `seen` tells it which one can work).
    // The keys go up on the session's SECOND stream: the first one may still hold the previous call's insert kernel (left
    // running behind that call, waiting for the other session's hit gather to release the cache), and nothing of this
    // upload depends on it.  The probe waits for the event behind the last piece.
    hipStream_t ks = copy_stream_;
    // ... by copy-engine copies, or read out of the staging buffer by a kernel (hps_pull_bytes).  While the session's calls miss
    // much, the link is full of the OTHER session's 4-MB row copies, and a key copy queues behind them: the probe started 0.3 ms
    // late, the two sessions' uploads stopped overlapping (one session's copies alone move 34 GB/s, two overlapping 51), and the
    // sessions stayed in that step for whole blocks — 20 steps in 20.3 ms instead of 16.3, in 1-5 blocks of 12
    // (profiles/round5/slow_blocks_keys_behind_the_other_sessions_rows.txt).  The kernel's reads share the link packet by packet:
    // no such block in 72, counts on the host after 0.20 instead of 0.22-0.29 ms.  While calls miss little it is the other way
    // round — nothing is on the link, and a kernel would wait for CUs behind the other session's gather (every key resident
    // 6.35 -> 5.4 G lookups/s): copies.  Same bound as the second-stream scatter and the probe's place in the lane.
    // (HPS_ZC_CONTROL=0 — no kernel of this library reads or writes host memory — keeps the copies)
    const bool pull_keys = zc_control_ && h_keys_dev_ && (keys_by_kernel_ == 1 || (keys_by_kernel_ == 2 && miss_much_.high));
    uint64_t seen = 0;
    auto stage = [&](int width) -> Status {
      std::atomic<uint64_t> high_or{0};
      uint8_t* dst8 = reinterpret_cast<uint8_t*>(h_keys_pinned_);
      uint8_t* dev8 = reinterpret_cast<uint8_t*>(d_keys_);
      auto body = [&](size_t i) {
        const Task& tk = tasks[i];
        if (width == 8) { CopyKeys64Streaming(tk.src, tk.n, h_keys_pinned_ + tk.off); StreamFence(); return; }
        const uint64_t high = width == 4 ? PackKeys32(tk.src, tk.n, reinterpret_cast<uint32_t*>(dst8) + tk.off, tk.base)
                                         : PackKeys24(tk.src, tk.n, dst8 + 3 * tk.off, tk.base);
        if (high >> (8 * width)) high_or.fetch_or(high, std::memory_order_relaxed);
      };
      size_t g0 = 0;
      while (g0 < tasks.size()) {
        size_t g1 = g0, keys_in_group = 0;
        while (g1 < tasks.size() && (keys_in_group == 0 || keys_in_group + tasks[g1].n <= kGroupKeys)) keys_in_group += tasks[g1++].n;
        const auto tp0 = std::chrono::steady_clock::now();
        if (g1 - g0 <= 2) for (size_t i = g0; i < g1; ++i) body(i);
        else ThreadPool::Serving().ParallelFor(g1 - g0, [&](size_t i) { body(g0 + i); });
        const auto tp1 = std::chrono::steady_clock::now();
        if (width < 8 && (seen = high_or.load(std::memory_order_relaxed)) != 0) return Status::Ok();   // caller restages wider
        const size_t first = tasks[g0].off, count = tasks[g1 - 1].off + tasks[g1 - 1].n - first;
        if (pull_keys)
          HIP_TRY(LaunchPullBytes(reinterpret_cast<const uint8_t*>(h_keys_dev_) + first * (size_t)width, dev8 + first * (size_t)width, count * (size_t)width, ks));
        else
          HIP_TRY(hipMemcpyAsync(dev8 + first * (size_t)width, dst8 + first * (size_t)width, count * (size_t)width, hipMemcpyHostToDevice, ks));
        const auto tp2 = std::chrono::steady_clock::now();
        stage_pool_ms_ += std::chrono::duration<float, std::milli>(tp1 - tp0).count();
        stage_enqueue_ms_ += std::chrono::duration<float, std::milli>(tp2 - tp1).count();
        g0 = g1;
      }
      key_bytes_ = width;
      keys_narrow_ = width < 8;
      return Status::Ok();
    };
    key_bytes_ = 8;
    bool staged = false;
    if (try_pack24) {
      HPS_RETURN_IF_ERROR(stage(3));
      staged = keys_narrow_;
      if (!staged) {
        NarrowFailed(true);   // keys of more than 24 bits in this traffic
        HIP_TRY(hipStreamSynchronize(ks));   // groups already in flight read the staging buffer we are about to rewrite
      } else narrow24_streak_ = 0;
    }
    if (!staged && try_narrow && (seen >> 32) == 0) {
      seen = 0;
      HPS_RETURN_IF_ERROR(stage(4));
      staged = keys_narrow_;
      if (!staged) HIP_TRY(hipStreamSynchronize(ks));
      else narrow_streak_ = 0;
    }
    if (!staged) {
      if (try_narrow) NarrowFailed(false);   // wide (or negative) keys in this traffic: plain copies for the next calls
      if (direct_dma) HIP_TRY(hipMemcpyAsync(d_keys_, base, N * sizeof(int64_t), hipMemcpyHostToDevice, ks));
      else HPS_RETURN_IF_ERROR(stage(8));
    }
    const auto te0 = std::chrono::steady_clock::now();
    HIP_TRY(hipEventRecord(ev_keys_, ks));
    keys_wait_pending_ = true;   // the first stream waits for the keys AFTER the call block's pull has been enqueued (PrepareCall):
                                 // the pull needs nothing of the keys and runs under the tail of their upload
    stage_event_ms_ = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - te0).count();
  }
  key_stage_ms_ = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tk0).count();
  const Status st_ = TimedLookupDevice(d_keys_, vectors_per_table, num_keys_per_table, num_tables);
  // diagnostic: where did a slow call spend its time (HPS_TRACE_TAIL=<ms>: calls longer than that; no number: 5 ms)
  static const float kTraceCallsMs = [] { const char* e = std::getenv("HPS_TRACE_TAIL"); if (!e) return -1.f; const float v = std::strtof(e, nullptr); return v > 0.f ? v : 5.f; }();
  if (kTraceCallsMs > 0.f) {
    const float all = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tk0).count();
    if (all > kTraceCallsMs)
      fprintf(stderr, "[hps call] %.2f ms: key staging %.2f (pool %.2f, H2D enqueue %.2f, event %.2f), counts on host %.2f, ps fetch %.2f, tail %.2f, engine call %.2f, GPU span %.2f (direct_dma %d narrow %d) session %p began at %.3f ms\n",
              all, key_stage_ms_, stage_pool_ms_, stage_enqueue_ms_, stage_event_ms_, phase_ms_[0], phase_ms_[1], phase_ms_[2], phase_ms_[3], last_gpu_call_ms_, (int)direct_dma, (int)keys_narrow_,
              (void*)this, std::chrono::duration<double, std::milli>(tk0.time_since_epoch()).count());
  }
  return st_;
}

// LookupDevice + (option "timing") the GPU-side span of the call: probe start (after the waits on other
// sessions' kernels) to the last kernel of the call, by HIP events on the session's stream.
Status LookupSession::TimedLookupDevice(const int64_t* d_keys_flat, float* const* d_out, const size_t* n, size_t T) {
  const Status st = LookupDevice(d_keys_flat, d_out, n, T);
  last_gpu_call_ms_ = 0.f;
  // ev_c1_ was recorded behind the last kernel of whichever exit the call took, before its final synchronisation
  if (timing_ && st.ok()) (void)hipEventElapsedTime(&last_gpu_call_ms_, ev_t0_, ev_c1_);
  return st;
}

Status LookupSession::lookup_from_device(const int64_t* d_keys_flat, float* const* d_vectors_per_table,
                                         const size_t* num_keys_per_table, size_t num_tables) {
  if (!cache_) return Error(Code::kUnsupported, "lookup_from_device needs a GPU-cache session (gpucache=true)");
  if (num_tables != tables_.size())
    return Error(Code::kInvalidArg, "lookup: got ", num_tables, " tables, model '", params_.model_name, "' has ",
                 tables_.size());
  size_t N = 0;
  for (size_t t = 0; t < num_tables; ++t) N += num_keys_per_table[t];
  if (N > max_keys_) return Error(Code::kInvalidArg, "lookup: ", N, " keys exceed the session capacity of ", max_keys_);
  if (N == 0) return Status::Ok();
  HIP_TRY(hipSetDevice(device_));
  key_stage_ms_ = 0.f;
  keys_narrow_ = false;
  key_bytes_ = 8;
  return TimedLookupDevice(d_keys_flat, d_vectors_per_table, num_keys_per_table, num_tables);
}

Status LookupSession::lookup_from_device_padded(const int64_t* d_keys_flat, float* const* d_vectors_per_table,
                                                const size_t* num_keys_per_table, size_t num_tables) {
  skip_empty_next_ = true;
  const Status st = lookup_from_device(d_keys_flat, d_vectors_per_table, num_keys_per_table, num_tables);
  skip_empty_next_ = false;
  return st;
}

Status LookupSession::lookup_from_device_indexed(const int64_t* d_keys_flat, const uint32_t* d_dst_index, float* const* d_vectors_per_table,
                                                 const size_t* num_keys_per_table, size_t num_tables) {
  if (!d_dst_index) return Error(Code::kInvalidArg, "lookup_from_device_indexed: null destination index");
  dst_index_next_ = d_dst_index;
  const Status st = lookup_from_device(d_keys_flat, d_vectors_per_table, num_keys_per_table, num_tables);
  dst_index_next_ = nullptr;
  return st;
}

void LookupSession::discount_padding(uint64_t padding_keys) {
  if (!cache_ || padding_keys == 0) return;
  std::lock_guard<std::mutex> lk(cache_->stat_mu_);
  cache_->counters_.keys -= std::min<uint64_t>(padding_keys, cache_->counters_.keys);
}

Status LookupSession::LookupHostTier(const void* const* h_keys_per_table, float* const* h_vectors_per_table,
                                     const size_t* num_keys_per_table, size_t num_tables) {
  // gpucache=false: rows come straight from the parameter server into host memory
  // (docs/architecture.md:72; model_instance_state.cpp:114-133).
  std::vector<HierParameterServer::FetchJob> jobs;
  for (size_t t = 0; t < num_tables; ++t) {
    const size_t n = num_keys_per_table[t];
    if (n == 0) continue;
    if (!h_keys_per_table[t] || !h_vectors_per_table[t]) return Error(Code::kInvalidArg, "lookup: null pointer for table ", t);
    jobs.push_back({tables_[t].get(), (const int64_t*)h_keys_per_table[t], n, h_vectors_per_table[t],
                    tables_[t]->dim(), params_.default_value_for_each_table[t], nullptr});
  }
  return ps_->FetchMulti(jobs);
}

// Call descriptor (the per-table slicing of ProcessRequest, model_instance_state.cpp:180-193), the probe tiles and the
// zeroed accumulator block, uploaded with one copy.
Status LookupSession::PrepareCall(const int64_t* d_keys_flat, float* const* d_out, const size_t* n, size_t T, bool probe_only,
                                  uint64_t* N_out) {
  // the previous call's insert (left running behind it) has landed by now in all but pathological cases: its statistics
  // are read before this call's first push overwrites the words, and the page-locked staging it may have read in place
  // is free for this call's host gather
  HPS_RETURN_IF_ERROR(CollectDeferred());
  CallDesc& c = *h_call_;
  c.num_tables = (uint32_t)T;
  c.keys = d_keys_flat;
  const bool staged_narrow = keys_narrow_ && d_keys_flat == d_keys_;
  c.keys32 = (staged_narrow && key_bytes_ == 4) ? reinterpret_cast<const uint32_t*>(d_keys_) : nullptr;
  c.keys24 = (staged_narrow && key_bytes_ == 3) ? reinterpret_cast<const uint8_t*>(d_keys_) : nullptr;
  uint64_t N = 0;
  uint32_t tiles = 0;
  // Small requests get smaller tiles (a tile = one workgroup of the probe: a 28,672-key request in 1,024-key tiles would
  // put 28 workgroups on 256 CUs).  The tile REGIONS stay kTileKeys apart, so nothing else changes.
  uint64_t total = 0;
  for (size_t t = 0; t < T; ++t) total += n[t];
  const uint64_t tile_keys = total <= kSmallRequestKeys ? kSmallTileKeys : (uint64_t)kTileKeys;
  for (size_t t = 0; t < T; ++t) {
    c.key_start[t] = N;
    c.key_base[t] = (staged_narrow && t < key_base_.size()) ? key_base_[t] : 0;
    c.out[t] = probe_only ? nullptr : d_out[t];
    if (!probe_only && n[t] && !d_out[t]) return Error(Code::kInvalidArg, "lookup: null output pointer for table ", t);
    const uint32_t D = tables_[t]->dim();
    c.vec_ok[t] = (!probe_only && (D & 3u) == 0 && ((uintptr_t)d_out[t] & 15u) == 0) ? 1 : 0;
    for (uint64_t b = 0; b < n[t]; b += tile_keys)
      h_tiles_[tiles++] = TileDesc{N + b, (uint32_t)std::min<uint64_t>(tile_keys, n[t] - b), (uint32_t)t};
    N += n[t];
  }
  c.key_start[T] = N;
  c.total_keys = N;
  c.epoch = cache_->NextEpoch();
  c.stamp8 = cache_->Stamp8(c.epoch);
  c.skip_empty_keys = skip_empty_next_ ? 1u : 0u;
  c.dst_index = probe_only ? nullptr : dst_index_next_;
  if (++call_tag_ == 0) {  // 2^32 calls later: entries of the first calls would look like this call's
    HIP_TRY(hipStreamSynchronize(stream_));
    HIP_TRY(hipMemsetAsync(work_.set, 0, (work_.set_mask + 1) * sizeof(unsigned long long), stream_));
    call_tag_ = 1;
  }
  work_.num_tiles = tiles;
  work_.xcd_tiles = (tiles >= 64 && probe_xcd_tiles_) ? 1u : 0u;
  work_.call_tag = call_tag_;
  const size_t block_bytes = block_tiles_off_ + (size_t)tiles * sizeof(TileDesc);
  if (zc_control_) {
    const hipError_t pe = LaunchPull16(h_block_dev_, d_block_, block_bytes, stream_);
    if (pe != hipSuccess) return Error(Code::kInternal, "call block pull launch failed: ", hipGetErrorString(pe));
  } else {
    HIP_TRY(hipMemcpyAsync(d_block_, h_block_, block_bytes, hipMemcpyHostToDevice, stream_));
  }
  if (keys_wait_pending_) {
    keys_wait_pending_ = false;
    HIP_TRY(hipStreamWaitEvent(stream_, ev_keys_, 0));
  }
  *N_out = N;
  return Status::Ok();
}

Status LookupSession::PushWords(uint32_t words, hipEvent_t ev, uint32_t* seq_out, hipStream_t on) {
  hipStream_t st = on ? on : stream_;
  if (zc_control_) {
    if (++push_seq_ == 0) push_seq_ = 1;
    const hipError_t e = LaunchPushWords(d_acc_, h_acc_dev_, words, h_seq_dev_, push_seq_, st);
    if (e != hipSuccess) return Error(Code::kInternal, "accumulator push launch failed: ", hipGetErrorString(e));
  } else if (words) {
    HIP_TRY(hipMemcpyAsync(h_acc_, d_acc_, (size_t)words * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  }
  if (seq_out) *seq_out = push_seq_;
  HIP_TRY(hipEventRecord(ev ? ev : ev_done_, st));
  return Status::Ok();
}

static inline void SpinPause() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}

Status LookupSession::WaitPushed() { return WaitPushedSeq(push_seq_, ev_done_); }

Status LookupSession::WaitPushedSeq(uint32_t want, hipEvent_t ev) {
  if (!zc_control_) { HIP_TRY(hipEventSynchronize(ev)); return Status::Ok(); }
  // the word only moves forward (a later push may have overtaken the one waited for): signed distance, wrap-safe
  auto landed = [&]() { return (int32_t)(__atomic_load_n(h_seq_, __ATOMIC_ACQUIRE) - want) >= 0; };
  // poll the sequence word for a while (the usual wait is tens of microseconds), looking at the event now and then so that a
  // failed stream ends the wait; a long wait (a millisecond of uploads ahead of the push) goes to the runtime's own wait
  const auto t0 = std::chrono::steady_clock::now();
  constexpr long kSpinUs = 300;
  for (uint32_t i = 1;; ++i) {
    if (landed()) return Status::Ok();
    SpinPause();
    if ((i & 127u) == 0) sched_yield();   // a waiter must not keep a CPU from a pool worker that has work (thread_pool.cpp)
    if ((i & 511u) == 0) {
      const hipError_t q = hipEventQuery(ev);
      if (q != hipSuccess && q != hipErrorNotReady) return Error(Code::kInternal, "lookup stream failed: ", hipGetErrorString(q));
      // (Not longer: polling for 6 ms instead — tried in round 4 against the runtime's occasional 4-ms wake-ups — made things
      //  worse: 9-ms waits for the miss counts.  Work queued behind a cross-stream event wait is released by a thread of the
      //  HIP runtime; a caller that spins keeps that thread off the CPU, a caller that blocks in hipEventSynchronize lets it run.)
      if (q == hipSuccess || std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(kSpinUs)) break;
    }
  }
  HIP_TRY(hipEventSynchronize(ev));
  // the push kernel has retired: its stores are on their way; give the last one the time to land
  for (uint32_t i = 0; i < (1u << 24); ++i) {
    if (landed()) return Status::Ok();
    SpinPause();
  }
  return Error(Code::kInternal, "accumulator push did not arrive");
}

// Statistics (and duration) of the insert kernel a call left running behind it.
Status LookupSession::CollectDeferred() {
  std::lock_guard<std::mutex> lk(deferred_mu_);
  if (!deferred_pending_) return Status::Ok();
  (void)hipSetDevice(device_);
  HPS_RETURN_IF_ERROR(WaitPushedSeq(deferred_seq_, ev_done2_));   // (a failed wait leaves the statistics pending: not lost silently)
  deferred_pending_ = false;
  AddInsertStats();
  if (deferred_timed_) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev_i0_, ev_i1_) == hipSuccess) last_insert_ms_ = ms;
    else (void)hipGetLastError();
  }
  return Status::Ok();
}

// After the accumulator block has come back: per-table unique miss counts, the call's statistics.
Status LookupSession::ReadBackCounts(size_t T, uint64_t N, bool exact) {
  uint64_t misses = 0, uniq = 0, uniq_keys = 0, row_bytes = 0;
  for (size_t t = 0; t < T; ++t) {
    const uint32_t um = h_acc_[AccTableWord((uint32_t)t, kAccUniqMiss)];
    uniq_miss_[t] = um;
    uniq += um;
    row_bytes += (uint64_t)um * tables_[t]->dim() * sizeof(float);
    misses += h_acc_[AccTableWord((uint32_t)t, kAccSentMiss)];
    if (exact) uniq_keys += (uint64_t)um + h_acc_[AccTableWord((uint32_t)t, kAccUniqHit)];
  }
  last_misses_ = misses;
  last_miss_row_bytes_ = row_bytes;
  miss_much_.Update(row_bytes, side_bytes_);
  last_unique_ = uniq;
  last_unique_keys_ = uniq_keys;
  cache_->AdvanceClock(uniq);
  std::lock_guard<std::mutex> lk(cache_->stat_mu_);
  cache_->counters_.lookups += 1;
  cache_->counters_.keys += N;
  cache_->counters_.misses += misses;
  cache_->counters_.unique_misses += uniq;
  return Status::Ok();
}

void LookupSession::AddInsertStats() { cache_->AddStatLines(h_acc_); }

Status LookupSession::LookupDevice(const int64_t* d_keys_flat, float* const* d_out, const size_t* n, size_t T) {
  // direct mode: no table reload may replace a pinned slab while our kernels read it
  std::shared_lock<std::shared_mutex> direct_lock;
  if (cache_->direct()) {
    while (cache_->direct_writers().load(std::memory_order_acquire) > 0) std::this_thread::yield();
    direct_lock = std::shared_lock<std::shared_mutex>(cache_->direct_mutex());
  }
  const auto tc0 = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t).count();
  };
  phase_ms_[0] = phase_ms_[1] = phase_ms_[2] = phase_ms_[3] = 0.f;
  last_gather_ms_ = last_scatter_ms_ = 0.f;
  if (!defer_insert_) last_insert_ms_ = 0.f;   // (deferred: the duration of the most recent insert that has finished)
  // The insertion policy compares the table's hit rate over UNIQUE keys with the threshold
  // (docs/hierarchical_parameter_server.md:69: "first determines the associated unique embedding keys";
  // docs/architecture.md:66: "the real hit rate of the GPU embedding cache lookup"); a threshold outside (0,1) decides
  // without the rate, and then the unique-hit count is not taken.
  const bool exact = params_.hit_rate_threshold > 0.0f && params_.hit_rate_threshold < 1.0f;
  uint64_t N = 0;
  HPS_RETURN_IF_ERROR(PrepareCall(d_keys_flat, d_out, n, T, /*probe_only=*/false, &N));
  const CallDesc& c = *h_call_;
  const uint32_t epoch = c.epoch;
  const CallWork& w = work_;
  bool all128 = true;
  for (size_t t = 0; t < T; ++t) all128 &= tables_[t]->dim() == 128 && c.vec_ok[t];

  // (option "host_gather": serve this session's misses the reference's way — host threads + H2D copy — although the
  //  cache is in ps_direct_access mode; the pinned tables serve both paths)
  const bool use_direct = cache_->direct() && !force_host_gather_;
  const bool fast_direct = use_direct && params_.hit_rate_threshold >= 1.0f && last_misses_ > 0;
  // Split arrangement (host-gather tier, while calls keep missing): the miss counts go to the host right behind the
  // probe, and the hit rows are gathered by K_G while the host threads gather the missed rows and the DMA engine
  // uploads them — HBM-bound and PCIe-bound halves of one call side by side.  Otherwise K_G runs before the counts
  // are read (a call that missed nothing then needs no second synchronisation).
  // (device-driven tier: its own split, direct_split_ — staging layout + fetch kernel on the second stream next to K_G on
  //  the session's stream, HandleMissesDirect.  Round 1 measured this arrangement 8 % slower with the fetch kernel of the
  //  time, 2,048 single-row groups that shared the memory system with K_G; with round 3's small fetch grid it is 4 % faster:
  //  1.74 / 1.75 -> 1.83 / 1.81 G lookups/s, profiles/round3/ab_direct_fetch_grid.txt.  Putting the fetch kernel into the
  //  lane instead — nothing next to it at all — gave 1.34 against 1.59 G.)
  const bool split = split_probe_ && !use_direct && last_misses_ > 0;
  const int cu = cache_->cu_count();
  const uint32_t gather_blocks = GatherGridBlocks(N, cu);
  CallWork wk = work_;
  // device-driven tier with synchronous insertion: nobody on the host reads the unique keys — spare the PCIe writes
  if (use_direct && params_.hit_rate_threshold >= 1.0f) wk.uniq_keys_host = nullptr;
  // a narrowed request (every key below 2^32, checked while staging) gets its unique missed keys back as uint32
  uniq_narrow_ = !use_direct && keys_narrow_ && d_keys_flat == d_keys_ && narrow_publish_;
  wk.uniq_keys_host32 = uniq_narrow_ ? reinterpret_cast<uint32_t*>(wk.uniq_keys_host) : nullptr;
  // (Tried and withdrawn: K_M without the zero-copy host stores — 12 us of its 23 inside a busy link — and a publish kernel
  //  for the keys on the second stream next to K_G.  The persistent K_G owns every CU by the time the publish kernel is
  //  released, so the counts reached the host after the gather: 0.38 instead of 0.22 ms.)

  // ---- K_P: tile dedup + probe;  K_M: call-wide unique misses (+ unique hits) ----
  cache_->BeginRead(stream_);
  // The kernel lane keeps the HBM-bound kernels of a cache's sessions from running into each other.  K_P is not one of them
  // (~120 MB of bucket lines in 42 us), but in the lane it waits for the other session's whole gather — and with it this call's
  // miss counts, host gather and upload.  While the session's calls miss little (the same bound as the second-stream scatter:
  // the last call's missed rows fit side_bytes_) the probe therefore runs NEXT TO that gather and only K_G takes a turn: at
  // 99 % hit the call's chain is 0.73 instead of 0.81 ms and two sessions deliver 4.3 instead of 3.6-3.9 G lookups/s, every key
  // resident 6.0-6.2 instead of 5.6-5.8 G (profiles/round5/ab_probe_outside_the_lane.txt).  Calls that miss more (the
  // headline's 37 MB) are bound by PCIe either way and keep the probe in the lane, where it runs in 42 us instead of 50.
  const bool probe_lane = exclusive_ && (probe_in_lane_ == 1 || (probe_in_lane_ == 2 && miss_much_.high));
  if (probe_lane) cache_->LaneEnter(stream_);
  Mark(ev_t0_);
  const bool tail = fused_unique_ && ProbeTailAvailable(probe_variant_);
  hipError_t e = LaunchProbeTiles(d_call_, cache_->device_tables(), wk, probe_variant_, tail, stream_, Kt(ev_t0_, tail && !exact ? ev_t1_ : nullptr));
  if (e == hipSuccess && !tail) e = LaunchMissUnique(d_call_, cache_->device_tables(), wk, stream_, Kt(nullptr, exact ? nullptr : ev_t1_));
  if (e == hipSuccess && exact) {
    // K_H: unique hit keys per table = distinct slots among the slot words K_P has just written (LDS bitmaps, kernels.hip)
    uint32_t parts = 0;
    for (size_t t = 0; t < T; ++t)
      if (n[t]) parts += UniqueHitsParts((uint64_t)cache_->host_tables()[t].num_buckets * kBucketSlots);
    e = LaunchUniqueHits(d_call_, cache_->device_tables(), parts, w.slot, d_acc_, stream_, Kt(nullptr, ev_t1_));
  }
  Mark(ev_t1_);
  // K_G follows K_P at once when the call does not wait for the counts first (no misses lately, or the device-driven tier):
  // the lane is kept across both — handing it to the other session in between costs two more switches of the GPU between
  // streams per pair of calls for nothing (every key resident, two sessions: 20 us of idle GPU per step)
  // (split: the counts leave on the session's SECOND stream, released by the probe's event, so K_G follows K_P on this stream
  //  just the same — round 3 pushed them between the two kernels: two more dependent packets and ~35 us of idle GPU per call)
  const bool side_push = split && zc_control_;
  const bool hold_lane = probe_lane && (!split || side_push) && e == hipSuccess;
  if (probe_lane && !hold_lane) cache_->LaneLeave(stream_, ev_lane_[0]);
  // other sessions' probes chain behind ours (K_P, and K_M / K_H where they are launches of their own)
  (void)hipEventRecord(ev_probe_, stream_);
  auto gather = [&]() -> hipError_t {
    if (exclusive_ && !hold_lane) cache_->LaneEnter(stream_);
    Mark(ev_g0_);
    const hipError_t ge = LaunchGatherHits(d_call_, cache_->device_tables(), (uint32_t)T, N, w.slot, gather_blocks, all128, xcd_walk_, stream_,
                                           Kt(ev_g0_, ev_g1_), c.dst_index != nullptr);
    Mark(ev_g1_);
    if (exclusive_) cache_->LaneLeave(stream_, ev_lane_[1]);
    return ge;
  };
  bool read_open = true;
  // the cache stays read-locked until K_G has read the last slot: writers wait for ev_read_, probes for ev_probe_
  // (option chain_gather: other sessions' probes wait for our gather too — every HBM-bound kernel then runs alone)
  auto end_read = [&]() {
    if (!read_open) return;
    if (chain_gather_) cache_->EndRead(stream_, ev_read_);
    else cache_->EndReadFused(stream_, ev_probe_, ev_read_);
    read_open = false;
  };
  if (e != hipSuccess) { end_read(); return Error(Code::kInternal, "probe launch failed: ", hipGetErrorString(e)); }
  if (side_push) {
    HIP_TRY(hipStreamWaitEvent(copy_stream_, ev_probe_, 0));
    const Status ps = PushWords((uint32_t)acc_words_, ev_done_, nullptr, copy_stream_);
    if (!ps.ok()) { end_read(); return ps; }
  }
  if (!split || side_push) {
    e = gather();
    end_read();
    if (e != hipSuccess) return Error(Code::kInternal, "hit gather launch failed: ", hipGetErrorString(e));
  }
  last_async_ = false;
  table_async_.assign(T, 0);
  if (fast_direct) {
    // Device-driven miss path with the insertion policy fixed to "synchronous": nothing on the host depends
    // on the miss counts, so the whole call is enqueued without a round trip and the counts come back at the end.
    // (Only while the previous call of this session missed something: a fully resident working set is served
    // faster by reading the counts first and stopping there.)
    const Status st = HandleMissesDirect(N, epoch, /*counts_known=*/false, nullptr);
    if (timing_) { (void)hipEventElapsedTime(&last_gpu_ms_, ev_t0_, ev_t1_); (void)hipEventElapsedTime(&last_gather_ms_, ev_g0_, ev_g1_); }
    if (st.ok()) HPS_RETURN_IF_ERROR(ReadBackCounts(T, N, exact));
    phase_ms_[3] = ms_since(tc0);
    phase_ms_[2] = phase_ms_[3];  // no host phases on this path; [1] holds the fetch kernel's GPU time
    return st;
  }
  if (timing_) (void)hipEventRecord(ev_c1_, stream_);
  if (!side_push) {
    const Status ps = PushWords((uint32_t)acc_words_);
    if (!ps.ok()) { end_read(); return ps; }
  }
  if (split && !side_push) {
    // K_G behind the counts: it runs while the host reads them and works on the misses
    e = gather();
    end_read();
    if (e != hipSuccess) return Error(Code::kInternal, "hit gather launch failed: ", hipGetErrorString(e));
  }
  HPS_RETURN_IF_ERROR(WaitPushed());
  if (timing_) (void)hipEventElapsedTime(&last_gpu_ms_, ev_t0_, ev_t1_);
  HPS_RETURN_IF_ERROR(ReadBackCounts(T, N, exact));
  phase_ms_[0] = phase_ms_[3] = ms_since(tc0);
  split_call_ = split;
  if (last_unique_ == 0) {
    if (split) HIP_TRY(hipStreamSynchronize(stream_));   // the hit rows are still being written
    if (timing_) (void)hipEventElapsedTime(&last_gather_ms_, ev_g0_, ev_g1_);
    return Status::Ok();
  }

  // ---- insertion policy, decided per table as the reference's per-table lookup loop does
  //      (docs/architecture.md:65-67): a table whose hit rate over the call's unique keys
  //      reaches hit_rate_threshold returns the default vector for its misses now and has them inserted in the
  //      background; the other tables fetch, return and insert their missed rows before the call returns ----
  bool any_async = false, any_sync = false;
  for (size_t t = 0; t < T; ++t) {
    const uint64_t um = uniq_miss_[t];
    if (um == 0) continue;
    bool as;
    if (params_.hit_rate_threshold >= 1.0f) as = false;
    else if (params_.hit_rate_threshold <= 0.0f) as = true;
    else {
      const uint64_t uk = um + h_acc_[AccTableWord((uint32_t)t, kAccUniqHit)];
      as = 1.0 - (double)um / (double)uk >= (double)params_.hit_rate_threshold;
    }
    table_async_[t] = as ? 1 : 0;
    any_async |= as;
    any_sync |= !as;
  }
  const uint32_t* d_mode = nullptr;
  if (any_async && any_sync) {
    // mixed call: the kernels learn the per-table mode from a small device array
    for (size_t t = 0; t < T; ++t) h_mode_[t] = table_async_[t];
    HIP_TRY(hipMemcpyAsync(d_mode_, h_mode_, T * sizeof(uint32_t), hipMemcpyHostToDevice, stream_));
    d_mode = d_mode_;
  }
  if (any_async) {
    last_async_ = true;
    e = LaunchMissFillDefault(d_call_, cache_->device_tables(), w, d_mode, stream_);
    if (e != hipSuccess) return Error(Code::kInternal, "default fill launch failed: ", hipGetErrorString(e));
    // hand the async tables' unique missed keys to the background inserter (best effort)
    if (use_direct) {
      // device-driven tier: the keys never leave the GPU; fetch + insert run on the cache's own stream
      uint64_t uniq = 0, floats = 0;
      uint32_t* job_lines = h_mode_ + kMaxTables;   // T accumulator lines: unique misses of the async tables only
      for (size_t t = 0; t < T; ++t) {
        const uint32_t cnt = table_async_[t] ? uniq_miss_[t] : 0u;
        job_lines[t * kAccStride + kAccUniqMiss] = cnt;
        uniq += cnt;
        floats = ((floats + 3) & ~(uint64_t)3) + (uint64_t)cnt * tables_[t]->dim();
      }
      bool accepted = false;
      HPS_RETURN_IF_ERROR(cache_->SubmitDirectInsert(stream_, d_call_->key_start, w.uniq_keys, d_acc_ + (size_t)kStatLines * kAccStride,
                                                     any_sync ? job_lines : nullptr, N, uniq, floats, &accepted));
      if (accepted) ps_->RunDirectInsert(cache_);
    } else {
      std::vector<std::vector<int64_t>> job(T);
      for (size_t t = 0; t < T; ++t) {
        if (!table_async_[t]) continue;
        if (uniq_narrow_) {
          const uint32_t* k32 = reinterpret_cast<const uint32_t*>(h_uniq_keys_) + c.key_start[t];
          job[t].resize(uniq_miss_[t]);
          for (size_t i = 0; i < job[t].size(); ++i) job[t][i] = c.key_base[t] + (int64_t)(uint64_t)k32[i];
        } else {
          job[t].assign(h_uniq_keys_ + c.key_start[t], h_uniq_keys_ + c.key_start[t] + uniq_miss_[t]);
        }
      }
      ps_->SubmitAsyncInsert(cache_, std::move(job));
    }
    {
      std::lock_guard<std::mutex> lk(cache_->stat_mu_);
      cache_->counters_.async_calls += 1;
    }
    if (!any_sync) {
      if (timing_) (void)hipEventRecord(ev_c1_, stream_);
      HIP_TRY(hipStreamSynchronize(stream_));
      if (timing_) (void)hipEventElapsedTime(&last_gather_ms_, ev_g0_, ev_g1_);
      return Status::Ok();
    }
  }
  const Status st = use_direct ? HandleMissesDirect(N, epoch, /*counts_known=*/true, d_mode) : HandleMisses(N, epoch);
  if (timing_ && st.ok()) (void)hipEventElapsedTime(&last_gather_ms_, ev_g0_, ev_g1_);
  phase_ms_[3] = ms_since(tc0);
  phase_ms_[2] = phase_ms_[3] - phase_ms_[0] - phase_ms_[1];
  return st;
}

// BASELINE config 5, fused arrangement: probe only (no OUTPUT0), unique misses fetched by the device-driven tier,
// then the dot interaction reads every row from where it is — cache slot or miss staging — and only afterwards are
// the missed rows inserted.  The cache stays read-locked (reader event) from the probe to the interaction, so no
// other session's insert can recycle a slot this call still has to read; probes of other sessions chain behind OUR
// probe only (its own event), not behind the whole call.
Status LookupSession::lookup_interact(DenseInteraction* dense, const int64_t* d_keys_flat, uint64_t batch, const float* d_dense_features,
                                      void* d_out_f16) {
  if (!cache_ || !cache_->direct())
    return Error(Code::kUnsupported, "lookup_interact needs a GPU-cache session of a ps_direct_access model");
  if (!dense || !d_keys_flat || !d_dense_features || !d_out_f16) return Error(Code::kInvalidArg, "lookup_interact: null argument");
  const size_t T = tables_.size();
  if (params_.hit_rate_threshold < 1.0f)
    return Error(Code::kUnsupported, "lookup_interact serves exact rows only (hit_rate_threshold must be 1.0; it is ",
                 params_.hit_rate_threshold, ")");
  if (dense->num_tables() != T || dense->device() != device_)
    return Error(Code::kInvalidArg, "lookup_interact: the dense step was built for ", dense->num_tables(), " tables on device ",
                 dense->device(), ", the model has ", T, " on device ", device_);
  for (size_t t = 0; t < T; ++t)
    if (tables_[t]->dim() != dense->emb_dim())
      return Error(Code::kInvalidArg, "lookup_interact: table ", t, " is ", tables_[t]->dim(), " wide, the dense step expects ", dense->emb_dim());
  const uint64_t N = batch * T;
  if (N > max_keys_) return Error(Code::kInvalidArg, "lookup_interact: ", N, " keys exceed the session capacity of ", max_keys_);
  if ((uint64_t)T * (dense->emb_dim() / 4) > 1024) return Error(Code::kUnsupported, "lookup_interact: more than 4096 floats per sample");
  if (batch == 0) return Status::Ok();
  HIP_TRY(hipSetDevice(device_));
  std::shared_lock<std::shared_mutex> direct_lock;
  while (cache_->direct_writers().load(std::memory_order_acquire) > 0) std::this_thread::yield();
  direct_lock = std::shared_lock<std::shared_mutex>(cache_->direct_mutex());

  std::vector<size_t> n(T, (size_t)batch);
  // Which arrangement serves this call (session option "interact_mode"): the FUSED one below — probe, fetch, interaction reading
  // cache slots and staged rows, OUTPUT0 never written — or the SEPARATE steps: an ordinary lookup into a buffer of the session's
  // own, then the dense step's two kernels.  Fused saves the 2 x N x 4D bytes of OUTPUT0 (all keys resident: 0.34 against 0.56 ms
  // per step) but keeps the cache read-locked from the probe to the interaction and fetches the misses before anything is
  // gathered; while calls miss much the separate steps overlap the fetch with the hit gather (HandleMissesDirect's split) and come
  // out ahead.  Adaptive (the default): separate while the last call's missed rows exceeded side_scatter_mb — the bound the other
  // per-call switches use.
  const bool separate = interact_mode_ == 0 || (interact_mode_ == 2 && miss_much_.high);
  last_interact_separate_ = separate;
  if (separate) {
    direct_lock.unlock();   // (LookupDevice takes it itself)
    const size_t D = dense->emb_dim();
    if (interact_emb_floats_ < N * D) {
      if (d_interact_emb_) HIP_TRY(hipFree(d_interact_emb_));
      d_interact_emb_ = nullptr;
      interact_emb_floats_ = 0;
      const size_t want = std::max<size_t>(N * D, max_keys_ * D);
      if (hipMalloc((void**)&d_interact_emb_, want * sizeof(float)) != hipSuccess)
        return Error(Code::kInternal, "lookup_interact: out of device memory for the embedding buffer of the separate arrangement (", want * sizeof(float), " bytes)");
      interact_emb_floats_ = want;
    }
    std::vector<float*> outs(T);
    for (size_t t = 0; t < T; ++t) outs[t] = d_interact_emb_ + t * batch * D;
    key_stage_ms_ = 0.f;
    keys_narrow_ = false;
    key_bytes_ = 8;
    // The bottom MLP needs nothing of the lookup: it goes down the session's SECOND stream first and runs next to the probe
    // (a caller that runs lookup and dense step itself has it behind the lookup: 40 us of the call); the interaction waits for it.
    const void* d_bottom = nullptr;
    HPS_RETURN_IF_ERROR(dense->BottomMlp(d_dense_features, batch, copy_stream_, &d_bottom));
    HPS_RETURN_IF_ERROR(TimedLookupDevice(d_keys_flat, outs.data(), n.data(), T));
    // The interaction follows the MLP on the SECOND stream (in order behind it, no event needed) — not on the first: that one
    // still holds the insert kernel the lookup left running behind it, which waits for other sessions' gathers to release the
    // cache; the rows the interaction reads are complete (the lookup returned).
    HPS_RETURN_IF_ERROR(dense->Interact(d_interact_emb_, d_bottom, batch, d_out_f16, copy_stream_));
    HIP_TRY(hipStreamSynchronize(copy_stream_));
    return Status::Ok();
  }
  uint64_t N2 = 0;
  HPS_RETURN_IF_ERROR(PrepareCall(d_keys_flat, nullptr, n.data(), T, /*probe_only=*/true, &N2));
  const uint32_t epoch = h_call_->epoch;
  CallWork w = work_;
  w.uniq_keys_host = nullptr;     // direct-only path: nobody on the host reads the unique keys — spare the PCIe stores
  w.uniq_keys_host32 = nullptr;
  const int cu = cache_->cu_count();
  // bottom MLP first: it needs nothing from the lookup and leaves the stream before the cache is read-locked
  const void* d_bottom = nullptr;
  HPS_RETURN_IF_ERROR(dense->BottomMlp(d_dense_features, batch, stream_, &d_bottom));

  cache_->BeginRead(stream_);   // ---- read lock: held (order mutex + reader event) until the interaction is enqueued ----
  Mark(ev_t0_);
  const bool tail = fused_unique_ && ProbeTailAvailable(probe_variant_);
  hipError_t e = LaunchProbeTiles(d_call_, cache_->device_tables(), w, probe_variant_, tail, stream_, Kt(ev_t0_, tail ? ev_t1_ : nullptr));
  if (e == hipSuccess && !tail) e = LaunchMissUnique(d_call_, cache_->device_tables(), w, stream_, Kt(nullptr, ev_t1_));
  Mark(ev_t1_);
  (void)hipEventRecord(ev_probe_, stream_);
  if (e == hipSuccess) e = LaunchMissDescBuild(cache_->device_tables(), (uint32_t)T, d_acc_, d_md_, /*clear_stats=*/false, nullptr, stream_);
  if (e == hipSuccess) {
    cache_->BeginFetch(stream_);
    if (timing_) (void)hipEventRecord(ev_f0_, stream_);
    e = LaunchPsFetchDirect(cache_->device_index(), (uint32_t)T, d_md_, d_call_->key_start, w.uniq_keys, d_staging_, d_found_, N, 0,
                            stream_);
    if (timing_) (void)hipEventRecord(ev_f1_, stream_);
    cache_->EndFetch(stream_, ev_fetch_);
  }
  if (e == hipSuccess)
    e = LaunchLookupInteract(cache_->device_tables(), d_md_, w.slot, w.rep_of, w.uidx_of, d_staging_, d_bottom, batch, (uint32_t)T,
                             dense->emb_dim(), dense->out_stride(), d_out_f16, cu, stream_);
  cache_->EndReadFused(stream_, ev_probe_, ev_read_);   // ---- read lock released (enqueue side) ----
  if (e != hipSuccess) return Error(Code::kInternal, "lookup_interact launch failed: ", hipGetErrorString(e));
  last_async_ = false;
  table_async_.assign(T, 0);
  // The output is complete behind this point of the stream: that is what the caller waits for.  The insert of the missed rows
  // is enqueued behind it and left running (defer_insert_, as in HandleMissesDirect): later readers of the cache are ordered
  // behind it by the writer event, its statistics are read at the session's next call (round 5 waited for it here: 35 us of
  // kernel + a writer window on the return path of every call that missed — 0.990 against 0.941 ms for the separate steps).
  const bool defer = defer_insert_ && zc_control_;
  uint32_t rows_seq = 0;
  if (defer) {
    if (timing_) (void)hipEventRecord(ev_c1_, stream_);
    HPS_RETURN_IF_ERROR(PushWords((uint32_t)acc_words_, ev_done_, &rows_seq));
  }
  // (writer window = the insert kernel alone, as in HandleMissesDirect: the stream is drained before other sessions are made to wait)
  {
    const uint64_t est_bytes = (uint64_t)last_unique_ * dense->emb_dim() * sizeof(float);
    if (est_bytes > (2u << 20)) HIP_TRY(hipStreamSynchronize(stream_));
  }
  cache_->BeginWrite(stream_);
  Mark(ev_i0_);
  e = LaunchCacheInsert(cache_->device_tables(), (uint32_t)T, d_md_, N, d_call_->key_start, w.uniq_keys, d_staging_, d_found_,
                        cache_->InsertStamps(epoch), d_acc_, cu, stream_, Kt(ev_i0_, ev_i1_));
  Mark(ev_i1_);
  cache_->EndWrite(stream_);
  if (e != hipSuccess) return Error(Code::kInternal, "cache insert launch failed: ", hipGetErrorString(e));
  if (defer) {
    {
      std::lock_guard<std::mutex> lk(deferred_mu_);
      const Status ps2 = PushWords((uint32_t)kStatLines * kAccStride, ev_done2_, &deferred_seq_);
      if (!ps2.ok()) return ps2;
      deferred_pending_ = true;
      deferred_timed_ = timing_;
    }
    HPS_RETURN_IF_ERROR(WaitPushedSeq(rows_seq, ev_done_));
  } else {
    if (timing_) (void)hipEventRecord(ev_c1_, stream_);
    HPS_RETURN_IF_ERROR(PushWords((uint32_t)acc_words_));
    HPS_RETURN_IF_ERROR(WaitPushed());
  }
  if (timing_) {
    (void)hipEventElapsedTime(&last_gpu_ms_, ev_t0_, ev_t1_);
    (void)hipEventElapsedTime(&phase_ms_[1], ev_f0_, ev_f1_);
    (void)hipEventElapsedTime(&last_gpu_call_ms_, ev_t0_, ev_c1_);
  }
  HPS_RETURN_IF_ERROR(ReadBackCounts(T, N, false));
  if (!defer) AddInsertStats();
  return Status::Ok();
}

// Synchronous miss path, device-driven ("ps_direct_access"): the GPU resolves the unique missed keys through the
// device-resident index of the host tier and pulls the rows out of pinned host memory itself.
Status LookupSession::HandleMissesDirect(uint64_t N, uint32_t epoch, bool counts_known, const uint32_t* d_table_mode) {
  const size_t T = tables_.size();
  const int cu = cache_->cu_count();
  const uint64_t max_unique = counts_known ? (uint64_t)last_unique_ : N;
  // direct_split_: staging layout + PCIe fetch on the session's second stream, released by the probe alone — they run next
  // to this call's own hit gather; the scatter waits for both
  hipStream_t fs = stream_;
  if (direct_split_ && !counts_known && N > kSmallRequestKeys) {   // (a small request's gather is a few microseconds: the hop to the second stream costs more)
    fs = copy_stream_;
    HIP_TRY(hipStreamWaitEvent(fs, ev_probe_, 0));
  }
  hipError_t e = LaunchMissDescBuild(cache_->device_tables(), (uint32_t)T, d_acc_, d_md_, /*clear_stats=*/false, d_table_mode, fs);
  if (e == hipSuccess) {
    cache_->BeginFetch(fs);
    if (timing_) (void)hipEventRecord(ev_f0_, fs);
    e = LaunchPsFetchDirect(cache_->device_index(), (uint32_t)T, d_md_, d_call_->key_start, work_.uniq_keys, d_staging_,
                            d_found_, max_unique, N <= kSmallRequestKeys ? -1 : 0, fs);
    if (timing_) (void)hipEventRecord(ev_f1_, fs);
    cache_->EndFetch(fs, ev_fetch_);
    if (fs != stream_) (void)hipStreamWaitEvent(stream_, ev_fetch_, 0);
  }
  if (e != hipSuccess) return Error(Code::kInternal, "direct miss path launch failed: ", hipGetErrorString(e));
  // Keep the window in which other sessions' kernels wait for our scatter (lane) and our writer event down to those two
  // kernels: drain the stream first, so that their events are not recorded behind a millisecond of PCIe fetch.
  // Small requests skip the drain (their fetch is a few tens of microseconds, less than the host round trip):
  // the estimate is this call's unique-miss count when known, else the previous call's.
  {
    uint64_t row_bytes = 0;
    for (const auto& tb : tables_) row_bytes = std::max<uint64_t>(row_bytes, (uint64_t)tb->dim() * sizeof(float));
    const uint64_t est_bytes = (uint64_t)last_unique_ * row_bytes;
    if (est_bytes > (2u << 20)) HIP_TRY(hipStreamSynchronize(stream_));
  }
  if (exclusive_) cache_->LaneEnter(stream_);
  Mark(ev_s0_);
  e = LaunchMissScatter(d_call_, cache_->device_tables(), d_md_, work_, d_staging_, stream_, Kt(ev_s0_, ev_s1_));
  Mark(ev_s1_);
  if (exclusive_) cache_->LaneLeave(stream_, ev_lane_[2]);
  if (e != hipSuccess) return Error(Code::kInternal, "direct miss path launch failed: ", hipGetErrorString(e));
  const bool defer = defer_insert_ && zc_control_;
  uint32_t rows_seq = 0;
  if (defer) {
    // rows complete (and, when the call ran ahead of its counts, the counts with them): what the caller waits for
    if (timing_) (void)hipEventRecord(ev_c1_, stream_);
    HPS_RETURN_IF_ERROR(PushWords((uint32_t)acc_words_, ev_done_, &rows_seq));
  }
  cache_->BeginWrite(stream_);
  if (exclusive_) cache_->LaneEnter(stream_);
  Mark(ev_i0_);
  e = LaunchCacheInsert(cache_->device_tables(), (uint32_t)T, d_md_, max_unique, d_call_->key_start, work_.uniq_keys,
                        d_staging_, d_found_, cache_->InsertStamps(epoch), d_acc_, cu, stream_, Kt(ev_i0_, ev_i1_));
  Mark(ev_i1_);
  if (exclusive_) cache_->LaneLeave(stream_, ev_lane_[3]);
  cache_->EndWrite(stream_);
  if (e != hipSuccess) return Error(Code::kInternal, "cache insert launch failed: ", hipGetErrorString(e));
  if (defer) {
    {
      std::lock_guard<std::mutex> lk(deferred_mu_);
      const Status ps2 = PushWords((uint32_t)kStatLines * kAccStride, ev_done2_, &deferred_seq_);
      if (!ps2.ok()) return ps2;
      deferred_pending_ = true;
      deferred_timed_ = timing_;
    }
    HPS_RETURN_IF_ERROR(WaitPushedSeq(rows_seq, ev_done_));
    if (timing_) {
      (void)hipEventElapsedTime(&phase_ms_[1], ev_f0_, ev_f1_);  // direct path: [1] = the fetch kernel (GPU time)
      (void)hipEventElapsedTime(&last_scatter_ms_, ev_s0_, ev_s1_);
    }
    return Status::Ok();
  }
  if (timing_) (void)hipEventRecord(ev_c1_, stream_);
  HPS_RETURN_IF_ERROR(PushWords((uint32_t)acc_words_));
  HPS_RETURN_IF_ERROR(WaitPushed());
  if (timing_) {
    (void)hipEventElapsedTime(&phase_ms_[1], ev_f0_, ev_f1_);  // direct path: [1] = the fetch kernel (GPU time)
    (void)hipEventElapsedTime(&last_scatter_ms_, ev_s0_, ev_s1_);
    (void)hipEventElapsedTime(&last_insert_ms_, ev_i0_, ev_i1_);
  }
  AddInsertStats();
  return Status::Ok();
}

// Synchronous miss path: parameter-server gather of the unique missed keys into pinned staging,
// one H2D copy per chunk, missed rows scattered to the output, then inserted into the cache.
Status LookupSession::HandleMisses(uint64_t N, uint32_t epoch) {
  const size_t T = tables_.size();
  const CallDesc& c = *h_call_;
  const int cu = cache_->cu_count();
  std::vector<uint32_t> ucnt(T), done(T, 0);
  size_t total_floats = 0, total_uniq = 0;
  for (size_t t = 0; t < T; ++t) {
    // tables in async-insert mode take no part here: their misses got the default vector (K_D)
    ucnt[t] = (t < table_async_.size() && table_async_[t]) ? 0u : uniq_miss_[t];
    total_floats += (size_t)ucnt[t] * tables_[t]->dim();
    total_uniq += ucnt[t];
  }
  const size_t cap_floats = kStagingCapBytes / sizeof(float);
  HPS_RETURN_IF_ERROR(EnsureStaging(std::min(total_floats, cap_floats), total_uniq));

  for (;;) {
    // ---- assemble the next chunk: whole tables while they fit, else a slice of one table ----
    MissDesc& md = *h_md_;
    size_t fl = 0, uq = 0;
    bool any = false;
    for (size_t t = 0; t < T; ++t) {
      const uint32_t D = tables_[t]->dim();
      md.useg_start[t] = uq;
      md.chunk_lo[t] = done[t];
      // keep every table's staging offset 16-B aligned so the float4 path stays usable
      fl = (fl + 3) & ~(size_t)3;
      md.stage_off[t] = fl;
      const size_t room_rows = fl < staging_floats_ ? (staging_floats_ - fl) / D : 0;
      const uint32_t take = (uint32_t)std::min<size_t>(ucnt[t] - done[t], room_rows);
      md.chunk_hi[t] = done[t] + take;
      fl += (size_t)take * D;
      uq += take;
      any |= take > 0;
    }
    md.useg_start[T] = uq;
    if (!any) break;
    bool last = true;
    for (size_t t = 0; t < T; ++t) last &= md.chunk_hi[t] == ucnt[t];
    // The insert kernel of the call's last chunk is left running BEHIND the call (defer_insert_): the rows are exact without
    // it, later readers of the cache are ordered behind it by the writer event, its statistics are read at the next call.
    const bool defer = defer_insert_ && last && zc_control_;

    // ---- host parameter-server gather (multi-threaded) into pinned staging ----
    // Gather and upload in pieces of a few MB (runs of consecutive tables): the H2D copy of piece p runs
    // on the copy engine while the host threads gather piece p+1, so the PCIe time (the floor of this
    // path: every missed row crosses the link once) hides most of the DRAM-latency-bound gather.
    // A small chunk (a small request, or a big one that missed little) is not uploaded at all: the scatter and insert kernels
    // read the gathered rows out of the page-locked staging buffer themselves, and the descriptor is pulled by a kernel — three
    // SDMA copies and their queue hand-offs less on the path of a request that takes 0.15 ms in all.
    const bool in_place = zc_control_ && fl * sizeof(float) <= in_place_bytes_ && h_staging_dev_ && h_found_dev_ && h_md_dev_;
    // A chunk read in place needs nothing of this call's hit gather and nothing the gather writes (missed and hit rows are
    // disjoint): descriptor pull and scatter go down the session's SECOND stream, released by the probe alone, and run next
    // to K_G instead of behind it — and outside the kernel lane (a few hundred rows do not disturb an HBM-bound kernel,
    // while waiting for the lane puts the other session's 230-us gather on this call's return path).  Round 3 had
    // near-all-hit calls (hit 0.9996) at 1.5 x their kernel time because of exactly that wait plus the writer lock.
    // (a small request's gather is a few microseconds: the hop to the second stream costs more)
    // The same arrangement for a chunk that IS uploaded, up to side_bytes_ of rows (a call at 99 % hit: 17 K rows, 9 MB): the
    // uploads already go down the second stream; descriptor, found flags and the scatter follow them there, so nothing waits
    // for a host-side drain of K_G and nothing waits for the lane.  Beyond that size (the headline's 37 MB at 95 % hit) the
    // scatter is an HBM-bound kernel of its own and keeps its turn in the lane behind a drained stream.
    const bool side = defer && N > kSmallRequestKeys && (in_place || fl * sizeof(float) <= side_bytes_);
    hipStream_t ss = side ? copy_stream_ : stream_;
    if (side) HIP_TRY(hipStreamWaitEvent(ss, ev_probe_, 0));
    if (in_place || (side && h_md_dev_ && zc_control_)) {
      const hipError_t pe = LaunchPull16(h_md_dev_, d_md_, sizeof(MissDesc), ss);
      if (pe != hipSuccess) return Error(Code::kInternal, "miss descriptor pull launch failed: ", hipGetErrorString(pe));
    } else {
      HIP_TRY(hipMemcpyAsync(d_md_, h_md_, sizeof(MissDesc), hipMemcpyHostToDevice, ss));
    }
    // upload piece: 4 MB (8- and 16-MB pieces: no gain, profiles/round4/ab_upload_piece_size_and_sessions.txt) — reached by doubling
    // from a first piece of 512 KB: the link starts after 1,000 gathered rows instead of 8,000.  A lone request's p50 1.18 -> 1.14 ms
    // (four interleaved pairs), two requests in flight 1.57 -> 1.55 ms at the same rate (profiles/round6/ab_first_upload_piece.txt)
    constexpr size_t kPieceFloats = (size_t)4 * (1u << 20) / sizeof(float), kFirstPieceFloats = (size_t)512 * 1024 / sizeof(float);
    size_t piece_floats = kFirstPieceFloats;
    std::vector<HierParameterServer::FetchJob> jobs;
    size_t piece_begin = SIZE_MAX, piece_end = 0;
    bool used_copy_stream = false;
    auto flush = [&]() -> Status {
      if (jobs.empty()) return Status::Ok();
      const auto tf0 = std::chrono::steady_clock::now();
      HPS_RETURN_IF_ERROR(ps_->FetchMulti(jobs));
      phase_ms_[1] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tf0).count();
      // split call: the session's stream is busy with K_G, the pieces go down the copy stream
      // (A/B on the MI355X box: alternating the pieces between two copy streams was slower — 1.24 vs 1.06 ms/step.
      //  Round 4: "upload turns" — one call's missed rows on the link at a time, the session that finds the turn taken gathers
      //  everything and sends ONE copy when its turn comes — 1.70-1.72 G lookups/s and p50 1.93-1.99 ms against 2.00 G and
      //  1.60 ms: a single stream of copies leaves the link idle between a call's pieces, two streams fill each other's gaps.
      //  Withdrawn; profiles/round4/ab_upload_turns_withdrawn.txt.)
      if (!in_place) {
        hipStream_t cs = (split_call_ || side) ? copy_stream_ : stream_;
        HIP_TRY(hipMemcpyAsync(d_staging_ + piece_begin, h_staging_ + piece_begin, (piece_end - piece_begin) * sizeof(float),
                               hipMemcpyHostToDevice, cs));
        used_copy_stream |= (cs == copy_stream_);
      }
      jobs.clear();
      piece_begin = SIZE_MAX; piece_end = 0;
      return Status::Ok();
    };
    for (size_t t = 0; t < T; ++t) {
      const uint32_t lo = md.chunk_lo[t], hi = md.chunk_hi[t];
      if (hi == lo) continue;
      const uint32_t D = tables_[t]->dim();
      // a big table is cut into several pieces of its own
      for (uint32_t r = lo; r < hi;) {
        // (the first pieces of a call are smaller: the link starts sooner; doubling up to 4 MB)
        const size_t room = piece_begin == SIZE_MAX ? piece_floats : piece_floats - std::min(piece_floats, piece_end - piece_begin);
        const uint32_t rows_per_piece = (uint32_t)std::max<size_t>(1, std::max<size_t>(room, D) / D);
        const uint32_t re = std::min(hi, r + rows_per_piece);
        const size_t off = md.stage_off[t] + (size_t)(r - lo) * D;
        jobs.push_back({tables_[t].get(), uniq_narrow_ ? nullptr : h_uniq_keys_ + c.key_start[t] + r, re - r, h_staging_ + off, D,
                        params_.default_value_for_each_table[t], h_found_ + md.useg_start[t] + (r - lo),
                        uniq_narrow_ ? reinterpret_cast<const uint32_t*>(h_uniq_keys_) + c.key_start[t] + r : nullptr,
                        uniq_narrow_ ? c.key_base[t] : 0});
        piece_begin = std::min(piece_begin, off);
        piece_end = std::max(piece_end, off + (size_t)(re - r) * D);
        if (piece_end - piece_begin >= piece_floats) {
          HPS_RETURN_IF_ERROR(flush());
          piece_floats = std::min(kPieceFloats, piece_floats * 2);
        }
        r = re;
      }
    }
    HPS_RETURN_IF_ERROR(flush());
    if (!in_place) HIP_TRY(hipMemcpyAsync(d_found_, h_found_, uq, hipMemcpyHostToDevice, ss));
    const float* rows_src = in_place ? h_staging_dev_ : d_staging_;
    const uint8_t* found_src = in_place ? h_found_dev_ : d_found_;
    if (used_copy_stream && !side) {
      HIP_TRY(hipEventRecord(ev_copy_, copy_stream_));
      HIP_TRY(hipStreamWaitEvent(stream_, ev_copy_, 0));
    }

    static const bool kTrace = std::getenv("HPS_TRACE_TAIL") != nullptr;
    const auto tt0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tt0).count(); };
    float tr[6] = {0, 0, 0, 0, 0, 0};
    hipError_t e = hipSuccess;
    if (side) {
      // scatter next to K_G, outside the lane; the session's stream then waits for it
      Mark(ev_s0_, ss);
      e = LaunchMissScatter(d_call_, cache_->device_tables(), d_md_, work_, rows_src, ss, Kt(ev_s0_, ev_s1_));
      Mark(ev_s1_, ss);
      if (e != hipSuccess) return Error(Code::kInternal, "miss scatter launch failed: ", hipGetErrorString(e));
      HIP_TRY(hipEventRecord(ev_copy_, ss));
      HIP_TRY(hipStreamWaitEvent(stream_, ev_copy_, 0));
      tr[1] = since();
    } else {
      // Other sessions' kernels wait for our scatter (the lane) and their probes for our writer event.  Let the PCIe copies
      // (and this call's own hit gather) drain first, so that those waits cover the scatter and the insert kernel alone
      // (tens of microseconds) and not the millisecond of H2D queued ahead of them on this stream.
      // (a chunk read in place has nothing queued ahead of its scatter: no drain, one host wait for the whole call)
      if (!in_place) HIP_TRY(hipStreamSynchronize(stream_));
      tr[0] = since();
      if (exclusive_) cache_->LaneEnter(stream_);
      Mark(ev_s0_);
      e = LaunchMissScatter(d_call_, cache_->device_tables(), d_md_, work_, rows_src, stream_, Kt(ev_s0_, ev_s1_));
      Mark(ev_s1_);
      if (exclusive_) cache_->LaneLeave(stream_, ev_lane_[2]);
      tr[1] = since();
      if (e != hipSuccess) return Error(Code::kInternal, "miss scatter launch failed: ", hipGetErrorString(e));
    }
    uint32_t rows_seq = 0;
    if (defer) {
      // the call's rows are complete behind this point of the stream: that is what the caller waits for
      if (timing_) (void)hipEventRecord(ev_c1_, stream_);
      HPS_RETURN_IF_ERROR(PushWords(0, ev_done_, &rows_seq));
    }
    // Few missed rows (they stayed where the host gathered them): only every n-th such call of the session pays for the writer
    // window (config.h: gpucache_small_miss_insert_interval); the others return their rows and leave them uncached.
    // (missed rows that were uploaded but still scattered on the second stream — up to side_scatter_mb, a call at 99 % hit: every
    //  n/2-th call)
    // Only NEAR-ALL-HIT calls skip (at most one key in 64 missed): a cold or low-hit-rate cache — small online requests whose missed
    // rows fit in_place_kb whatever the hit rate — inserts on every call, as the reference does below its hit_rate_threshold
    // (docs/architecture.md:65-67).
    const bool near_all_hit = last_misses_ * 64 <= N;
    const uint32_t ins_every = !(defer && side && near_all_hit) ? 1u : in_place ? cache_->small_insert_interval() : std::max(1u, cache_->small_insert_interval() / 2);
    if (ins_every > 1 && (++small_calls_ % ins_every) != 0) {
      cache_->AddDropped(uq);
      HPS_RETURN_IF_ERROR(WaitPushedSeq(rows_seq, ev_done_));
      if (timing_) (void)hipEventElapsedTime(&last_scatter_ms_, ev_s0_, ev_s1_);
      return Status::Ok();
    }
    cache_->BeginWrite(stream_);
    tr[2] = since();
    if (exclusive_) cache_->LaneEnter(stream_);
    Mark(ev_i0_);
    e = LaunchCacheInsert(cache_->device_tables(), (uint32_t)T, d_md_, uq, d_call_->key_start, work_.uniq_keys,
                          rows_src, found_src, cache_->InsertStamps(epoch), d_acc_, cu, stream_, Kt(ev_i0_, ev_i1_));
    Mark(ev_i1_);
    if (exclusive_) cache_->LaneLeave(stream_, ev_lane_[3]);
    cache_->EndWrite(stream_);
    tr[3] = since();
    if (e != hipSuccess) return Error(Code::kInternal, "cache insert launch failed: ", hipGetErrorString(e));
    if (defer) {
      {
        std::lock_guard<std::mutex> lk(deferred_mu_);
        const Status ps2 = PushWords((uint32_t)kStatLines * kAccStride, ev_done2_, &deferred_seq_);
        if (!ps2.ok()) return ps2;
        deferred_pending_ = true;
        deferred_timed_ = timing_;
      }
      HPS_RETURN_IF_ERROR(WaitPushedSeq(rows_seq, ev_done_));
      tr[4] = since();
      if (kTrace && tr[4] > 3.0f)
        fprintf(stderr, "[hps tail] drained %.2f  scatter-enqueued %.2f  write-lock %.2f  insert-enqueued %.2f  rows done %.2f ms (fetch %.2f ms before; insert left behind)\n",
                tr[0], tr[1], tr[2], tr[3], tr[4], phase_ms_[1]);
      if (timing_) (void)hipEventElapsedTime(&last_scatter_ms_, ev_s0_, ev_s1_);
      return Status::Ok();
    }
    // staging is reused by the next chunk
    if (timing_) (void)hipEventRecord(ev_c1_, stream_);
    if (last) {   // the insert statistics ride the call's final synchronisation
      HPS_RETURN_IF_ERROR(PushWords((uint32_t)kStatLines * kAccStride));
      HPS_RETURN_IF_ERROR(WaitPushed());
    } else {
      HIP_TRY(hipStreamSynchronize(stream_));
    }
    tr[4] = since();
    if (kTrace && tr[4] > 3.0f)
      fprintf(stderr, "[hps tail] drained %.2f  scatter-enqueued %.2f  write-lock %.2f  insert-enqueued %.2f  done %.2f ms (fetch %.2f ms before)\n",
              tr[0], tr[1], tr[2], tr[3], tr[4], phase_ms_[1]);
    if (timing_) {
      (void)hipEventElapsedTime(&last_scatter_ms_, ev_s0_, ev_s1_);
      (void)hipEventElapsedTime(&last_insert_ms_, ev_i0_, ev_i1_);
    }
    for (size_t t = 0; t < T; ++t) done[t] = md.chunk_hi[t];
    if (last) break;
  }
  AddInsertStats();
  return Status::Ok();
}

}  // namespace hps
