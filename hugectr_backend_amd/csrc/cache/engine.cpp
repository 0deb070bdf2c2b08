#include "engine.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "kernels.h"

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
  std::lock_guard<std::mutex> lk(stat_mu_);
  return counters_;
}

// LRU epochs: one per lookup call, 32 bits.  Long before the counter wraps (at 10 k lookups/s that is five
// days) all stamps are folded back so that "smaller = older" keeps holding.  Rare and heavy-handed on purpose:
// the device is drained, every table's stamps are rewritten, the counter restarts above the kept span.
constexpr uint32_t kEpochRenormAt = 0xF0000000u;
constexpr uint32_t kEpochKeepSpan = 1u << 30;

uint32_t EmbeddingCache::NextEpoch() {
  const uint32_t e = epoch_.fetch_add(1, std::memory_order_relaxed) + 1;
  if (e >= kEpochRenormAt) {
    std::lock_guard<std::mutex> lk(order_mu_);
    const uint32_t cur = epoch_.load(std::memory_order_relaxed);
    if (cur >= kEpochRenormAt) {
      (void)hipSetDevice(cfg_.device_id_);
      (void)hipDeviceSynchronize();
      const uint32_t keep_from = cur - kEpochKeepSpan;
      for (const TableCacheDev& tb : h_tables_)
        (void)LaunchCacheRenorm(tb.stamps, (uint64_t)tb.num_buckets * kBucketSlots, keep_from, nullptr);
      (void)hipDeviceSynchronize();
      epoch_.store(kEpochKeepSpan + 2, std::memory_order_relaxed);
    }
    return epoch_.fetch_add(1, std::memory_order_relaxed) + 1;
  }
  return e;
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
                            const std::vector<std::shared_ptr<HostTable>>& tables, int device) {
  HPS_RETURN_IF_ERROR(RequireDevice(device));
  HIP_TRY(hipSetDevice(device));
  model_ = model;
  const size_t T = tables.size();
  if (T == 0 || T > (size_t)kMaxTables)
    return Error(Code::kInvalidArg, "model '", model, "': ", T, " tables (supported: 1..", kMaxTables, ")");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  cu_count_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  static_ = p.embedding_cache_type == EmbeddingCacheType::Static;

  cfg_.num_emb_table_ = T;
  cfg_.use_gpu_embedding_cache_ = true;
  cfg_.device_id_ = device;
  h_tables_.resize(T);
  HIP_TRY(hipEventCreateWithFlags(&last_write_, hipEventDisableTiming));

  for (size_t t = 0; t < T; ++t) {
    const uint32_t D = tables[t]->dim();
    const size_t R = tables[t]->size();
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
    int64_t* dk = nullptr; uint32_t* ds = nullptr; float* dr = nullptr;
    HPS_RETURN_IF_ERROR(DevAlloc(&dk, slots)); allocations_.push_back(dk);
    HPS_RETURN_IF_ERROR(DevAlloc(&ds, slots)); allocations_.push_back(ds);
    HPS_RETURN_IF_ERROR(DevAlloc(&dr, slots * (size_t)D)); allocations_.push_back(dr);
    HIP_TRY(LaunchCacheClear(dk, ds, slots, nullptr));
    tb.bucket_keys = dk; tb.stamps = ds; tb.rows = dr;
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
  HPS_RETURN_IF_ERROR(DevAlloc(&d_stats, 4));
  HIP_TRY(hipMemset(d_zero_ks, 0, sizeof(uint64_t) * ((size_t)kMaxTables + 1)));
  HIP_TRY(hipMemset(d_stats, 0, 4 * sizeof(uint32_t)));
  int64_t* d_keys = nullptr; float* d_rows = nullptr;
  size_t maxD = 1;
  for (size_t t = 0; t < T; ++t) maxD = std::max<size_t>(maxD, tables[t]->dim());
  size_t chunk_rows = std::min(chunk_rows_max, std::max<size_t>(1, kStagingCapBytes / (maxD * sizeof(float))));
  HPS_RETURN_IF_ERROR(DevAlloc(&d_keys, chunk_rows));
  HPS_RETURN_IF_ERROR(DevAlloc(&d_rows, chunk_rows * maxD));
  std::vector<int64_t> hk;
  std::vector<float> hr;
  const uint32_t epoch = 1;
  Status st = Status::Ok();
  for (size_t t = 0; t < T && st.ok(); ++t) {
    const HostTable& ht = *tables[t];
    const uint32_t D = ht.dim();
    const size_t want = std::min(cfg_.capacity_rows_[t], ht.size());
    for (size_t r0 = 0; r0 < want && st.ok(); r0 += chunk_rows) {
      const size_t n = std::min(chunk_rows, want - r0);
      // canonical rows only: a key repeated in the file is represented by its last row
      hk.clear(); hr.clear();
      const int64_t* src_keys = ht.keys() + r0;
      const float* src_rows = ht.row_at(r0);
      const bool contiguous = !ht.has_duplicate_keys();
      size_t m = n;
      if (!contiguous) {
        for (size_t i = 0; i < n; ++i) {
          if (ht.Find(src_keys[i]) != (int64_t)(r0 + i)) continue;
          hk.push_back(src_keys[i]);
          hr.insert(hr.end(), ht.row_at(r0 + i), ht.row_at(r0 + i) + D);
        }
        m = hk.size(); src_keys = hk.data(); src_rows = hr.data();
      }
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
      const hipError_t e = LaunchCacheInsert(d_warm, (uint32_t)T, d_md, m, d_zero_ks, d_keys, d_rows, nullptr, epoch,
                                             d_stats, cu_count_, nullptr);
      if (e != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        st = Error(Code::kInternal, "cache warm-up: insert kernel failed: ", hipGetErrorString(e));
        break;
      }
    }
  }
  uint32_t stats[4] = {0, 0, 0, 0};
  (void)hipMemcpy(stats, d_stats, sizeof stats, hipMemcpyDeviceToHost);
  {
    std::lock_guard<std::mutex> lk(stat_mu_);
    counters_.dropped += stats[0];
    counters_.inserted += stats[1];
    counters_.refreshed += stats[2];
  }
  (void)hipFree(d_keys); (void)hipFree(d_rows); (void)hipFree(d_md); (void)hipFree(d_zero_ks); (void)hipFree(d_stats);
  (void)hipFree(d_warm);
  epoch_.store(1);
  if (const char* e = std::getenv("HPS_TEST_EPOCH_START")) epoch_.store((uint32_t)std::strtoul(e, nullptr, 0));  // test hook: epoch wrap
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
  uint32_t* d_counts = nullptr;   // [0] misses, [1..T] unique per table, [kMaxTables+1..+4] insert statistics
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
  for (void* p : {(void*)I.d_keys, (void*)I.d_key_start, (void*)I.d_counts, (void*)I.d_md, (void*)I.d_staging, (void*)I.d_found})
    if (p) (void)hipFree(p);
  delete dins_;
  dins_ = nullptr;
}

// Part 1, on the calling lookup's thread: claim the (single) job slot and snapshot the session's unique missed keys
// with copies enqueued on the session's stream (ordered after its dedup kernels, before its next call reuses them).
Status EmbeddingCache::SubmitDirectInsert(hipStream_t session_stream, const uint64_t* d_key_start, const int64_t* d_uniq_keys,
                                          const uint32_t* d_counts, const uint32_t* h_counts_override, uint64_t N,
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
    HPS_RETURN_IF_ERROR(DevAlloc(&dins_->d_counts, (size_t)kMaxTables + 8));
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
  if (h_counts_override)  // mixed call: only the async tables' counts (pinned host words, stable until the call's final sync)
    HIP_TRY(hipMemcpyAsync(I.d_counts, h_counts_override, (1 + T) * sizeof(uint32_t), hipMemcpyHostToDevice, session_stream));
  else
    HIP_TRY(hipMemcpyAsync(I.d_counts, d_counts, (1 + T) * sizeof(uint32_t), hipMemcpyDeviceToDevice, session_stream));
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
  hipError_t e = LaunchMissDescBuild(d_tables_, (uint32_t)T, I.d_counts, I.d_md, I.d_counts + kMaxTables + 1, nullptr, I.stream);
  // not part of the foreground fetch chain: a small grid that takes its time must not hold up a lookup's fetch
  if (e == hipSuccess)
    e = LaunchPsFetchDirect(d_index_, (uint32_t)T, I.d_md, I.d_key_start, I.d_keys, I.d_staging, I.d_found, I.unique_total,
                            /*grid_blocks=*/32, I.stream);
  if (e != hipSuccess) return Error(Code::kInternal, "direct background fetch launch failed: ", hipGetErrorString(e));
  HIP_TRY(hipStreamSynchronize(I.stream));
  BeginWrite(I.stream);
  e = LaunchCacheInsert(d_tables_, (uint32_t)T, I.d_md, I.unique_total, I.d_key_start, I.d_keys, I.d_staging, I.d_found, epoch,
                        I.d_counts + kMaxTables + 1, cu_count_, I.stream);
  EndWrite(I.stream);
  if (e != hipSuccess) return Error(Code::kInternal, "direct background insert launch failed: ", hipGetErrorString(e));
  uint32_t st[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(st, I.d_counts + kMaxTables + 1, sizeof st, hipMemcpyDeviceToHost, I.stream));
  HIP_TRY(hipStreamSynchronize(I.stream));
  std::lock_guard<std::mutex> lk2(stat_mu_);
  counters_.dropped += st[0];
  counters_.inserted += st[1];
  counters_.refreshed += st[2];
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
  const size_t slots = (size_t)h_tables_[table].num_buckets * kBucketSlots;
  std::vector<int64_t> all(slots);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(all.data(), h_tables_[table].bucket_keys, slots * sizeof(int64_t), hipMemcpyDeviceToHost));
  keys->clear();
  for (int64_t k : all) if (k != HPS_EMPTY_KEY) keys->push_back(k);
  return Status::Ok();
}

// =================================================================================================
// LookupSession
// =================================================================================================
LookupSession::~LookupSession() { Release(); }

void LookupSession::Release() {
  if (!cache_) return;
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  cache_->ForgetReader(ev_read_);  // our reader event may still be registered with the cache
  cache_->ForgetReader(ev_probe_);
  cache_->ForgetFetch(ev_fetch_);
  auto hfree = [](void* p) { if (p) (void)hipHostFree(p); };
  auto dfree = [](void* p) { if (p) (void)hipFree(p); };
  hfree(h_keys_pinned_); dfree(d_keys_); hfree(h_call_); dfree(d_call_); hfree(h_call_probe_); dfree(d_call_probe_);
  if (ev_g0_) (void)hipEventDestroy(ev_g0_);
  if (ev_g1_) (void)hipEventDestroy(ev_g1_);
  hfree(h_md_); dfree(d_md_);
  dfree(d_slot_); dfree(d_block_miss_); dfree(d_set_); dfree(d_counts_); hfree(h_counts_); hfree(h_mode_);
  dfree(d_uniq_keys_); hfree(h_uniq_keys_); hfree(h_staging_); dfree(d_staging_); hfree(h_found_); dfree(d_found_);
  if (ev_done_) (void)hipEventDestroy(ev_done_);
  if (ev_read_) (void)hipEventDestroy(ev_read_);
  if (ev_fetch_) (void)hipEventDestroy(ev_fetch_);
  if (ev_t0_) (void)hipEventDestroy(ev_t0_);
  if (ev_t1_) (void)hipEventDestroy(ev_t1_);
  if (ev_f0_) (void)hipEventDestroy(ev_f0_);
  if (ev_f1_) (void)hipEventDestroy(ev_f1_);
  if (ev_c1_) (void)hipEventDestroy(ev_c1_);
  if (ev_probe_) (void)hipEventDestroy(ev_probe_);
  if (ev_copy_) (void)hipEventDestroy(ev_copy_);
  if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
  if (stream_) (void)hipStreamDestroy(stream_);
  stream_ = nullptr;
  cache_.reset();
}

Status LookupSession::Init(HierParameterServer* ps, const InferenceParams& p, std::shared_ptr<EmbeddingCache> cache) {
  ps_ = ps;
  params_ = p;
  tables_ = ps->tables_of(p.model_name);
  const size_t T = tables_.size();
  if (T == 0) return Error(Code::kNotFound, "model '", p.model_name, "' has no tables loaded in the parameter server");
  size_t per_sample = 0;
  for (size_t c : p.maxnum_catfeature_query_per_table_per_sample) per_sample += c;
  max_keys_ = p.max_batchsize * per_sample;  // model_instance_state.cpp:98-99
  if (max_keys_ == 0) return Error(Code::kInvalidArg, "model '", p.model_name, "': max_batch_size * sum(maxnum_catfeature...) is 0");
  if (max_keys_ >= (1ull << 31) - 2) return Error(Code::kUnsupported, "more than 2^31 keys per request are not supported");
  if (!p.use_gpu_embedding_cache) return Status::Ok();  // host-tier session: no device state at all

  if (!cache) return Error(Code::kInvalidArg, "model '", p.model_name, "' uses the GPU cache but no EmbeddingCache was given");
  cache_ = std::move(cache);
  device_ = cache_->device();
  HIP_TRY(hipSetDevice(device_));
  HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&ev_copy_, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&ev_done_, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&ev_read_, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&ev_fetch_, hipEventDisableTiming));
  HIP_TRY(hipEventCreate(&ev_t0_));
  HIP_TRY(hipEventCreate(&ev_t1_));
  HIP_TRY(hipEventCreate(&ev_f0_));
  HIP_TRY(hipEventCreate(&ev_f1_));
  HIP_TRY(hipEventCreate(&ev_c1_));
  HIP_TRY(hipEventCreateWithFlags(&ev_probe_, hipEventDisableTiming));

  HPS_RETURN_IF_ERROR(PinAlloc(&h_keys_pinned_, max_keys_));
  HPS_RETURN_IF_ERROR(DevAlloc(&d_keys_, max_keys_));
  HPS_RETURN_IF_ERROR(PinAlloc(&h_call_, 1));
  HPS_RETURN_IF_ERROR(DevAlloc(&d_call_, 1));
  HPS_RETURN_IF_ERROR(PinAlloc(&h_call_probe_, 1));
  HPS_RETURN_IF_ERROR(DevAlloc(&d_call_probe_, 1));
  HIP_TRY(hipEventCreate(&ev_g0_));
  HIP_TRY(hipEventCreate(&ev_g1_));
  if (const char* e = std::getenv("HPS_SPLIT_PROBE")) split_probe_ = std::strtol(e, nullptr, 10) != 0;   // A/B switch
  HPS_RETURN_IF_ERROR(PinAlloc(&h_md_, 1));
  HPS_RETURN_IF_ERROR(DevAlloc(&d_md_, 1));
  HPS_RETURN_IF_ERROR(DevAlloc(&d_slot_, max_keys_));
  probe_blocks_cap_ = ProbeGridBlocks(max_keys_, cache_->cu_count());
  HPS_RETURN_IF_ERROR(DevAlloc(&d_block_miss_, probe_blocks_cap_));
  set_cap_ = 1024;
  while (set_cap_ < 2 * (uint64_t)max_keys_) set_cap_ <<= 1;
  HPS_RETURN_IF_ERROR(DevAlloc(&d_set_, set_cap_));
  HPS_RETURN_IF_ERROR(DevAlloc(&d_counts_, (size_t)kCountWords));
  HPS_RETURN_IF_ERROR(PinAlloc(&h_counts_, (size_t)kCountWords));
  HPS_RETURN_IF_ERROR(PinAlloc(&h_mode_, (size_t)2 * kMaxTables + 2));
  HIP_TRY(hipMemset(d_counts_, 0, (size_t)kCountWords * sizeof(uint32_t)));
  HPS_RETURN_IF_ERROR(DevAlloc(&d_uniq_keys_, max_keys_));
  HPS_RETURN_IF_ERROR(PinAlloc(&h_uniq_keys_, max_keys_, hipHostMallocMapped));
  void* dv = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&dv, h_uniq_keys_, 0));
  h_uniq_keys_devptr_ = (int64_t*)dv;
  if (cache_->direct()) {
    // the device sizes the staging layout itself (hps_missdesc_build), so the buffer must hold the worst case:
    // every key of a full batch missing.  Device memory only — no pinned host staging in this mode.
    size_t worst = 4 * T;
    for (size_t t = 0; t < T; ++t)
      worst += p.max_batchsize * p.maxnum_catfeature_query_per_table_per_sample[t] * (size_t)tables_[t]->dim();
    HPS_RETURN_IF_ERROR(DevAlloc(&d_staging_, worst));
    HPS_RETURN_IF_ERROR(DevAlloc(&d_found_, max_keys_));
    staging_floats_ = worst;
    staging_uniq_ = max_keys_;
  }
  HIP_TRY(hipDeviceSynchronize());
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
      want = std::max(floats, staging_floats_ * 2);
      want = std::max<size_t>(want, 1u << 16);
      HPS_RETURN_IF_ERROR(DevAlloc(&d_staging_, want));
      staging_floats_ = want;
    }
    HPS_RETURN_IF_ERROR(PinAlloc(&h_staging_, staging_floats_));
  }
  if (uniq > staging_uniq_ || (!h_found_ && uniq > 0)) {
    HIP_TRY(hipStreamSynchronize(stream_));
    if (h_found_) (void)hipHostFree(h_found_);
    h_found_ = nullptr;
    if (uniq > staging_uniq_) {
      if (d_found_) (void)hipFree(d_found_);
      d_found_ = nullptr;
      size_t want = std::max(uniq, staging_uniq_ * 2);
      want = std::max<size_t>(want, 1u << 12);
      HPS_RETURN_IF_ERROR(DevAlloc(&d_found_, want));
      staging_uniq_ = want;
    }
    HPS_RETURN_IF_ERROR(PinAlloc(&h_found_, staging_uniq_));
  }
  return Status::Ok();
}

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
  // stage keys: pageable -> pinned (the reference memcpy's into its key buffer too, hps.cc:595-597,
  // but its "PIN" buffer is plain malloc: hps_buffer.hpp:114-123) -> one async H2D copy.
  size_t off = 0;
  for (size_t t = 0; t < num_tables; ++t) {
    if (num_keys_per_table[t]) {
      if (!h_keys_per_table[t]) return Error(Code::kInvalidArg, "lookup: null key pointer for table ", t);
      memcpy(h_keys_pinned_ + off, h_keys_per_table[t], num_keys_per_table[t] * sizeof(int64_t));
    }
    off += num_keys_per_table[t];
  }
  HIP_TRY(hipMemcpyAsync(d_keys_, h_keys_pinned_, N * sizeof(int64_t), hipMemcpyHostToDevice, stream_));
  return TimedLookupDevice(d_keys_, vectors_per_table, num_keys_per_table, num_tables);
}

// LookupDevice + (option "timing") the GPU-side span of the call: probe+gather start (after the waits on other
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
  return TimedLookupDevice(d_keys_flat, d_vectors_per_table, num_keys_per_table, num_tables);
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

Status LookupSession::LookupDevice(const int64_t* d_keys_flat, float* const* d_out, const size_t* n, size_t T) {
  // ---- call descriptor: the per-table slicing of ProcessRequest (model_instance_state.cpp:180-193) ----
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
  CallDesc& c = *h_call_;
  c.num_tables = (uint32_t)T;
  c.keys = d_keys_flat;
  uint64_t N = 0;
  for (size_t t = 0; t < T; ++t) {
    c.key_start[t] = N;
    N += n[t];
    c.out[t] = d_out[t];
    if (n[t] && !d_out[t]) return Error(Code::kInvalidArg, "lookup: null output pointer for table ", t);
    const uint32_t D = tables_[t]->dim();
    c.vec_ok[t] = ((D & 3u) == 0 && ((uintptr_t)d_out[t] & 15u) == 0) ? 1 : 0;
  }
  c.key_start[T] = N;
  c.total_keys = N;
  const uint32_t epoch = cache_->NextEpoch();
  c.epoch = epoch;
  const size_t desc_bytes = sizeof(CallDesc);
  HIP_TRY(hipMemcpyAsync(d_call_, h_call_, desc_bytes, hipMemcpyHostToDevice, stream_));

  // (option "host_gather": serve this session's misses the reference's way — host threads + H2D copy — although the
  //  cache is in ps_direct_access mode; the pinned tables serve both paths)
  const bool use_direct = cache_->direct() && !force_host_gather_;
  const bool fast_direct = use_direct && params_.hit_rate_threshold >= 1.0f && last_misses_ > 0;
  // Split probe (host-gather tier, while calls keep missing): K_A only probes (75 us instead of 290), the miss counts
  // reach the host right after the dedup, and the hit rows are gathered by K_G while the host threads gather the
  // missed rows and the DMA engine uploads them — HBM-bound and PCIe-bound halves of one call side by side.
  // (device-driven tier: measured and left fused.  With K_G on the session's stream and the staging layout + fetch kernel
  //  on the second one, config 2 dropped from 1.85 to 1.70 G lookups/s: the fetch kernel then shares the memory system with
  //  its own session's K_G as well as the other session's kernels, and falls from 0.78 to 0.70 of the PCIe peak.)
  const bool split = split_probe_ && !use_direct && last_misses_ > 0 && N > 0;
  bool all128 = true;
  const CallDesc* d_probe_call = d_call_;
  last_gather_ms_ = 0.f;
  if (split) {
    *h_call_probe_ = c;
    for (size_t t = 0; t < T; ++t) {
      h_call_probe_->out[t] = nullptr;
      all128 &= tables_[t]->dim() == 128 && c.vec_ok[t];
    }
    HIP_TRY(hipMemcpyAsync(d_call_probe_, h_call_probe_, desc_bytes, hipMemcpyHostToDevice, stream_));
    d_probe_call = d_call_probe_;
  }

  // ---- K_A: probe + gather hits ----
  const int cu = cache_->cu_count();
  const uint32_t probe_blocks = ProbeGridBlocks(N, cu, probe_balanced_);
  cache_->BeginRead(stream_);
  if (timing_) (void)hipEventRecord(ev_t0_, stream_);
  hipError_t e = LaunchProbeGather(d_probe_call, cache_->device_tables(), (uint32_t)T, N, d_slot_, d_block_miss_, probe_blocks,
                                   probe_unroll_, stream_);
  if (timing_) (void)hipEventRecord(ev_t1_, stream_);
  if (split) (void)hipEventRecord(ev_probe_, stream_);   // other sessions' probes chain behind the probe alone
  else cache_->EndRead(stream_, ev_read_);
  // (split: the cache stays read-locked until K_G has read the slots; released below on every path)
  auto end_split_read = [&]() { if (split) cache_->EndReadFused(stream_, ev_probe_, ev_read_); };
  if (e != hipSuccess) { end_split_read(); return Error(Code::kInternal, "probe/gather launch failed: ", hipGetErrorString(e)); }

  // ---- K_B: unique missed keys (all three kernels exit at once when nothing missed) ----
  e = LaunchMissDedup(d_call_, c.key_start, (uint32_t)T, probe_blocks, d_slot_, d_block_miss_, d_set_, set_cap_,
                      d_counts_, d_uniq_keys_, h_uniq_keys_devptr_, cu, stream_);
  if (e != hipSuccess) { end_split_read(); return Error(Code::kInternal, "miss dedup launch failed: ", hipGetErrorString(e)); }
  last_async_ = false;
  auto account = [&]() {
    const uint64_t misses = h_counts_[0];
    uint64_t uniq = 0;
    for (size_t t = 0; t < T; ++t) uniq += h_counts_[1 + t];
    last_misses_ = misses;
    last_unique_ = uniq;
    std::lock_guard<std::mutex> lk(cache_->stat_mu_);
    cache_->counters_.lookups += 1;
    cache_->counters_.keys += N;
    cache_->counters_.misses += misses;
    cache_->counters_.unique_misses += uniq;
  };
  if (fast_direct) {
    // Device-driven miss path with the insertion policy fixed to "synchronous": nothing on the host depends
    // on the miss counts, so the whole call is enqueued without a round trip and the counts come back at the end.
    // (Only while the previous call of this session missed something: a fully resident working set is served
    // faster by reading the counts first and stopping there — four empty kernels and a sync less.)
    table_async_.assign(T, 0);
    const Status st = HandleMissesDirect(N, epoch, /*counts_known=*/false, nullptr);
    if (timing_) (void)hipEventElapsedTime(&last_gpu_ms_, ev_t0_, ev_t1_);
    if (st.ok()) account();
    phase_ms_[3] = ms_since(tc0);
    phase_ms_[2] = phase_ms_[3];  // no host phases on this path; [1] holds the fetch kernel's GPU time
    return st;
  }
  if (timing_) (void)hipEventRecord(ev_c1_, stream_);
  HIP_TRY(hipMemcpyAsync(h_counts_, d_counts_, ((size_t)kTableMissBase + T) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
  (void)hipEventRecord(ev_done_, stream_);
  if (split) {
    // K_G behind the counts: it runs while the host reads them and works on the misses
    if (timing_) (void)hipEventRecord(ev_g0_, stream_);
    e = LaunchGatherHits(d_call_, cache_->device_tables(), (uint32_t)T, N, d_slot_, probe_blocks, all128, stream_);
    if (timing_) (void)hipEventRecord(ev_g1_, stream_);
    end_split_read();
    if (e != hipSuccess) return Error(Code::kInternal, "hit gather launch failed: ", hipGetErrorString(e));
  }
  HIP_TRY(hipEventSynchronize(ev_done_));
  if (timing_) (void)hipEventElapsedTime(&last_gpu_ms_, ev_t0_, ev_t1_);
  account();
  const uint64_t misses = h_counts_[0];
  phase_ms_[0] = phase_ms_[3] = ms_since(tc0);
  table_async_.assign(T, 0);
  split_call_ = split;
  if (misses == 0) {
    if (split) {
      HIP_TRY(hipStreamSynchronize(stream_));   // the hit rows are still being written
      if (timing_) (void)hipEventElapsedTime(&last_gather_ms_, ev_g0_, ev_g1_);
    }
    return Status::Ok();
  }

  // ---- insertion policy, decided per table as the reference's per-table lookup loop does
  //      (docs/architecture.md:65-67): a table whose hit rate in this call reaches
  //      hit_rate_threshold returns the default vector for its misses now and has them inserted in the
  //      background; the other tables fetch, return and insert their missed rows before the call returns ----
  bool any_async = false, any_sync = false;
  for (size_t t = 0; t < T; ++t) {
    const uint64_t n_t = c.key_start[t + 1] - c.key_start[t];
    const uint64_t m_t = h_counts_[kTableMissBase + t];
    if (m_t == 0 || n_t == 0) continue;
    const bool as = 1.0 - (double)m_t / (double)n_t >= (double)params_.hit_rate_threshold;
    table_async_[t] = as ? 1 : 0;
    any_async |= as;
    any_sync |= !as;
  }
  const uint32_t* d_mode = nullptr;
  if (any_async && any_sync) {
    // mixed call: the per-table mode goes to the kernels through the words the per-table miss counts came back in
    for (size_t t = 0; t < T; ++t) h_mode_[t] = table_async_[t];
    HIP_TRY(hipMemcpyAsync(d_counts_ + kTableMissBase, h_mode_, T * sizeof(uint32_t), hipMemcpyHostToDevice, stream_));
    d_mode = d_counts_ + kTableMissBase;
  }
  if (any_async) {
    last_async_ = true;
    e = LaunchMissFillDefault(d_call_, cache_->device_tables(), N, d_slot_, d_mode, cu, stream_);
    if (e != hipSuccess) return Error(Code::kInternal, "default fill launch failed: ", hipGetErrorString(e));
    // hand the async tables' unique missed keys to the background inserter (best effort)
    if (use_direct) {
      // device-driven tier: the keys never leave the GPU; fetch + insert run on the cache's own stream
      uint64_t uniq = 0, floats = 0;
      uint32_t* job_counts = h_mode_ + kMaxTables;   // [0] any misses, [1 + t] unique misses of the async tables
      job_counts[0] = 1;
      for (size_t t = 0; t < T; ++t) {
        const uint32_t cnt = table_async_[t] ? h_counts_[1 + t] : 0u;
        job_counts[1 + t] = cnt;
        uniq += cnt;
        floats = ((floats + 3) & ~(uint64_t)3) + (uint64_t)cnt * tables_[t]->dim();
      }
      bool accepted = false;
      HPS_RETURN_IF_ERROR(cache_->SubmitDirectInsert(stream_, d_call_->key_start, d_uniq_keys_, d_counts_, any_sync ? job_counts : nullptr,
                                                     N, uniq, floats, &accepted));
      if (accepted) ps_->RunDirectInsert(cache_);
    } else {
      std::vector<std::vector<int64_t>> job(T);
      for (size_t t = 0; t < T; ++t) {
        if (!table_async_[t]) continue;
        const uint32_t cnt = h_counts_[1 + t];
        job[t].assign(h_uniq_keys_ + c.key_start[t], h_uniq_keys_ + c.key_start[t] + cnt);
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
      return Status::Ok();
    }
  }
  const Status st = use_direct ? HandleMissesDirect(N, epoch, /*counts_known=*/true, d_mode) : HandleMisses(N, epoch);
  if (split && timing_ && st.ok()) (void)hipEventElapsedTime(&last_gather_ms_, ev_g0_, ev_g1_);
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

  CallDesc& c = *h_call_;
  c.num_tables = (uint32_t)T;
  c.keys = d_keys_flat;
  for (size_t t = 0; t < T; ++t) {
    c.key_start[t] = (uint64_t)t * batch;
    c.out[t] = nullptr;   // probe only
    c.vec_ok[t] = 0;
  }
  c.key_start[T] = N;
  c.total_keys = N;
  const uint32_t epoch = cache_->NextEpoch();
  c.epoch = epoch;
  HIP_TRY(hipMemcpyAsync(d_call_, h_call_, sizeof(CallDesc), hipMemcpyHostToDevice, stream_));
  const int cu = cache_->cu_count();
  const uint32_t probe_blocks = ProbeGridBlocks(N, cu, probe_balanced_);
  // bottom MLP first: it needs nothing from the lookup and leaves the stream before the cache is read-locked
  const void* d_bottom = nullptr;
  HPS_RETURN_IF_ERROR(dense->BottomMlp(d_dense_features, batch, stream_, &d_bottom));

  cache_->BeginRead(stream_);   // ---- read lock: held (order mutex + reader event) until the interaction is enqueued ----
  if (timing_) (void)hipEventRecord(ev_t0_, stream_);
  hipError_t e = LaunchProbeGather(d_call_, cache_->device_tables(), (uint32_t)T, N, d_slot_, d_block_miss_, probe_blocks,
                                   probe_unroll_, stream_);
  if (timing_) (void)hipEventRecord(ev_t1_, stream_);
  (void)hipEventRecord(ev_probe_, stream_);
  if (e == hipSuccess)
    e = LaunchMissDedup(d_call_, c.key_start, (uint32_t)T, probe_blocks, d_slot_, d_block_miss_, d_set_, set_cap_, d_counts_,
                        d_uniq_keys_, h_uniq_keys_devptr_, cu, stream_);
  if (e == hipSuccess)
    e = LaunchMissDescBuild(cache_->device_tables(), (uint32_t)T, d_counts_, d_md_, d_counts_ + kMaxTables + 1, nullptr, stream_);
  if (e == hipSuccess) {
    cache_->BeginFetch(stream_);
    if (timing_) (void)hipEventRecord(ev_f0_, stream_);
    e = LaunchPsFetchDirect(cache_->device_index(), (uint32_t)T, d_md_, d_call_->key_start, d_uniq_keys_, d_staging_, d_found_, N, 0,
                            stream_);
    if (timing_) (void)hipEventRecord(ev_f1_, stream_);
    cache_->EndFetch(stream_, ev_fetch_);
  }
  if (e == hipSuccess)
    e = LaunchLookupInteract(cache_->device_tables(), d_md_, d_slot_, d_staging_, d_bottom, batch, (uint32_t)T, dense->emb_dim(),
                             dense->out_stride(), d_out_f16, cu, stream_);
  cache_->EndReadFused(stream_, ev_probe_, ev_read_);   // ---- read lock released (enqueue side) ----
  if (e != hipSuccess) return Error(Code::kInternal, "lookup_interact launch failed: ", hipGetErrorString(e));
  last_async_ = false;
  table_async_.assign(T, 0);
  // insert the missed rows (writer window = the insert kernel alone, as in HandleMissesDirect)
  {
    const uint64_t est_bytes = (uint64_t)last_unique_ * dense->emb_dim() * sizeof(float);
    if (est_bytes > (2u << 20)) HIP_TRY(hipStreamSynchronize(stream_));
  }
  cache_->BeginWrite(stream_);

  e = LaunchCacheInsert(cache_->device_tables(), (uint32_t)T, d_md_, N, d_call_->key_start, d_uniq_keys_, d_staging_, d_found_, epoch,
                        d_counts_ + kMaxTables + 1, cu, stream_);
  cache_->EndWrite(stream_);
  if (e != hipSuccess) return Error(Code::kInternal, "cache insert launch failed: ", hipGetErrorString(e));
  if (timing_) (void)hipEventRecord(ev_c1_, stream_);
  HIP_TRY(hipMemcpyAsync(h_counts_, d_counts_, ((size_t)kMaxTables + 5) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
  HIP_TRY(hipStreamSynchronize(stream_));
  if (timing_) {
    (void)hipEventElapsedTime(&last_gpu_ms_, ev_t0_, ev_t1_);
    (void)hipEventElapsedTime(&phase_ms_[1], ev_f0_, ev_f1_);
    (void)hipEventElapsedTime(&last_gpu_call_ms_, ev_t0_, ev_c1_);
  }
  uint64_t uniq = 0;
  for (size_t t = 0; t < T; ++t) uniq += h_counts_[1 + t];
  last_misses_ = h_counts_[0];
  last_unique_ = uniq;
  std::lock_guard<std::mutex> lk(cache_->stat_mu_);
  cache_->counters_.lookups += 1;
  cache_->counters_.keys += N;
  cache_->counters_.misses += last_misses_;
  cache_->counters_.unique_misses += uniq;
  cache_->counters_.dropped += h_counts_[kMaxTables + 1];
  cache_->counters_.inserted += h_counts_[kMaxTables + 2];
  cache_->counters_.refreshed += h_counts_[kMaxTables + 3];
  return Status::Ok();
}

// Synchronous miss path, device-driven ("ps_direct_access"): the GPU resolves the unique missed keys through the
// device-resident index of the host tier and pulls the rows out of pinned host memory itself.
Status LookupSession::HandleMissesDirect(uint64_t N, uint32_t epoch, bool counts_known, const uint32_t* d_table_mode) {
  const size_t T = tables_.size();
  const int cu = cache_->cu_count();
  const uint64_t max_unique = counts_known ? (uint64_t)last_unique_ : N;
  hipStream_t fs = stream_;
  hipError_t e = LaunchMissDescBuild(cache_->device_tables(), (uint32_t)T, d_counts_, d_md_, d_counts_ + kMaxTables + 1,
                                     d_table_mode, fs);
  if (e == hipSuccess) {
    cache_->BeginFetch(fs);
    if (timing_) (void)hipEventRecord(ev_f0_, fs);
    e = LaunchPsFetchDirect(cache_->device_index(), (uint32_t)T, d_md_, d_call_->key_start, d_uniq_keys_, d_staging_,
                            d_found_, max_unique, 0, fs);
    if (timing_) (void)hipEventRecord(ev_f1_, fs);
    cache_->EndFetch(fs, ev_fetch_);
  }
  if (e == hipSuccess) e = LaunchMissScatter(d_call_, cache_->device_tables(), d_md_, N, d_slot_, d_staging_, cu, stream_);
  if (e != hipSuccess) return Error(Code::kInternal, "direct miss path launch failed: ", hipGetErrorString(e));
  // Keep the window in which other sessions' probes wait for our writer event down to the insert kernel: drain the
  // stream first, so the event is recorded behind the insert alone and not behind a millisecond of PCIe fetch.
  // Small requests skip the drain (their fetch is a few tens of microseconds, less than the host round trip):
  // the estimate is this call's unique-miss count when known, else the previous call's.
  {
    uint64_t row_bytes = 0;
    for (const auto& tb : tables_) row_bytes = std::max<uint64_t>(row_bytes, (uint64_t)tb->dim() * sizeof(float));
    const uint64_t est_bytes = (uint64_t)last_unique_ * row_bytes;
    if (est_bytes > (2u << 20)) HIP_TRY(hipStreamSynchronize(stream_));
  }
  cache_->BeginWrite(stream_);
  e = LaunchCacheInsert(cache_->device_tables(), (uint32_t)T, d_md_, max_unique, d_call_->key_start, d_uniq_keys_,
                        d_staging_, d_found_, epoch, d_counts_ + kMaxTables + 1, cu, stream_);
  cache_->EndWrite(stream_);
  if (e != hipSuccess) return Error(Code::kInternal, "cache insert launch failed: ", hipGetErrorString(e));
  if (timing_) (void)hipEventRecord(ev_c1_, stream_);
  HIP_TRY(hipMemcpyAsync(h_counts_, d_counts_, ((size_t)kMaxTables + 5) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
  HIP_TRY(hipStreamSynchronize(stream_));
  if (timing_) (void)hipEventElapsedTime(&phase_ms_[1], ev_f0_, ev_f1_);  // direct path: [1] = the fetch kernel (GPU time)
  std::lock_guard<std::mutex> lk(cache_->stat_mu_);
  cache_->counters_.dropped += h_counts_[kMaxTables + 1];
  cache_->counters_.inserted += h_counts_[kMaxTables + 2];
  cache_->counters_.refreshed += h_counts_[kMaxTables + 3];
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
    ucnt[t] = (t < table_async_.size() && table_async_[t]) ? 0u : h_counts_[1 + t];
    total_floats += (size_t)ucnt[t] * tables_[t]->dim();
    total_uniq += ucnt[t];
  }
  const size_t cap_floats = kStagingCapBytes / sizeof(float);
  HPS_RETURN_IF_ERROR(EnsureStaging(std::min(total_floats, cap_floats), total_uniq));
  HIP_TRY(hipMemsetAsync(d_counts_ + kMaxTables + 1, 0, 4 * sizeof(uint32_t), stream_));

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

    // ---- host parameter-server gather (multi-threaded) into pinned staging ----
    // Gather and upload in pieces of a few MB (runs of consecutive tables): the H2D copy of piece p runs
    // on the copy engine while the host threads gather piece p+1, so the PCIe time (the floor of this
    // path: every missed row crosses the link once) hides most of the DRAM-latency-bound gather.
    HIP_TRY(hipMemcpyAsync(d_md_, h_md_, sizeof(MissDesc), hipMemcpyHostToDevice, stream_));
    constexpr size_t kPieceFloats = (4u << 20) / sizeof(float);
    std::vector<HierParameterServer::FetchJob> jobs;
    size_t piece_begin = SIZE_MAX, piece_end = 0;
    unsigned piece_no = 0;
    bool used_copy_stream = false;
    auto flush = [&]() -> Status {
      if (jobs.empty()) return Status::Ok();
      const auto tf0 = std::chrono::steady_clock::now();
      HPS_RETURN_IF_ERROR(ps_->FetchMulti(jobs));
      phase_ms_[1] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tf0).count();
      // (A/B on the MI355X box: alternating the pieces between two copy streams was slower — 1.24 vs 1.06
      //  ms/step at two sessions — so every piece goes down the session's own stream.)
      constexpr bool kTwoCopyStreams = false;
      // split call: the session's stream is busy with K_G, the pieces go down the copy stream
      hipStream_t cs = (split_call_ || (kTwoCopyStreams && (piece_no++ & 1))) ? copy_stream_ : stream_;
      HIP_TRY(hipMemcpyAsync(d_staging_ + piece_begin, h_staging_ + piece_begin, (piece_end - piece_begin) * sizeof(float),
                             hipMemcpyHostToDevice, cs));
      used_copy_stream |= (cs == copy_stream_);
      jobs.clear();
      piece_begin = SIZE_MAX; piece_end = 0;
      return Status::Ok();
    };
    for (size_t t = 0; t < T; ++t) {
      const uint32_t lo = md.chunk_lo[t], hi = md.chunk_hi[t];
      if (hi == lo) continue;
      const uint32_t D = tables_[t]->dim();
      // a big table is cut into several pieces of its own
      const uint32_t rows_per_piece = (uint32_t)std::max<size_t>(1, kPieceFloats / D);
      for (uint32_t r = lo; r < hi; r += rows_per_piece) {
        const uint32_t re = std::min(hi, r + rows_per_piece);
        const size_t off = md.stage_off[t] + (size_t)(r - lo) * D;
        jobs.push_back({tables_[t].get(), h_uniq_keys_ + c.key_start[t] + r, re - r, h_staging_ + off, D,
                        params_.default_value_for_each_table[t], h_found_ + md.useg_start[t] + (r - lo)});
        piece_begin = std::min(piece_begin, off);
        piece_end = std::max(piece_end, off + (size_t)(re - r) * D);
        if (piece_end - piece_begin >= kPieceFloats) HPS_RETURN_IF_ERROR(flush());
      }
    }
    HPS_RETURN_IF_ERROR(flush());
    HIP_TRY(hipMemcpyAsync(d_found_, h_found_, uq, hipMemcpyHostToDevice, stream_));
    if (used_copy_stream) {
      HIP_TRY(hipEventRecord(ev_copy_, copy_stream_));
      HIP_TRY(hipStreamWaitEvent(stream_, ev_copy_, 0));
    }

    hipError_t e = LaunchMissScatter(d_call_, cache_->device_tables(), d_md_, N, d_slot_, d_staging_, cu, stream_);
    if (e != hipSuccess) return Error(Code::kInternal, "miss scatter launch failed: ", hipGetErrorString(e));
    // Other sessions' probes wait for our writer event.  Let the PCIe copy and the scatter drain first,
    // so that the window in which the cache is "being written" is the insert kernel alone (tens of
    // microseconds) and not insert + the millisecond of H2D queued ahead of it on this stream.
    HIP_TRY(hipStreamSynchronize(stream_));
    cache_->BeginWrite(stream_);
    e = LaunchCacheInsert(cache_->device_tables(), (uint32_t)T, d_md_, uq, d_call_->key_start, d_uniq_keys_,
                          d_staging_, d_found_, epoch, d_counts_ + kMaxTables + 1, cu, stream_);
    cache_->EndWrite(stream_);
    if (e != hipSuccess) return Error(Code::kInternal, "cache insert launch failed: ", hipGetErrorString(e));
    // staging is reused by the next chunk
    if (timing_) (void)hipEventRecord(ev_c1_, stream_);
    HIP_TRY(hipStreamSynchronize(stream_));
    for (size_t t = 0; t < T; ++t) done[t] = md.chunk_hi[t];
  }
  HIP_TRY(hipMemcpyAsync(h_counts_ + kMaxTables + 1, d_counts_ + kMaxTables + 1, 4 * sizeof(uint32_t),
                         hipMemcpyDeviceToHost, stream_));
  HIP_TRY(hipStreamSynchronize(stream_));
  std::lock_guard<std::mutex> lk(cache_->stat_mu_);
  cache_->counters_.dropped += h_counts_[kMaxTables + 1];
  cache_->counters_.inserted += h_counts_[kMaxTables + 2];
  cache_->counters_.refreshed += h_counts_[kMaxTables + 3];
  return Status::Ok();
}

}  // namespace hps
