// The HPS engine: the three objects the reference's backend shell talks to
// (/root/reference/hps_backend/include/backend.hpp:31-32, model_instance_state.hpp:33-35):
//
//   HierParameterServer  <->  HugeCTR::HierParameterServerBase   (call sites: backend.cpp:68-71,
//                              model_state.cpp:111,132,135,160,379-392,411)
//   EmbeddingCache       <->  HugeCTR::EmbeddingCacheBase        (model_instance_state.cpp:107-109,168-169)
//   LookupSession        <->  HugeCTR::LookupSessionBase         (model_instance_state.cpp:170-171,194-195)
//
// The implementation behind them is new: host tier = partitioned hash tables (ps/host_table.h),
// device tier = bucketed HBM cache driven by the HIP kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <regex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../common/config.h"
#include "../common/status.h"
#include "../ps/host_table.h"
#include "../ps/update_source.h"
#include "../ps/thread_pool.h"
#include "../dense/dense.h"
#include "device_types.h"
#include "kernels.h"
#include "direct_kernels.h"

#include <shared_mutex>

namespace hps {

class HierParameterServer;
class LookupSession;

struct EmbeddingCacheConfig {  // get_cache_config() of the reference (only num_emb_table_ is read there)
  size_t num_emb_table_ = 0;
  std::vector<uint32_t> embedding_vec_size_;
  std::vector<size_t> num_set_in_cache_;       // buckets per table
  std::vector<size_t> capacity_rows_;          // ceil(gpucacheper * R_t)
  std::vector<float> default_value_;
  bool use_gpu_embedding_cache_ = true;
  int device_id_ = 0;
};

struct CacheCounters {
  uint64_t lookups = 0;           // lookup calls
  uint64_t keys = 0;              // keys looked up
  uint64_t misses = 0;            // keys not resident at probe time
  uint64_t unique_misses = 0;
  uint64_t inserted = 0;
  uint64_t refreshed = 0;         // key already resident at insert time: row refreshed in place
  uint64_t dropped = 0;           // insert skipped: every slot of the bucket was hit in the current recency unit, or (admission) more recently than a new key's nominal age
  uint64_t async_calls = 0;       // lookups answered in async-insert mode
};

// Per-(model, device) GPU embedding cache shared by every lookup session of that model on that device
// (docs/architecture.md:20,29).
class EmbeddingCache {
 public:
  ~EmbeddingCache();
  const EmbeddingCacheConfig& get_cache_config() const { return cfg_; }
  int device() const { return cfg_.device_id_; }
  int shard() const { return shard_; }                 // -1: a replica (holds any key)
  uint32_t num_shards() const { return num_shards_; }
  const std::string& model_name() const { return model_; }
  uint32_t num_tables() const { return (uint32_t)cfg_.num_emb_table_; }
  const TableCacheDev* device_tables() const { return d_tables_; }
  const std::vector<TableCacheDev>& host_tables() const { return h_tables_; }
  int cu_count() const { return cu_count_; }
  // (collects the statistics of inserts that lookup sessions left running behind their last call first)
  CacheCounters counters() const;
  void AddStatLines(const uint32_t* lines);   // insert statistics of one launch series: kStatLines accumulator lines

  // slot index (>=0) or -1 per key, straight from the device tables; no LRU side effect (tests, refresh)
  Status Query(uint32_t table, const int64_t* h_keys, size_t n, int32_t* h_slots);
  // every resident key of one table (order unspecified)
  Status DumpKeys(uint32_t table, std::vector<int64_t>* keys);
  // Fetch `keys_per_table` from the parameter server and insert them (keys already resident get their
  // row refreshed in place).  Used by the async-insert path and by refresh_embedding_cache; runs on
  // the cache's own stream and staging buffers, serialised by a mutex.
  // pacing (refresh): pieces of at most piece_keys keys; while lookup sessions are calling, a piece that took t is followed by a
  // pause of t x (1 / link_share - 1) — the refresh takes link_share of the PCIe link and of the writer windows, the sessions
  // the rest — and the host gather runs on at most max_threads threads of the serving pool.  *row_bytes: bytes uploaded.
  struct InsertPacing { size_t piece_keys = 0; double link_share = 1.0; size_t max_threads = 0; };
  Status InsertKeys(HierParameterServer* ps, const std::vector<std::vector<int64_t>>& keys_per_table, const InsertPacing* pacing = nullptr,
                    uint64_t* row_bytes = nullptr);
  uint64_t calls_so_far() const { return calls_.load(std::memory_order_relaxed); }
  // rows a refresh has uploaded so far, live (piece by piece; CacheCounters::refreshed moves only at the end of a slice)
  uint64_t refresh_rows_uploaded() const { return refresh_rows_.load(std::memory_order_relaxed); }   // lookup calls of all sessions (the refresh paces itself by it)
  // blocks until every queued async insertion has finished
  void WaitAsync();

 private:
  friend class HierParameterServer;
  friend class LookupSession;
  EmbeddingCache() = default;
  // refresh bookkeeping (HierParameterServer::RefreshOne; guarded by refresh_mu_): per table the host table's load epoch and the
  // positions in its change log as of the warm-up / the last two refreshes (HostTable::ChangeMark)
  std::mutex refresh_mu_;
  std::vector<uint64_t> seen_epoch_, seen_prev_, seen_last_;
  std::atomic<uint64_t> refresh_rows_{0};
  // shard >= 0: this cache is shard `shard` of `num_shards` of a table-sharded model (ps.json "table_sharding": "hash"): it
  // is sized for, warmed with and only ever asked for the keys with mix64(key) mod num_shards == shard
  Status Init(const std::string& model, const InferenceParams& p, const std::vector<std::shared_ptr<HostTable>>& tables,
              int device, int shard = -1, uint32_t num_shards = 1);
  void Release();
  void FreeInserter();

  // ---- ordering of kernels from different sessions on the shared device tables ----
  // readers = probe/gather kernels, writers = insert/refresh kernels.  Enqueue-side only: the lock is
  // held while stream-waits and the kernel launch are enqueued, never while the GPU runs.
  void BeginRead(hipStream_t stream);                       // stream waits for the last writer
  void EndRead(hipStream_t stream, hipEvent_t reader_done); // records + registers the reader event
  void EndReadFused(hipStream_t stream, hipEvent_t probe_done, hipEvent_t reader_done);
  void BeginWrite(hipStream_t stream);                      // stream waits for last writer + all readers
  void EndWrite(hipStream_t stream);                        // records the writer event
  void ForgetReader(hipEvent_t reader_done);                // a session is going away: drop its event
  // ps_direct_access: one PCIe fetch kernel at a time per cache.  The link is the bottleneck of the miss path;
  // two fetches side by side only share it, while queued behind each other they alternate and every session
  // does its HBM work (probe, dedup, scatter, insert) under the other session's fetch.
  void BeginFetch(hipStream_t stream);
  void EndFetch(hipStream_t stream, hipEvent_t fetch_done);
  void ForgetFetch(hipEvent_t fetch_done);
  // One HBM-bound kernel group at a time per cache (the "lane"): probe pair, hit gather, miss scatter and insert of all
  // sessions are chained through events.  Two such kernels side by side only share the memory system — each takes as
  // long as both together — and a short one (a scatter of 10 us of work) that starts under another session's 220-us
  // gather is scheduled into the gather's last free wave slots and takes until its end.  Enqueue side only, like the
  // read/write brackets; lock order: order mutex, then lane mutex.
  void LaneEnter(hipStream_t stream);
  void LaneLeave(hipStream_t stream, hipEvent_t done);
  void ForgetLane(hipEvent_t done);
  // The cache's clock.  A call takes a TIME TOKEN at its start: (recency unit << 8) | (call counter & 0xFF).  The recency unit
  // advances with the cache's TURNOVER — the unique rows its lookups missed, in units of total slots / units_per_turnover_
  // (AdvanceClock) — not with the number of calls: how long ago "an old key" was last hit only means something relative to
  // how fast new rows arrive, and a horizon counted in calls that suits 1.7 M-key requests is nothing for 28 K-key ones.
  // Measured (tools/hit_rate_long.py on the GPU, tools/lru_sim.py as the model; headline workload, thousands of calls): with
  // the unit at 8 calls and new keys entering 256 calls old (rounds 2-3) the hit rate settles at 0.9491 after ~1,000 calls
  // — the 260 calls of bench.py's timed region never get there —; with new keys entering 2,560 calls old at 0.9589 (19 %
  // fewer missed rows); in turnovers (one = ~950 calls of that workload) that is 2.5.
  // HPS_LRU_AGE_SHIFT=s in the environment: the unit is 2^s calls instead (rounds 2-3; tests that count calls).
  uint32_t NextEpoch();
  void AdvanceClock(uint64_t missed_rows) {
    clock_rows_.fetch_add(missed_rows < total_slots_ ? missed_rows : total_slots_, std::memory_order_relaxed);   // (at most one turnover per call)
  }
  // recency stamp the kernels write for a call with time token `epoch` (device_types.h)
  uint32_t Stamp8(uint32_t epoch) const { return (epoch >> 8) % kStampMod; }
  // what the insert kernel gets: the current unit in byte 0, the stamp of a newly inserted key in byte 1 (insert_age_ units
  // in the past: scan-resistant insertion — a key seen once must be seen again before it outranks keys that were hit)
  // bits 16..23: the call counter's low byte, bits 24..27: admit_log2_ (the insert kernel's admission rule, kernels.hip)
  uint32_t InsertStamps(uint32_t epoch) const {
    const uint32_t now8 = Stamp8(epoch);
    return now8 | (((now8 + kStampMod - insert_age_) % kStampMod) << 8) | ((epoch & 0xFFu) << 16) | ((admit_log2_ & 15u) << 24);
  }
  // every n-th small-miss call of a session inserts (InferenceParams::small_miss_insert_interval; 1 when the admission rule is off)
  uint32_t small_insert_interval() const { return admit_log2_ == 0 ? 1u : small_interval_; }
  void AddDropped(uint64_t n) { std::lock_guard<std::mutex> lk(stat_mu_); counters_.dropped += n; }
  uint32_t small_interval_ = 4;
  uint32_t admit_log2_ = 4;   // HPS_LRU_ADMIT: a new key does not take a slot hit more recently than the insert age, except one
                              // new key in 2^this (0 = every new key takes the bucket's oldest slot, rounds 1-3's behaviour)
  uint32_t insert_age_ = 160; // recency units a newly inserted key is aged by (HPS_LRU_INSERT_AGE; 0 = plain LRU insertion),
                              // < kAgeSaturate: 2.5 turnovers (call clock: 32 units unless given)
  uint32_t units_per_turnover_ = 64;   // 192 usable units = 3 turnovers of horizon
  bool call_clock_ = false;   // HPS_LRU_AGE_SHIFT given: recency unit = 2^age_shift_ calls
  uint32_t age_shift_ = 3;
  uint64_t total_slots_ = 1, rows_per_unit_ = 1, call_start_ = 0;
  std::atomic<uint64_t> calls_{0}, clock_rows_{0};

  std::string model_;
  int shard_ = -1;
  uint32_t num_shards_ = 1;
  EmbeddingCacheConfig cfg_;
  std::vector<TableCacheDev> h_tables_;
  TableCacheDev* d_tables_ = nullptr;
  std::vector<void*> allocations_;
  int cu_count_ = 256;
  bool static_ = false;

  std::mutex order_mu_;
  hipEvent_t last_write_ = nullptr;
  bool has_write_ = false;
  std::vector<hipEvent_t> readers_;
  hipEvent_t last_reader_ = nullptr;        // most recent probe/gather of any session (probes are chained)
  hipStream_t last_reader_stream_ = nullptr;
  std::mutex lane_mu_;
  hipEvent_t last_lane_ = nullptr;
  hipStream_t last_lane_stream_ = nullptr;
  std::mutex fetch_mu_;
  hipEvent_t last_fetch_ = nullptr;         // most recent direct PCIe fetch of any session (fetches are chained)
  hipStream_t last_fetch_stream_ = nullptr;

  mutable std::mutex stat_mu_;
  CacheCounters counters_;
  // lookup sessions of this cache (registered in LookupSession::Init, removed in Release): counters() asks each for the
  // statistics of an insert kernel it left behind its last call (LookupSession::CollectDeferred)
  mutable std::mutex sess_mu_;
  mutable std::condition_variable sess_cv_;
  mutable int collectors_ = 0;   // counters() calls working on a copy of sessions_
  std::vector<LookupSession*> sessions_;
  void RegisterSession(LookupSession* s);
  void UnregisterSession(LookupSession* s);

  // ---- device-driven parameter-server tier ("ps_direct_access", direct_kernels.hip) ----
 public:
  bool direct() const { return direct_; }
  const PsIndexDev* device_index() const { return d_index_; }
  // (re)build the device index of every table whose host copy changed since the last build
  Status SyncDirectIndex(const std::vector<std::shared_ptr<HostTable>>& tables);
  // lookups hold this shared for the duration of a call; a table reload takes it exclusively, so that no kernel
  // is reading a pinned slab while it is being replaced
  std::shared_mutex& direct_mutex() { return direct_mu_; }
  // a pending reload holds new lookups at the door (glibc's rwlock prefers readers: overlapping sessions would
  // starve the writer otherwise)
  std::atomic<int>& direct_writers() { return direct_writers_; }
  // Async-insert mode of a direct cache: snapshot the calling session's unique missed keys (enqueued on the session's
  // stream) and fetch + insert them on the cache's own stream, no host thread involved.  Best effort like the host
  // inserter: *accepted = false when the previous job is still running (the batch's misses stay uncached).
  // d_acc_tables / h_acc_tables_override: the T per-table lines of a call's accumulator block (device_types.h); the
  // override (pinned host memory) carries the async tables' counts only when the call is mixed.
  Status SubmitDirectInsert(hipStream_t session_stream, const uint64_t* d_key_start, const int64_t* d_uniq_keys,
                            const uint32_t* d_acc_tables, const uint32_t* h_acc_tables_override /*pinned, optional*/,
                            uint64_t N, uint64_t unique_total, uint64_t staging_floats, bool* accepted);

 private:
  struct DirectInserter;
  DirectInserter* dins_ = nullptr;
  std::mutex dins_mu_;
  void FreeDirectInserter();
  Status FinishDirectInsert();  // second half of an accepted job; runs on a pool thread (HierParameterServer::RunDirectInsert)
  bool direct_ = false;
  std::atomic<int> direct_writers_{0};
  std::vector<PsIndexDev> h_index_;
  PsIndexDev* d_index_ = nullptr;
  std::vector<uint64_t> index_generation_;
  std::vector<std::pair<void*, void*>> index_mem_;  // per table: device keys[], rows[]
  std::shared_mutex direct_mu_;

  struct Inserter;  // stream + staging of the background insert path
  Inserter* ins_ = nullptr;
  std::mutex ins_mu_;
  std::mutex pend_mu_;
  std::condition_variable pend_cv_;
  int pending_async_ = 0;
};

// One lookup session = one worker buffer set + one HIP stream.  Thread-compatible: Triton never runs
// one instance concurrently (hps.cc:353-359); different sessions run concurrently.
class LookupSession {
 public:
  ~LookupSession();

  // The reference signature (docs/architecture.md:308-323, model_instance_state.cpp:194-195):
  // host key pointers in, device (gpucache) or host (no gpucache) vector pointers out; blocking.
  Status lookup(const void* const* h_keys_per_table, float* const* vectors_per_table,
                const size_t* num_keys_per_table, size_t num_tables);

  // Keys already resident in HBM, flat + table-major (the bench path: nothing crosses PCIe except
  // missed rows).  d_vectors_per_table are device pointers.  Blocking unless `stream_out` is given,
  // in which case the hit path is enqueued and the call returns after the miss path completed.
  Status lookup_from_device(const int64_t* d_keys_flat, float* const* d_vectors_per_table,
                            const size_t* num_keys_per_table, size_t num_tables);
  // BASELINE config 5 fused: `batch` samples, one key per table per sample (d_keys_flat table-major, T*batch keys);
  // result = the dense step's output [batch][out_stride] f16, no OUTPUT0.  ps_direct_access models, threshold 1.0.
  Status lookup_interact(DenseInteraction* dense, const int64_t* d_keys_flat, uint64_t batch, const float* d_dense_features,
                         void* d_out_f16);

  const InferenceParams& params() const { return params_; }
  bool uses_gpu_cache() const { return cache_ != nullptr; }
  // for the sharded session built on top of a lookup session (shard_session.h)
  hipStream_t stream() const { return stream_; }
  int device() const { return device_; }
  size_t max_keys() const { return max_keys_; }
  size_t num_tables() const { return tables_.size(); }
  uint32_t table_dim(size_t t) const { return tables_[t]->dim(); }
  float table_default(size_t t) const { return params_.default_value_for_each_table[t]; }
  // lookup_from_device for the padded key array of the sharded exchange: keys equal to HPS_EMPTY_KEY are padding (no probe,
  // no row written, no statistics); `padding` of the `num_keys` are such keys as far as the caller knows (0 if unknown:
  // the cache's key counter then includes them)
  Status lookup_from_device_padded(const int64_t* d_keys_flat, float* const* d_vectors_per_table, const size_t* num_keys_per_table,
                                   size_t num_tables);
  void discount_padding(uint64_t padding_keys);
  // lookup_from_device for ONE OWNER'S BUCKET of a table-sharded request (shard_entry.h): the row of key i is written to
  // d_vectors_per_table[t] + d_dst_index[i] * D_t — its place in the entry instance's output, which may sit on another GPU
  // (peer-mapped: the gather / scatter / default-fill kernels store over xGMI).  d_keys_flat / d_dst_index may be peer memory too.
  Status lookup_from_device_indexed(const int64_t* d_keys_flat, const uint32_t* d_dst_index, float* const* d_vectors_per_table,
                                    const size_t* num_keys_per_table, size_t num_tables);
  // last call's numbers
  uint64_t last_miss_count() const { return last_misses_; }
  uint64_t last_unique_miss_count() const { return last_unique_; }
  bool last_call_async() const { return last_async_; }
  uint64_t last_unique_key_count() const { return last_unique_keys_; }  // 0 unless the policy needed it
  float last_gpu_ms() const { return last_gpu_ms_; }      // probe + miss-unique kernels of the last call (HIP events)
  float last_gather_ms() const { return last_gather_ms_; }  // hit-gather kernel of the last call
  void set_split_probe(bool b) { split_probe_ = b; }
  void set_xcd_walk(bool b) { xcd_walk_ = b; }
  void set_chain_gather(bool b) { chain_gather_ = b; }
  void set_probe_in_lane(int v) { probe_in_lane_ = v; }
  void set_keys_by_kernel(int v) { keys_by_kernel_ = v; }
  void set_interact_mode(int v) { interact_mode_ = v; }        // lookup_interact: 0 separate steps, 1 fused, 2 by the last call's missed rows
  bool last_interact_separate() const { return last_interact_separate_; }
  void set_narrow_publish(bool b) { narrow_publish_ = b; }
  void set_exclusive_kernels(bool b) { exclusive_ = b; }
  void set_fused_unique(bool b) { fused_unique_ = b; }
  float last_key_stage_ms() const { return key_stage_ms_; }
  float last_scatter_ms() const { return last_scatter_ms_; }   // miss-scatter kernel of the last call (last chunk)
  float last_insert_ms() const { return last_insert_ms_; }     // cache-insert kernel of the last call (last chunk)
  void set_keys_pinned_check(bool b) { keys_pinned_hint_ = b ? 1 : 0; }
  void set_narrow_keys(int mode) { narrow_keys_ = mode != 0; pack24_keys_ = mode == 1; narrow_backoff_ = narrow24_backoff_ = 0; narrow_streak_ = narrow24_streak_ = 0; }   // 0 off, 1 uint32 + 3-byte packing, 2 uint32 only
  bool last_keys_narrow() const { return keys_narrow_; }
  int last_key_bytes() const { return key_bytes_; }   // bytes per key the last host-keys call moved over PCIe: 8, 4 or 3
  float last_gpu_call_ms() const { return last_gpu_call_ms_; }  // first kernel to last of the last call (HIP events)
  // host wall-clock phases of the last call (ms): [0] enqueue -> miss counts known, [1] parameter-server
  // gather, [2] H2D + scatter + insert until the stream drained, [3] whole call
  const float* last_phase_ms() const { return phase_ms_; }
  void set_probe_variant(int v) { probe_variant_ = v; }
  void set_force_host_gather(bool b) { force_host_gather_ = b; }
  void set_timing(bool on) { timing_ = on; }
  void set_defer_insert(bool b) { defer_insert_ = b; }
  void set_in_place_bytes(size_t b) { in_place_bytes_ = b; }
  void set_side_bytes(size_t b) { side_bytes_ = b; }
  // The insert kernel of a synchronous call is enqueued behind the call and NOT waited for (defer_insert_): the rows the
  // call returns are exact without it, and every later reader of the cache is ordered behind it by the cache's writer
  // event.  Its statistics (and, with option "timing", its duration) are collected here — at the start of the session's
  // next call, by EmbeddingCache::counters(), and when the session goes away.  Thread-safe.
  Status CollectDeferred();
  // per-session override of the model's hit_rate_threshold (sync vs async insertion, docs/architecture.md:65-67)
  void set_hit_rate_threshold(float v) { params_.hit_rate_threshold = v; }

 private:
  friend class HierParameterServer;
  LookupSession() = default;
  Status Init(HierParameterServer* ps, const InferenceParams& p, std::shared_ptr<EmbeddingCache> cache, size_t max_keys_override = 0);
  Status LookupHostTier(const void* const* h_keys_per_table, float* const* h_vectors_per_table,
                        const size_t* num_keys_per_table, size_t num_tables);
  Status LookupDevice(const int64_t* d_keys_flat, float* const* d_vectors_per_table, const size_t* n, size_t T);
  Status HandleMisses(uint64_t N, uint32_t epoch);
  Status HandleMissesDirect(uint64_t N, uint32_t epoch, bool counts_known, const uint32_t* d_table_mode);
  std::vector<uint8_t> table_async_;   // this call: 1 = the table's misses are served in async-insert mode
  uint32_t* h_mode_ = nullptr;         // pinned: [0, kMaxTables) per-table mode words, then kMaxTables accumulator lines
                                       // (the async tables' unique counts of a mixed call, for the background inserter)
  Status EnsureStaging(size_t floats, size_t uniq);
  void Release();

  HierParameterServer* ps_ = nullptr;
  InferenceParams params_;
  std::shared_ptr<EmbeddingCache> cache_;
  std::vector<std::shared_ptr<HostTable>> tables_;
  int device_ = 0;
  hipStream_t stream_ = nullptr;
  hipStream_t copy_stream_ = nullptr;  // second H2D queue for the missed-row pieces
  bool side_hi_ = false;               // copy_stream_ is a high-priority queue (the first two sessions of a device only)
  hipEvent_t ev_copy_ = nullptr;
  hipEvent_t ev_keys_ = nullptr;       // behind the key upload (second stream)
  bool keys_wait_pending_ = false;     // lookup() recorded ev_keys_; PrepareCall makes the first stream wait for it behind the block pull
  hipEvent_t ev_done_ = nullptr, ev_read_ = nullptr, ev_fetch_ = nullptr, ev_t0_ = nullptr, ev_t1_ = nullptr,
             ev_f0_ = nullptr, ev_f1_ = nullptr, ev_c1_ = nullptr, ev_probe_ = nullptr;
  float last_gpu_call_ms_ = 0.f;
  float stage_event_ms_ = 0.f;
  float stage_pool_ms_ = 0.f, stage_enqueue_ms_ = 0.f;   // key staging: time in the pool loops / in the H2D enqueues
  float key_stage_ms_ = 0.f;      // host side of lookup(): staging the keys and enqueueing their H2D copies
  bool narrow_keys_ = true;       // option "narrow_keys": stage pageable keys as uint32 when they all fit
  bool pack24_keys_ = true;       // ... and at 3 bytes each when they all fit 24 bits (option value 2 turns only this off)
  bool keys_narrow_ = false;      // this call's staged keys are uint32
  bool skip_empty_next_ = false;  // the call being prepared treats HPS_EMPTY_KEY as padding (lookup_from_device_padded)
  const uint32_t* dst_index_next_ = nullptr;   // the call being prepared writes its rows through this index (lookup_from_device_indexed)
  int narrow_backoff_ = 0;        // calls left before narrowing is tried again after a wide key was seen
  int narrow24_backoff_ = 0;      // the same for the 3-byte packing after a key of 25..32 bits
  int narrow_streak_ = 0, narrow24_streak_ = 0;   // failed attempts in a row: the pause doubles with each, 256 .. 65,536 calls
  void NarrowFailed(bool pack24) {
    int& streak = pack24 ? narrow24_streak_ : narrow_streak_;
    (pack24 ? narrow24_backoff_ : narrow_backoff_) = 256 << (streak < 8 ? streak : 8);
    if (streak < 8) ++streak;
  }
  int key_bytes_ = 8;
  int keys_pinned_hint_ = 1;      // 1: flat key arrays in page-locked memory are DMA'd in place (option "keys_pinned_check")
  Status TimedLookupDevice(const int64_t* d_keys_flat, float* const* d_out, const size_t* n, size_t T);

  size_t max_keys_ = 0;           // max_batch * sum(maxnum_catfeature)
  size_t max_tiles_ = 0;          // max_keys_ / kTileKeys + T (tiles never straddle tables)
  int64_t* h_keys_pinned_ = nullptr;
  const int64_t* h_keys_dev_ = nullptr;   // device view of h_keys_pinned_ (small requests: the probe reads the keys from there)
  int64_t* d_keys_ = nullptr;
  // The call block: CallDesc | accumulator block (zeros) | TileDesc[max_tiles_], one pinned image and one device
  // copy, uploaded with ONE H2D copy per call (every extra small copy or memset is a blit kernel on the stream).
  char* h_block_ = nullptr;       // pinned
  char* d_block_ = nullptr;
  size_t block_acc_off_ = 0, block_tiles_off_ = 0, acc_words_ = 0;
  CallDesc* h_call_ = nullptr;    // = h_block_
  CallDesc* d_call_ = nullptr;    // = d_block_
  TileDesc* h_tiles_ = nullptr;
  uint32_t* d_acc_ = nullptr;
  uint32_t* h_acc_ = nullptr;     // pinned mirror the accumulator block is copied back to
  // Control words over the compute queue (kernels.hip: hps_pull16 / hps_push_words) instead of SDMA copies: device views of
  // the two pinned images, the sequence word a push publishes last and the host polls.  HPS_ZC_CONTROL=0: hipMemcpyAsync.
  bool zc_control_ = true;
  const void* h_block_dev_ = nullptr;
  uint32_t* h_acc_dev_ = nullptr;
  uint32_t* h_seq_ = nullptr;      // = h_acc_ + acc_words_ (own 128-B line)
  uint32_t* h_seq_dev_ = nullptr;
  uint32_t push_seq_ = 0;
  // enqueue: d_acc_[0..words) -> h_acc_, then the sequence word; records `ev` (default ev_done_); *seq_out = the word's value
  Status PushWords(uint32_t words, hipEvent_t ev = nullptr, uint32_t* seq_out = nullptr, hipStream_t on = nullptr);
  Status WaitPushed();                // host: until the last PushWords has landed (or the stream reports an error)
  Status WaitPushedSeq(uint32_t seq, hipEvent_t ev);   // ... until the push that carried `seq` (or a later one) has landed
  uint32_t small_calls_ = 0;          // small-miss calls of this session so far (every small_insert_interval()-th inserts)
  bool defer_insert_ = true;          // option "defer_insert": the insert kernel is not on the call's return path
  size_t in_place_bytes_ = 1u << 20;  // option "in_place_kb": missed rows of a chunk up to this size are read by
                                      // the kernels where the host gathered them (page-locked staging), no upload
  size_t side_bytes_ = 16u << 20;     // option "side_scatter_mb": missed rows of a call up to this size are uploaded AND scattered on the
                                      // second stream, next to the hit gather and outside the kernel lane
  std::mutex deferred_mu_;
  bool deferred_pending_ = false;     // an insert + statistics push is in flight behind the last call
  uint32_t deferred_seq_ = 0;
  bool deferred_timed_ = false;
  hipEvent_t ev_done2_ = nullptr;     // behind the deferred statistics push
  CallWork work_{};               // device pointers of the per-call work arrays
  uint32_t* d_mode_ = nullptr;    // per-table insertion mode of a mixed call (1 = async)
  uint32_t call_tag_ = 0;
  hipEvent_t ev_g0_ = nullptr, ev_g1_ = nullptr;   // around the hit-gather kernel
  hipEvent_t ev_s0_ = nullptr, ev_s1_ = nullptr, ev_i0_ = nullptr, ev_i1_ = nullptr;   // around the miss scatter / the cache insert
  float last_scatter_ms_ = 0.f, last_insert_ms_ = 0.f;
  float last_gather_ms_ = 0.f;
  bool split_call_ = false;      // the call in progress gathers its hits on stream_ while the miss path runs (copies go down copy_stream_)
  bool split_probe_ = true;      // host-gather tier: start the miss path behind the probe, gather the hits meanwhile (§3.4c);
                                 // session option split_probe=0: gather first, then the counts
  // option "timing": the per-kernel times of a call are the kernels' OWN start / stop timestamps (KTimer, kernels.h), as a
  // profiler reports them (hipEventRecord pairs around the launches also time two packet hand-offs per kernel: round 3's A/B)
  bool kernel_stamps_ = true;
  KTimer Kt(hipEvent_t a, hipEvent_t b) const { return (timing_ && kernel_stamps_) ? KTimer{a, b} : KTimer{}; }
  void Mark(hipEvent_t e, hipStream_t s = nullptr) { if (timing_ && !kernel_stamps_) (void)hipEventRecord(e, s ? s : stream_); }
  bool fused_unique_ = true;     // the call-wide unique misses are found in the probe kernel's tail (option "fused_unique")
  bool exclusive_ = true;        // the HBM-bound kernels of this session take the cache's lane (option "exclusive_kernels")
  hipEvent_t ev_lane_[4] = {nullptr, nullptr, nullptr, nullptr};   // probe pair, hit gather, miss scatter, insert
  bool probe_xcd_tiles_ = true;      // probe kernel: XCD x takes the x-th eighth of the tiles (round 3's A/B: profiles/round3/ab_probe_xcd_tiles.txt)
  bool frame_of_reference_ = true;   // narrowed keys are offsets from their table's smallest key (key_pack.h)
  std::vector<int64_t> key_base_;    // this call's per-table bases
  bool direct_split_ = true;     // device-driven tier: the fetch kernel runs next to the call's own hit gather (round 3: +4 %)
  bool narrow_publish_ = true;   // a narrowed request's unique missed keys come back to the host as uint32 (option "narrow_publish")
  bool uniq_narrow_ = false;     // this call: h_uniq_keys_ holds uint32 keys
  int interact_mode_ = 2;         // option "interact_mode" (lookup_interact): 0 = lookup into a buffer of the session + the dense step, 1 = the fused
                                 // arrangement, 2 (default) = separate while the session's calls miss much (last call's missed rows > side_bytes_)
  bool last_interact_separate_ = false;
  float* d_interact_emb_ = nullptr;      // OUTPUT0 of the separate arrangement (allocated at its first use)
  size_t interact_emb_floats_ = 0;
  int keys_by_kernel_ = 2;       // option "keys_by_kernel": staged keys are pulled into HBM by a kernel instead of copy-engine copies:
                                 // 0 never, 1 always, 2 (default) while the session's calls miss much (last call's missed rows > side_bytes_)
  int probe_in_lane_ = 2;        // option "probe_in_lane": 1 = K_P takes its turn in the kernel lane, 0 = it runs next to another session's
                                 // K_G, 2 (default) = next to it while the session's calls miss little (last call's missed rows <= side_bytes_)
  bool chain_gather_ = false;    // other sessions' probes queue behind this session's gather as well as its probe
  bool xcd_walk_ = true;         // K_G: each XCD sweeps its own eighth of the key range (option "xcd_walk" 0: plain grid stride)
  MissDesc* h_md_ = nullptr;      // pinned
  const MissDesc* h_md_dev_ = nullptr;     // device views of the pinned miss descriptor / staging rows / found flags (small chunks)
  const float* h_staging_dev_ = nullptr;
  const uint8_t* h_found_dev_ = nullptr;
  MissDesc* d_md_ = nullptr;
  int64_t* h_uniq_keys_ = nullptr;        // pinned, device-mapped (work_.uniq_keys_host is its device view)
  std::vector<uint32_t> uniq_miss_;       // this call: unique missed keys per table
  float* h_staging_ = nullptr;    // pinned
  float* d_staging_ = nullptr;
  uint8_t* h_found_ = nullptr;    // pinned
  uint8_t* d_found_ = nullptr;
  size_t staging_floats_ = 0, staging_uniq_ = 0;
  Status PrepareCall(const int64_t* d_keys_flat, float* const* d_out, const size_t* n, size_t T, bool probe_only, uint64_t* N_out);
  Status ReadBackCounts(size_t T, uint64_t N, bool exact);
  void AddInsertStats();

  uint64_t last_misses_ = 0, last_unique_ = 0, last_unique_keys_ = 0, last_miss_row_bytes_ = 0;
  // "This session's calls miss much": what the per-call switches steer by (keys_by_kernel 2, probe_in_lane 2, interact_mode 2).
  // Round 5 compared the LAST call's missed rows with side_bytes_ on every call — traffic that sits on that bound flipped the
  // arrangement call by call, and two sessions doing so in opposite phase was exactly the kind of stable bad mode the 20-ms
  // blocks were.  Now a switch with hysteresis: up when a call's missed rows exceed the bound, down only when they fall below
  // three quarters of it, and never sooner than kDwell calls after the last change.
  struct MissMuch {
    static constexpr uint32_t kDwell = 8;
    bool high = false;
    uint32_t since = kDwell;     // calls since the last change
    uint64_t flips = 0;
    void Update(uint64_t bytes, uint64_t bound) {
      if (since < kDwell) { ++since; return; }
      const bool want = high ? bytes * 4 >= bound * 3 : bytes > bound;
      if (want != high) { high = want; since = 0; ++flips; }
    }
  };
  MissMuch miss_much_;

 public:
  bool miss_much_mode() const { return miss_much_.high; }
  uint64_t mode_flips() const { return miss_much_.flips; }

 private:
  bool last_async_ = false;
  float last_gpu_ms_ = 0.f;
  float phase_ms_[4] = {0, 0, 0, 0};
  int probe_variant_ = 1002;  // K_P: U + 100 * no_dedup + 1000 * wide (512 threads per tile).  Two bucket lines in flight per 8-lane
                              // group: 43 us against 46.5 (U = 4) and 45 (U = 1) alone, 50 against 56 us in the timed region (tools/kbench.py)
  bool force_host_gather_ = false;  // option "host_gather": host-thread gather + H2D even on a ps_direct_access cache
  bool timing_ = false;
};

// The process-wide parameter server (one per backend: backend.cpp:68-69).
class HierParameterServer : public std::enable_shared_from_this<HierParameterServer> {
 public:
  // HierParameterServerBase::create(ps_json_config_file): parse, load every model's tables into the host
  // tier, build + warm GPU caches on each model's deployed devices (docs/architecture.md:246-260).
  static Status create(const std::string& ps_json_config_file, std::shared_ptr<HierParameterServer>* out);
  static Status create_from_text(const std::string& ps_json_text, std::shared_ptr<HierParameterServer>* out);
  // No files: models get their tables later through load_table_* (tests, bench).
  static Status create_from_config(const ParameterServerConfig& cfg, bool load_tables,
                                   std::shared_ptr<HierParameterServer>* out);
  ~HierParameterServer();

  const std::map<std::string, InferenceParams>& get_hps_model_configuration_map() const { return cfg_.models; }
  const ParameterServerConfig& config() const { return cfg_; }
  // thread-safe copy of one model's parameters; false when the model is not configured
  bool model_params(const std::string& model, InferenceParams* out);

  std::shared_ptr<EmbeddingCache> get_embedding_cache(const std::string& model, int device);
  // table-sharded model (ps.json "table_sharding": "hash"): the cache of shard s (on deployed_device_list[s])
  std::shared_ptr<EmbeddingCache> get_shard_cache(const std::string& model, uint32_t shard);
  // a lookup session sized for `max_keys` keys per call instead of the model's request capacity (shard sessions of an entry instance)
  Status create_lookup_session_sized(const std::string& model, std::shared_ptr<EmbeddingCache> cache, size_t max_keys,
                                     std::unique_ptr<LookupSession>* out);
  Status update_database_per_model(const InferenceParams& p);          // (re)load sparse files into the host tier
  Status create_embedding_cache_per_model(const InferenceParams& p);   // build caches on deployed_devices
  Status destory_embedding_cache_per_model(const std::string& model);  // [sic] reference spelling
  // Rows of resident keys are taken again from the parameter server (docs/hierarchical_parameter_server.md:234-238, 259-265).
  // By default only rows that CAN differ: a table that was neither reloaded nor updated since the cache last looked costs
  // nothing, an updated one costs its changed keys that are resident (HostTable's change log; every change is replayed by two
  // consecutive refreshes, which closes the race with a lookup that fetched the old row just before the update and inserts it
  // just after the refresh).  full = true (or ps.json "gpucache_refresh_changed_only": false): every resident row, as the
  // reference does.  Either way the upload is paced (InsertPacing) while sessions are serving.
  struct RefreshStats {
    uint64_t tables = 0, tables_unchanged = 0, tables_full = 0;
    uint64_t keys_dumped = 0;      // resident keys read back for a full pass
    uint64_t keys_changed = 0;     // change-log entries looked at
    uint64_t rows_refreshed = 0, row_bytes = 0;
    double seconds = 0.0;
  };
  Status refresh_embedding_cache(const std::string& model, int device, bool full = false, RefreshStats* stats = nullptr);
  Status add_model(const InferenceParams& p);  // online deployment: register a model parsed later
  // Re-read ps.json and (re)register every model in it (HPSBackend::ParseParameterServer, hps.cc:210-219)
  Status parse_config(const std::string& ps_json_config_file);

  Status create_lookup_session(const std::string& model, std::shared_ptr<EmbeddingCache> cache,
                               std::unique_ptr<LookupSession>* out);

  // table injection without files
  // online update hook: insert-or-overwrite rows of one table (fences ps_direct_access caches like a reload)
  Status upsert_table(const std::string& model, size_t table, const int64_t* keys, const float* rows, size_t n,
                      unsigned layers = HostTable::kLayerVolatile | HostTable::kLayerPersistent);
  Status load_table_from_arrays(const std::string& model, size_t table, const int64_t* keys, const float* rows,
                                size_t R, bool borrow);
  Status load_table_synthetic(const std::string& model, size_t table, uint64_t seed, int64_t key0, size_t R,
                              uint32_t shard = 0, uint32_t num_shards = 1);
  std::vector<std::shared_ptr<HostTable>> tables_of(const std::string& model);

  // Host-tier fetch of one table's keys: rows or default, multi-threaded.
  Status Fetch(const HostTable& tb, const int64_t* keys, size_t n, float* out, size_t stride, float default_value,
               uint8_t* found, size_t* nfound);
  // Several tables' fetches as ONE fork-join over the pool (a request touches every table of the model;
  // 26 separate fork-joins of a few thousand keys each would leave most cores idle).
  struct FetchJob {
    const HostTable* table;
    const int64_t* keys;
    size_t n;
    float* out;
    size_t stride;
    float default_value;
    uint8_t* found;  // optional
    const uint32_t* keys32 = nullptr;  // when `keys` is null: the same keys as uint32 offsets (widened task by task)
    int64_t key_base = 0;              // ... from this base
  };
  Status FetchMulti(const std::vector<FetchJob>& jobs, size_t max_threads = 0);   // max_threads 0: the whole serving pool

  // async-insert mode: queue "fetch these keys and insert them" for a cache; dropped (best effort)
  // when more than number_of_worker_buffers_in_pool jobs are already waiting.
  void RunDirectInsert(std::shared_ptr<EmbeddingCache> cache);  // device-driven tier: finish an accepted background job
  void SubmitAsyncInsert(std::shared_ptr<EmbeddingCache> cache, std::vector<std::vector<int64_t>> keys_per_table);

  ThreadPool* pool() { return pool_; }

  // ---- online updates (ps.json "update_source", csrc/ps/update_source.h) ----
  // One chunk of an update message into the database layers (insert-or-overwrite in the host tier, written through to the
  // persistent store); the keys are remembered until the consumer commits.
  Status ApplyUpdate(const std::string& model, uint32_t table, uint32_t dim, const int64_t* keys, const float* rows, size_t n);
  // The consumer committed: rows of updated keys that are RESIDENT in a GPU cache of these models are replaced there
  // (keys that are not resident stay out: an update is not a request).
  void OnUpdatesCommitted(const std::set<std::string>& models);
  bool update_source_stats(UpdateSourceStats* out) const;
  uint64_t filtered_update_count() const { return filtered_updates_.load(std::memory_order_relaxed); }
  Status drain_update_source(size_t timeout_ms);
  Status stop_update_source();   // joins the consumer thread; pending messages stay uncommitted

 private:
  HierParameterServer() = default;
  Status Build(bool load_tables);
  Status EnsureTables(const InferenceParams& p, bool load);
  Status MutateTables(const std::string& model, const std::function<Status()>& fn);
  Status RefreshOne(const std::string& model, const std::shared_ptr<EmbeddingCache>& cache, bool full, RefreshStats* stats);

  ParameterServerConfig cfg_;
  ThreadPool* pool_ = nullptr;
  std::mutex mu_;
  std::map<std::string, std::vector<std::shared_ptr<HostTable>>> tables_;
  // (model, device, shard): shard = -1 for the replicas of an ordinary model, 0..P-1 for the shards of a table-sharded one
  std::map<std::tuple<std::string, int, int>, std::shared_ptr<EmbeddingCache>> caches_;
  std::mutex upd_mu_;
  std::map<std::string, std::vector<std::vector<int64_t>>> updated_keys_;   // model -> per table: keys applied since the last commit
  // volatile_db.update_filters / persistent_db.update_filters, compiled (set before the consumer starts, then read-only): each
  // layer takes the updates ITS list selects (backend.cpp:207-216, 250-259)
  std::vector<std::regex> update_filters_, persistent_update_filters_;
  bool persistent_subscribes_ = false;        // a persistent database is configured
  // filtered: update messages no update_filters entry selected (skipped silently, as a subscription filter does)
  std::atomic<uint64_t> filtered_updates_{0};
  mutable std::mutex updates_mu_;             // guards the POINTER: stats / drain take a reference under it, stop moves it out
  std::shared_ptr<UpdateConsumer> updates_;   // last member: its thread stops before anything it uses goes away
};

}  // namespace hps
