// Table-sharded lookup behind ONE instance — BASELINE config 3 through the reference's own boundary.
//
// The reference is replicas only (docs/architecture.md:11,29) and Triton hands ONE request to ONE instance and blocks on it
// (/root/reference/hps_backend/src/hps.cc:353-369, 406): a sharded table therefore has to be served by whichever instance
// the request lands on ("single-entry": SURVEY.md 5.8).  ps.json `"table_sharding": "hash"` makes entry s of a model's
// deployed_device_list SHARD s: its GPU cache holds (gpucacheper of) the keys with mix64(key) mod P == s; the host tier below
// stays whole and process-wide, as in the reference (backend.cpp:68-69).  An entry session on device g then serves a request:
//
//   keys -> HBM of g (host keys: staged through page-locked memory)
//   hps_entry_dedup / hist / scan / scatter      representatives bucketed by owner, + every bucket key's row position in OUTPUT0
//   P x LookupSession::lookup_from_device_indexed   one lookup session per shard ON THE SHARD'S DEVICE, driven by this session's
//                                                own worker threads side by side: probe / gather / miss path as for any
//                                                request; the rows are stored straight into g's output buffer through
//                                                peer-mapped pointers (xGMI), the bucket keys are read from g the same way
//   hps_entry_expand                             rows of repeated keys, copied locally on g
//
// No collective and no lock-step: every instance of the model drives its own P shard sessions, and the shard caches order
// the sessions of all instances as they order any sessions (EmbeddingCache::BeginRead / BeginWrite).  The SPMD variant — one
// process per GPU, RCCL send/recv groups — is shard_session.h.
//
// One host round trip sits between the bucket step and the lookups (P x T counts, ~20 us): exact bucket sizes mean no
// padding, no overflow/retry, any number of tables.  An owner that gets more keys than its session holds (skew; the session is
// sized shard_capacity_factor x request capacity / P) is served in several passes (PlanShardPasses).
#pragma once
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "engine.h"
#include "entry_kernels.h"

namespace hps {

// One pass of one owner: keys [offset, offset + sum n) of the owner's bucket, n[t] of table t.
struct ShardPass {
  uint64_t offset = 0;
  std::vector<size_t> n;
};
// The passes that serve a bucket of counts[t] keys per table (table-major) with a session that holds `capacity` keys per
// call: consecutive ranges of at most `capacity` keys.  Pure host logic (tests/test_shard_entry_cpu.py drives it through the C ABI).
std::vector<ShardPass> PlanShardPasses(const uint32_t* counts, size_t num_tables, size_t capacity);

struct ShardEntryStats {
  uint64_t keys = 0;             // keys of the last request
  uint64_t unique_keys = 0;      // keys that travelled: distinct (table, key) pairs (= keys without shard_dedup; per tile when dedup_level is 1)
  int dedup_level = 2;           // 0: none, 1: within tiles of 1,024 keys only, 2: call-wide
  std::vector<uint64_t> sent;    // keys each shard was asked for
  std::vector<uint32_t> passes;  // lookup calls per shard (1 unless its bucket exceeded the session's capacity)
  std::vector<float> shard_ms;   // wall time of each shard's lookups
  float bucket_ms = 0.f;         // descriptor upload .. counts on the host (HIP events + the wait)
  float lookup_ms = 0.f;         // dispatch .. last shard done (wall)
  float expand_ms = 0.f;         // repeated keys' rows (wall, including the synchronisation)
  float key_stage_ms = 0.f;      // host keys: staging + upload enqueue
  int key_bytes = 8;             // bytes per key that crossed PCIe: 8, 4 (uint32 offsets) or 3 (packed)
  uint64_t misses = 0, unique_misses = 0;   // summed over the shards' lookups
};

class ShardedEntrySession {
 public:
  // `entry_device` must be one of the model's deployed devices; the session's own stream and buffers live there.
  static Status Create(std::shared_ptr<HierParameterServer> ps, const std::string& model, int entry_device,
                       std::unique_ptr<ShardedEntrySession>* out);
  ~ShardedEntrySession();

  // The reference's contract (docs/architecture.md:308-323): host key pointers in, device vector pointers (entry device) out.
  Status lookup(const void* const* h_keys_per_table, float* const* d_vectors_per_table, const size_t* num_keys_per_table,
                size_t num_tables);
  // KEYS already in the entry device's memory, flat + table-major.
  Status lookup_from_device(const int64_t* d_keys_flat, float* const* d_vectors_per_table, const size_t* num_keys_per_table,
                            size_t num_tables);

  const ShardEntryStats& last_stats() const { return stats_; }
  uint32_t num_shards() const { return P_; }
  int device() const { return device_; }
  size_t max_keys() const { return max_keys_; }
  size_t shard_capacity() const { return shard_cap_; }
  LookupSession* shard_session(uint32_t s) { return s < sessions_.size() ? sessions_[s].get() : nullptr; }
  void set_dedup(int level) { dedup_ = level; tile_only_left_ = 0; }   // 0 off, 1 adaptive (default), 2 always both levels
  void set_timing(bool b);   // forwards to the shard sessions (per-kernel times in their own statistics)

 private:
  ShardedEntrySession() = default;
  // narrow_bytes 3 / 4: d_narrow_ holds the request's keys as offsets from key_base_ (they are widened into d_keys_ first)
  Status Run(const int64_t* d_keys_flat, float* const* d_out, const size_t* n, size_t T, uint32_t narrow_bytes = 0);

  struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool has_job = false, stop = false, done = true;
    const int64_t* keys = nullptr;
    const uint32_t* idx = nullptr;
    float* const* out = nullptr;
    std::vector<ShardPass> plan;
    Status st = Status::Ok();
    float ms = 0.f, wake_ms = 0.f;   // lookups' wall time; job posted -> worker running
    std::chrono::steady_clock::time_point posted;
    uint64_t misses = 0, unique = 0;
  };
  void WorkerMain(uint32_t s);

  std::shared_ptr<HierParameterServer> ps_;
  InferenceParams params_;
  uint32_t P_ = 1;
  int device_ = 0;
  size_t max_keys_ = 0, max_tiles_ = 0, shard_cap_ = 0;
  int dedup_ = 1;
  uint32_t tile_only_left_ = 0;    // adaptive dedup: requests left that skip the call-wide level
  hipStream_t stream_ = nullptr;
  hipEvent_t ev_[2] = {nullptr, nullptr};
  std::vector<std::unique_ptr<LookupSession>> sessions_;   // one per shard, on the shard's device
  std::vector<int> shard_device_;
  std::vector<std::unique_ptr<Worker>> workers_;
  std::vector<uint32_t> dims_;

  // entry device
  char* h_block_ = nullptr;        // pinned: EntryDesc | TileDesc[max_tiles_]
  char* d_block_ = nullptr;
  size_t tiles_off_ = 0;
  int64_t* h_keys_ = nullptr;      // pinned staging of host keys
  int64_t* d_keys_ = nullptr;
  uint8_t* d_narrow_ = nullptr;    // narrowed host keys as they crossed PCIe (4 bytes per key at most)
  std::vector<std::shared_ptr<HostTable>> tables_;
  std::vector<int64_t> key_base_;  // this request's per-table bases (frame of reference: the tables' smallest keys)
  int narrow_backoff_ = 0, narrow_streak_ = 0;   // calls left before narrowing is tried again after a key too wide; failures in a row
  uint32_t* d_rep_ = nullptr;
  unsigned long long* d_set_ = nullptr;
  uint64_t set_mask_ = 0;
  uint32_t set_tag_ = 0;
  uint32_t *d_hist_ = nullptr, *d_within_ = nullptr, *d_counts_ = nullptr;   // d_counts_: base[P + 1] | counts[P][T]
  uint32_t* h_counts_ = nullptr;   // pinned mirror
  int64_t* d_bkeys_ = nullptr;
  uint32_t* d_bidx_ = nullptr;
  ShardEntryStats stats_;
};

}  // namespace hps
