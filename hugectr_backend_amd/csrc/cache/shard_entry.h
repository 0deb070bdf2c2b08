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
// TWO TRANSPORTS for the rows (ps.json "shard_transport", session option "transport"; chosen per session, both bit-exact):
//   "peer_store" (default)  the owners' kernels store each row straight into the entry GPU's OUTPUT0 through peer-mapped pointers
//                           (fine-grained xGMI stores, no extra copy of any row, needs peer access between the devices)
//   "staged_copy"           an owner gathers a PIECE of its bucket (<= shard_copy_piece_keys keys) into a LOCAL block with an ordinary
//                           lookup, a copy engine ships the block to the entry GPU's receive buffer (hipMemcpyPeerAsync: SDMA over
//                           xGMI, bulk transfers) while the owner already gathers the next piece, and hps_entry_place puts the rows of
//                           a delivered block into OUTPUT0 (local HBM of the entry GPU).  The bucket keys travel the same way, in one
//                           copy: no kernel of this transport touches another GPU's memory, so it also works where peer access
//                           is not available (the runtime then stages the copies itself).
// Which one fills the entry GPU's seven inbound links better is for the first multi-GPU run to say (bench.py runs both on the same
// requests); on one GPU staged_copy costs one more pass over the rows.
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
  uint64_t dedup_flips = 0;      // adaptive dedup: changes of level since the session was created
  int transport = 0;             // 0 peer_store, 1 staged_copy
  uint64_t copied_bytes = 0;     // staged_copy: row bytes the copy engines shipped into the entry GPU
  std::vector<float> copy_wait_ms;   // staged_copy: time each shard's worker spent waiting for its copies to land
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
  hipStream_t stream() const { return stream_; }   // the entry device's stream (idle between requests)
  size_t max_keys() const { return max_keys_; }
  size_t shard_capacity() const { return shard_cap_; }
  LookupSession* shard_session(uint32_t s) { return s < sessions_.size() ? sessions_[s].get() : nullptr; }
  void set_dedup(int level) { dedup_ = level; tile_only_left_ = 0; level_hold_ = 0; }   // 0 off, 1 adaptive (default), 2 always both levels
  // 0 peer_store, 1 staged_copy; peer_store is refused when a shard's device cannot store into the entry device
  Status set_transport(int transport);
  int transport() const { return transport_; }
  void set_piece_keys(size_t keys) { if (keys == 0 || keys >= 1024) piece_keys_ = keys; }   // staged_copy: keys per piece, 0 = automatic (from the next request on)
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
    // staged_copy
    bool staged = false;
    uint64_t total = 0;              // keys of the bucket
    float* recv = nullptr;           // this shard's region of the entry GPU's receive buffer
    const EntryDesc* desc = nullptr; // the request's descriptor on the entry GPU
    float copy_wait_ms = 0.f;
    uint64_t copied_bytes = 0;
  };
  void WorkerMain(uint32_t s);
  Status ServeStaged(uint32_t s, Worker& w, uint64_t* misses, uint64_t* unique);

  // staged_copy: what one shard needs on its own device and on the entry device (allocated at the first staged request)
  struct StagedShard {
    float* stage[2] = {nullptr, nullptr};   // shard device: two blocks of piece rows
    size_t stage_floats = 0;
    int64_t* okeys = nullptr;                // shard device: the bucket's keys
    hipStream_t copy_stream = nullptr;       // shard device
    hipEvent_t copied[2] = {nullptr, nullptr};
    hipStream_t place_stream = nullptr;      // entry device
  };
  Status EnsureStaged();
  void FreeStaged();
  static size_t PieceFloats(const ShardPass& pass, const std::vector<uint32_t>& dims);
  std::vector<StagedShard> staged_;
  float* d_recv_ = nullptr;         // entry device: the shards' blocks, shard-major, piece by piece
  size_t recv_floats_ = 0;
  size_t piece_keys_ = 0;           // 0: automatic (kAutoPieceKeys for a remote shard, one piece for a shard on the entry GPU)
  static constexpr size_t kAutoPieceKeys = 131072;
  int transport_ = 0;
  bool peers_ok_ = true;            // every shard device can store into the entry device

  std::shared_ptr<HierParameterServer> ps_;
  InferenceParams params_;
  uint32_t P_ = 1;
  int device_ = 0;
  size_t max_keys_ = 0, max_tiles_ = 0, shard_cap_ = 0;
  int dedup_ = 1;
  uint32_t tile_only_left_ = 0;    // adaptive dedup: requests left that skip the call-wide level
  uint32_t level_hold_ = 0;        // ... requests left before the level may change again
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
