// One embedding table in the host (CPU RAM) tier of the parameter server: the build's restatement of
// the reference's `hash_map` / `parallel_hash_map` volatile database
// (/root/reference/docs/hierarchical_parameter_server.md:380-412; README.md:127-135).
//
// Layout: rows live in one slab in load order (R x D fp32); the index is `num_partitions`
// open-addressing tables (key -> row number), partition = key mod num_partitions ("the last couple
// of bits of your embedding keys", docs/architecture.md:131).  Readers are lock-free; loads and
// upserts take the table's writer lock.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdint>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "../common/config.h"
#include "../common/status.h"
#include "thread_pool.h"
#include "volatile_tier.h"

namespace hps {

// Host tier smaller than the table (docs/hierarchical_parameter_server.md:460-507, 520-569): the rows stay in a
// memory-mapped row store on disk, the key index stays in RAM (16 B per row against 4*D B per row), and a bounded
// volatile tier (volatile_tier.h) holds the rows that are served from RAM.
struct HostTierOptions {
  bool tiered = false;          // false: the whole table in RAM (the fast path, BASELINE configs 1-5)
  bool persistent = false;      // a persistent database stands behind the volatile tier: its misses are read from the
                                // row store; false: what the volatile tier does not hold is not found (default vector)
  bool store_writable = false;  // online updates are written through to the row store (persistent_db.read_only=false)
  VolatileDatabaseParams vdb;   // overflow_margin, overflow_policy, overflow_resolution_target, initial_cache_rate,
                                // cache_missed_embeddings
};

struct HostTierStats {
  VolatileTierStats vdb;
  uint64_t persistent_hits = 0, not_found = 0, persistent_rows = 0;
};

class HostTable {
 public:
  // Takes effect at the next load.
  void SetTierOptions(const HostTierOptions& o) { tier_opt_ = o; }
  bool tiered() const { return vt_ != nullptr; }
  HostTierStats tier_stats() const;
  void DumpVolatileKeys(std::vector<int64_t>* out) const;

  // pinned: keys and rows live in page-locked, device-mapped host memory (hipHostMalloc) so that the GPU can
  // read them in place over PCIe ("ps_direct_access": the miss path then needs no host threads at all).
  HostTable(std::string name, uint32_t dim, size_t num_partitions, bool pinned = false);
  bool pinned() const { return pinned_; }
  // bumped by every (re)load / growing upsert: device-side indexes built from this table compare it
  uint64_t generation() const { return generation_.load(std::memory_order_acquire); }
  // ---- what changed, for whoever holds copies of rows (the GPU caches' refresh, cache/parameter_server.cpp) ----
  // Every (re)load starts a new load epoch; every Upsert appends its keys to a bounded change log.  A holder remembers
  // (load epoch, log position) as of the moment its copies were taken:
  //   ChangeMark(&e, &s)                       the current position (take it BEFORE reading rows: changes racing with the read
  //                                            are then replayed, never lost)
  //   ChangesSince(e, s, &keys, &e2, &s2)      true: `keys` = every key upserted since (e, s) (duplicates possible), (e2, s2) the
  //                                            new mark; false: the table was reloaded since, or the log no longer reaches back
  //                                            that far — every row may differ
  void ChangeMark(uint64_t* load_epoch, uint64_t* log_seq) const;
  bool ChangesSince(uint64_t load_epoch, uint64_t log_seq, std::vector<int64_t>* keys, uint64_t* new_epoch, uint64_t* new_seq) const;
  static constexpr size_t kChangeLogMax = (size_t)4 << 20;   // keys kept (32 MB); older entries are dropped, their holders refresh fully
  ~HostTable();
  HostTable(const HostTable&) = delete;
  HostTable& operator=(const HostTable&) = delete;

  // "<dir>/key" = R native-endian int64, "<dir>/emb_vector" = R*D native-endian fp32, same order
  // (docs/architecture.md:185-218).  Replaces the current contents.
  Status LoadFromDir(const std::string& dir, ThreadPool* pool);
  // Copies (or, with borrow=true, references) caller memory.  Replaces the current contents.
  Status LoadFromArrays(const int64_t* keys, const float* rows, size_t R, bool borrow, ThreadPool* pool);
  // Synthetic table (bench): keys key0..key0+R-1, rows from the SURVEY.md §8d recipe, generated in
  // parallel straight into the slab.
  // num_shards > 1: only the keys of key0..key0+R-1 owned by `shard` (mix64(key) mod num_shards), BASELINE config 3
  Status LoadSynthetic(uint64_t seed, uint32_t table_id, int64_t key0, size_t R, ThreadPool* pool, uint32_t shard = 0,
                       uint32_t num_shards = 1);
  // Insert-or-overwrite rows (online update path; duplicate keys: last wins).
  // layers: which database layers take the update (each layer subscribes with its own update_filters, backend.cpp:207-216,
  // 250-259) — kLayerVolatile the in-memory tier, kLayerPersistent the row store behind it.  A table that is ONE store (the whole
  // table in memory, no volatile tier in front of a persistent database) is updated when either bit is set.
  static constexpr unsigned kLayerVolatile = 1u, kLayerPersistent = 2u;
  Status Upsert(const int64_t* keys, const float* rows, size_t n, unsigned layers = kLayerVolatile | kLayerPersistent);

  const std::string& name() const { return name_; }
  uint32_t dim() const { return dim_; }
  size_t size() const { return num_rows_; }          // rows in the slab (file order)
  size_t num_partitions() const { return parts_.size(); }
  // true when the loaded data held the same key more than once (only the last row is live)
  bool has_duplicate_keys() const { return has_dups_; }
  // smallest key the table holds (0 for an empty table; the sentinel key HPS_EMPTY_KEY is left out): the frame of reference
  // a lookup session narrows a request's keys against (cache/engine.cpp) — a hint only, a key below it just does not narrow
  int64_t min_key() const { return min_key_.load(std::memory_order_relaxed); }
  int64_t key_at(size_t r) const { return keys_[r]; }
  const int64_t* keys() const { return keys_; }
  const float* row_at(size_t r) const { return rows_ + r * dim_; }

  // Row number of `key`, or -1.
  int64_t Find(int64_t key) const;

  // out + i*stride  <-  row(keys[i])  or  default_value broadcast.  found[i] (optional) = 1/0.
  // Single-threaded; callers parallelise over key ranges (ParameterServer::Fetch).
  // Returns the number of keys found.
  size_t Fetch(const int64_t* keys, size_t n, float* out, size_t stride, float default_value,
               uint8_t* found) const;

 private:
  struct Entry { int64_t key; int64_t row; };
  struct Partition {
    Entry* slots = nullptr;
    uint64_t mask = 0;  // capacity - 1
    std::atomic<size_t> used{0};
  };
  void FreeAll();
  void SetupTier();
  Status FinishLoad(ThreadPool* pool);
  size_t FetchTiered(const int64_t* keys, size_t n, float* out, size_t stride, float default_value, uint8_t* found) const;
  Status UpsertTiered(const int64_t* keys, const float* rows, size_t n, unsigned layers);
  Status AppendRows(const int64_t* keys, const float* rows, const std::vector<size_t>& fresh);
  Status IndexAppended(size_t first_new);
  double index_headroom_ = 1.0;   // AllocPartitions sizes the index for this many times the keys it is given
  void* DataAlloc(size_t bytes);
  void DataFree(void* p);
  bool pinned_ = false;
  std::atomic<uint64_t> generation_{0};
  mutable std::mutex log_mu_;
  uint64_t load_epoch_ = 0;            // (re)loads so far
  uint64_t log_base_ = 0;              // sequence number of change_log_[0]
  std::vector<int64_t> change_log_;    // keys of the upserts since log_base_
  void LogChanges(const int64_t* keys, size_t n);
  std::atomic<int64_t> min_key_{0};
  Status BuildIndex(ThreadPool* pool);
  Status AllocPartitions(const std::vector<size_t>& counts);
  static uint64_t SlotOf(int64_t key, uint64_t mask);
  size_t PartitionOf(int64_t key) const;
  void InsertConcurrent(int64_t key, int64_t row);
  int64_t FindUnlocked(int64_t key) const;

  std::string name_;
  uint32_t dim_;
  std::vector<std::unique_ptr<Partition>> parts_;
  bool pow2_parts_;
  int64_t* keys_ = nullptr;
  float* rows_ = nullptr;
  size_t num_rows_ = 0;
  size_t cap_rows_ = 0;
  bool owns_keys_ = false, owns_rows_ = false;
  bool has_dups_ = false;
  bool has_sentinel_ = false;  // HPS_EMPTY_KEY itself stored as a legal key
  int64_t sentinel_row_ = -1;
  mutable std::shared_mutex mu_;
  // glibc's rwlock prefers readers: a reload or an online update would wait for as long as lookups keep overlapping.
  // Readers therefore stand aside while a writer is waiting or working.
  mutable std::atomic<int> writers_{0};
  struct ReadLock {
    explicit ReadLock(const HostTable& t) : t_(t) {
      // (a reload takes seconds and runs on the pool: waiting readers must not eat the CPU quota it needs)
      for (unsigned spins = 0; t.writers_.load(std::memory_order_acquire) > 0; ++spins) {
        if (spins < 64) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(100));
      }
      t.mu_.lock_shared();
    }
    ~ReadLock() { t_.mu_.unlock_shared(); }
    const HostTable& t_;
  };
  struct WriteLock {
    explicit WriteLock(HostTable& t) : t_(t) {
      t.writers_.fetch_add(1, std::memory_order_acq_rel);
      t.mu_.lock();
    }
    ~WriteLock() {
      t_.mu_.unlock();
      t_.writers_.fetch_sub(1, std::memory_order_acq_rel);
    }
    HostTable& t_;
  };
  // tiered mode
  HostTierOptions tier_opt_;
  std::unique_ptr<VolatileTier> vt_;
  mutable std::atomic<uint64_t> clock_{0};     // one tick per fetch call: the access stamp of evict_oldest
  mutable std::atomic<uint64_t> persistent_hits_{0}, not_found_{0};
  std::string map_dir_;                        // row store directory when rows_ is a file mapping
  size_t map_bytes_ = 0;
  bool rows_writable_ = true;
};

}  // namespace hps
