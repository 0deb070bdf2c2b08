// Bounded volatile database in front of a persistent row store: the host tier when RAM < table.
// Restates the overflow handling of the reference's CPU-memory database
// (/root/reference/docs/hierarchical_parameter_server.md:460-507): at most `overflow_margin` embeddings per
// partition; going over it prunes the partition to `overflow_margin * overflow_resolution_target` entries chosen by
// `overflow_policy` (evict_random / evict_least_used / evict_oldest); `cache_missed_embeddings` decides whether rows
// found behind this tier are inserted into it (docs:497-500).  Decisions the documentation leaves open are fixed in
// SURVEY.md Appendix C style and restated in oracle/hps_oracle.py (VolatileDbModel):
//   * a partition never holds more than overflow_margin entries: the insert that would exceed it prunes first
//     (the margin is the limit whatever the table held at load time: keys added later by online updates or appended
//     rows count against the same margin; storage grows on demand up to it);
//   * keep = max(1, floor(overflow_margin * overflow_resolution_target)); an insert into a full partition whose
//     prune frees nothing (margin 1) evicts one entry by the policy;
//   * evict_oldest orders by (last access stamp, key), evict_least_used by (access count, last access stamp, key),
//     smallest first; a stamp is one tick per fetch call; an insert counts as one access;
//   * evict_random draws from a per-partition xorshift64* seeded with the partition number.
#pragma once
#include <atomic>
#include <cstdint>
#include <shared_mutex>
#include <vector>

#include "../common/config.h"

namespace hps {

struct VolatileTierStats {
  uint64_t entries = 0, capacity = 0, max_partition_entries = 0;
  uint64_t lookups = 0, hits = 0, inserts = 0, evictions = 0, overflows = 0;
};

class VolatileTier {
 public:
  // partition_keys[p]: distinct keys of partition p at load time — a sizing hint only, the limit is overflow_margin
  VolatileTier(uint32_t dim, const std::vector<size_t>& partition_keys, const VolatileDatabaseParams& params);
  ~VolatileTier();
  VolatileTier(const VolatileTier&) = delete;
  VolatileTier& operator=(const VolatileTier&) = delete;

  // One reader/writer lock per partition, taken by the caller around the calls below (a request is bucketed by
  // partition, so a lock is taken once per partition and request, not once per key).
  std::shared_mutex& mutex(size_t partition) const { return parts_[partition]->mu; }

  // --- shared side (any number of threads, none inside the exclusive side) ---
  // Copies the row of `key` to dst and records the access; false if the key is not held.
  bool Lookup(size_t partition, int64_t key, float* dst, uint64_t now);
  bool Contains(size_t partition, int64_t key) const;

  // --- exclusive side ---
  // Insert or overwrite.  Returns the number of entries evicted to make room.
  size_t Insert(size_t partition, int64_t key, const float* row, uint64_t now);
  // Overwrite only if present (online update of a cached row); keeps the access statistics.
  bool Overwrite(size_t partition, int64_t key, const float* row);

  // These two take the partition locks themselves (shared), one partition at a time.
  void DumpKeys(std::vector<int64_t>* out) const;
  VolatileTierStats stats() const;
  size_t num_partitions() const { return parts_.size(); }

 private:
  static constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
  struct Cell { int64_t key; uint32_t slot; };
  struct Partition {
    mutable std::shared_mutex mu;
    std::vector<Cell> index;        // open addressing, linear probing, backward-shift deletion; slot == kNoSlot: empty
    uint64_t mask = 0;
    size_t limit = 0;               // overflow_margin: most entries the partition may hold
    size_t cap = 0;                 // slots allocated so far (<= limit, grown on demand)
    size_t keep = 0;                // entries left after a prune
    size_t size = 0;
    int64_t* keys = nullptr;        // [cap]
    std::atomic<uint64_t>* stamp = nullptr;   // [cap] last access
    std::atomic<uint32_t>* count = nullptr;   // [cap] accesses
    float* rows = nullptr;          // [cap][dim]
    std::vector<uint32_t> free_slots;
    uint64_t rng = 0;
    std::atomic<uint64_t> lookups{0}, hits{0};
    uint64_t inserts = 0, evictions = 0, overflows = 0;
  };
  static uint64_t Home(int64_t key, uint64_t mask);
  uint32_t FindSlot(const Partition& P, int64_t key) const;
  void Erase(Partition& P, int64_t key);
  size_t Prune(Partition& P, size_t at_least);
  void Grow(Partition& P, size_t new_cap);

  uint32_t dim_;
  DatabaseOverflowPolicy policy_;
  std::vector<Partition*> parts_;
};

}  // namespace hps
