#include "volatile_tier.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <tuple>

#include "../common/hps_hash.h"

namespace hps {

VolatileTier::VolatileTier(uint32_t dim, const std::vector<size_t>& partition_keys, const VolatileDatabaseParams& params)
    : dim_(dim), policy_(params.overflow_policy) {
  double target = params.overflow_resolution_target;
  if (!(target > 0.0) || !(target < 1.0)) target = 0.8;   // docs:487-489: strictly between 0 and 1
  for (size_t p = 0; p < partition_keys.size(); ++p) {
    Partition* P = new Partition();
    P->limit = std::max<size_t>(1, params.overflow_margin);
    // floor(limit * target) without leaving size_t for the "no margin" default (SIZE_MAX)
    P->keep = P->limit >= (1ull << 52) ? P->limit : std::max<size_t>(1, (size_t)std::floor((double)P->limit * target));
    // storage for what the table holds now; online updates and appended rows grow it (Grow) up to the limit
    Grow(*P, std::max<size_t>(1, std::min(P->limit, partition_keys[p])));
    P->rng = hps_mix64(0x9E3779B97F4A7C15ull + p) | 1ull;
    parts_.push_back(P);
  }
}

// (Re)allocates the slot slab for new_cap > cap slots, keeps every live slot where it is, rebuilds the index when
// it would pass load 0.5.  Exclusive side only.
void VolatileTier::Grow(Partition& P, size_t new_cap) {
  if (new_cap <= P.cap) return;
  int64_t* keys = new int64_t[new_cap];
  std::atomic<uint64_t>* stamp = new std::atomic<uint64_t>[new_cap];
  std::atomic<uint32_t>* count = new std::atomic<uint32_t>[new_cap];
  void* mem = nullptr;
  const size_t bytes = std::max<size_t>(64, new_cap * (size_t)dim_ * sizeof(float));
  if (posix_memalign(&mem, 64, (bytes + 63) / 64 * 64) != 0) mem = nullptr;
  if (!mem) { delete[] keys; delete[] stamp; delete[] count; throw std::bad_alloc(); }
  float* rows = (float*)mem;   // pages are touched only as slots fill
  for (const Cell& c : P.index) {
    if (c.slot == kNoSlot) continue;
    const uint32_t s = c.slot;
    keys[s] = P.keys[s];
    stamp[s].store(P.stamp[s].load(std::memory_order_relaxed), std::memory_order_relaxed);
    count[s].store(P.count[s].load(std::memory_order_relaxed), std::memory_order_relaxed);
    memcpy(rows + (size_t)s * dim_, P.rows + (size_t)s * dim_, (size_t)dim_ * sizeof(float));
  }
  delete[] P.keys; delete[] P.stamp; delete[] P.count; free(P.rows);
  P.keys = keys; P.stamp = stamp; P.count = count; P.rows = rows;
  // new slots are handed out lowest first, after the ones already free
  std::vector<uint32_t> fresh;
  fresh.reserve(new_cap - P.cap + P.free_slots.size());
  for (size_t s = new_cap; s-- > P.cap;) fresh.push_back((uint32_t)s);
  fresh.insert(fresh.end(), P.free_slots.begin(), P.free_slots.end());
  P.free_slots.swap(fresh);
  P.cap = new_cap;
  uint64_t icap = 16;
  while (icap < new_cap * 2) icap <<= 1;
  if (icap > P.index.size()) {
    std::vector<Cell> old;
    old.swap(P.index);
    P.index.assign(icap, Cell{0, kNoSlot});
    P.mask = icap - 1;
    for (const Cell& c : old) {
      if (c.slot == kNoSlot) continue;
      uint64_t i = Home(c.key, P.mask);
      while (P.index[i].slot != kNoSlot) i = (i + 1) & P.mask;
      P.index[i] = c;
    }
  }
}

VolatileTier::~VolatileTier() {
  for (Partition* P : parts_) {
    delete[] P->keys;
    delete[] P->stamp;
    delete[] P->count;
    free(P->rows);
    delete P;
  }
}

uint64_t VolatileTier::Home(int64_t key, uint64_t mask) { return (hps_mix64((uint64_t)key) >> 17) & mask; }

uint32_t VolatileTier::FindSlot(const Partition& P, int64_t key) const {
  uint64_t i = Home(key, P.mask);
  for (;;) {
    const Cell& c = P.index[i];
    if (c.slot == kNoSlot) return kNoSlot;
    if (c.key == key) return c.slot;
    i = (i + 1) & P.mask;
  }
}

bool VolatileTier::Lookup(size_t partition, int64_t key, float* dst, uint64_t now) {
  Partition& P = *parts_[partition];
  P.lookups.fetch_add(1, std::memory_order_relaxed);
  const uint32_t s = FindSlot(P, key);
  if (s == kNoSlot) return false;
  memcpy(dst, P.rows + (size_t)s * dim_, (size_t)dim_ * sizeof(float));
  P.stamp[s].store(now, std::memory_order_relaxed);
  P.count[s].fetch_add(1, std::memory_order_relaxed);
  P.hits.fetch_add(1, std::memory_order_relaxed);
  return true;
}

bool VolatileTier::Contains(size_t partition, int64_t key) const { return FindSlot(*parts_[partition], key) != kNoSlot; }

void VolatileTier::Erase(Partition& P, int64_t key) {
  uint64_t i = Home(key, P.mask);
  for (;;) {
    Cell& c = P.index[i];
    if (c.slot == kNoSlot) return;
    if (c.key == key) break;
    i = (i + 1) & P.mask;
  }
  P.free_slots.push_back(P.index[i].slot);
  --P.size;
  // backward shift: pull later cells of the probe run into the hole while that keeps them reachable from their home
  uint64_t hole = i, j = i;
  for (;;) {
    j = (j + 1) & P.mask;
    const Cell c = P.index[j];
    if (c.slot == kNoSlot) break;
    const uint64_t home = Home(c.key, P.mask);
    // c may move to `hole` iff home is cyclically outside (hole, j]
    const bool stays = hole <= j ? (home > hole && home <= j) : (home > hole || home <= j);
    if (!stays) { P.index[hole] = c; hole = j; }
  }
  P.index[hole] = Cell{0, kNoSlot};
}

// Drops the partition to `keep` entries, and at least `at_least` of them (a full partition whose keep equals its
// limit still has to make room for the insert that called).
size_t VolatileTier::Prune(Partition& P, size_t at_least) {
  size_t drop = P.size > P.keep ? P.size - P.keep : 0;
  drop = std::min(P.size, std::max(drop, at_least));
  if (drop == 0) return 0;
  std::vector<uint32_t> live;
  live.reserve(P.size);
  for (const Cell& c : P.index) if (c.slot != kNoSlot) live.push_back(c.slot);
  if (policy_ == DatabaseOverflowPolicy::EvictRandom) {
    std::sort(live.begin(), live.end());   // slot order: the draw does not depend on the index layout
    for (size_t i = 0; i < drop; ++i) {    // partial Fisher-Yates
      P.rng ^= P.rng >> 12; P.rng ^= P.rng << 25; P.rng ^= P.rng >> 27;
      const uint64_t r = (P.rng * 0x2545F4914F6CDD1Dull) >> 11;
      const size_t j = i + (size_t)(r % (live.size() - i));
      std::swap(live[i], live[j]);
    }
  } else {
    auto rank = [&](uint32_t s) {
      const uint64_t st = P.stamp[s].load(std::memory_order_relaxed);
      const uint64_t ct = policy_ == DatabaseOverflowPolicy::EvictLeastUsed ? P.count[s].load(std::memory_order_relaxed) : 0;
      return std::make_tuple(ct, st, P.keys[s]);
    };
    std::nth_element(live.begin(), live.begin() + (drop - 1), live.end(),
                     [&](uint32_t a, uint32_t b) { return rank(a) < rank(b); });
  }
  for (size_t i = 0; i < drop; ++i) Erase(P, P.keys[live[i]]);
  P.evictions += drop;
  ++P.overflows;
  return drop;
}

size_t VolatileTier::Insert(size_t partition, int64_t key, const float* row, uint64_t now) {
  Partition& P = *parts_[partition];
  uint32_t s = FindSlot(P, key);
  size_t evicted = 0;
  if (s == kNoSlot) {
    if (P.size >= P.limit) evicted = Prune(P, 1);
    else if (P.free_slots.empty()) Grow(P, std::min(P.limit, std::max(P.cap * 2, P.cap + 16)));
    s = P.free_slots.back();
    P.free_slots.pop_back();
    uint64_t i = Home(key, P.mask);
    while (P.index[i].slot != kNoSlot) i = (i + 1) & P.mask;
    P.index[i] = Cell{key, s};
    P.keys[s] = key;
    P.count[s].store(0, std::memory_order_relaxed);
    ++P.size;
    ++P.inserts;
  }
  memcpy(P.rows + (size_t)s * dim_, row, (size_t)dim_ * sizeof(float));
  P.stamp[s].store(now, std::memory_order_relaxed);
  P.count[s].fetch_add(1, std::memory_order_relaxed);
  return evicted;
}

bool VolatileTier::Overwrite(size_t partition, int64_t key, const float* row) {
  Partition& P = *parts_[partition];
  const uint32_t s = FindSlot(P, key);
  if (s == kNoSlot) return false;
  memcpy(P.rows + (size_t)s * dim_, row, (size_t)dim_ * sizeof(float));
  return true;
}

void VolatileTier::DumpKeys(std::vector<int64_t>* out) const {
  out->clear();
  for (const Partition* P : parts_) {
    std::shared_lock<std::shared_mutex> lk(P->mu);
    for (const Cell& c : P->index) if (c.slot != kNoSlot) out->push_back(c.key);
  }
  std::sort(out->begin(), out->end());
}

VolatileTierStats VolatileTier::stats() const {
  VolatileTierStats s;
  for (const Partition* P : parts_) {
    std::shared_lock<std::shared_mutex> lk(P->mu);
    s.entries += P->size;
    s.capacity += P->limit >= (1ull << 52) ? P->cap : P->limit;
    s.max_partition_entries = std::max<uint64_t>(s.max_partition_entries, P->size);
    s.lookups += P->lookups.load(std::memory_order_relaxed);
    s.hits += P->hits.load(std::memory_order_relaxed);
    s.inserts += P->inserts;
    s.evictions += P->evictions;
    s.overflows += P->overflows;
  }
  return s;
}

}  // namespace hps
