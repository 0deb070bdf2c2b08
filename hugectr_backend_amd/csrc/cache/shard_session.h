// Native sharded (model-parallel) lookup — BASELINE config 3, a north-star addition (the reference is replicas-only:
// docs/architecture.md:11,29).  One table, rows partitioned over the P ranks of a node by
// owner(key) = mix64(key) mod P; one process per GPU.  Per call and rank, everything on the lookup session's stream:
//
//   bucket local keys by owner into P fixed-capacity blocks   (shard_kernels.hip; block = [count, largest block this rank needed, keys...])
//   all-to-all of the blocks                                   RCCL: ncclGroupStart + ncclSend/ncclRecv per peer, over xGMI
//   received blocks -> one padded key array                    (unused slots carry the cache's reserved key: the probe skips them)
//   local lookup                                               LookupSession::lookup_from_device_padded on this rank's shard
//   all-to-all of the padded rows back                         RCCL
//   rows -> input order                                        gather through the positions the bucket step recorded
//
// The capacity of a block is fixed before the call (mean + slack), so there is NO count exchange and NO device->host read-back
// between the two exchanges; the host waits once in the middle only when the local lookup has misses to fetch (as any lookup
// does) and once at the end.  If some rank's block overflowed, every rank learns the largest block anybody needed from the
// headers that travel with the keys, and all ranks repeat the call with exactly that capacity: two attempts at most.
// The row exchange ships whole blocks (cap x D floats per peer, ~5 % more than the rows in them at P = 8): exact sizes
// would need the received counts on the host, i.e. a device->host round trip (~20 us) between the two exchanges to save
// ~4 us of link time.
#pragma once
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <atomic>
#include <memory>
#include <mutex>
#include <vector>

#include "engine.h"

namespace hps {

// Moves P equal blocks between P ranks: block p of `send` goes to rank p, block p of `recv` comes from rank p.
class ShardTransport {
 public:
  virtual ~ShardTransport() = default;
  virtual uint32_t rank() const = 0;
  virtual uint32_t size() const = 0;
  virtual Status AllToAll(const void* d_send, void* d_recv, size_t bytes_per_peer, hipStream_t stream) = 0;
  virtual const char* name() const = 0;
  // This rank cannot go on with the collective call it is in (a launch failed, an argument was bad): peers that are waiting
  // for it inside AllToAll return an error instead of waiting for ever.  In-process transport: immediate.  RCCL: the
  // communicator is aborted (ncclCommAbort) — peers already inside a send/recv kernel are beyond reach of this process;
  // deployments bound the call with a watchdog (bench.py does).
  virtual void Abort() {}
};

// RCCL (librccl.so, loaded on first use).  unique_id: the 128 bytes of ncclGetUniqueId from rank 0, distributed by the
// caller (torch.distributed broadcast, a file, MPI ...).
Status ShardUniqueId(uint8_t out[128]);
Status MakeRcclTransport(uint32_t rank, uint32_t world, const uint8_t unique_id[128], int device, std::unique_ptr<ShardTransport>* out);
// First contact (multi_gpu_probe.h): one communicator over `devices` — one rank per entry, the ranks being threads of this
// process — and one all-reduce of one word.  `phase` (devices.size() words) is written as the ranks advance: 1 communicator
// being created, 2 all-reduce enqueued, 3 result back and right, 4 communicator destroyed; a caller watching from another
// thread can tell where a hang sits.  *ms: the slowest rank's time from entering ncclCommInitRank to the checked result.
Status RcclAllReduceSelfTest(const std::vector<int>& devices, std::atomic<int>* phase, float* ms);

// P endpoints inside ONE process on one device, for tests and single-process deployments: blocks are copied device to
// device on the callers' streams, ordered by events; the P callers must run on P threads (the exchange is a rendezvous).
class LocalShardGroup;
std::shared_ptr<LocalShardGroup> MakeLocalShardGroup(uint32_t world);
Status MakeLocalTransport(std::shared_ptr<LocalShardGroup> group, uint32_t rank, std::unique_ptr<ShardTransport>* out);

struct ShardCallStats {
  uint64_t capacity = 0;        // keys per block of the last call
  uint32_t attempts = 0;        // 1 unless a block overflowed (then 2: the second attempt runs with the capacity that was needed)
  uint64_t received = 0;        // keys this rank's shard was asked for in the last call (without padding)
  float keys_exchange_ms = 0, lookup_ms = 0, rows_exchange_ms = 0;   // last attempt, HIP events on the session's stream
  int key_bytes = 8;            // width at which a host request's keys crossed PCIe (Lookup: 8, device keys)
  std::vector<uint64_t> sent;   // keys this rank sent to every rank in the last call (distinct keys when the input dedup is on)
  uint64_t unique_keys = 0;     // distinct keys of this rank's request in the last call (= sum of sent; without dedup: keys as sent)
};

class ShardedSession {
 public:
  // `session`: lookup session of a ONE-table GPU-cache model that holds this rank's shard (shared: it lives at least as long
  // as this object); it must not be used directly while a sharded lookup runs.  max_local_keys: most keys a rank passes to
  // Lookup — the same on every rank (checked with one small exchange at the start of the first lookup).
  static Status Create(std::shared_ptr<LookupSession> session, std::unique_ptr<ShardTransport> transport, size_t max_local_keys,
                       std::unique_ptr<ShardedSession>* out);
  ~ShardedSession();
  // d_keys: n int64 on the session's device; d_out: n x D fp32.  Collective: every rank of the group calls it.  Blocking.
  Status Lookup(const int64_t* d_keys, size_t n, float* d_out);
  // The reference's contract (LookupSession::lookup, docs/architecture.md:308-323): keys in HOST memory.  Staged through
  // page-locked memory by the serving pool, as uint32 when every key fits (half the PCIe bytes), else as they are.
  Status LookupHost(const int64_t* h_keys, size_t n, float* d_out);
  const ShardCallStats& last_stats() const { return stats_; }
  uint32_t dim() const { return dim_; }

 private:
  ShardedSession() = default;
  Status Attempt(const void* d_keys, uint32_t key_bytes, size_t n, float* d_out, uint64_t cap, uint64_t* need);
  Status Run(const void* d_keys, uint32_t key_bytes, size_t n, float* d_out);
  Status VerifyGeometry();
  Status Refuse(Status st);
  bool verified_ = false;

  std::shared_ptr<LookupSession> session_;
  std::unique_ptr<ShardTransport> transport_;
  hipStream_t stream_ = nullptr;
  int device_ = 0;
  uint32_t P_ = 1, dim_ = 0;
  size_t max_local_ = 0;
  uint64_t cap_max_ = 0, cap_ = 0;
  float default_value_ = 0.f;
  int64_t *d_send_ = nullptr, *d_recv_ = nullptr, *d_keys_pad_ = nullptr, *d_keys_in_ = nullptr;
  int64_t* h_keys_in_ = nullptr;    // pinned staging of LookupHost
  hipEvent_t ev_[4] = {nullptr, nullptr, nullptr, nullptr};   // around the two exchanges
  float *d_rows_pad_ = nullptr, *d_rows_back_ = nullptr;
  uint32_t *d_pos_ = nullptr, *d_flags_ = nullptr;
  // input dedup in front of the exchange (HPS_SHARD_DEDUP, default on): a key the request repeats travels once
  bool dedup_ = true;
  uint32_t* d_rep_ = nullptr;
  unsigned long long* d_set_ = nullptr;
  uint64_t set_mask_ = 0;
  uint32_t set_tag_ = 0;
  // the block capacity follows the traffic down as well as up: every rank sees the same largest need per call (it travels in
  // the block headers), so every rank takes the same decision after the same number of calls
  uint64_t recent_need_ = 0;
  uint32_t calls_since_resize_ = 0;
  uint64_t* d_totals_ = nullptr;
  void* d_ws_ = nullptr;
  uint32_t* h_flags_ = nullptr;     // pinned: [0] largest block any rank needed, [1] keys received
  uint64_t* h_totals_ = nullptr;    // pinned
  ShardCallStats stats_;
};

}  // namespace hps
