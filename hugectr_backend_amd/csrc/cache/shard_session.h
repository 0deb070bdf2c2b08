// Native sharded (model-parallel) lookup — BASELINE config 3, a north-star addition (the reference is replicas-only:
// docs/architecture.md:11,29).  One table, rows partitioned over the P ranks of a node by
// owner(key) = mix64(key) mod P; one process per GPU.  Per call and rank, everything on the lookup session's stream:
//
//   bucket local keys by owner into P fixed-capacity blocks   (shard_kernels.hip; block = [count, overflow flag, keys...])
//   all-to-all of the blocks                                   RCCL: ncclGroupStart + ncclSend/ncclRecv per peer, over xGMI
//   received blocks -> one padded key array                    (unused slots carry a key the shard holds: they hit the cache)
//   local lookup                                               LookupSession::lookup_from_device on this rank's shard
//   all-to-all of the padded rows back                         RCCL
//   rows -> input order                                        gather through the positions the bucket step recorded
//
// The capacity of a block is fixed before the call (mean + slack), so there is NO count exchange, NO device->host read-back
// and NO stream synchronisation between the steps; one synchronisation ends the call.  If some rank's block overflowed
// (every rank learns it from the flags that travel with the keys) all ranks double the capacity and repeat the call.
#pragma once
#include <hip/hip_runtime_api.h>

#include <condition_variable>
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
};

// RCCL (librccl.so, loaded on first use).  unique_id: the 128 bytes of ncclGetUniqueId from rank 0, distributed by the
// caller (torch.distributed broadcast, a file, MPI ...).
Status ShardUniqueId(uint8_t out[128]);
Status MakeRcclTransport(uint32_t rank, uint32_t world, const uint8_t unique_id[128], int device, std::unique_ptr<ShardTransport>* out);

// P endpoints inside ONE process on one device, for tests and single-process deployments: blocks are copied device to
// device on the callers' streams, ordered by events; the P callers must run on P threads (the exchange is a rendezvous).
class LocalShardGroup;
std::shared_ptr<LocalShardGroup> MakeLocalShardGroup(uint32_t world);
Status MakeLocalTransport(std::shared_ptr<LocalShardGroup> group, uint32_t rank, std::unique_ptr<ShardTransport>* out);

struct ShardCallStats {
  uint64_t capacity = 0;        // keys per block of the last call
  uint32_t attempts = 0;        // 1 unless a block overflowed
  std::vector<uint64_t> sent;   // keys this rank sent to every rank in the last call
};

class ShardedSession {
 public:
  // `session`: lookup session of a ONE-table GPU-cache model that holds this rank's shard; it must outlive this object and
  // must not be used directly while a sharded lookup runs.  max_local_keys: most keys a rank passes to Lookup.
  static Status Create(LookupSession* session, std::unique_ptr<ShardTransport> transport, size_t max_local_keys,
                       std::unique_ptr<ShardedSession>* out);
  ~ShardedSession();
  // d_keys: n int64 on the session's device; d_out: n x D fp32.  Collective: every rank of the group calls it.  Blocking.
  Status Lookup(const int64_t* d_keys, size_t n, float* d_out);
  const ShardCallStats& last_stats() const { return stats_; }
  uint32_t dim() const { return dim_; }

 private:
  ShardedSession() = default;
  Status Attempt(const int64_t* d_keys, size_t n, float* d_out, uint64_t cap, bool* overflow);

  LookupSession* session_ = nullptr;
  std::unique_ptr<ShardTransport> transport_;
  hipStream_t stream_ = nullptr;
  int device_ = 0;
  uint32_t P_ = 1, dim_ = 0;
  size_t max_local_ = 0;
  uint64_t cap_max_ = 0, cap_ = 0;
  int64_t pad_key_ = 0;
  int64_t *d_send_ = nullptr, *d_recv_ = nullptr, *d_keys_pad_ = nullptr;
  float *d_rows_pad_ = nullptr, *d_rows_back_ = nullptr;
  uint32_t *d_pos_ = nullptr, *d_flags_ = nullptr;
  uint64_t* d_totals_ = nullptr;
  void* d_ws_ = nullptr;
  uint32_t* h_flags_ = nullptr;     // pinned: [0] overflow anywhere
  uint64_t* h_totals_ = nullptr;    // pinned
  ShardCallStats stats_;
};

}  // namespace hps
