#include "shard_session.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "shard_kernels.h"

namespace hps {

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return ::hps::Error(::hps::Code::kInternal, #expr, " failed: ", hipGetErrorString(_e), " (", \
                          __FILE__, ":", __LINE__, ")");                                           \
  } while (0)

// =================================================================================================
// RCCL transport.  The handful of entry points used are resolved from librccl.so at first use, so that the engine
// library itself loads on machines without RCCL (and in the CPU-only test container).  Signatures: rccl/rccl.h.
// =================================================================================================
namespace {

struct RcclApi {
  typedef struct { char internal[128]; } UniqueId;
  typedef void* Comm;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string load_error;

  static RcclApi& Get() {
    static RcclApi api = [] {
      RcclApi a;
      void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) { a.load_error = dlerror(); return a; }
      auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p && a.load_error.empty()) a.load_error = std::string("missing symbol ") + n; return p; };
      a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
      a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
      a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
      a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
      a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
      a.Send = (decltype(a.Send))sym("ncclSend");
      a.Recv = (decltype(a.Recv))sym("ncclRecv");
      a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
      return a;
    }();
    return api;
  }
  Status Ready() const {
    if (!load_error.empty()) return Error(Code::kUnavailable, "RCCL is not available: ", load_error);
    return Status::Ok();
  }
};

constexpr int kNcclUint8 = 1;   // ncclUint8 (rccl.h: ncclDataType_t)

#define RCCL_TRY(api, expr)                                                                         \
  do {                                                                                              \
    const int _r = (expr);                                                                          \
    if (_r != 0) return Error(Code::kInternal, #expr, " failed: ", (api).GetErrorString ? (api).GetErrorString(_r) : "?"); \
  } while (0)

class RcclTransport : public ShardTransport {
 public:
  RcclTransport(uint32_t rank, uint32_t world, RcclApi::Comm comm) : rank_(rank), world_(world), comm_(comm) {}
  ~RcclTransport() override { if (comm_) (void)RcclApi::Get().CommDestroy(comm_); }
  uint32_t rank() const override { return rank_; }
  uint32_t size() const override { return world_; }
  const char* name() const override { return "rccl"; }
  Status AllToAll(const void* d_send, void* d_recv, size_t bytes, hipStream_t stream) override {
    RcclApi& api = RcclApi::Get();
    // one group = one fused launch: P sends and P receives progress together over the xGMI links
    RCCL_TRY(api, api.GroupStart());
    for (uint32_t p = 0; p < world_; ++p) {
      RCCL_TRY(api, api.Send((const char*)d_send + (size_t)p * bytes, bytes, kNcclUint8, (int)p, comm_, stream));
      RCCL_TRY(api, api.Recv((char*)d_recv + (size_t)p * bytes, bytes, kNcclUint8, (int)p, comm_, stream));
    }
    RCCL_TRY(api, api.GroupEnd());
    return Status::Ok();
  }

 private:
  uint32_t rank_, world_;
  RcclApi::Comm comm_;
};

}  // namespace

Status ShardUniqueId(uint8_t out[128]) {
  RcclApi& api = RcclApi::Get();
  HPS_RETURN_IF_ERROR(api.Ready());
  RcclApi::UniqueId id;
  RCCL_TRY(api, api.GetUniqueId(&id));
  memcpy(out, id.internal, 128);
  return Status::Ok();
}

Status MakeRcclTransport(uint32_t rank, uint32_t world, const uint8_t unique_id[128], int device, std::unique_ptr<ShardTransport>* out) {
  RcclApi& api = RcclApi::Get();
  HPS_RETURN_IF_ERROR(api.Ready());
  if (world == 0 || rank >= world) return Error(Code::kInvalidArg, "rank ", rank, " of ", world);
  HIP_TRY(hipSetDevice(device));
  RcclApi::UniqueId id;
  memcpy(id.internal, unique_id, 128);
  RcclApi::Comm comm = nullptr;
  RCCL_TRY(api, api.CommInitRank(&comm, (int)world, id, (int)rank));
  out->reset(new RcclTransport(rank, world, comm));
  return Status::Ok();
}

// =================================================================================================
// In-process transport
// =================================================================================================
class LocalShardGroup {
 public:
  explicit LocalShardGroup(uint32_t world) : world_(world), send_(world, nullptr), ready_(world, nullptr), done_(world, nullptr) {}
  ~LocalShardGroup() {
    for (hipEvent_t e : ready_) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : done_) if (e) (void)hipEventDestroy(e);
  }
  uint32_t world() const { return world_; }
  // generation barrier over the P calling threads
  void Arrive() {
    std::unique_lock<std::mutex> lk(mu_);
    const uint64_t gen = gen_;
    if (++arrived_ == world_) { arrived_ = 0; ++gen_; cv_.notify_all(); }
    else cv_.wait(lk, [&] { return gen_ != gen; });
  }
  Status Exchange(uint32_t rank, const void* d_send, void* d_recv, size_t bytes, hipStream_t stream) {
    if (!ready_[rank]) {
      HIP_TRY(hipEventCreateWithFlags(&ready_[rank], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&done_[rank], hipEventDisableTiming));
    }
    send_[rank] = d_send;
    HIP_TRY(hipEventRecord(ready_[rank], stream));   // my blocks are complete once my stream gets here
    Arrive();
    for (uint32_t p = 0; p < world_; ++p) {          // pull my block out of every peer's send buffer
      HIP_TRY(hipStreamWaitEvent(stream, ready_[p], 0));
      HIP_TRY(hipMemcpyAsync((char*)d_recv + (size_t)p * bytes, (const char*)send_[p] + (size_t)rank * bytes, bytes, hipMemcpyDeviceToDevice, stream));
    }
    HIP_TRY(hipEventRecord(done_[rank], stream));
    Arrive();
    for (uint32_t p = 0; p < world_; ++p) HIP_TRY(hipStreamWaitEvent(stream, done_[p], 0));   // peers are done reading my send buffer
    Arrive();   // nobody re-records an event another rank has not waited on yet
    return Status::Ok();
  }

 private:
  uint32_t world_;
  std::vector<const void*> send_;
  std::vector<hipEvent_t> ready_, done_;
  std::mutex mu_;
  std::condition_variable cv_;
  uint32_t arrived_ = 0;
  uint64_t gen_ = 0;
};

namespace {
class LocalTransport : public ShardTransport {
 public:
  LocalTransport(std::shared_ptr<LocalShardGroup> g, uint32_t rank) : g_(std::move(g)), rank_(rank) {}
  uint32_t rank() const override { return rank_; }
  uint32_t size() const override { return g_->world(); }
  const char* name() const override { return "in-process"; }
  Status AllToAll(const void* d_send, void* d_recv, size_t bytes, hipStream_t stream) override {
    return g_->Exchange(rank_, d_send, d_recv, bytes, stream);
  }

 private:
  std::shared_ptr<LocalShardGroup> g_;
  uint32_t rank_;
};
}  // namespace

std::shared_ptr<LocalShardGroup> MakeLocalShardGroup(uint32_t world) { return std::make_shared<LocalShardGroup>(world); }

Status MakeLocalTransport(std::shared_ptr<LocalShardGroup> group, uint32_t rank, std::unique_ptr<ShardTransport>* out) {
  if (!group || rank >= group->world()) return Error(Code::kInvalidArg, "bad local shard group / rank");
  out->reset(new LocalTransport(std::move(group), rank));
  return Status::Ok();
}

// =================================================================================================
// ShardedSession
// =================================================================================================
Status ShardedSession::Create(LookupSession* session, std::unique_ptr<ShardTransport> transport, size_t max_local_keys,
                              std::unique_ptr<ShardedSession>* out) {
  if (!session || !transport) return Error(Code::kInvalidArg, "null argument");
  if (!session->uses_gpu_cache()) return Error(Code::kUnsupported, "the native sharded lookup needs a GPU-cache session");
  if (session->num_tables() != 1) return Error(Code::kUnsupported, "the sharded lookup handles one table per session, the model has ", session->num_tables());
  if (max_local_keys == 0) return Error(Code::kInvalidArg, "max_local_keys is 0");
  std::unique_ptr<ShardedSession> s(new ShardedSession());
  s->session_ = session;
  s->P_ = transport->size();
  s->transport_ = std::move(transport);
  s->stream_ = session->stream();
  s->device_ = session->device();
  s->dim_ = session->table_dim(0);
  s->max_local_ = max_local_keys;
  s->pad_key_ = session->any_key_of_table(0);
  if (s->P_ == 0 || s->P_ > 64) return Error(Code::kUnsupported, "1..64 shards are supported, got ", s->P_);
  // the local lookup sees P blocks of `cap` keys: the session's request capacity bounds the block size
  s->cap_max_ = session->max_keys() / s->P_;
  const double mean = (double)max_local_keys / s->P_;
  // start at mean + 8 sigma of a binomial split (hashed owners) + a little: uniform traffic never overflows; skewed
  // traffic (one hot key = one owner) overflows once and the capacity doubles
  uint64_t cap = (uint64_t)std::ceil(mean + 8.0 * std::sqrt(mean) + 64.0);
  cap = std::min<uint64_t>(std::max<uint64_t>(cap, 1), std::min<uint64_t>(s->cap_max_, max_local_keys));
  if (s->cap_max_ == 0)
    return Error(Code::kInvalidArg, "the session's request capacity (", session->max_keys(), " keys) is smaller than the number of shards");
  s->cap_ = std::max<uint64_t>(cap, 1);
  HIP_TRY(hipSetDevice(s->device_));
  const uint64_t cap_alloc = std::min<uint64_t>(s->cap_max_, max_local_keys);   // no block can ever need more
  const size_t P = s->P_, D = s->dim_;
  auto dev = [](auto** p, size_t count) -> Status {
    void* v = nullptr;
    if (hipMalloc(&v, std::max<size_t>(count, 1) * sizeof(**p)) != hipSuccess) return Error(Code::kInternal, "sharded session: out of device memory");
    *p = (std::remove_reference_t<decltype(**p)>*)v;
    return Status::Ok();
  };
  HPS_RETURN_IF_ERROR(dev(&s->d_send_, P * (cap_alloc + 2)));
  HPS_RETURN_IF_ERROR(dev(&s->d_recv_, P * (cap_alloc + 2)));
  HPS_RETURN_IF_ERROR(dev(&s->d_keys_pad_, P * cap_alloc));
  HPS_RETURN_IF_ERROR(dev(&s->d_rows_pad_, P * cap_alloc * D));
  HPS_RETURN_IF_ERROR(dev(&s->d_rows_back_, P * cap_alloc * D));
  HPS_RETURN_IF_ERROR(dev(&s->d_pos_, max_local_keys));
  HPS_RETURN_IF_ERROR(dev(&s->d_flags_, 4));
  HPS_RETURN_IF_ERROR(dev(&s->d_totals_, P));
  {
    void* v = nullptr;
    HIP_TRY(hipMalloc(&v, ShardBucketWorkspaceBytes(max_local_keys, s->P_) + 64));
    s->d_ws_ = v;
    HIP_TRY(hipHostMalloc(&v, 4 * sizeof(uint32_t), hipHostMallocDefault));
    s->h_flags_ = (uint32_t*)v;
    HIP_TRY(hipHostMalloc(&v, P * sizeof(uint64_t), hipHostMallocDefault));
    s->h_totals_ = (uint64_t*)v;
  }
  s->cap_max_ = cap_alloc;
  *out = std::move(s);
  return Status::Ok();
}

ShardedSession::~ShardedSession() {
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  for (void* p : {(void*)d_send_, (void*)d_recv_, (void*)d_keys_pad_, (void*)d_rows_pad_, (void*)d_rows_back_, (void*)d_pos_, (void*)d_flags_,
                  (void*)d_totals_, d_ws_})
    if (p) (void)hipFree(p);
  if (h_flags_) (void)hipHostFree(h_flags_);
  if (h_totals_) (void)hipHostFree(h_totals_);
}

Status ShardedSession::Attempt(const int64_t* d_keys, size_t n, float* d_out, uint64_t cap, bool* overflow) {
  const uint64_t stride = cap + 2;
  const size_t D = dim_;
  HIP_TRY(hipMemsetAsync(d_flags_, 0, 4 * sizeof(uint32_t), stream_));
  HIP_TRY(LaunchShardBucketPadded(d_keys, n, P_, cap, d_send_, d_pos_, d_totals_, d_ws_, stream_));
  HPS_RETURN_IF_ERROR(transport_->AllToAll(d_send_, d_recv_, stride * sizeof(int64_t), stream_));
  HIP_TRY(LaunchShardPrepare(d_recv_, P_, cap, pad_key_, d_keys_pad_, d_flags_, stream_));
  // the local lookup: one table, P * cap keys, rows in the same padded layout (its own kernels follow ours on the stream)
  float* rows = d_rows_pad_;
  const size_t nk = (size_t)P_ * cap;
  HPS_RETURN_IF_ERROR(session_->lookup_from_device(d_keys_pad_, &rows, &nk, 1));
  HPS_RETURN_IF_ERROR(transport_->AllToAll(d_rows_pad_, d_rows_back_, cap * D * sizeof(float), stream_));
  HIP_TRY(LaunchShardGatherBack(d_rows_back_, d_pos_, n, dim_, d_out, stream_));
  HIP_TRY(hipMemcpyAsync(h_flags_, d_flags_, sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
  HIP_TRY(hipMemcpyAsync(h_totals_, d_totals_, P_ * sizeof(uint64_t), hipMemcpyDeviceToHost, stream_));
  HIP_TRY(hipStreamSynchronize(stream_));
  *overflow = h_flags_[0] != 0;
  return Status::Ok();
}

Status ShardedSession::Lookup(const int64_t* d_keys, size_t n, float* d_out) {
  if (n > max_local_) return Error(Code::kInvalidArg, "sharded lookup: ", n, " keys exceed max_local_keys = ", max_local_);
  if (n && (!d_keys || !d_out)) return Error(Code::kInvalidArg, "null argument");
  HIP_TRY(hipSetDevice(device_));
  stats_.attempts = 0;
  for (;;) {
    bool overflow = false;
    ++stats_.attempts;
    HPS_RETURN_IF_ERROR(Attempt(d_keys, n, d_out, cap_, &overflow));
    stats_.capacity = cap_;
    stats_.sent.assign(h_totals_, h_totals_ + P_);
    if (!overflow) return Status::Ok();
    // some rank's block was too small: every rank saw the flag (it travels with the keys), every rank doubles
    if (cap_ >= cap_max_)
      return Error(Code::kInvalidArg, "sharded lookup: a shard received more than ", cap_max_,
                   " keys from one rank; raise the model's max_batch_size (request capacity / shards bounds the block size)");
    cap_ = std::min<uint64_t>(cap_max_, cap_ * 2);
  }
}

}  // namespace hps
