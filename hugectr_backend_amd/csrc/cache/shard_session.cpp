#include "shard_session.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>

#include "../ps/thread_pool.h"
#include "key_pack.h"
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
  int (*CommAbort)(Comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;   // (self-test only)
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
      a.CommAbort = (decltype(a.CommAbort))dlsym(h, "ncclCommAbort");   // optional
      a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
      a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
      a.Send = (decltype(a.Send))sym("ncclSend");
      a.Recv = (decltype(a.Recv))sym("ncclRecv");
      a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
      a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");   // optional
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
  void Abort() override {
    if (comm_ && RcclApi::Get().CommAbort) { (void)RcclApi::Get().CommAbort(comm_); comm_ = nullptr; }
  }
  uint32_t rank() const override { return rank_; }
  uint32_t size() const override { return world_; }
  const char* name() const override { return "rccl"; }
  Status AllToAll(const void* d_send, void* d_recv, size_t bytes, hipStream_t stream) override {
    RcclApi& api = RcclApi::Get();
    if (!comm_) return Error(Code::kUnavailable, "the RCCL communicator of this sharded session was aborted");
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

Status RcclAllReduceSelfTest(const std::vector<int>& devices, std::atomic<int>* phase, float* ms) {
  RcclApi& api = RcclApi::Get();
  HPS_RETURN_IF_ERROR(api.Ready());
  if (!api.AllReduce) return Error(Code::kUnavailable, "librccl has no ncclAllReduce");
  const int n = (int)devices.size();
  if (n == 0) return Error(Code::kInvalidArg, "no devices");
  RcclApi::UniqueId id;
  RCCL_TRY(api, api.GetUniqueId(&id));
  std::vector<Status> st(n, Status::Ok());
  std::vector<float> took(n, 0.f);
  std::vector<std::thread> th;
  for (int r = 0; r < n; ++r) {
    th.emplace_back([&, r]() {
      st[r] = [&]() -> Status {
        const auto t0 = std::chrono::steady_clock::now();
        HIP_TRY(hipSetDevice(devices[r]));
        if (phase) phase[r].store(1);
        RcclApi::Comm comm = nullptr;
        RCCL_TRY(api, api.CommInitRank(&comm, n, id, r));
        hipStream_t s = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        int32_t* d = nullptr;
        HIP_TRY(hipMalloc((void**)&d, sizeof(int32_t)));
        const int32_t mine = r + 1;
        HIP_TRY(hipMemcpyAsync(d, &mine, sizeof mine, hipMemcpyHostToDevice, s));
        constexpr int kNcclInt32 = 2, kNcclSum = 0;   // rccl.h: ncclDataType_t / ncclRedOp_t
        RCCL_TRY(api, api.AllReduce(d, d, 1, kNcclInt32, kNcclSum, comm, s));
        if (phase) phase[r].store(2);
        int32_t got = 0;
        HIP_TRY(hipMemcpyAsync(&got, d, sizeof got, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (got != n * (n + 1) / 2) return Error(Code::kInternal, "all-reduce over ", n, " ranks gave ", got, " on rank ", r, ", expected ", n * (n + 1) / 2);
        took[r] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (phase) phase[r].store(3);
        (void)hipFree(d);
        (void)hipStreamDestroy(s);
        (void)api.CommDestroy(comm);
        if (phase) phase[r].store(4);
        return Status::Ok();
      }();
    });
  }
  for (auto& t : th) t.join();
  float worst = 0.f;
  for (int r = 0; r < n; ++r) {
    if (!st[r].ok()) return Error(st[r].code(), "rank ", r, " (device ", devices[r], "): ", st[r].message());
    worst = std::max(worst, took[r]);
  }
  if (ms) *ms = worst;
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
  // generation barrier over the P calling threads; false: some rank gave up (Abort) — nobody waits for it any more
  bool Arrive() {
    std::unique_lock<std::mutex> lk(mu_);
    if (aborted_) return false;
    const uint64_t gen = gen_;
    if (++arrived_ == world_) { arrived_ = 0; ++gen_; cv_.notify_all(); }
    else cv_.wait(lk, [&] { return gen_ != gen || aborted_; });
    return !aborted_;
  }
  void Abort() {
    std::lock_guard<std::mutex> lk(mu_);
    aborted_ = true;
    cv_.notify_all();
  }
  Status Exchange(uint32_t rank, const void* d_send, void* d_recv, size_t bytes, hipStream_t stream) {
    const auto gone = [] { return Error(Code::kUnavailable, "sharded lookup: another rank of the in-process group failed; the group is unusable"); };
    if (!ready_[rank]) {
      HIP_TRY(hipEventCreateWithFlags(&ready_[rank], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&done_[rank], hipEventDisableTiming));
    }
    send_[rank] = d_send;
    HIP_TRY(hipEventRecord(ready_[rank], stream));   // my blocks are complete once my stream gets here
    if (!Arrive()) return gone();
    for (uint32_t p = 0; p < world_; ++p) {          // pull my block out of every peer's send buffer
      HIP_TRY(hipStreamWaitEvent(stream, ready_[p], 0));
      HIP_TRY(hipMemcpyAsync((char*)d_recv + (size_t)p * bytes, (const char*)send_[p] + (size_t)rank * bytes, bytes, hipMemcpyDeviceToDevice, stream));
    }
    HIP_TRY(hipEventRecord(done_[rank], stream));
    if (!Arrive()) return gone();
    for (uint32_t p = 0; p < world_; ++p) HIP_TRY(hipStreamWaitEvent(stream, done_[p], 0));   // peers are done reading my send buffer
    if (!Arrive()) return gone();   // nobody re-records an event another rank has not waited on yet
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
  bool aborted_ = false;
};

namespace {
class LocalTransport : public ShardTransport {
 public:
  LocalTransport(std::shared_ptr<LocalShardGroup> g, uint32_t rank) : g_(std::move(g)), rank_(rank) {}
  uint32_t rank() const override { return rank_; }
  uint32_t size() const override { return g_->world(); }
  const char* name() const override { return "in-process"; }
  Status AllToAll(const void* d_send, void* d_recv, size_t bytes, hipStream_t stream) override {
    const Status st = g_->Exchange(rank_, d_send, d_recv, bytes, stream);
    if (!st.ok()) g_->Abort();   // a HIP call of mine failed in the middle of the rendezvous: release the others
    return st;
  }
  void Abort() override { g_->Abort(); }

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
Status ShardedSession::Create(std::shared_ptr<LookupSession> session, std::unique_ptr<ShardTransport> transport, size_t max_local_keys,
                              std::unique_ptr<ShardedSession>* out) {
  if (!session || !transport) return Error(Code::kInvalidArg, "null argument");
  if (!session->uses_gpu_cache()) return Error(Code::kUnsupported, "the native sharded lookup needs a GPU-cache session");
  if (session->num_tables() != 1) return Error(Code::kUnsupported, "the sharded lookup handles one table per session, the model has ", session->num_tables());
  if (max_local_keys == 0) return Error(Code::kInvalidArg, "max_local_keys is 0");
  if (max_local_keys >= 0xFFFFFFFFull) return Error(Code::kInvalidArg, "max_local_keys must be below 2^32 - 1");
  std::unique_ptr<ShardedSession> s(new ShardedSession());
  s->P_ = transport->size();
  s->transport_ = std::move(transport);
  s->stream_ = session->stream();
  s->device_ = session->device();
  s->dim_ = session->table_dim(0);
  s->default_value_ = session->table_default(0);
  s->max_local_ = max_local_keys;
  if (s->P_ == 0 || s->P_ > 64) return Error(Code::kUnsupported, "1..64 shards are supported, got ", s->P_);
  // the local lookup sees P blocks of `cap` keys: the session's request capacity bounds the block size
  s->cap_max_ = session->max_keys() / s->P_;
  const double mean = (double)max_local_keys / s->P_;
  // start at mean + 8 sigma of a binomial split (hashed owners) + a little: uniform traffic never overflows; skewed
  // traffic (one hot key = one owner) overflows once and the capacity becomes what was needed
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
  HPS_RETURN_IF_ERROR(dev(&s->d_send_, P * std::max<uint64_t>(cap_alloc + 2, 4)));   // (4 words per peer: VerifyGeometry)
  HPS_RETURN_IF_ERROR(dev(&s->d_recv_, P * std::max<uint64_t>(cap_alloc + 2, 4)));
  HPS_RETURN_IF_ERROR(dev(&s->d_keys_pad_, P * cap_alloc));
  HPS_RETURN_IF_ERROR(dev(&s->d_rows_pad_, P * cap_alloc * D));
  HPS_RETURN_IF_ERROR(dev(&s->d_rows_back_, P * cap_alloc * D));
  HPS_RETURN_IF_ERROR(dev(&s->d_pos_, max_local_keys));
  HPS_RETURN_IF_ERROR(dev(&s->d_flags_, 4));
  HPS_RETURN_IF_ERROR(dev(&s->d_totals_, P));
  HPS_RETURN_IF_ERROR(dev(&s->d_keys_in_, max_local_keys));
  if (const char* e = std::getenv("HPS_SHARD_DEDUP")) s->dedup_ = std::strtol(e, nullptr, 10) != 0;
  if (s->dedup_) {
    uint64_t set_cap = 1024;
    while (set_cap < 2 * (uint64_t)max_local_keys) set_cap <<= 1;
    HPS_RETURN_IF_ERROR(dev(&s->d_rep_, max_local_keys));
    HPS_RETURN_IF_ERROR(dev(&s->d_set_, set_cap));
    HIP_TRY(hipMemset(s->d_set_, 0, set_cap * sizeof(unsigned long long)));   // tag 0 is never used by a call
    s->set_mask_ = set_cap - 1;
  }
  {
    void* v = nullptr;
    HIP_TRY(hipMalloc(&v, ShardBucketWorkspaceBytes(max_local_keys, s->P_) + 64));
    s->d_ws_ = v;
    HIP_TRY(hipHostMalloc(&v, 4 * sizeof(uint32_t), hipHostMallocDefault));
    s->h_flags_ = (uint32_t*)v;
    HIP_TRY(hipHostMalloc(&v, std::max<size_t>(P, 4) * sizeof(uint64_t), hipHostMallocDefault));
    s->h_totals_ = (uint64_t*)v;
    HIP_TRY(hipHostMalloc(&v, max_local_keys * sizeof(int64_t), hipHostMallocDefault));
    s->h_keys_in_ = (int64_t*)v;
  }
  for (hipEvent_t& e : s->ev_) HIP_TRY(hipEventCreate(&e));
  s->cap_max_ = cap_alloc;
  s->session_ = std::move(session);
  *out = std::move(s);
  return Status::Ok();
}

ShardedSession::~ShardedSession() {
  (void)hipSetDevice(device_);
  if (stream_ && session_) (void)hipStreamSynchronize(stream_);   // (the stream belongs to the session, alive through session_)
  for (void* p : {(void*)d_send_, (void*)d_recv_, (void*)d_keys_pad_, (void*)d_rows_pad_, (void*)d_rows_back_, (void*)d_pos_, (void*)d_flags_,
                  (void*)d_totals_, (void*)d_keys_in_, (void*)d_rep_, (void*)d_set_, d_ws_})
    if (p) (void)hipFree(p);
  if (h_flags_) (void)hipHostFree(h_flags_);
  if (h_totals_) (void)hipHostFree(h_totals_);
  if (h_keys_in_) (void)hipHostFree(h_keys_in_);
  for (hipEvent_t e : ev_) if (e) (void)hipEventDestroy(e);
}

Status ShardedSession::Attempt(const void* d_keys, uint32_t key_bytes, size_t n, float* d_out, uint64_t cap, uint64_t* need) {
  const uint64_t stride = cap + 2;
  const size_t D = dim_;
  HIP_TRY(hipMemsetAsync(d_flags_, 0, 4 * sizeof(uint32_t), stream_));
  const uint32_t* rep = nullptr;
  if (dedup_ && n) {
    if (++set_tag_ == 0) {   // 2^32 attempts later: entries of the first ones would look like this one's
      HIP_TRY(hipMemsetAsync(d_set_, 0, (set_mask_ + 1) * sizeof(unsigned long long), stream_));
      set_tag_ = 1;
    }
    HIP_TRY(LaunchShardDedup(d_keys, key_bytes, n, d_set_, set_mask_, set_tag_, d_rep_, stream_));
    rep = d_rep_;
  }
  HIP_TRY(LaunchShardBucketPadded(d_keys, key_bytes, n, P_, cap, d_send_, d_pos_, d_totals_, d_ws_, stream_, rep));
  HIP_TRY(hipEventRecord(ev_[0], stream_));
  HPS_RETURN_IF_ERROR(transport_->AllToAll(d_send_, d_recv_, stride * sizeof(int64_t), stream_));
  HIP_TRY(hipEventRecord(ev_[1], stream_));
  HIP_TRY(LaunchShardPrepare(d_recv_, P_, cap, d_keys_pad_, d_flags_, stream_));
  // the local lookup: one table, P * cap keys, rows in the same padded layout (its own kernels follow ours on the stream);
  // the padding is skipped by the probe
  float* rows = d_rows_pad_;
  const size_t nk = (size_t)P_ * cap;
  HPS_RETURN_IF_ERROR(session_->lookup_from_device_padded(d_keys_pad_, &rows, &nk, 1));
  HIP_TRY(hipEventRecord(ev_[2], stream_));
  HPS_RETURN_IF_ERROR(transport_->AllToAll(d_rows_pad_, d_rows_back_, cap * D * sizeof(float), stream_));
  HIP_TRY(hipEventRecord(ev_[3], stream_));
  HIP_TRY(LaunchShardGatherBack(d_rows_back_, d_pos_, n, dim_, d_out, default_value_, stream_, rep));
  HIP_TRY(hipMemcpyAsync(h_flags_, d_flags_, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
  HIP_TRY(hipMemcpyAsync(h_totals_, d_totals_, P_ * sizeof(uint64_t), hipMemcpyDeviceToHost, stream_));
  HIP_TRY(hipStreamSynchronize(stream_));
  *need = h_flags_[0];
  stats_.received = h_flags_[1];
  session_->discount_padding(nk - std::min<uint64_t>(nk, h_flags_[1]));
  (void)hipEventElapsedTime(&stats_.keys_exchange_ms, ev_[0], ev_[1]);
  (void)hipEventElapsedTime(&stats_.lookup_ms, ev_[1], ev_[2]);
  (void)hipEventElapsedTime(&stats_.rows_exchange_ms, ev_[2], ev_[3]);
  return Status::Ok();
}

// Every rank must have been created with the same geometry: one exchange of (max_local_keys, first capacity, row width,
// block limit) per peer at the start of the first collective call, compared on the host.  A mismatch would otherwise show up
// as mismatched send/recv sizes — a hang.
Status ShardedSession::VerifyGeometry() {
  uint64_t* mine = h_totals_;
  mine[0] = max_local_; mine[1] = cap_; mine[2] = dim_; mine[3] = cap_max_;
  for (uint32_t p = 0; p < P_; ++p)
    HIP_TRY(hipMemcpyAsync(d_send_ + (size_t)p * 4, mine, 4 * sizeof(uint64_t), hipMemcpyHostToDevice, stream_));
  HPS_RETURN_IF_ERROR(transport_->AllToAll(d_send_, d_recv_, 4 * sizeof(uint64_t), stream_));
  std::vector<uint64_t> got((size_t)P_ * 4);
  HIP_TRY(hipMemcpyAsync(got.data(), d_recv_, got.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, stream_));
  HIP_TRY(hipStreamSynchronize(stream_));
  for (uint32_t p = 0; p < P_; ++p)
    if (memcmp(&got[(size_t)p * 4], mine, 4 * sizeof(uint64_t)) != 0)
      return Error(Code::kInvalidArg, "sharded session: rank ", p, " was created with max_local_keys ", got[(size_t)p * 4], ", first capacity ",
                   got[(size_t)p * 4 + 1], ", row width ", got[(size_t)p * 4 + 2], ", block limit ", got[(size_t)p * 4 + 3], "; this rank (",
                   transport_->rank(), ") with ", mine[0], ", ", mine[1], ", ", mine[2], ", ", mine[3], " — they must agree");
  verified_ = true;
  return Status::Ok();
}

Status ShardedSession::Run(const void* d_keys, uint32_t key_bytes, size_t n, float* d_out) {
  stats_.attempts = 0;
  if (!verified_) {
    const Status st = VerifyGeometry();
    if (!st.ok()) { transport_->Abort(); return st; }
  }
  for (;;) {
    uint64_t need = 0;
    ++stats_.attempts;
    const Status st = Attempt(d_keys, key_bytes, n, d_out, cap_, &need);
    if (!st.ok()) {
      transport_->Abort();   // this rank is out of the collective: nobody may wait for it
      return st;
    }
    stats_.capacity = cap_;
    stats_.sent.assign(h_totals_, h_totals_ + P_);
    stats_.unique_keys = 0;
    for (uint64_t v : stats_.sent) stats_.unique_keys += v;
    if (need <= cap_) {
      // The capacity follows the traffic DOWN too (blocks travel whole: capacity is what the links carry).  Deduplicated Zipf
      // traffic needs half the blocks uniform traffic of the same size does; after 32 calls that all fitted a capacity a tenth
      // smaller, the capacity becomes the largest need of those calls (+ 1/16 + 64).  `need` is the same number on every
      // rank, so is the call count: every rank resizes in the same call.  A burst that no longer fits costs one repeated
      // call (the growth below), never a wrong row.
      recent_need_ = std::max(recent_need_, need);
      if (++calls_since_resize_ >= 32) {
        const uint64_t target = std::min<uint64_t>(cap_max_, recent_need_ + recent_need_ / 16 + 64);
        if (target * 10 < cap_ * 9) cap_ = std::max<uint64_t>(target, 1);
        recent_need_ = 0;
        calls_since_resize_ = 0;
      }
      return Status::Ok();
    }
    // some rank's block was too small.  Every rank has seen the same maximum (each rank's largest need travels in every
    // block header it sends), so every rank picks the same new capacity and the second attempt fits.
    if (need > cap_max_) {
      // No abort here: `need` is the maximum every rank computed from the same block headers of an exchange that has
      // COMPLETED, so every rank arrives at this line together and returns the same error; nobody is left inside a
      // collective.  (Round 3 aborted the transport — ncclCommAbort / the group's aborted flag — and one oversized or
      // skewed request killed the sharded session on all ranks for good.)  The session stays usable: the next request
      // that fits is served.
      return Error(Code::kInvalidArg, "sharded lookup: one rank has ", need, " keys for one shard, more than the block limit of ", cap_max_,
                   " keys; raise the model's max_batch_size (request capacity / shards bounds the block size)");
    }
    cap_ = std::min<uint64_t>(cap_max_, need + need / 16 + 64);   // a little headroom: the next call's hot key may be hotter
    recent_need_ = 0;
    calls_since_resize_ = 0;
  }
}

// A call that is refused before it reaches the exchange still leaves the other ranks inside theirs: every refusal goes
// through here, which takes this rank's endpoint out of the group so that nobody waits for it.
Status ShardedSession::Refuse(Status st) {
  transport_->Abort();
  return st;
}

Status ShardedSession::Lookup(const int64_t* d_keys, size_t n, float* d_out) {
  if (n > max_local_) return Refuse(Error(Code::kInvalidArg, "sharded lookup: ", n, " keys exceed max_local_keys = ", max_local_));
  if (n && (!d_keys || !d_out)) return Refuse(Error(Code::kInvalidArg, "null argument"));
  if (hipSetDevice(device_) != hipSuccess) return Refuse(Error(Code::kInternal, "hipSetDevice(", device_, ") failed"));
  stats_.key_bytes = 8;
  return Run(d_keys, 8, n, d_out);
}

Status ShardedSession::LookupHost(const int64_t* h_keys, size_t n, float* d_out) {
  if (n > max_local_) return Refuse(Error(Code::kInvalidArg, "sharded lookup: ", n, " keys exceed max_local_keys = ", max_local_));
  if (n && (!h_keys || !d_out)) return Refuse(Error(Code::kInvalidArg, "null argument"));
  if (hipSetDevice(device_) != hipSuccess) return Refuse(Error(Code::kInternal, "hipSetDevice(", device_, ") failed"));
  // stage + narrow, as LookupSession::lookup does for a replica's request: 32 K-key tasks on the serving pool copy the keys
  // into page-locked memory as uint32 while OR-ing them together; a key that does not fit ends the attempt and the
  // request goes as it is (8 bytes per key)
  constexpr size_t kTaskKeys = 32768;
  const size_t tasks = (n + kTaskKeys - 1) / kTaskKeys;
  uint32_t width = 4;
  {
    uint64_t sample = 0;
    for (size_t t = 0; t < tasks; ++t) sample |= (uint64_t)h_keys[t * kTaskKeys] | (uint64_t)h_keys[std::min(n, (t + 1) * kTaskKeys) - 1];
    if (sample >> 32) width = 8;
  }
  for (;;) {
    std::atomic<uint64_t> high{0};
    auto body = [&](size_t t) {
      const size_t b = t * kTaskKeys, e = std::min(n, b + kTaskKeys);
      if (width == 8) memcpy(h_keys_in_ + b, h_keys + b, (e - b) * sizeof(int64_t));
      else {
        const uint64_t h = PackKeys32(h_keys + b, e - b, reinterpret_cast<uint32_t*>(h_keys_in_) + b);
        if (h >> 32) high.fetch_or(h, std::memory_order_relaxed);
      }
    };
    if (tasks <= 2) for (size_t t = 0; t < tasks; ++t) body(t);
    else ThreadPool::Serving().ParallelFor(tasks, body);
    if (width == 4 && high.load() != 0) { width = 8; continue; }
    break;
  }
  if (n && hipMemcpyAsync(d_keys_in_, h_keys_in_, n * width, hipMemcpyHostToDevice, stream_) != hipSuccess)
    return Refuse(Error(Code::kInternal, "sharded lookup: key upload failed"));
  stats_.key_bytes = (int)width;
  return Run(d_keys_in_, width, n, d_out);
}

}  // namespace hps
