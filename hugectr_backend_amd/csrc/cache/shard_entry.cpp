#include "shard_entry.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

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

namespace {
float MsSince(std::chrono::steady_clock::time_point t) {
  return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t).count();
}
}  // namespace

std::vector<ShardPass> PlanShardPasses(const uint32_t* counts, size_t T, size_t capacity) {
  std::vector<ShardPass> plan;
  if (capacity == 0) return plan;
  uint64_t offset = 0;
  ShardPass cur;
  cur.n.assign(T, 0);
  size_t in_pass = 0;
  for (size_t t = 0; t < T; ++t) {
    size_t left = counts[t];
    while (left) {
      const size_t take = std::min(left, capacity - in_pass);
      cur.n[t] += take;
      in_pass += take;
      left -= take;
      if (in_pass == capacity) {
        cur.offset = offset;
        plan.push_back(cur);
        offset += in_pass;
        cur.n.assign(T, 0);
        in_pass = 0;
      }
    }
  }
  if (in_pass) {
    cur.offset = offset;
    plan.push_back(cur);
  }
  return plan;
}

Status ShardedEntrySession::Create(std::shared_ptr<HierParameterServer> ps, const std::string& model, int entry_device,
                                   std::unique_ptr<ShardedEntrySession>* out) {
  if (!ps || !out) return Error(Code::kInvalidArg, "null argument");
  InferenceParams p;
  if (!ps->model_params(model, &p)) return Error(Code::kNotFound, "model '", model, "' is not in the parameter server configuration");
  if (!p.table_sharding)
    return Error(Code::kInvalidArg, "model '", model, "' is not table-sharded (ps.json \"table_sharding\": \"hash\")");
  if (std::find(p.deployed_devices.begin(), p.deployed_devices.end(), entry_device) == p.deployed_devices.end())
    return Error(Code::kInvalidArg, "model '", model, "': device ", entry_device, " is not in deployed_device_list");
  const size_t T = p.num_tables();
  if (T == 0 || T > (size_t)kMaxTables) return Error(Code::kInvalidArg, "model '", model, "': ", T, " tables (supported: 1..", kMaxTables, ")");
  std::unique_ptr<ShardedEntrySession> s(new ShardedEntrySession());
  s->ps_ = ps;
  s->params_ = p;
  s->P_ = (uint32_t)p.deployed_devices.size();
  s->device_ = entry_device;
  s->dedup_ = p.shard_dedup ? 1 : 0;
  s->transport_ = p.shard_transport_staged ? 1 : 0;
  s->piece_keys_ = p.shard_copy_piece_keys;
  size_t per_sample = 0;
  for (size_t c : p.maxnum_catfeature_query_per_table_per_sample) per_sample += c;
  s->max_keys_ = p.max_batchsize * per_sample;
  if (s->max_keys_ == 0) return Error(Code::kInvalidArg, "model '", model, "': max_batch_size * sum(maxnum_catfeature...) is 0");
  if (s->max_keys_ >= (1ull << 31) - 2) return Error(Code::kUnsupported, "more than 2^31 keys per request are not supported");
  s->max_tiles_ = s->max_keys_ / kTileKeys + T;
  // one owner's session: its fair share of a full request times the configured slack; more than that goes in passes
  const double share = std::ceil((double)s->max_keys_ / s->P_ * p.shard_capacity_factor);
  s->shard_cap_ = (size_t)std::min<double>((double)s->max_keys_, share + 1024.0);
  auto tabs = ps->tables_of(model);
  if (tabs.size() != T) return Error(Code::kNotFound, "model '", model, "': tables are not loaded");
  for (size_t t = 0; t < T; ++t) s->dims_.push_back(tabs[t]->dim());
  s->tables_ = tabs;

  // ---- the P shard sessions, each on its shard's device; their kernels read the bucket keys out of, and store the rows
  //      into, the entry device's memory: peer access from every other shard device to the entry device ----
  for (uint32_t sh = 0; sh < s->P_; ++sh) {
    auto cache = ps->get_shard_cache(model, sh);
    if (!cache)
      return Error(Code::kNotFound, "model '", model, "': shard ", sh, " has no embedding cache (create_embedding_cache_per_model first)");
    const int dev = cache->device();
    s->shard_device_.push_back(dev);
    if (dev != entry_device) {
      HIP_TRY(hipSetDevice(dev));
      int can = 0;
      HIP_TRY(hipDeviceCanAccessPeer(&can, dev, entry_device));
      if (!can) {
        // the staged_copy transport moves keys and rows with copies only and does without the mapping
        if (s->transport_ == 0)
          return Error(Code::kUnavailable, "model '", model, "': device ", dev, " (shard ", sh, ") cannot access device ", entry_device,
                       " as a peer; the peer_store transport of the table-sharded lookup needs peer access between the deployed devices"
                       " (ps.json \"shard_transport\": \"staged_copy\" does not)");
        s->peers_ok_ = false;
      } else {
        const hipError_t pe = hipDeviceEnablePeerAccess(entry_device, 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
          return Error(Code::kInternal, "hipDeviceEnablePeerAccess(", entry_device, ") from device ", dev, " failed: ", hipGetErrorString(pe));
        (void)hipGetLastError();
      }
    }
    std::unique_ptr<LookupSession> ls;
    HPS_RETURN_IF_ERROR(ps->create_lookup_session_sized(model, cache, s->shard_cap_, &ls));
    s->sessions_.push_back(std::move(ls));
  }

  // ---- the entry device's side ----
  HIP_TRY(hipSetDevice(entry_device));
  HIP_TRY(hipStreamCreateWithFlags(&s->stream_, hipStreamNonBlocking));
  for (hipEvent_t& e : s->ev_) HIP_TRY(hipEventCreate(&e));
  auto dev_alloc = [](auto** ptr, size_t count) -> Status {
    void* v = nullptr;
    if (hipMalloc(&v, std::max<size_t>(count, 1) * sizeof(**ptr)) != hipSuccess)
      return Error(Code::kInternal, "sharded entry session: out of device memory");
    *ptr = (std::remove_reference_t<decltype(**ptr)>*)v;
    return Status::Ok();
  };
  auto pin_alloc = [](auto** ptr, size_t count) -> Status {
    void* v = nullptr;
    if (hipHostMalloc(&v, std::max<size_t>(count, 1) * sizeof(**ptr), hipHostMallocDefault) != hipSuccess)
      return Error(Code::kInternal, "sharded entry session: out of page-locked memory");
    *ptr = (std::remove_reference_t<decltype(**ptr)>*)v;
    return Status::Ok();
  };
  const size_t N = s->max_keys_, P = s->P_;
  s->tiles_off_ = (sizeof(EntryDesc) + 127) & ~(size_t)127;
  const size_t block_bytes = s->tiles_off_ + s->max_tiles_ * sizeof(TileDesc);
  HPS_RETURN_IF_ERROR(pin_alloc(&s->h_block_, block_bytes));
  memset(s->h_block_, 0, block_bytes);
  HPS_RETURN_IF_ERROR(dev_alloc(&s->d_block_, block_bytes));
  HPS_RETURN_IF_ERROR(pin_alloc(&s->h_keys_, N));
  HPS_RETURN_IF_ERROR(dev_alloc(&s->d_keys_, N));
  HPS_RETURN_IF_ERROR(dev_alloc(&s->d_narrow_, N * 4));
  HPS_RETURN_IF_ERROR(dev_alloc(&s->d_rep_, N));
  uint64_t set_cap = 1024;
  while (set_cap < 2 * (uint64_t)N) set_cap <<= 1;
  HPS_RETURN_IF_ERROR(dev_alloc(&s->d_set_, set_cap));
  HIP_TRY(hipMemset(s->d_set_, 0, set_cap * sizeof(unsigned long long)));   // tag 0 is never used by a call
  s->set_mask_ = set_cap - 1;
  HPS_RETURN_IF_ERROR(dev_alloc(&s->d_hist_, s->max_tiles_ * P));
  HPS_RETURN_IF_ERROR(dev_alloc(&s->d_within_, s->max_tiles_ * P));
  HPS_RETURN_IF_ERROR(dev_alloc(&s->d_counts_, (P + 1) + P * T));
  HPS_RETURN_IF_ERROR(pin_alloc(&s->h_counts_, (P + 1) + P * T));
  HPS_RETURN_IF_ERROR(dev_alloc(&s->d_bkeys_, N));
  HPS_RETURN_IF_ERROR(dev_alloc(&s->d_bidx_, N));
  HIP_TRY(hipDeviceSynchronize());

  for (uint32_t sh = 0; sh < s->P_; ++sh) s->workers_.emplace_back(new Worker());
  ShardedEntrySession* raw = s.get();
  for (uint32_t sh = 0; sh < s->P_; ++sh) s->workers_[sh]->th = std::thread([raw, sh] { raw->WorkerMain(sh); });
  s->stats_.sent.assign(P, 0);
  s->stats_.passes.assign(P, 0);
  s->stats_.shard_ms.assign(P, 0.f);
  s->stats_.copy_wait_ms.assign(P, 0.f);
  *out = std::move(s);
  return Status::Ok();
}

ShardedEntrySession::~ShardedEntrySession() {
  for (auto& w : workers_) {
    if (!w) continue;
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->stop = true;
    }
    w->cv.notify_all();
  }
  for (auto& w : workers_) if (w && w->th.joinable()) w->th.join();
  FreeStaged();
  sessions_.clear();
  (void)hipSetDevice(device_);
  if (stream_) { (void)hipStreamSynchronize(stream_); (void)hipStreamDestroy(stream_); }
  for (hipEvent_t e : ev_) if (e) (void)hipEventDestroy(e);
  for (void* p : {(void*)d_block_, (void*)d_keys_, (void*)d_narrow_, (void*)d_rep_, (void*)d_set_, (void*)d_hist_, (void*)d_within_, (void*)d_counts_,
                  (void*)d_bkeys_, (void*)d_bidx_})
    if (p) (void)hipFree(p);
  for (void* p : {(void*)h_block_, (void*)h_keys_, (void*)h_counts_}) if (p) (void)hipHostFree(p);
}

void ShardedEntrySession::set_timing(bool b) {
  for (auto& s : sessions_) s->set_timing(b);
}

// One worker per shard: drives that shard's lookups on the shard's device while the others drive theirs.
void ShardedEntrySession::WorkerMain(uint32_t s) {
  (void)hipSetDevice(shard_device_[s]);
  Worker& w = *workers_[s];
  const size_t T = dims_.size();
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(w.mu);
      w.cv.wait(lk, [&] { return w.has_job || w.stop; });
      if (w.stop) return;
      w.has_job = false;
    }
    const auto t0 = std::chrono::steady_clock::now();
    w.wake_ms = std::chrono::duration<float, std::milli>(t0 - w.posted).count();
    Status st = Status::Ok();
    uint64_t misses = 0, unique = 0;
    if (w.staged) {
      st = ServeStaged(s, w, &misses, &unique);
    } else {
      for (const ShardPass& pass : w.plan) {
        st = sessions_[s]->lookup_from_device_indexed(w.keys + pass.offset, w.idx + pass.offset, w.out, pass.n.data(), T);
        if (!st.ok()) break;
        misses += sessions_[s]->last_miss_count();
        unique += sessions_[s]->last_unique_miss_count();
      }
    }
    {
      std::lock_guard<std::mutex> lk(w.mu);
      w.st = st;
      w.ms = MsSince(t0);
      w.misses = misses;
      w.unique = unique;
      w.done = true;
    }
    w.cv.notify_all();
  }
}

Status ShardedEntrySession::set_transport(int transport) {
  if (transport != 0 && transport != 1) return Error(Code::kInvalidArg, "transport must be 0 (peer_store) or 1 (staged_copy)");
  if (transport == 0 && !peers_ok_)
    return Error(Code::kUnavailable, "model '", params_.model_name, "': the peer_store transport needs peer access from every shard's device to device ",
                 device_, ", which this machine does not grant");
  transport_ = transport;
  return Status::Ok();
}

// floats of one piece's block: table-major, every table's rows on a 16-byte boundary (the gather kernels' float4 path)
size_t ShardedEntrySession::PieceFloats(const ShardPass& pass, const std::vector<uint32_t>& dims) {
  size_t fl = 0;
  for (size_t t = 0; t < dims.size(); ++t) fl = ((fl + 3) & ~(size_t)3) + pass.n[t] * dims[t];
  return (fl + 3) & ~(size_t)3;
}

Status ShardedEntrySession::EnsureStaged() {
  uint32_t maxd = 1;
  for (uint32_t d : dims_) maxd = std::max(maxd, d);
  // a block holds the largest piece any shard may be asked for (automatic mode: a co-located shard's whole bucket)
  size_t piece_max = 1024;
  for (uint32_t s = 0; s < P_; ++s) {
    const size_t w = piece_keys_ ? piece_keys_ : (shard_device_[s] == device_ ? sessions_[s]->max_keys() : kAutoPieceKeys);
    piece_max = std::max(piece_max, std::min(sessions_[s]->max_keys(), w));
  }
  const size_t want = piece_max * maxd + 4 * dims_.size() + 4;
  if (staged_.size() == P_ && staged_[0].stage_floats >= want) return Status::Ok();
  FreeStaged();
  staged_.resize(P_);
  for (uint32_t s = 0; s < P_; ++s) {
    StagedShard& g = staged_[s];
    HIP_TRY(hipSetDevice(shard_device_[s]));
    for (int b = 0; b < 2; ++b) {
      void* v = nullptr;
      if (hipMalloc(&v, want * sizeof(float)) != hipSuccess) return Error(Code::kInternal, "sharded entry session: out of device memory (piece blocks)");
      g.stage[b] = (float*)v;
      HIP_TRY(hipEventCreateWithFlags(&g.copied[b], hipEventDisableTiming));
    }
    g.stage_floats = want;
    void* v = nullptr;
    if (hipMalloc(&v, max_keys_ * sizeof(int64_t)) != hipSuccess) return Error(Code::kInternal, "sharded entry session: out of device memory (bucket keys)");
    g.okeys = (int64_t*)v;
    HIP_TRY(hipStreamCreateWithFlags(&g.copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipSetDevice(device_));
    HIP_TRY(hipStreamCreateWithFlags(&g.place_stream, hipStreamNonBlocking));
  }
  HIP_TRY(hipSetDevice(device_));
  return Status::Ok();
}

void ShardedEntrySession::FreeStaged() {
  for (uint32_t s = 0; s < staged_.size(); ++s) {
    StagedShard& g = staged_[s];
    (void)hipSetDevice(shard_device_[s]);
    if (g.copy_stream) { (void)hipStreamSynchronize(g.copy_stream); (void)hipStreamDestroy(g.copy_stream); }
    for (int b = 0; b < 2; ++b) {
      if (g.copied[b]) (void)hipEventDestroy(g.copied[b]);
      if (g.stage[b]) (void)hipFree(g.stage[b]);
    }
    if (g.okeys) (void)hipFree(g.okeys);
    (void)hipSetDevice(device_);
    if (g.place_stream) { (void)hipStreamSynchronize(g.place_stream); (void)hipStreamDestroy(g.place_stream); }
  }
  staged_.clear();
  (void)hipSetDevice(device_);
  if (d_recv_) { (void)hipFree(d_recv_); d_recv_ = nullptr; recv_floats_ = 0; }
}

// staged_copy, one owner's side (runs on the owner's worker thread): the bucket's keys come over in one copy; then piece by
// piece — ordinary lookup into a local block, the block shipped by a copy engine while the next piece is looked up, and the
// rows of a delivered block put into OUTPUT0 by a kernel on the entry GPU.
Status ShardedEntrySession::ServeStaged(uint32_t s, Worker& w, uint64_t* misses, uint64_t* unique) {
  StagedShard& g = staged_[s];
  const int dev = shard_device_[s];
  const size_t T = dims_.size();
  auto ship = [&](void* dst, int dst_dev, const void* src, int src_dev, size_t bytes) -> hipError_t {
    if (dst_dev == src_dev) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g.copy_stream);
    return hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, g.copy_stream);
  };
  HIP_TRY(hipSetDevice(dev));
  HIP_TRY(ship(g.okeys, dev, w.keys, device_, (size_t)w.total * sizeof(int64_t)));
  HIP_TRY(hipStreamSynchronize(g.copy_stream));
  struct Shipped { bool valid = false; int buf = 0; const ShardPass* pass = nullptr; size_t recv_off = 0; } prev;
  float wait_ms = 0.f;
  // the rows of a shipped block go to their places in OUTPUT0 (entry GPU) once the copy has landed
  auto deliver = [&](const Shipped& x) -> Status {
    const auto tw = std::chrono::steady_clock::now();
    HIP_TRY(hipEventSynchronize(g.copied[x.buf]));
    wait_ms += MsSince(tw);
    HIP_TRY(hipSetDevice(device_));
    PlaceArgs a;
    a.num_segments = 0;
    a.num_keys = 0;
    size_t fl = 0, first_key = x.pass->offset;
    auto flush = [&]() -> Status {
      if (a.num_segments == 0) return Status::Ok();
      a.start[a.num_segments] = a.num_keys;
      const hipError_t e = LaunchEntryPlace(w.desc, a, w.idx + first_key, w.recv + x.recv_off, g.place_stream);
      if (e != hipSuccess) return Error(Code::kInternal, "row placement launch failed: ", hipGetErrorString(e));
      first_key += a.num_keys;
      a.num_segments = 0;
      a.num_keys = 0;
      return Status::Ok();
    };
    for (size_t t = 0; t < T; ++t) {
      fl = (fl + 3) & ~(size_t)3;
      const size_t nt = x.pass->n[t];
      if (nt == 0) continue;
      if (a.num_segments == (uint32_t)kPlaceMaxSegments) HPS_RETURN_IF_ERROR(flush());
      a.start[a.num_segments] = a.num_keys;
      a.table[a.num_segments] = (uint32_t)t;
      a.src_off[a.num_segments] = (uint32_t)fl;
      ++a.num_segments;
      a.num_keys += (uint32_t)nt;
      fl += nt * dims_[t];
    }
    HPS_RETURN_IF_ERROR(flush());
    HIP_TRY(hipSetDevice(dev));
    return Status::Ok();
  };
  size_t recv_off = 0;
  std::vector<float*> outs(T, nullptr);
  uint32_t p = 0;
  for (const ShardPass& pass : w.plan) {
    const int b = (int)(p & 1u);
    size_t fl = 0;
    for (size_t t = 0; t < T; ++t) {
      fl = (fl + 3) & ~(size_t)3;
      outs[t] = g.stage[b] + fl;
      fl += pass.n[t] * dims_[t];
    }
    fl = (fl + 3) & ~(size_t)3;
    if (fl > g.stage_floats) return Error(Code::kInternal, "staged_copy: a piece of ", fl, " floats does not fit its block of ", g.stage_floats);
    // (block b's previous copy — piece p - 2 — was waited for when that piece was delivered)
    HPS_RETURN_IF_ERROR(sessions_[s]->lookup_from_device(g.okeys + pass.offset, outs.data(), pass.n.data(), T));
    *misses += sessions_[s]->last_miss_count();
    *unique += sessions_[s]->last_unique_miss_count();
    HIP_TRY(hipSetDevice(dev));
    HIP_TRY(ship(w.recv + recv_off, device_, g.stage[b], dev, fl * sizeof(float)));
    HIP_TRY(hipEventRecord(g.copied[b], g.copy_stream));
    w.copied_bytes += fl * sizeof(float);
    if (prev.valid) HPS_RETURN_IF_ERROR(deliver(prev));
    prev.valid = true; prev.buf = b; prev.pass = &pass; prev.recv_off = recv_off;
    recv_off += fl;
    ++p;
  }
  if (prev.valid) HPS_RETURN_IF_ERROR(deliver(prev));
  HIP_TRY(hipSetDevice(device_));
  HIP_TRY(hipStreamSynchronize(g.place_stream));
  HIP_TRY(hipSetDevice(dev));
  w.copy_wait_ms = wait_ms;
  return Status::Ok();
}

Status ShardedEntrySession::lookup_from_device(const int64_t* d_keys_flat, float* const* d_out, const size_t* n, size_t T) {
  if (T != dims_.size()) return Error(Code::kInvalidArg, "lookup: got ", T, " tables, model '", params_.model_name, "' has ", dims_.size());
  if (!n || !d_out) return Error(Code::kInvalidArg, "null argument");
  HIP_TRY(hipSetDevice(device_));
  stats_.key_stage_ms = 0.f;
  stats_.key_bytes = 8;
  return Run(d_keys_flat, d_out, n, T);
}

Status ShardedEntrySession::lookup(const void* const* h_keys_per_table, float* const* d_out, const size_t* n, size_t T) {
  if (T != dims_.size()) return Error(Code::kInvalidArg, "lookup: got ", T, " tables, model '", params_.model_name, "' has ", dims_.size());
  if (!h_keys_per_table || !n || !d_out) return Error(Code::kInvalidArg, "null argument");
  size_t N = 0;
  for (size_t t = 0; t < T; ++t) N += n[t];
  if (N > max_keys_)
    return Error(Code::kInvalidArg, "lookup: ", N, " keys exceed the request capacity of ", max_keys_,
                 " (max_batch_size x sum(maxnum_catfeature_query_per_table_per_sample))");
  HIP_TRY(hipSetDevice(device_));
  stats_.key_stage_ms = 0.f;
  if (N == 0) return Run(nullptr, d_out, n, T);
  const auto t0 = std::chrono::steady_clock::now();
  // one flat array in page-locked memory (Triton's pinned input pool): DMA in place; else staged through page-locked memory
  // in 32 K-key tasks on the serving pool, 4-MB groups, each group's upload enqueued while the next is staged
  bool flat = true;
  const int64_t* base = nullptr;
  {
    const int64_t* expect = nullptr;
    for (size_t t = 0; t < T; ++t) {
      if (n[t] == 0) continue;
      const int64_t* p = (const int64_t*)h_keys_per_table[t];
      if (!p) return Error(Code::kInvalidArg, "lookup: null key pointer for table ", t);
      if (!base) base = p;
      else if (p != expect) flat = false;
      expect = p + n[t];
    }
  }
  bool in_place = false;
  if (flat) {
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof attr);
    if (hipPointerGetAttributes(&attr, base) == hipSuccess && attr.type == hipMemoryTypeHost) in_place = true;
    else (void)hipGetLastError();
  }
  if (in_place) {
    HIP_TRY(hipMemcpyAsync(d_keys_, base, N * sizeof(int64_t), hipMemcpyHostToDevice, stream_));
  } else {
    constexpr size_t kTaskKeys = 32768, kGroupKeys = (4u << 20) / sizeof(int64_t);
    struct Task { const int64_t* src; size_t off, n; uint64_t base; };
    std::vector<Task> tasks;
    size_t off = 0;
    key_base_.assign(T, 0);
    for (size_t t = 0; t < T; ++t) {
      const int64_t* p = (const int64_t*)h_keys_per_table[t];
      key_base_[t] = tables_[t]->min_key();
      for (size_t b = 0; b < n[t]; b += kTaskKeys) tasks.push_back({p + b, off + b, std::min(kTaskKeys, n[t] - b), (uint64_t)key_base_[t]});
      off += n[t];
    }
    // The keys cross PCIe at the width they need, as a replica's do (LookupSession::lookup, key_pack.h): offsets from their
    // table's smallest key, 3 bytes each (packed) when they all fit 24 bits, uint32 when 32, else 8 bytes as they are; a look at
    // three keys of every task decides the attempt, the copy loop ORs everything it sees and a key too wide restages the
    // request at 8 bytes (and narrowing pauses for 256 calls, doubling per failure in a row).  On the entry GPU a small kernel
    // widens them again (hps_entry_widen) — the bucket kernels and the owners read int64.
    uint32_t width = 8;
    if (narrow_backoff_ > 0) --narrow_backoff_;
    else if (N >= 4 * kTaskKeys) {
      uint64_t sample = 0;
      for (const Task& tk : tasks)
        sample |= ((uint64_t)tk.src[0] - tk.base) | ((uint64_t)tk.src[tk.n / 2] - tk.base) | ((uint64_t)tk.src[tk.n - 1] - tk.base);
      width = (sample >> 32) ? 8u : (sample >> 24) ? 4u : 3u;
    }
    for (; width < 8;) {
      std::atomic<uint64_t> high_or{0};
      uint8_t* dst8 = reinterpret_cast<uint8_t*>(h_keys_);
      bool failed = false;
      size_t g0n = 0;
      while (g0n < tasks.size() && !failed) {
        size_t g1 = g0n, keys_in_group = 0;
        while (g1 < tasks.size() && (keys_in_group == 0 || keys_in_group + tasks[g1].n <= kGroupKeys)) keys_in_group += tasks[g1++].n;
        auto body = [&](size_t i) {
          const Task& tk = tasks[g0n + i];
          const uint64_t high = width == 4 ? PackKeys32(tk.src, tk.n, reinterpret_cast<uint32_t*>(dst8) + tk.off, tk.base)
                                           : PackKeys24(tk.src, tk.n, dst8 + 3 * tk.off, tk.base);
          if (high >> (8 * width)) high_or.fetch_or(high, std::memory_order_relaxed);
        };
        if (g1 - g0n <= 2) for (size_t i = 0; i < g1 - g0n; ++i) body(i);
        else ThreadPool::Serving().ParallelFor(g1 - g0n, body);
        if (high_or.load(std::memory_order_relaxed) != 0) { failed = true; break; }
        const size_t first = tasks[g0n].off, count = tasks[g1 - 1].off + tasks[g1 - 1].n - first;
        HIP_TRY(hipMemcpyAsync(d_narrow_ + first * width, dst8 + first * width, count * width, hipMemcpyHostToDevice, stream_));
        g0n = g1;
      }
      if (!failed) {
        narrow_streak_ = 0;
        stats_.key_bytes = (int)width;
        stats_.key_stage_ms = MsSince(t0);
        return Run(d_keys_, d_out, n, T, width);
      }
      // a key too wide for the attempt: the groups in flight read the staging buffer that is about to be rewritten
      HIP_TRY(hipStreamSynchronize(stream_));
      const uint64_t seen = high_or.load(std::memory_order_relaxed);
      if (width == 3 && (seen >> 32) == 0) { width = 4; continue; }
      narrow_backoff_ = 256 << (narrow_streak_ < 8 ? narrow_streak_ : 8);
      if (narrow_streak_ < 8) ++narrow_streak_;
      width = 8;
    }
    size_t g0 = 0;
    while (g0 < tasks.size()) {
      size_t g1 = g0, keys_in_group = 0;
      while (g1 < tasks.size() && (keys_in_group == 0 || keys_in_group + tasks[g1].n <= kGroupKeys)) keys_in_group += tasks[g1++].n;
      auto body = [&](size_t i) { const Task& tk = tasks[g0 + i]; CopyKeys64Streaming(tk.src, tk.n, h_keys_ + tk.off); StreamFence(); };
      if (g1 - g0 <= 2) for (size_t i = 0; i < g1 - g0; ++i) body(i);
      else ThreadPool::Serving().ParallelFor(g1 - g0, body);
      const size_t first = tasks[g0].off, count = tasks[g1 - 1].off + tasks[g1 - 1].n - first;
      HIP_TRY(hipMemcpyAsync(d_keys_ + first, h_keys_ + first, count * sizeof(int64_t), hipMemcpyHostToDevice, stream_));
      g0 = g1;
    }
  }
  stats_.key_bytes = 8;
  stats_.key_stage_ms = MsSince(t0);
  return Run(d_keys_, d_out, n, T);
}

Status ShardedEntrySession::Run(const int64_t* d_keys_flat, float* const* d_out, const size_t* n, size_t T, uint32_t narrow_bytes) {
  const auto t0 = std::chrono::steady_clock::now();
  EntryDesc& d = *reinterpret_cast<EntryDesc*>(h_block_);
  TileDesc* tiles = reinterpret_cast<TileDesc*>(h_block_ + tiles_off_);
  uint64_t N = 0;
  uint32_t nt = 0;
  // (before any tile descriptor is written: the page-locked block holds max_tiles_ of them)
  for (size_t t = 0; t < T; ++t) N += n[t];
  if (N > max_keys_)
    return Error(Code::kInvalidArg, "lookup: ", N, " keys exceed the request capacity of ", max_keys_,
                 " (max_batch_size x sum(maxnum_catfeature_query_per_table_per_sample))");
  N = 0;
  for (size_t t = 0; t < T; ++t) {
    if (n[t] && !d_out[t]) return Error(Code::kInvalidArg, "lookup: null output pointer for table ", t);
    d.key_start[t] = N;
    d.first_tile[t] = nt;
    d.out[t] = d_out[t];
    d.dim[t] = dims_[t];
    d.key_base[t] = (narrow_bytes && t < key_base_.size()) ? key_base_[t] : 0;
    for (uint64_t b = 0; b < n[t]; b += kTileKeys) tiles[nt++] = TileDesc{N + b, (uint32_t)std::min<uint64_t>(kTileKeys, n[t] - b), (uint32_t)t};
    N += n[t];
  }
  d.key_start[T] = N;
  d.first_tile[T] = nt;
  d.num_tables = (uint32_t)T;
  d.num_shards = P_;
  d.num_tiles = nt;
  d.total_keys = N;
  stats_.keys = N;
  stats_.unique_keys = 0;
  stats_.misses = stats_.unique_misses = 0;
  stats_.bucket_ms = stats_.lookup_ms = stats_.expand_ms = 0.f;
  std::fill(stats_.sent.begin(), stats_.sent.end(), 0);
  std::fill(stats_.passes.begin(), stats_.passes.end(), 0);
  std::fill(stats_.shard_ms.begin(), stats_.shard_ms.end(), 0.f);
  std::fill(stats_.copy_wait_ms.begin(), stats_.copy_wait_ms.end(), 0.f);
  stats_.transport = transport_;
  stats_.copied_bytes = 0;
  if (N == 0) return Status::Ok();
  if (!d_keys_flat) return Error(Code::kInvalidArg, "lookup: null key pointer");

  // ---- bucket the request by owner on the entry device ----
  const EntryDesc* dd = reinterpret_cast<const EntryDesc*>(d_block_);
  const TileDesc* dt = reinterpret_cast<const TileDesc*>(d_block_ + tiles_off_);
  HIP_TRY(hipMemcpyAsync(d_block_, h_block_, tiles_off_ + (size_t)nt * sizeof(TileDesc), hipMemcpyHostToDevice, stream_));
  if (narrow_bytes) HIP_TRY(LaunchEntryWiden(dd, dt, nt, d_narrow_, narrow_bytes, d_keys_, stream_));
  // Input dedup, two levels (entry_kernels.hip): inside tiles of 1,024 keys (LDS), then call-wide — one device-scope CAS per tile
  // representative: 1.7 M of them for a request that repeats little, 113 us at the chip's ~15 G atomics/s to spare 5 % of the
  // rows.  Adaptive (shard_dedup, the default): a big request of which more than 90 % travelled anyway sends the next 31 requests
  // through the tile level only; one that then finds repeats inside its tiles (< 80 % travel) brings the call-wide level back
  // at once.  Rows are exact at every level: a key that travels twice is looked up twice.
  const bool dedup = dedup_ != 0;
  const int level = !dedup ? 0 : (dedup_ == 1 && tile_only_left_ > 0) ? 1 : 2;
  stats_.dedup_level = level;
  if (level == 2) {
    if (++set_tag_ == 0) {   // 2^32 requests later: entries of the first ones would look like this one's
      HIP_TRY(hipMemsetAsync(d_set_, 0, (set_mask_ + 1) * sizeof(unsigned long long), stream_));
      set_tag_ = 1;
    }
    HIP_TRY(LaunchEntryDedup(dd, dt, nt, d_keys_flat, N, d_set_, set_mask_, set_tag_, d_rep_, stream_));
  } else if (level == 1) {
    HIP_TRY(LaunchEntryDedup(dd, dt, nt, d_keys_flat, N, nullptr, 0, 0, d_rep_, stream_));
  }
  uint32_t* d_base = d_counts_;
  uint32_t* d_cnt = d_counts_ + (P_ + 1);
  HIP_TRY(LaunchEntryBucket(dd, dt, nt, P_, d_keys_flat, dedup ? d_rep_ : nullptr, d_hist_, d_within_, d_base, d_cnt, d_bkeys_, d_bidx_, stream_));
  HIP_TRY(hipMemcpyAsync(h_counts_, d_counts_, ((size_t)(P_ + 1) + (size_t)P_ * T) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
  // the bucket arrays are complete (and visible to the peers' kernels) once this returns
  HIP_TRY(hipStreamSynchronize(stream_));
  stats_.bucket_ms = MsSince(t0);
  const uint32_t* base = h_counts_;
  const uint32_t* counts = h_counts_ + (P_ + 1);
  stats_.unique_keys = base[P_];
  if (dedup_ == 1) {
    constexpr uint64_t kAdaptiveMinKeys = 1u << 16;
    // (back to both levels at once — the safe direction —, but then for at least 8 requests: traffic that alternates between the
    //  two bounds, one request repeating little, the next much, would otherwise change the arrangement with every request)
    if (level_hold_ > 0) --level_hold_;
    if (level == 2 && N >= kAdaptiveMinKeys && stats_.unique_keys * 10 > N * 9 && level_hold_ == 0) { tile_only_left_ = 31; ++stats_.dedup_flips; }
    else if (level == 1) {
      const bool back = stats_.unique_keys * 10 < N * 8;
      tile_only_left_ = back ? 0 : tile_only_left_ - 1;
      if (tile_only_left_ == 0) { ++stats_.dedup_flips; if (back) level_hold_ = 8; }
    }
  }

  // ---- every owner looks its bucket up, all of them side by side; rows land in d_out over the peer mappings (peer_store) or
  //      come over block by block through the receive buffer (staged_copy) ----
  const auto t1 = std::chrono::steady_clock::now();
  const bool staged = transport_ == 1;
  std::vector<std::vector<ShardPass>> plans(P_);
  std::vector<size_t> recv_base(P_ + 1, 0);
  for (uint32_t s = 0; s < P_; ++s) {
    const uint32_t total = base[s + 1] - base[s];
    if (total && !staged) plans[s] = PlanShardPasses(counts + (size_t)s * T, T, sessions_[s]->max_keys());
    if (total && staged) {
      // pieces of equal size: as many as shard_copy_piece_keys asks for, none of them a short tail (a piece costs ~0.15 ms of call
      // overhead whatever it holds)
      // (automatic: a shard on the entry GPU itself ships with a local copy — nothing to put the next lookup under: one piece)
      const size_t want = piece_keys_ ? piece_keys_ : (shard_device_[s] == device_ ? sessions_[s]->max_keys() : kAutoPieceKeys);
      const size_t cap = std::min(sessions_[s]->max_keys(), want);
      const size_t pieces = (total + cap - 1) / cap;
      plans[s] = PlanShardPasses(counts + (size_t)s * T, T, (total + pieces - 1) / pieces);
    }
    size_t fl = 0;
    if (staged) for (const ShardPass& pass : plans[s]) fl += PieceFloats(pass, dims_);
    recv_base[s + 1] = recv_base[s] + fl;
  }
  if (staged) {
    HPS_RETURN_IF_ERROR(EnsureStaged());
    if (recv_base[P_] > recv_floats_) {
      // (grows with the largest request seen; nothing of an earlier request is in flight here)
      if (d_recv_) { HIP_TRY(hipFree(d_recv_)); d_recv_ = nullptr; recv_floats_ = 0; }
      const size_t want = recv_base[P_] + recv_base[P_] / 8 + 1024;
      void* v = nullptr;
      if (hipMalloc(&v, want * sizeof(float)) != hipSuccess) return Error(Code::kInternal, "sharded entry session: out of device memory (receive buffer)");
      d_recv_ = (float*)v;
      recv_floats_ = want;
    }
  }
  for (uint32_t s = 0; s < P_; ++s) {
    const uint32_t total = base[s + 1] - base[s];
    stats_.sent[s] = total;
    if (total == 0) continue;
    Worker& w = *workers_[s];
    {
      std::lock_guard<std::mutex> lk(w.mu);
      w.plan = std::move(plans[s]);
      w.keys = d_bkeys_ + base[s];
      w.idx = d_bidx_ + base[s];
      w.out = d_out;
      w.staged = staged;
      w.total = total;
      w.recv = staged ? d_recv_ + recv_base[s] : nullptr;
      w.desc = dd;
      w.copy_wait_ms = 0.f;
      w.copied_bytes = 0;
      w.done = false;
      w.has_job = true;
      w.posted = std::chrono::steady_clock::now();
      stats_.passes[s] = (uint32_t)w.plan.size();
    }
    w.cv.notify_all();
  }
  Status first = Status::Ok();
  for (uint32_t s = 0; s < P_; ++s) {
    if (stats_.sent[s] == 0) continue;
    Worker& w = *workers_[s];
    std::unique_lock<std::mutex> lk(w.mu);
    w.cv.wait(lk, [&] { return w.done; });
    if (!w.st.ok() && first.ok()) first = Error(w.st.code(), "shard ", s, " (device ", shard_device_[s], "): ", w.st.message());
    stats_.shard_ms[s] = w.ms;
    stats_.copy_wait_ms[s] = w.copy_wait_ms;
    stats_.copied_bytes += w.copied_bytes;
    stats_.misses += w.misses;
    stats_.unique_misses += w.unique;
  }
  stats_.lookup_ms = MsSince(t1);
  HPS_RETURN_IF_ERROR(first);
  static const bool kTrace = std::getenv("HPS_TRACE_TAIL") != nullptr;   // the slow-call trace also covers entry requests
  if (kTrace) {
    std::string per;
    char buf[192];
    for (uint32_t s = 0; s < P_; ++s) {
      const float* ph = sessions_[s]->last_phase_ms();   // (of the shard's last pass)
      snprintf(buf, sizeof buf, " [%u: woke %.3f, lookups %.3f (counts on the host %.3f, host gather %.3f, call %.3f)]", s, workers_[s]->wake_ms,
               workers_[s]->ms, ph[0], ph[1], ph[3]);
      per += buf;
    }
    fprintf(stderr, "[hps entry] bucket %.3f ms, shard lookups %.3f ms:%s\n", stats_.bucket_ms, stats_.lookup_ms, per.c_str());
  }

  // ---- the request's repeated keys take their representative's row (local copy on the entry device) ----
  if (dedup && stats_.unique_keys < N) {
    const auto t2 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(device_));
    HIP_TRY(LaunchEntryExpand(dd, d_rep_, N, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    stats_.expand_ms = MsSince(t2);
  }
  return Status::Ok();
}

}  // namespace hps
