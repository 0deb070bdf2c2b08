// HierParameterServer + the background insert path of EmbeddingCache.
#include <hip/hip_runtime.h>

#include <chrono>
#include <thread>

#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <filesystem>
#include <cstring>

#include "engine.h"

#include <regex>
#include "kernels.h"

namespace hps {

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return ::hps::Error(::hps::Code::kInternal, #expr, " failed: ", hipGetErrorString(_e), " (", \
                          __FILE__, ":", __LINE__, ")");                                           \
  } while (0)

// =================================================================================================
// EmbeddingCache: background insertion (async-insert mode, refresh)
// =================================================================================================
struct EmbeddingCache::Inserter {
  hipStream_t stream = nullptr;
  int64_t* h_keys = nullptr;  // pinned
  int64_t* d_keys = nullptr;
  float* h_rows = nullptr;    // pinned
  float* d_rows = nullptr;
  uint8_t* h_found = nullptr; // pinned
  uint8_t* d_found = nullptr;
  MissDesc* h_md = nullptr;   // pinned
  MissDesc* d_md = nullptr;
  uint64_t* h_ks = nullptr;   // pinned  key_start for the insert kernel (= useg_start)
  uint64_t* d_ks = nullptr;
  uint32_t* d_stats = nullptr;
  uint32_t* h_stats = nullptr;
  size_t cap_keys = 0, cap_floats = 0;
};

static constexpr size_t kInsChunkKeys = 1u << 18;

void EmbeddingCache::FreeInserter() {
  if (!ins_) return;
  Inserter& I = *ins_;
  if (I.stream) { (void)hipStreamSynchronize(I.stream); (void)hipStreamDestroy(I.stream); }
  for (void* p : {(void*)I.h_keys, (void*)I.h_rows, (void*)I.h_found, (void*)I.h_md, (void*)I.h_ks, (void*)I.h_stats})
    if (p) (void)hipHostFree(p);
  for (void* p : {(void*)I.d_keys, (void*)I.d_rows, (void*)I.d_found, (void*)I.d_md, (void*)I.d_ks, (void*)I.d_stats})
    if (p) (void)hipFree(p);
  delete ins_;
  ins_ = nullptr;
}

void EmbeddingCache::WaitAsync() {
  std::unique_lock<std::mutex> lk(pend_mu_);
  pend_cv_.wait(lk, [&] { return pending_async_ == 0; });
}

Status EmbeddingCache::InsertKeys(HierParameterServer* ps, const std::vector<std::vector<int64_t>>& keys_per_table, const InsertPacing* pacing,
                                  uint64_t* row_bytes) {
  const size_t T = num_tables();
  if (keys_per_table.size() != T) return Error(Code::kInvalidArg, "InsertKeys: table count mismatch");
  if (static_) return Status::Ok();
  auto tables = ps->tables_of(model_);
  if (tables.size() != T) return Error(Code::kNotFound, "InsertKeys: model '", model_, "' is not loaded");
  std::unique_lock<std::mutex> lk(ins_mu_);
  HIP_TRY(hipSetDevice(cfg_.device_id_));
  size_t maxD = 1;
  for (size_t t = 0; t < T; ++t) maxD = std::max<size_t>(maxD, cfg_.embedding_vec_size_[t]);
  if (!ins_) {
    ins_ = new Inserter();
    Inserter& I = *ins_;
    HIP_TRY(hipStreamCreateWithFlags(&I.stream, hipStreamNonBlocking));
    I.cap_keys = kInsChunkKeys;
    I.cap_floats = kInsChunkKeys * maxD + 4 * T;
    HIP_TRY(hipHostMalloc((void**)&I.h_keys, I.cap_keys * sizeof(int64_t), hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&I.d_keys, I.cap_keys * sizeof(int64_t)));
    HIP_TRY(hipHostMalloc((void**)&I.h_rows, I.cap_floats * sizeof(float), hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&I.d_rows, I.cap_floats * sizeof(float)));
    HIP_TRY(hipHostMalloc((void**)&I.h_found, I.cap_keys, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&I.d_found, I.cap_keys));
    HIP_TRY(hipHostMalloc((void**)&I.h_md, sizeof(MissDesc), hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&I.d_md, sizeof(MissDesc)));
    HIP_TRY(hipHostMalloc((void**)&I.h_ks, sizeof(uint64_t) * ((size_t)kMaxTables + 1), hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&I.d_ks, sizeof(uint64_t) * ((size_t)kMaxTables + 1)));
    HIP_TRY(hipMalloc((void**)&I.d_stats, (size_t)kStatLines * kAccStride * sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc((void**)&I.h_stats, (size_t)kStatLines * kAccStride * sizeof(uint32_t), hipHostMallocDefault));
  }
  Inserter& I = *ins_;
  std::vector<size_t> done(T, 0);
  HIP_TRY(hipMemsetAsync(I.d_stats, 0, (size_t)kStatLines * kAccStride * sizeof(uint32_t), I.stream));
  const size_t piece_cap = (pacing && pacing->piece_keys) ? std::min(I.cap_keys, std::max<size_t>(pacing->piece_keys, 256)) : I.cap_keys;
  uint64_t calls_seen = calls_so_far();
  for (;;) {
    const auto piece_t0 = std::chrono::steady_clock::now();
    MissDesc& md = *I.h_md;
    size_t uq = 0, fl = 0;
    std::vector<HierParameterServer::FetchJob> jobs;
    for (size_t t = 0; t < T; ++t) {
      const uint32_t D = cfg_.embedding_vec_size_[t];
      fl = (fl + 3) & ~(size_t)3;
      md.useg_start[t] = uq;
      I.h_ks[t] = uq;
      md.chunk_lo[t] = 0;
      md.stage_off[t] = fl;
      const size_t take = std::min(keys_per_table[t].size() - done[t], piece_cap - uq);
      md.chunk_hi[t] = (uint32_t)take;
      if (take) {
        memcpy(I.h_keys + uq, keys_per_table[t].data() + done[t], take * sizeof(int64_t));
        jobs.push_back({tables[t].get(), I.h_keys + uq, take, I.h_rows + fl, D, cfg_.default_value_[t], I.h_found + uq});
      }
      done[t] += take;
      uq += take;
      fl += take * D;
    }
    md.useg_start[T] = uq;
    I.h_ks[T] = uq;
    if (uq == 0) break;
    HPS_RETURN_IF_ERROR(ps->FetchMulti(jobs, pacing ? pacing->max_threads : 0));
    if (row_bytes) *row_bytes += fl * sizeof(float);
    if (pacing) refresh_rows_.fetch_add(uq, std::memory_order_relaxed);
    HIP_TRY(hipMemcpyAsync(I.d_md, I.h_md, sizeof(MissDesc), hipMemcpyHostToDevice, I.stream));
    HIP_TRY(hipMemcpyAsync(I.d_ks, I.h_ks, sizeof(uint64_t) * (T + 1), hipMemcpyHostToDevice, I.stream));
    HIP_TRY(hipMemcpyAsync(I.d_keys, I.h_keys, uq * sizeof(int64_t), hipMemcpyHostToDevice, I.stream));
    HIP_TRY(hipMemcpyAsync(I.d_rows, I.h_rows, fl * sizeof(float), hipMemcpyHostToDevice, I.stream));
    HIP_TRY(hipMemcpyAsync(I.d_found, I.h_found, uq, hipMemcpyHostToDevice, I.stream));
    const uint32_t epoch = NextEpoch();
    HIP_TRY(hipStreamSynchronize(I.stream));  // copies done: keep the writer window = the insert kernel only
    BeginWrite(I.stream);
    const hipError_t e = LaunchCacheInsert(d_tables_, (uint32_t)T, I.d_md, uq, I.d_ks, I.d_keys, I.d_rows, I.d_found,
                                           InsertStamps(epoch), I.d_stats, cu_count_, I.stream);
    EndWrite(I.stream);
    if (e != hipSuccess) return Error(Code::kInternal, "cache insert launch failed: ", hipGetErrorString(e));
    HIP_TRY(hipStreamSynchronize(I.stream));
    if (pacing && pacing->link_share < 1.0) {
      // sessions called since the last piece: they are serving — leave them their share of the link, the CPUs and the cache
      const uint64_t calls_now = calls_so_far();
      if (calls_now != calls_seen) {
        const double took = std::chrono::duration<double>(std::chrono::steady_clock::now() - piece_t0).count();
        const double pause = std::min(0.05, took * (1.0 / pacing->link_share - 1.0));
        // The pause is spent WITHOUT the inserter: the background inserts of async-insert models and the update consumer use the
        // same stream and staging, and would otherwise stand still behind a refresh that sleeps most of the time.  Nothing of this
        // call is in flight here (the stream was drained); its statistics so far are handed in before the lock goes.
        HIP_TRY(hipMemcpyAsync(I.h_stats, I.d_stats, (size_t)kStatLines * kAccStride * sizeof(uint32_t), hipMemcpyDeviceToHost, I.stream));
        HIP_TRY(hipStreamSynchronize(I.stream));
        AddStatLines(I.h_stats);
        lk.unlock();
        std::this_thread::sleep_for(std::chrono::duration<double>(pause));
        lk.lock();
        HIP_TRY(hipSetDevice(cfg_.device_id_));
        HIP_TRY(hipMemsetAsync(I.d_stats, 0, (size_t)kStatLines * kAccStride * sizeof(uint32_t), I.stream));
      }
      calls_seen = calls_so_far();
    }
  }
  HIP_TRY(hipMemcpyAsync(I.h_stats, I.d_stats, (size_t)kStatLines * kAccStride * sizeof(uint32_t), hipMemcpyDeviceToHost, I.stream));
  HIP_TRY(hipStreamSynchronize(I.stream));
  AddStatLines(I.h_stats);
  return Status::Ok();
}

// =================================================================================================
// HierParameterServer
// =================================================================================================
namespace {
// The NUMA node the host tier's worker pools (and with them the tables they load) should live on (thread_pool.h): the node the
// deployed GPUs hang off when they share one, the caller's own node for a deployment without GPU caches, -1 (no binding) when the
// GPUs span nodes or the machine has one node.
int NumaNodeOfCpu(int cpu) {
  for (int n = 0; n < 64; ++n) {
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpu%d", n, cpu);
    if (access(path, F_OK) == 0) return n;
  }
  return -1;
}
int PreferredNumaNode(const ParameterServerConfig& cfg) {
  if (access("/sys/devices/system/node/node1", F_OK) != 0) return -1;
  std::vector<int> devs;
  for (const auto& kv : cfg.models)
    if (kv.second.use_gpu_embedding_cache)
      for (int d : kv.second.deployed_devices) if (std::find(devs.begin(), devs.end(), d) == devs.end()) devs.push_back(d);
  // a CPU-only deployment is not bound: its host tier is all there is, and one node's CPUs would halve it (HPS_NUMA_NODE=<n>
  // still binds on request — ThreadPool::BindToNumaNode reads it)
  if (devs.empty()) return -1;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); return -1; }
  int node = -1;
  for (int d : devs) {
    if (d < 0 || d >= ndev) return -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, d) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char* c = bus; *c; ++c) *c = (char)tolower((unsigned char)*c);
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    int n = -1;
    const bool ok = fscanf(f, "%d", &n) == 1;
    fclose(f);
    if (!ok || n < 0) return -1;
    if (node >= 0 && n != node) return -1;     // replicas on both sockets: the one host tier serves them all
    node = n;
  }
  return node;
}
}  // namespace

HierParameterServer::~HierParameterServer() {
  {
    std::shared_ptr<UpdateConsumer> u;
    { std::lock_guard<std::mutex> lk(updates_mu_); u = std::move(updates_); }
    u.reset();   // the consumer thread calls back into this object
  }
  for (auto& kv : caches_) kv.second->WaitAsync();
  caches_.clear();
}

Status HierParameterServer::create(const std::string& path, std::shared_ptr<HierParameterServer>* out) {
  ParameterServerConfig cfg;
  HPS_RETURN_IF_ERROR(ParseParameterServerFile(path, &cfg));
  return create_from_config(cfg, true, out);
}

Status HierParameterServer::create_from_text(const std::string& text, std::shared_ptr<HierParameterServer>* out) {
  ParameterServerConfig cfg;
  HPS_RETURN_IF_ERROR(ParseParameterServerText(text, &cfg));
  return create_from_config(cfg, true, out);
}

Status HierParameterServer::create_from_config(const ParameterServerConfig& cfg, bool load_tables,
                                               std::shared_ptr<HierParameterServer>* out) {
  std::shared_ptr<HierParameterServer> ps(new HierParameterServer());
  ps->cfg_ = cfg;
  ThreadPool::BindToNumaNode(PreferredNumaNode(cfg));   // (before the pools start; the first server of the process decides)
  ps->pool_ = &ThreadPool::Global();
  if (!cfg.support_int64_key)
    return Error(Code::kUnsupported,
                 "supportlonglong=false: only 64-bit keys are supported (as in the reference backend, "
                 "model_state.cpp:213-218, hps.cc:573)");
  // Tiers this build does not have must not be configured silently away: a model that relies on a Redis cluster
  // or on RocksDB for rows the host tier does not hold would quietly serve default vectors.
  if (cfg.volatile_db.type == DatabaseType::RedisCluster)
    return Error(Code::kUnsupported, "volatile_db.type = redis_cluster is not implemented in this build "
                                     "(available: hash_map, parallel_hash_map: the in-process host tier)");
  if (cfg.volatile_db.type == DatabaseType::Disabled || cfg.volatile_db.type == DatabaseType::RocksDB)
    return Error(Code::kUnsupported, "volatile_db.type = ", ToString(cfg.volatile_db.type),
                 ": the in-process host tier (hash_map / parallel_hash_map) is the only volatile database of this build");
  // persistent_db.type = rocks_db selects this build's persistent tier: a memory-mapped row store under
  // persistent_db.path (csrc/ps/host_table.h); RocksDB's own file format is neither read nor written.
  if (cfg.persistent_db.type != DatabaseType::Disabled && cfg.persistent_db.type != DatabaseType::RocksDB)
    return Error(Code::kUnsupported, "persistent_db.type = ", ToString(cfg.persistent_db.type),
                 " is not a persistent database (available: disabled, rocks_db)");
  if (cfg.update_source.type == UpdateSourceType::KafkaMessageQueue)
    return Error(Code::kUnsupported, "update_source.type = kafka_message_queue: no Kafka client in this build; the consumer "
                                     "loop is there behind a transport interface (csrc/ps/update_source.h) — available: null, file_tail");
  if (!(cfg.volatile_db.initial_cache_rate >= 0.0) || cfg.volatile_db.initial_cache_rate > 1.0)
    return Error(Code::kInvalidArg, "volatile_db.initial_cache_rate = ", cfg.volatile_db.initial_cache_rate, " is outside [0, 1]");
  if (cfg.volatile_db.overflow_margin == 0) return Error(Code::kInvalidArg, "volatile_db.overflow_margin must be > 0");
  HPS_RETURN_IF_ERROR(ps->Build(load_tables));
  if (cfg.update_source.type == UpdateSourceType::FileTail) {
    // update_filters (docs/hierarchical_parameter_server.md:509-512, 570-573; parsed at backend.cpp:207-216, 250-259): regular
    // expressions over the update's tag "hps_<model>.<table name>" that decide which updates a database layer takes.  Each layer
    // subscribes with its OWN list, as in the reference: an update only the persistent list selects goes to the row store and
    // leaves the volatile tier and the GPU caches as they are (round 5 merged the two lists into one).
    ps->persistent_subscribes_ = cfg.persistent_db.type != DatabaseType::Disabled;
    std::vector<std::tuple<const char*, const std::vector<std::string>*, std::vector<std::regex>*>> lists{
        {"volatile_db", &cfg.volatile_db.update_filters, &ps->update_filters_}};
    if (ps->persistent_subscribes_) lists.push_back({"persistent_db", &cfg.persistent_db.update_filters, &ps->persistent_update_filters_});
    for (const auto& l : lists) {
      for (const std::string& f : *std::get<1>(l)) {
        try {
          std::get<2>(l)->emplace_back(f, std::regex::ECMAScript | std::regex::optimize);
        } catch (const std::regex_error& e) {
          return Error(Code::kInvalidArg, std::get<0>(l), ".update_filters: '", f, "' is not a regular expression (", e.what(), ")");
        }
      }
    }
    std::unique_ptr<UpdateTransport> tr;
    HPS_RETURN_IF_ERROR(MakeFileTailTransport(cfg.update_source.brokers, cfg.update_source.receive_buffer_size, &tr));
    HierParameterServer* raw = ps.get();   // the consumer is a member: it never outlives the server
    ps->updates_.reset(new UpdateConsumer(
        cfg.update_source, std::move(tr),
        [raw](const std::string& m, uint32_t t, uint32_t d, const int64_t* k, const float* r, size_t n) { return raw->ApplyUpdate(m, t, d, k, r, n); },
        [raw](const std::set<std::string>& models) { raw->OnUpdatesCommitted(models); }));
  }
  *out = std::move(ps);
  return Status::Ok();
}

Status HierParameterServer::ApplyUpdate(const std::string& model, uint32_t table, uint32_t dim, const int64_t* keys, const float* rows, size_t n) {
  auto tabs = tables_of(model);
  if (table >= tabs.size()) return Error(Code::kNotFound, "update for model '", model, "' table ", table, ": no such table in this server");
  if (tabs[table]->dim() != dim)
    return Error(Code::kInvalidArg, "update for model '", model, "' table ", table, ": rows are ", dim, " wide, the table is ", tabs[table]->dim());
  unsigned layers = 0;
  {
    // the layers take an update only if its tag matches one of their update_filters (default "^hps_.+$": every model)
    std::string tag = "hps_" + model + ".";
    InferenceParams ip;
    if (model_params(model, &ip) && table < ip.embedding_table_names.size()) tag += ip.embedding_table_names[table];
    else tag += "sparse_embedding" + std::to_string(table + 1);
    for (const std::regex& f : update_filters_) if (std::regex_search(tag, f)) { layers |= HostTable::kLayerVolatile; break; }
    if (persistent_subscribes_)
      for (const std::regex& f : persistent_update_filters_) if (std::regex_search(tag, f)) { layers |= HostTable::kLayerPersistent; break; }
    // not subscribed to: skipped silently, like a message on a topic nobody listens to (the consumer counts the message as dealt
    // with; hps_server_update_source_filtered says how many there were)
    if (!layers) { filtered_updates_.fetch_add(1, std::memory_order_relaxed); return Status(Code::kOk, kUpdateFiltered); }
  }
  HPS_RETURN_IF_ERROR(upsert_table(model, table, keys, rows, n, layers));
  // the GPU caches follow the volatile layer: an update only the persistent database subscribed to is not pushed into them
  if (!(layers & HostTable::kLayerVolatile)) return Status::Ok();
  std::lock_guard<std::mutex> lk(upd_mu_);
  auto& per_table = updated_keys_[model];
  if (per_table.size() < tabs.size()) per_table.resize(tabs.size());
  per_table[table].insert(per_table[table].end(), keys, keys + n);
  return Status::Ok();
}

void HierParameterServer::OnUpdatesCommitted(const std::set<std::string>& models) {
  for (const std::string& model : models) {
    std::vector<std::vector<int64_t>> keys;
    {
      std::lock_guard<std::mutex> lk(upd_mu_);
      auto it = updated_keys_.find(model);
      if (it == updated_keys_.end()) continue;
      keys = std::move(it->second);
      updated_keys_.erase(it);
    }
    std::vector<std::shared_ptr<EmbeddingCache>> caches;
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (auto& kv : caches_) if (std::get<0>(kv.first) == model) caches.push_back(kv.second);
    }
    for (auto& c : caches) {
      if (keys.size() != c->num_tables()) continue;
      // only what is resident: an updated row that nobody has asked for does not displace a row somebody has
      std::vector<std::vector<int64_t>> resident(keys.size());
      bool any = false;
      for (size_t t = 0; t < keys.size(); ++t) {
        std::vector<int64_t>& k = keys[t];
        std::sort(k.begin(), k.end());
        k.erase(std::unique(k.begin(), k.end()), k.end());
        if (k.empty()) continue;
        std::vector<int32_t> slots(k.size());
        if (!c->Query((uint32_t)t, k.data(), k.size(), slots.data()).ok()) continue;
        for (size_t i = 0; i < k.size(); ++i) if (slots[i] >= 0) resident[t].push_back(k[i]);
        any |= !resident[t].empty();
      }
      if (any) (void)c->InsertKeys(this, resident);
    }
  }
}

// (a monitoring thread polling the statistics, or a drain in progress, may run next to a stop: each takes its own reference to
//  the consumer under the lock; the consumer's thread is joined by whoever drops the last one)
bool HierParameterServer::update_source_stats(UpdateSourceStats* out) const {
  std::shared_ptr<UpdateConsumer> u;
  { std::lock_guard<std::mutex> lk(updates_mu_); u = updates_; }
  if (!u) return false;
  if (out) *out = u->stats();
  return true;
}

Status HierParameterServer::drain_update_source(size_t timeout_ms) {
  std::shared_ptr<UpdateConsumer> u;
  { std::lock_guard<std::mutex> lk(updates_mu_); u = updates_; }
  if (!u) return Error(Code::kUnavailable, "no update source is configured (ps.json update_source.type)");
  return u->Drain(timeout_ms);
}

Status HierParameterServer::stop_update_source() {
  std::shared_ptr<UpdateConsumer> u;
  { std::lock_guard<std::mutex> lk(updates_mu_); u = std::move(updates_); updates_.reset(); }
  if (!u) return Error(Code::kUnavailable, "no update source is configured (ps.json update_source.type)");
  u->Stop();   // joins the thread: what was applied is delivered to the caches and committed, nothing more is applied
  return Status::Ok();
}

// Copies a table's two files into the persistent store directory (created if needed).
static Status MaterializeStore(const std::string& src, const std::string& dst) {
  std::error_code ec;
  std::filesystem::create_directories(dst, ec);
  if (ec) return Error(Code::kInternal, "persistent_db: cannot create '", dst, "': ", ec.message());
  for (const char* f : {"key", "emb_vector"}) {
    std::filesystem::copy_file(src + "/" + f, dst + "/" + f, std::filesystem::copy_options::overwrite_existing, ec);
    if (ec) return Error(Code::kNotFound, "persistent_db: cannot copy '", src, "/", f, "' to '", dst, "': ", ec.message());
    std::filesystem::permissions(dst + "/" + f, std::filesystem::perms::owner_read | std::filesystem::perms::owner_write,
                                 std::filesystem::perm_options::add, ec);
  }
  return Status::Ok();
}

// "synthetic://<rows>[?seed=<n>][&key0=<k>]"; false for anything else (then the string is a directory)
static bool ParseSyntheticSource(const std::string& src, uint64_t* rows, uint64_t* seed, int64_t* key0) {
  static const char kScheme[] = "synthetic://";
  if (src.compare(0, sizeof(kScheme) - 1, kScheme) != 0) return false;
  const char* c = src.c_str() + sizeof(kScheme) - 1;
  char* end = nullptr;
  const unsigned long long r = std::strtoull(c, &end, 10);
  if (end == c) return false;   // e.g. "synthetic://name" of a model whose tables are injected later
  *rows = r;
  *seed = 20260929ull;
  *key0 = 0;
  while (*end == '?' || *end == '&') {
    ++end;
    if (std::strncmp(end, "seed=", 5) == 0) *seed = std::strtoull(end + 5, &end, 10);
    else if (std::strncmp(end, "key0=", 5) == 0) *key0 = std::strtoll(end + 5, &end, 10);
    else return false;
  }
  return *end == 0;
}

Status HierParameterServer::EnsureTables(const InferenceParams& p, bool load) {
  std::vector<std::shared_ptr<HostTable>> tabs;
  {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = tables_.find(p.model_name);
    if (it != tables_.end()) tabs = it->second;
  }
  const size_t T = p.num_tables();
  // ps_direct_access: the tables live in device-mapped pinned memory so the GPU can read rows in place
  bool pinned = false;
  if (p.ps_direct_access && p.use_gpu_embedding_cache) {
    int ndev = 0;
    pinned = hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
    if (!pinned) return Error(Code::kUnavailable, "model '", p.model_name, "': ps_direct_access needs a GPU");
    (void)hipSetDevice(p.deployed_devices.empty() ? p.device_id : p.deployed_devices[0]);
  }
  // Host tier smaller than the table: a bounded volatile tier and/or a persistent row store behind it (host_table.h)
  HostTierOptions tier;
  tier.persistent = p.persistent_db.type != DatabaseType::Disabled;
  tier.store_writable = tier.persistent && !p.persistent_db.read_only;
  tier.vdb = p.volatile_db;
  tier.tiered = p.volatile_db.overflow_margin != SIZE_MAX || p.volatile_db.initial_cache_rate < 1.0;
  if (tier.tiered && pinned)
    return Error(Code::kUnsupported, "model '", p.model_name, "': ps_direct_access reads the host tier from the GPU and needs "
                 "the whole table in RAM (volatile_db.overflow_margin unlimited, initial_cache_rate 1.0)");
  const bool fresh = tabs.size() != T || (T && tabs[0]->pinned() != pinned);
  if (fresh) {
    tabs.clear();
    for (size_t t = 0; t < T; ++t)
      tabs.emplace_back(std::make_shared<HostTable>(p.embedding_table_names[t],
                                                    (uint32_t)p.embedding_vecsize_per_table[t],
                                                    p.volatile_db.num_partitions, pinned));
  }
  if (load) {
    HPS_RETURN_IF_ERROR(MutateTables(p.model_name, [&]() -> Status {
      for (size_t t = 0; t < T; ++t) {
        std::string dir = p.sparse_model_files[t];
        if (tier.persistent) {
          // the persistent database holds a full copy of every table (docs/hierarchical_parameter_server.md:520-569):
          // <path>/<model>/<table>/{key,emb_vector}, (re)written from the model files unless read_only
          const std::string base = p.persistent_db.path.empty() ? std::string("/tmp/rocksdb") : p.persistent_db.path;
          const std::string store = base + "/" + p.model_name + "/" + p.embedding_table_names[t];
          if (tier.store_writable) HPS_RETURN_IF_ERROR(MaterializeStore(dir, store));
          dir = store;
        }
        tabs[t]->SetTierOptions(tier);
        // "synthetic://<rows>[?seed=<n>][&key0=<k>]" instead of a directory: the table of SURVEY.md 8(d)'s recipe (keys
        // key0..key0+rows-1, hashed fp32 rows) generated straight into the host tier.  A deployment has files; this is how
        // a benchmark loads the 133-GB Criteo-shaped model THROUGH the plugin boundary without writing 133 GB first.
        uint64_t syn_rows = 0, syn_seed = 0;
        int64_t syn_key0 = 0;
        if (ParseSyntheticSource(p.sparse_model_files[t], &syn_rows, &syn_seed, &syn_key0)) {
          if (tier.persistent || tier.tiered)
            return Error(Code::kUnsupported, "model '", p.model_name, "': a synthetic:// table needs the plain in-memory host tier");
          HPS_RETURN_IF_ERROR(tabs[t]->LoadSynthetic(syn_seed, (uint32_t)t, syn_key0, (size_t)syn_rows, pool_));
          continue;
        }
        HPS_RETURN_IF_ERROR(tabs[t]->LoadFromDir(dir, pool_));
      }
      return Status::Ok();
    }));
  }
  std::lock_guard<std::mutex> lk(mu_);
  tables_[p.model_name] = tabs;
  return Status::Ok();
}

// Runs a table mutation.  Caches in ps_direct_access mode read the host tier from the device, so their
// lookups are fenced out (exclusive side of the cache's direct mutex) while rows move, and their device
// index of the tier is rebuilt before lookups resume.
Status HierParameterServer::MutateTables(const std::string& model, const std::function<Status()>& fn) {
  std::vector<std::shared_ptr<EmbeddingCache>> direct;
  {
    std::lock_guard<std::mutex> lk(mu_);
    for (auto& kv : caches_)
      if (std::get<0>(kv.first) == model && kv.second->direct()) direct.push_back(kv.second);
  }
  std::vector<std::unique_lock<std::shared_mutex>> locks;
  for (auto& c : direct) {
    c->direct_writers().fetch_add(1, std::memory_order_acq_rel);
    c->WaitAsync();
    locks.emplace_back(c->direct_mutex());
  }
  // (a lookup that was still in flight may have queued one more background job: it takes the shared side of the
  //  mutex itself, so it simply runs after the reload, against the new tables)
  struct Release {
    std::vector<std::shared_ptr<EmbeddingCache>>& v;
    ~Release() { for (auto& c : v) c->direct_writers().fetch_sub(1, std::memory_order_acq_rel); }
  } release{direct};
  Status st = fn();
  if (st.ok()) {
    auto tabs = tables_of(model);
    for (auto& c : direct) {
      if (tabs.size() != c->num_tables()) continue;
      Status s2 = c->SyncDirectIndex(tabs);
      if (!s2.ok()) { st = s2; break; }
    }
  }
  return st;
}

Status HierParameterServer::Build(bool load_tables) {
  for (const auto& name : cfg_.model_order) {
    const InferenceParams& p = cfg_.models.at(name);
    HPS_RETURN_IF_ERROR(EnsureTables(p, load_tables));
    if (load_tables && p.use_gpu_embedding_cache) HPS_RETURN_IF_ERROR(create_embedding_cache_per_model(p));
  }
  return Status::Ok();
}

Status HierParameterServer::add_model(const InferenceParams& p) {
  std::lock_guard<std::mutex> lk(mu_);
  if (!cfg_.models.count(p.model_name)) cfg_.model_order.push_back(p.model_name);
  cfg_.models[p.model_name] = p;
  return Status::Ok();
}

bool HierParameterServer::model_params(const std::string& model, InferenceParams* out) {
  std::lock_guard<std::mutex> lk(mu_);
  auto it = cfg_.models.find(model);
  if (it == cfg_.models.end()) return false;
  if (out) *out = it->second;
  return true;
}

Status HierParameterServer::parse_config(const std::string& path) {
  ParameterServerConfig fresh;
  HPS_RETURN_IF_ERROR(ParseParameterServerFile(path, &fresh));
  for (const auto& name : fresh.model_order) HPS_RETURN_IF_ERROR(add_model(fresh.models.at(name)));
  return Status::Ok();
}

std::vector<std::shared_ptr<HostTable>> HierParameterServer::tables_of(const std::string& model) {
  std::lock_guard<std::mutex> lk(mu_);
  auto it = tables_.find(model);
  return it == tables_.end() ? std::vector<std::shared_ptr<HostTable>>() : it->second;
}

std::shared_ptr<EmbeddingCache> HierParameterServer::get_embedding_cache(const std::string& model, int device) {
  std::lock_guard<std::mutex> lk(mu_);
  auto it = caches_.find({model, device, -1});
  if (it != caches_.end()) return it->second;
  // table-sharded model: the (first) shard that lives on that device
  for (auto& kv : caches_)
    if (std::get<0>(kv.first) == model && std::get<1>(kv.first) == device) return kv.second;
  return nullptr;
}

std::shared_ptr<EmbeddingCache> HierParameterServer::get_shard_cache(const std::string& model, uint32_t shard) {
  std::lock_guard<std::mutex> lk(mu_);
  for (auto& kv : caches_)
    if (std::get<0>(kv.first) == model && std::get<2>(kv.first) == (int)shard) return kv.second;
  return nullptr;
}

Status HierParameterServer::update_database_per_model(const InferenceParams& p) {
  HPS_RETURN_IF_ERROR(add_model(p));
  return EnsureTables(p, true);
}

Status HierParameterServer::create_embedding_cache_per_model(const InferenceParams& p) {
  if (!p.use_gpu_embedding_cache) return Status::Ok();
  auto tabs = tables_of(p.model_name);
  if (tabs.size() != p.num_tables())
    return Error(Code::kNotFound, "model '", p.model_name, "': tables are not loaded; call update_database_per_model first");
  if (p.table_sharding) {
    // entry s of deployed_device_list is shard s (a device may hold several: logical shards)
    const uint32_t P = (uint32_t)p.deployed_devices.size();
    for (uint32_t s = 0; s < P; ++s) {
      const int dev = p.deployed_devices[s];
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (caches_.count({p.model_name, dev, (int)s})) continue;
      }
      std::shared_ptr<EmbeddingCache> c(new EmbeddingCache());
      HPS_RETURN_IF_ERROR(c->Init(p.model_name, p, tabs, dev, (int)s, P));
      std::lock_guard<std::mutex> lk(mu_);
      caches_[{p.model_name, dev, (int)s}] = std::move(c);
    }
    return Status::Ok();
  }
  for (int dev : p.deployed_devices) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (caches_.count({p.model_name, dev, -1})) continue;
    }
    std::shared_ptr<EmbeddingCache> c(new EmbeddingCache());
    HPS_RETURN_IF_ERROR(c->Init(p.model_name, p, tabs, dev));
    std::lock_guard<std::mutex> lk(mu_);
    caches_[{p.model_name, dev, -1}] = std::move(c);
  }
  return Status::Ok();
}

Status HierParameterServer::destory_embedding_cache_per_model(const std::string& model) {
  std::vector<std::shared_ptr<EmbeddingCache>> victims;
  {
    std::lock_guard<std::mutex> lk(mu_);
    for (auto it = caches_.begin(); it != caches_.end();) {
      if (std::get<0>(it->first) == model) { victims.push_back(it->second); it = caches_.erase(it); }
      else ++it;
    }
  }
  for (auto& c : victims) c->WaitAsync();
  return Status::Ok();  // device memory goes when the last session drops its reference
}

Status HierParameterServer::refresh_embedding_cache(const std::string& model, int device, bool full, RefreshStats* stats) {
  std::vector<std::shared_ptr<EmbeddingCache>> on_device;   // one replica, or every shard of a table-sharded model on that device
  {
    std::lock_guard<std::mutex> lk(mu_);
    for (auto& kv : caches_)
      if (std::get<0>(kv.first) == model && std::get<1>(kv.first) == device) on_device.push_back(kv.second);
  }
  if (on_device.empty()) return Error(Code::kNotFound, "no embedding cache for model '", model, "' on device ", device);
  const auto t0 = std::chrono::steady_clock::now();
  if (stats) *stats = RefreshStats();
  for (auto& c : on_device) HPS_RETURN_IF_ERROR(RefreshOne(model, c, full, stats));
  if (stats) stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return Status::Ok();
}

Status HierParameterServer::RefreshOne(const std::string& model, const std::shared_ptr<EmbeddingCache>& cache, bool full, RefreshStats* stats) {
  InferenceParams p;
  {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = cfg_.models.find(model);
    if (it == cfg_.models.end()) return Error(Code::kNotFound, "model '", model, "' is not configured");
    p = it->second;
  }
  auto tabs = tables_of(model);
  const size_t T = cache->num_tables();
  if (tabs.size() != T) return Error(Code::kNotFound, "model '", model, "': tables are not loaded");
  std::lock_guard<std::mutex> rlk(cache->refresh_mu_);   // one refresh of a cache at a time
  if (!p.refresh_changed_only) full = true;
  if (cache->seen_epoch_.size() != T) full = true;         // (a cache that never recorded its marks)
  // ---- which keys to take again: per table either every resident key (reloaded table, log overrun, full pass) or the
  //      resident ones among the keys the table's change log names since the refresh BEFORE the last one ----
  std::vector<std::vector<int64_t>> todo(T);
  std::vector<uint64_t> mark_epoch(T, 0), mark_seq(T, 0);
  std::vector<uint8_t> whole_table(T, 0);
  RefreshStats local;
  for (size_t t = 0; t < T; ++t) {
    ++local.tables;
    std::vector<int64_t> changed;
    bool whole = full;
    if (!whole) whole = !tabs[t]->ChangesSince(cache->seen_epoch_[t], cache->seen_prev_[t], &changed, &mark_epoch[t], &mark_seq[t]);
    else tabs[t]->ChangeMark(&mark_epoch[t], &mark_seq[t]);     // (before the rows are read)
    whole_table[t] = whole ? 1 : 0;
    if (whole) {
      ++local.tables_full;
      HPS_RETURN_IF_ERROR(cache->DumpKeys((uint32_t)t, &todo[t]));
      local.keys_dumped += todo[t].size();
      continue;
    }
    if (changed.empty()) { ++local.tables_unchanged; continue; }
    local.keys_changed += changed.size();
    std::sort(changed.begin(), changed.end());
    changed.erase(std::unique(changed.begin(), changed.end()), changed.end());
    // only what is resident: an updated row nobody has asked for does not displace a row somebody has
    std::vector<int32_t> slots(changed.size());
    HPS_RETURN_IF_ERROR(cache->Query((uint32_t)t, changed.data(), changed.size(), slots.data()));
    for (size_t i = 0; i < changed.size(); ++i) if (slots[i] >= 0) todo[t].push_back(changed[i]);
  }
  // Re-read the vectors from the parameter server, a fraction of the CACHE per iteration
  // (cache_refresh_percentage_per_iteration, docs/hierarchical_parameter_server.md:234-238), in paced pieces.
  double frac = p.cache_refresh_percentage_per_iteration;
  if (!(frac > 0.0) || frac > 1.0) frac = 1.0;
  EmbeddingCache::InsertPacing pacing;   // (link share 1.0: unpaced — 262,144-row pieces, the whole serving pool, as before round 6)
  if (p.refresh_link_share < 1.0) {
    pacing.piece_keys = 32768;
    pacing.link_share = p.refresh_link_share;
    pacing.max_threads = 4;
  }
  std::vector<size_t> done(T, 0);
  for (;;) {
    std::vector<std::vector<int64_t>> part(T);
    bool any = false;
    for (size_t t = 0; t < T; ++t) {
      const size_t step = std::max<size_t>(1, (size_t)((double)cache->get_cache_config().capacity_rows_[t] * frac + 0.999999));
      const size_t n = std::min(step, todo[t].size() - done[t]);
      part[t].assign(todo[t].begin() + done[t], todo[t].begin() + done[t] + n);
      done[t] += n;
      local.rows_refreshed += n;
      any |= n > 0;
    }
    if (!any) break;
    HPS_RETURN_IF_ERROR(cache->InsertKeys(this, part, &pacing, &local.row_bytes));
  }
  // the marks move only after the rows are in: a refresh that failed half-way is repeated from where the last good one stood
  const bool had_marks = cache->seen_last_.size() == T && cache->seen_prev_.size() == T;
  std::vector<uint64_t> prev(T);
  for (size_t t = 0; t < T; ++t)   // (a whole-table pass has seen everything up to its mark)
    prev[t] = (whole_table[t] || !had_marks) ? mark_seq[t] : cache->seen_last_[t];
  cache->seen_epoch_ = mark_epoch;
  cache->seen_prev_ = std::move(prev);
  cache->seen_last_ = mark_seq;
  if (stats) {
    stats->tables += local.tables; stats->tables_unchanged += local.tables_unchanged; stats->tables_full += local.tables_full;
    stats->keys_dumped += local.keys_dumped; stats->keys_changed += local.keys_changed;
    stats->rows_refreshed += local.rows_refreshed; stats->row_bytes += local.row_bytes;
  }
  return Status::Ok();
}

Status HierParameterServer::create_lookup_session(const std::string& model, std::shared_ptr<EmbeddingCache> cache,
                                                  std::unique_ptr<LookupSession>* out) {
  InferenceParams p;
  {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = cfg_.models.find(model);
    if (it == cfg_.models.end()) return Error(Code::kNotFound, "model '", model, "' is not in the parameter server configuration");
    p = it->second;
  }
  std::unique_ptr<LookupSession> s(new LookupSession());
  HPS_RETURN_IF_ERROR(s->Init(this, p, std::move(cache)));
  *out = std::move(s);
  return Status::Ok();
}

Status HierParameterServer::create_lookup_session_sized(const std::string& model, std::shared_ptr<EmbeddingCache> cache, size_t max_keys,
                                                        std::unique_ptr<LookupSession>* out) {
  InferenceParams p;
  if (!model_params(model, &p)) return Error(Code::kNotFound, "model '", model, "' is not in the parameter server configuration");
  if (max_keys == 0) return Error(Code::kInvalidArg, "create_lookup_session_sized: max_keys is 0");
  std::unique_ptr<LookupSession> s(new LookupSession());
  HPS_RETURN_IF_ERROR(s->Init(this, p, std::move(cache), max_keys));
  *out = std::move(s);
  return Status::Ok();
}

Status HierParameterServer::load_table_from_arrays(const std::string& model, size_t table, const int64_t* keys,
                                                   const float* rows, size_t R, bool borrow) {
  auto tabs = tables_of(model);
  if (table >= tabs.size()) return Error(Code::kNotFound, "model '", model, "' has no table ", table);
  return MutateTables(model, [&]() { return tabs[table]->LoadFromArrays(keys, rows, R, borrow, pool_); });
}

Status HierParameterServer::upsert_table(const std::string& model, size_t table, const int64_t* keys, const float* rows,
                                         size_t n, unsigned layers) {
  auto tabs = tables_of(model);
  if (table >= tabs.size()) return Error(Code::kNotFound, "model '", model, "' has no table ", table);
  return MutateTables(model, [&]() { return tabs[table]->Upsert(keys, rows, n, layers); });
}

Status HierParameterServer::load_table_synthetic(const std::string& model, size_t table, uint64_t seed, int64_t key0,
                                                 size_t R, uint32_t shard, uint32_t num_shards) {
  auto tabs = tables_of(model);
  if (table >= tabs.size()) return Error(Code::kNotFound, "model '", model, "' has no table ", table);
  return MutateTables(model, [&]() {
    return tabs[table]->LoadSynthetic(seed, (uint32_t)table, key0, R, pool_, shard, num_shards);
  });
}

Status HierParameterServer::Fetch(const HostTable& tb, const int64_t* keys, size_t n, float* out, size_t stride,
                                  float default_value, uint8_t* found, size_t* nfound) {
  // Key ranges fan out over the pool; each task keeps 16 lookups in flight (host_table.cpp).
  const size_t chunk = 512;
  const size_t ntasks = (n + chunk - 1) / chunk;
  std::atomic<size_t> total{0};
  auto body = [&](size_t ti) {
    const size_t b = ti * chunk, e = std::min(n, b + chunk);
    const size_t f = tb.Fetch(keys + b, e - b, out + b * stride, stride, default_value, found ? found + b : nullptr);
    total.fetch_add(f, std::memory_order_relaxed);
  };
  if (ntasks <= 1) { if (ntasks) body(0); }
  else pool_->ParallelFor(ntasks, body);
  if (nfound) *nfound = total.load();
  return Status::Ok();
}

Status HierParameterServer::FetchMulti(const std::vector<FetchJob>& jobs, size_t max_threads) {
  struct Task { uint32_t job; size_t begin, end; };
  // Task size: about two tasks per thread — enough to even out the last round (a 4,096-key request cut into 256-key
  // tasks leaves two of 14 threads with a second task) without paying a contended claim per 100 keys (each claim is a
  // compare-and-swap on a line shared by every thread: 42 empty tasks cost 12 us on the box, 14 cost 6) — never below
  // 64 keys (the fetch pipeline needs a few blocks of 8 to fill), never above 256 (large requests balance anyway).
  size_t total = 0;
  for (const auto& j : jobs) total += j.n;
  const size_t threads = ThreadPool::Serving().size() + 1;
  constexpr size_t tasks_per_thread = 2;
  size_t chunk = total / (threads * tasks_per_thread);
  chunk = (chunk + 7) & ~(size_t)7;
  chunk = std::min<size_t>(256, std::max<size_t>(64, chunk));
  std::vector<Task> tasks;
  for (size_t j = 0; j < jobs.size(); ++j)
    for (size_t b = 0; b < jobs[j].n; b += chunk) tasks.push_back({(uint32_t)j, b, std::min(jobs[j].n, b + chunk)});
  auto body = [&](size_t ti) {
    const Task& k = tasks[ti];
    const FetchJob& J = jobs[k.job];
    int64_t wide[256];   // tasks are at most 256 keys
    const int64_t* keys = J.keys ? J.keys + k.begin : wide;
    if (!J.keys)
      for (size_t i = k.begin; i < k.end; ++i) wide[i - k.begin] = J.key_base + (int64_t)(uint64_t)J.keys32[i];
    J.table->Fetch(keys, k.end - k.begin, J.out + k.begin * J.stride, J.stride, J.default_value,
                   J.found ? J.found + k.begin : nullptr);
  };
  if (tasks.size() <= 1) { if (!tasks.empty()) body(0); }
  else ThreadPool::Serving().ParallelFor(tasks.size(), body, max_threads);
  return Status::Ok();
}

void HierParameterServer::RunDirectInsert(std::shared_ptr<EmbeddingCache> cache) {
  {
    std::lock_guard<std::mutex> lk(cache->pend_mu_);
    ++cache->pending_async_;
  }
  auto self = shared_from_this();
  pool_->Submit([self, cache]() {
    (void)cache->FinishDirectInsert();
    std::lock_guard<std::mutex> lk(cache->pend_mu_);
    --cache->pending_async_;
    cache->pend_cv_.notify_all();
  });
}

void HierParameterServer::SubmitAsyncInsert(std::shared_ptr<EmbeddingCache> cache,
                                            std::vector<std::vector<int64_t>> keys_per_table) {
  int limit = 2;
  {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = cfg_.models.find(cache->model_name());
    if (it != cfg_.models.end()) limit = std::max(1, it->second.number_of_worker_buffers_in_pool);
  }
  {
    std::lock_guard<std::mutex> lk(cache->pend_mu_);
    if (cache->pending_async_ >= limit) return;  // inserter is saturated: this batch's misses are not cached
    ++cache->pending_async_;
  }
  auto self = shared_from_this();
  pool_->Submit([self, cache, keys = std::move(keys_per_table)]() {
    (void)cache->InsertKeys(self.get(), keys);
    std::lock_guard<std::mutex> lk(cache->pend_mu_);
    --cache->pending_async_;
    cache->pend_cv_.notify_all();
  });
}

}  // namespace hps
