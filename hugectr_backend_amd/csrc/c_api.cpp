// extern "C" surface of libhps_amd.so — see include/hps_amd.h for the contract and the reference
// call site each function replaces.
#include "../../include/hps_amd.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <string>

#include "cache/copy_engines.h"
#include "cache/engine.h"
#include "cache/shard_kernels.h"
#include "cache/shard_entry.h"
#include "cache/multi_gpu_probe.h"
#include "cache/shard_session.h"
#include "dense/dense.h"

using namespace hps;

struct hps_server { std::shared_ptr<HierParameterServer> ps; std::vector<std::string> names; };
// The handle the shell gets from get_embedding_cache(model, device): like the reference's EmbeddingCacheBase it knows its
// parameter server and its model, so LookupSessionBase::create(params, cache) needs nothing else.  `cache` is null for a
// model that runs without GPU cache (the reference hands out a cache object with use_gpu_embedding_cache = false there).
struct hps_cache { std::shared_ptr<HierParameterServer> ps; std::string model; int device; std::shared_ptr<EmbeddingCache> cache; };
struct hps_session { std::shared_ptr<HierParameterServer> ps; std::shared_ptr<LookupSession> s; };   // shared: a sharded session built on it keeps it alive
struct hps_dense { std::unique_ptr<DenseInteraction> d; };
struct hps_shard_group { std::shared_ptr<LocalShardGroup> g; };
struct hps_shard_entry { std::shared_ptr<HierParameterServer> ps; std::unique_ptr<ShardedEntrySession> s; };   // (s goes first)
struct hps_shard_session { std::shared_ptr<HierParameterServer> ps; std::unique_ptr<ShardedSession> s; };   // (s goes first)

namespace {
// Runs when the library is loaded.  HIP maps a process's streams onto 4 hardware queues unless told otherwise; lookup
// sessions whose streams share a queue execute one after the other.  Only a hint: it has no effect if the hosting
// process initialised HIP earlier or set the variable itself (Triton deployments: export GPU_MAX_HW_QUEUES=8).
const int g_hw_queue_hint = (setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0), 0);
thread_local std::string g_err;

int Fail(const Status& st) {
  g_err = st.message();
  const int c = (int)st.code();
  return c < 0 ? HPS_ERR_UNKNOWN : c + 1;
}
int FailMsg(int code, const std::string& m) { g_err = m; return code; }

template <typename F>
int Guard(F&& f) {
  try {
    const Status st = f();
    if (st.ok()) return HPS_OK;
    return Fail(st);
  } catch (const std::exception& e) {
    return FailMsg(HPS_ERR_INTERNAL, std::string("exception: ") + e.what());
  } catch (...) {
    return FailMsg(HPS_ERR_INTERNAL, "unknown exception");
  }
}

Status FindModel(hps_server_t* sv, const char* model, const InferenceParams** out) {
  if (!sv || !model) return Error(Code::kInvalidArg, "null argument");
  const auto& m = sv->ps->get_hps_model_configuration_map();
  auto it = m.find(model);
  if (it == m.end()) return Error(Code::kNotFound, "model '", model, "' is not in the parameter server configuration");
  *out = &it->second;
  return Status::Ok();
}
}  // namespace

extern "C" {

const char* hps_last_error(void) { return g_err.c_str(); }

int hps_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

int hps_server_create(const char* path, hps_server_t** out) {
  return Guard([&]() -> Status {
    if (!path || !out) return Error(Code::kInvalidArg, "null argument");
    std::shared_ptr<HierParameterServer> ps;
    HPS_RETURN_IF_ERROR(HierParameterServer::create(path, &ps));
    *out = new hps_server{std::move(ps), {}};
    return Status::Ok();
  });
}

int hps_server_create_from_text(const char* text, int load_tables, hps_server_t** out) {
  return Guard([&]() -> Status {
    if (!text || !out) return Error(Code::kInvalidArg, "null argument");
    ParameterServerConfig cfg;
    HPS_RETURN_IF_ERROR(ParseParameterServerText(text, &cfg));
    std::shared_ptr<HierParameterServer> ps;
    HPS_RETURN_IF_ERROR(HierParameterServer::create_from_config(cfg, load_tables != 0, &ps));
    *out = new hps_server{std::move(ps), {}};
    return Status::Ok();
  });
}

void hps_server_destroy(hps_server_t* sv) { delete sv; }

int hps_server_model_count(hps_server_t* sv) {
  if (!sv) return 0;
  return (int)sv->ps->get_hps_model_configuration_map().size();
}

const char* hps_server_model_name(hps_server_t* sv, int index) {
  if (!sv) return nullptr;
  sv->names.clear();
  for (const auto& kv : sv->ps->get_hps_model_configuration_map()) sv->names.push_back(kv.first);
  if (index < 0 || (size_t)index >= sv->names.size()) return nullptr;
  return sv->names[(size_t)index].c_str();
}

int hps_server_model_info(hps_server_t* sv, const char* model, hps_model_info_t* out) {
  return Guard([&]() -> Status {
    const InferenceParams* p = nullptr;
    HPS_RETURN_IF_ERROR(FindModel(sv, model, &p));
    if (!out) return Error(Code::kInvalidArg, "null argument");
    memset(out, 0, sizeof *out);
    out->max_batch_size = p->max_batchsize;
    out->num_tables = (uint32_t)p->num_tables();
    out->use_gpu_embedding_cache = p->use_gpu_embedding_cache;
    out->hit_rate_threshold = p->hit_rate_threshold;
    out->cache_size_percentage = p->cache_size_percentage;
    out->i64_input_key = p->i64_input_key;
    out->number_of_worker_buffers_in_pool = p->number_of_worker_buffers_in_pool;
    out->number_of_refresh_buffers_in_pool = p->number_of_refresh_buffers_in_pool;
    out->cache_refresh_percentage_per_iteration = p->cache_refresh_percentage_per_iteration;
    out->device_id = p->device_id;
    out->num_deployed_devices = (uint32_t)p->deployed_devices.size();
    out->refresh_delay = p->refresh_delay;
    out->refresh_interval = p->refresh_interval;
    for (size_t c : p->maxnum_catfeature_query_per_table_per_sample) out->cat_num += c;
    for (size_t d : p->embedding_vecsize_per_table) out->embedding_size += d;
    return Status::Ok();
  });
}

int hps_server_table_info(hps_server_t* sv, const char* model, uint32_t table, hps_table_info_t* out) {
  return Guard([&]() -> Status {
    const InferenceParams* p = nullptr;
    HPS_RETURN_IF_ERROR(FindModel(sv, model, &p));
    if (!out || table >= p->num_tables()) return Error(Code::kInvalidArg, "table index out of range");
    out->embedding_vecsize = (uint32_t)p->embedding_vecsize_per_table[table];
    out->maxnum_catfeature = p->maxnum_catfeature_query_per_table_per_sample[table];
    out->default_value = p->default_value_for_each_table[table];
    auto tabs = sv->ps->tables_of(model);
    out->rows_loaded = table < tabs.size() ? tabs[table]->size() : 0;
    return Status::Ok();
  });
}

int hps_server_deployed_device(hps_server_t* sv, const char* model, uint32_t index, int32_t* device) {
  return Guard([&]() -> Status {
    const InferenceParams* p = nullptr;
    HPS_RETURN_IF_ERROR(FindModel(sv, model, &p));
    if (!device || index >= p->deployed_devices.size()) return Error(Code::kInvalidArg, "device index out of range");
    *device = p->deployed_devices[index];
    return Status::Ok();
  });
}

int hps_server_parse_config(hps_server_t* sv, const char* path) {
  return Guard([&]() -> Status {
    if (!sv || !path) return Error(Code::kInvalidArg, "null argument");
    return sv->ps->parse_config(path);
  });
}

int hps_server_update_database_per_model(hps_server_t* sv, const char* model) {
  return Guard([&]() -> Status {
    const InferenceParams* p = nullptr;
    HPS_RETURN_IF_ERROR(FindModel(sv, model, &p));
    const InferenceParams copy = *p;
    return sv->ps->update_database_per_model(copy);
  });
}

int hps_server_create_embedding_cache_per_model(hps_server_t* sv, const char* model) {
  return Guard([&]() -> Status {
    const InferenceParams* p = nullptr;
    HPS_RETURN_IF_ERROR(FindModel(sv, model, &p));
    const InferenceParams copy = *p;
    return sv->ps->create_embedding_cache_per_model(copy);
  });
}

int hps_server_destroy_embedding_cache_per_model(hps_server_t* sv, const char* model) {
  return Guard([&]() -> Status {
    if (!sv || !model) return Error(Code::kInvalidArg, "null argument");
    return sv->ps->destory_embedding_cache_per_model(model);
  });
}

int hps_server_refresh_embedding_cache(hps_server_t* sv, const char* model, int32_t device) {
  return Guard([&]() -> Status {
    if (!sv || !model) return Error(Code::kInvalidArg, "null argument");
    return sv->ps->refresh_embedding_cache(model, device);
  });
}

int hps_server_refresh_embedding_cache_ex(hps_server_t* sv, const char* model, int32_t device, int32_t full, hps_refresh_stats_t* out) {
  return Guard([&]() -> Status {
    if (!sv || !model) return Error(Code::kInvalidArg, "null argument");
    HierParameterServer::RefreshStats st;
    HPS_RETURN_IF_ERROR(sv->ps->refresh_embedding_cache(model, device, full != 0, &st));
    if (out) {
      out->tables = st.tables; out->tables_unchanged = st.tables_unchanged; out->tables_full = st.tables_full;
      out->keys_dumped = st.keys_dumped; out->keys_changed = st.keys_changed;
      out->rows_refreshed = st.rows_refreshed; out->row_bytes = st.row_bytes; out->seconds = st.seconds;
    }
    return Status::Ok();
  });
}

int hps_server_get_embedding_cache(hps_server_t* sv, const char* model, int32_t device, hps_cache_t** out) {
  return Guard([&]() -> Status {
    if (!sv || !model || !out) return Error(Code::kInvalidArg, "null argument");
    *out = nullptr;
    InferenceParams p;
    if (!sv->ps->model_params(model, &p)) return Status::Ok();   // unknown model: no cache (the reference returns nullptr)
    auto c = sv->ps->get_embedding_cache(model, device);
    if (!c && p.use_gpu_embedding_cache) return Status::Ok();     // GPU-cache model without a cache on this device
    *out = new hps_cache{sv->ps, model, device, std::move(c)};
    return Status::Ok();
  });
}

int hps_server_load_table_arrays(hps_server_t* sv, const char* model, uint32_t table, const int64_t* keys,
                                 const float* rows, uint64_t R, int borrow) {
  return Guard([&]() -> Status {
    if (!sv || !model || (R && (!keys || !rows))) return Error(Code::kInvalidArg, "null argument");
    return sv->ps->load_table_from_arrays(model, table, keys, rows, R, borrow != 0);
  });
}

int hps_server_load_table_synthetic(hps_server_t* sv, const char* model, uint32_t table, uint64_t seed, int64_t key0,
                                    uint64_t R) {
  return Guard([&]() -> Status {
    if (!sv || !model) return Error(Code::kInvalidArg, "null argument");
    return sv->ps->load_table_synthetic(model, table, seed, key0, R);
  });
}

int hps_server_update_source_stats(hps_server_t* sv, uint64_t* out6) {
  return Guard([&]() -> Status {
    if (!sv || !out6) return Error(Code::kInvalidArg, "null argument");
    UpdateSourceStats st;
    if (!sv->ps->update_source_stats(&st)) return Error(Code::kUnavailable, "no update source is configured (ps.json update_source.type)");
    out6[0] = st.messages; out6[1] = st.keys; out6[2] = st.dispatches; out6[3] = st.commits; out6[4] = st.dispatch_failures;
    out6[5] = st.rejected_messages;
    return Status::Ok();
  });
}

int hps_server_update_source_filtered(hps_server_t* sv, uint64_t* out) {
  return Guard([&]() -> Status {
    if (!sv || !out) return Error(Code::kInvalidArg, "null argument");
    *out = sv->ps->filtered_update_count();
    return Status::Ok();
  });
}

int hps_server_update_source_drain(hps_server_t* sv, uint32_t timeout_ms) {
  return Guard([&]() -> Status {
    if (!sv) return Error(Code::kInvalidArg, "null argument");
    return sv->ps->drain_update_source(timeout_ms);
  });
}

int hps_server_update_source_stop(hps_server_t* sv) {
  return Guard([&]() -> Status {
    if (!sv) return Error(Code::kInvalidArg, "null argument");
    return sv->ps->stop_update_source();
  });
}

int hps_update_message_encode(const char* model, uint32_t table, uint32_t dim, const int64_t* keys, const float* rows, uint64_t n,
                              void* out, uint64_t out_capacity, uint64_t* out_bytes) {
  return Guard([&]() -> Status {
    if (!model || !out_bytes || (n && (!keys || !rows))) return Error(Code::kInvalidArg, "null argument");
    const std::string m = EncodeUpdateMessage(model, table, dim, keys, rows, (size_t)n);
    *out_bytes = m.size();
    if (out) {
      if (out_capacity < m.size()) return Error(Code::kInvalidArg, "buffer of ", out_capacity, " bytes for a message of ", m.size());
      memcpy(out, m.data(), m.size());
    }
    return Status::Ok();
  });
}

int hps_server_load_table_synthetic_shard(hps_server_t* sv, const char* model, uint32_t table, uint64_t seed,
                                          int64_t key0, uint64_t R, uint32_t shard, uint32_t num_shards) {
  return Guard([&]() -> Status {
    if (!sv || !model) return Error(Code::kInvalidArg, "null argument");
    if (num_shards == 0) return Error(Code::kInvalidArg, "num_shards must be >= 1");
    return sv->ps->load_table_synthetic(model, table, seed, key0, R, shard, num_shards);
  });
}

int hps_server_fetch(hps_server_t* sv, const char* model, uint32_t table, const int64_t* keys, uint64_t n, float* out,
                     uint8_t* found) {
  return Guard([&]() -> Status {
    const InferenceParams* p = nullptr;
    HPS_RETURN_IF_ERROR(FindModel(sv, model, &p));
    auto tabs = sv->ps->tables_of(model);
    if (table >= tabs.size()) return Error(Code::kInvalidArg, "table index out of range");
    if (n && (!keys || !out)) return Error(Code::kInvalidArg, "null argument");
    return sv->ps->Fetch(*tabs[table], keys, n, out, tabs[table]->dim(), p->default_value_for_each_table[table], found,
                         nullptr);
  });
}

int hps_server_table_data(hps_server_t* sv, const char* model, uint32_t table, const int64_t** keys, const float** rows,
                          uint64_t* num_rows) {
  return Guard([&]() -> Status {
    if (!sv || !model || !keys || !rows || !num_rows) return Error(Code::kInvalidArg, "null argument");
    auto tabs = sv->ps->tables_of(model);
    if (table >= tabs.size()) return Error(Code::kInvalidArg, "table index out of range");
    *keys = tabs[table]->keys();
    *rows = tabs[table]->size() ? tabs[table]->row_at(0) : nullptr;
    *num_rows = tabs[table]->size();
    return Status::Ok();
  });
}

int hps_server_upsert(hps_server_t* sv, const char* model, uint32_t table, const int64_t* keys, const float* rows,
                      uint64_t n) {
  return Guard([&]() -> Status {
    const InferenceParams* p = nullptr;
    HPS_RETURN_IF_ERROR(FindModel(sv, model, &p));
    if (n && (!keys || !rows)) return Error(Code::kInvalidArg, "null argument");
    return sv->ps->upsert_table(model, table, keys, rows, n);
  });
}

int hps_server_host_tier_stats(hps_server_t* sv, const char* model, uint32_t table, hps_host_tier_stats_t* out) {
  return Guard([&]() -> Status {
    const InferenceParams* p = nullptr;
    HPS_RETURN_IF_ERROR(FindModel(sv, model, &p));
    auto tabs = sv->ps->tables_of(model);
    if (table >= tabs.size() || !out) return Error(Code::kInvalidArg, "bad table / null argument");
    const HostTierStats s = tabs[table]->tier_stats();
    *out = hps_host_tier_stats_t{};
    out->tiered = tabs[table]->tiered() ? 1 : 0;
    out->persistent_rows = s.persistent_rows;
    out->entries = s.vdb.entries; out->capacity = s.vdb.capacity; out->max_partition_entries = s.vdb.max_partition_entries;
    out->lookups = s.vdb.lookups; out->hits = s.vdb.hits;
    out->persistent_hits = s.persistent_hits; out->not_found = s.not_found;
    out->inserts = s.vdb.inserts; out->evictions = s.vdb.evictions; out->overflows = s.vdb.overflows;
    return Status::Ok();
  });
}

int hps_server_host_tier_keys(hps_server_t* sv, const char* model, uint32_t table, int64_t* out, uint64_t cap, uint64_t* n) {
  return Guard([&]() -> Status {
    const InferenceParams* p = nullptr;
    HPS_RETURN_IF_ERROR(FindModel(sv, model, &p));
    auto tabs = sv->ps->tables_of(model);
    if (table >= tabs.size() || !n) return Error(Code::kInvalidArg, "bad table / null argument");
    std::vector<int64_t> keys;
    tabs[table]->DumpVolatileKeys(&keys);
    *n = keys.size();
    if (out && cap >= keys.size() && !keys.empty()) memcpy(out, keys.data(), keys.size() * sizeof(int64_t));
    return Status::Ok();
  });
}

int hps_cache_num_tables(hps_cache_t* c) {
  if (!c) return 0;
  if (c->cache) return (int)c->cache->num_tables();
  InferenceParams p;
  return c->ps->model_params(c->model, &p) ? (int)p.num_tables() : 0;
}

int hps_cache_on_device(hps_cache_t* c) { return c && c->cache ? 1 : 0; }
uint64_t hps_cache_refresh_rows_uploaded(hps_cache_t* c) { return c && c->cache ? c->cache->refresh_rows_uploaded() : 0; }

int hps_pool_numa_node(void) { return ThreadPool::NumaNode(); }
uint64_t hps_pool_fast_overruns(void) { return ThreadPool::FastOverruns(); }
int hps_bind_calling_thread(void) { return ThreadPool::BindCallingThread() ? 1 : 0; }

int hps_wake_copy_engines(int device, char* buf, uint64_t cap) {
  const std::string report = WakeCopyEngines(device);
  if (buf && cap) {
    const size_t n = std::min<size_t>(report.size(), (size_t)cap - 1);
    memcpy(buf, report.data(), n);
    buf[n] = 0;
  }
  // "<up> engines host->device, <down> device->host, ..." or "skipped: ..."
  int up = 0, down = 0;
  if (sscanf(report.c_str(), "%d engines host->device, %d device->host", &up, &down) != 2) return 0;
  return up + down;
}

int hps_cache_table_info(hps_cache_t* c, uint32_t table, hps_cache_table_info_t* out) {
  return Guard([&]() -> Status {
    if (!c || !out) return Error(Code::kInvalidArg, "null argument");
    if (!c->cache) return Error(Code::kUnsupported, "model '", c->model, "' runs without GPU embedding cache");
    if (table >= c->cache->num_tables()) return Error(Code::kInvalidArg, "table index out of range");
    const auto& cfg = c->cache->get_cache_config();
    out->embedding_vecsize = cfg.embedding_vec_size_[table];
    out->num_buckets = cfg.num_set_in_cache_[table];
    out->capacity_rows = cfg.capacity_rows_[table];
    return Status::Ok();
  });
}

int hps_cache_counters(hps_cache_t* c, hps_cache_counters_t* out) {
  return Guard([&]() -> Status {
    if (!c || !out) return Error(Code::kInvalidArg, "null argument");
    if (!c->cache) return Error(Code::kUnsupported, "model '", c->model, "' runs without GPU embedding cache");
    const CacheCounters k = c->cache->counters();
    out->lookups = k.lookups; out->keys = k.keys; out->misses = k.misses; out->unique_misses = k.unique_misses;
    out->inserted = k.inserted; out->refreshed = k.refreshed; out->dropped = k.dropped; out->async_calls = k.async_calls;
    return Status::Ok();
  });
}

int hps_cache_query(hps_cache_t* c, uint32_t table, const int64_t* h_keys, uint64_t n, int32_t* h_slots) {
  return Guard([&]() -> Status {
    if (!c || (n && (!h_keys || !h_slots))) return Error(Code::kInvalidArg, "null argument");
    if (!c->cache) return Error(Code::kUnsupported, "model '", c->model, "' runs without GPU embedding cache");
    return c->cache->Query(table, h_keys, n, h_slots);
  });
}

int hps_cache_wait_async(hps_cache_t* c) {
  return Guard([&]() -> Status {
    if (!c) return Error(Code::kInvalidArg, "null argument");
    if (c->cache) c->cache->WaitAsync();
    return Status::Ok();
  });
}

void hps_cache_release(hps_cache_t* c) { delete c; }

int hps_session_create(hps_server_t* sv, const char* model, hps_cache_t* cache, hps_session_t** out) {
  return Guard([&]() -> Status {
    if (!sv || !model || !out) return Error(Code::kInvalidArg, "null argument");
    std::unique_ptr<LookupSession> s;
    HPS_RETURN_IF_ERROR(sv->ps->create_lookup_session(model, cache ? cache->cache : nullptr, &s));
    *out = new hps_session{sv->ps, std::move(s)};
    return Status::Ok();
  });
}

int hps_session_create_from_cache(hps_cache_t* cache, hps_session_t** out) {
  return Guard([&]() -> Status {
    if (!cache || !out) return Error(Code::kInvalidArg, "null argument");
    std::unique_ptr<LookupSession> s;
    HPS_RETURN_IF_ERROR(cache->ps->create_lookup_session(cache->model, cache->cache, &s));
    *out = new hps_session{cache->ps, std::move(s)};
    return Status::Ok();
  });
}

void hps_session_destroy(hps_session_t* s) { delete s; }

int hps_session_lookup(hps_session_t* s, const void* const* h_keys_per_table, float* const* vectors_per_table,
                       const size_t* num_keys_per_table, size_t num_tables) {
  return Guard([&]() -> Status {
    if (!s || !h_keys_per_table || !vectors_per_table || !num_keys_per_table) return Error(Code::kInvalidArg, "null argument");
    return s->s->lookup(h_keys_per_table, vectors_per_table, num_keys_per_table, num_tables);
  });
}

int hps_session_lookup_device(hps_session_t* s, const int64_t* d_keys_flat, float* const* d_vectors_per_table,
                              const size_t* num_keys_per_table, size_t num_tables) {
  return Guard([&]() -> Status {
    if (!s || !d_keys_flat || !d_vectors_per_table || !num_keys_per_table) return Error(Code::kInvalidArg, "null argument");
    return s->s->lookup_from_device(d_keys_flat, d_vectors_per_table, num_keys_per_table, num_tables);
  });
}

int hps_session_last_stats(hps_session_t* s, hps_lookup_stats_t* out) {
  return Guard([&]() -> Status {
    if (!s || !out) return Error(Code::kInvalidArg, "null argument");
    out->misses = s->s->last_miss_count();
    out->unique_misses = s->s->last_unique_miss_count();
    out->async_insert = s->s->last_call_async() ? 1 : 0;
    out->probe_gather_ms = s->s->last_gpu_ms();
    for (int i = 0; i < 4; ++i) out->phase_ms[i] = s->s->last_phase_ms()[i];
    out->gpu_call_ms = s->s->last_gpu_call_ms();
    out->hit_gather_ms = s->s->last_gather_ms();
    out->unique_keys = s->s->last_unique_key_count();
    out->key_stage_ms = s->s->last_key_stage_ms();
    out->keys_narrowed = s->s->last_keys_narrow() ? 1 : 0;
    out->key_bytes = s->s->last_key_bytes();
    out->scatter_ms = s->s->last_scatter_ms();
    out->insert_ms = s->s->last_insert_ms();
    out->miss_much_mode = s->s->miss_much_mode() ? 1 : 0;
    out->interact_separate = s->s->last_interact_separate() ? 1 : 0;
    out->mode_flips = s->s->mode_flips();
    return Status::Ok();
  });
}

int hps_session_set_option(hps_session_t* s, const char* name, int value) {
  return Guard([&]() -> Status {
    if (!s || !name) return Error(Code::kInvalidArg, "null argument");
    const std::string n(name);
    if (n == "timing") s->s->set_timing(value != 0);
    else if (n == "probe_variant" || n == "probe_unroll") {
      // 1002: tile-local input dedup (default); 1102: without it
      if (value != 1002 && value != 1102) return Error(Code::kInvalidArg, "probe_variant must be 1002 (default) or 1102 (no tile-local input dedup)");
      s->s->set_probe_variant(value);
    } else if (n == "exclusive_kernels") {
      s->s->set_exclusive_kernels(value != 0);
    } else if (n == "fused_unique") {
      s->s->set_fused_unique(value != 0);
    } else if (n == "narrow_publish") {
      s->s->set_narrow_publish(value != 0);
    } else if (n == "interact_mode") {
      if (value < 0 || value > 2) return Error(Code::kInvalidArg, "interact_mode must be 0 (separate steps), 1 (fused) or 2 (by the session's miss volume)");
      s->s->set_interact_mode(value);
    } else if (n == "keys_by_kernel") {
      s->s->set_keys_by_kernel(value);
    } else if (n == "probe_in_lane") {
      s->s->set_probe_in_lane(value);
    } else if (n == "chain_gather") {
      s->s->set_chain_gather(value != 0);
    } else if (n == "xcd_walk") {
      s->s->set_xcd_walk(value != 0);
    } else if (n == "keys_pinned_check") {
      s->s->set_keys_pinned_check(value != 0);
    } else if (n == "narrow_keys") {
      s->s->set_narrow_keys(value);
    } else if (n == "host_gather") {
      s->s->set_force_host_gather(value != 0);
    } else if (n == "split_probe") {
      s->s->set_split_probe(value != 0);
    } else if (n == "defer_insert") {
      s->s->set_defer_insert(value != 0);
    } else if (n == "side_scatter_mb") {
      if (value < 0) return Error(Code::kInvalidArg, "side_scatter_mb must be >= 0");
      s->s->set_side_bytes((size_t)value << 20);
    } else if (n == "in_place_kb") {
      if (value < 0) return Error(Code::kInvalidArg, "in_place_kb must be >= 0");
      s->s->set_in_place_bytes((size_t)value << 10);
    } else if (n == "hit_rate_threshold_permille") {
      if (value < 0) return Error(Code::kInvalidArg, "hit_rate_threshold_permille must be >= 0");
      s->s->set_hit_rate_threshold((float)value / 1000.0f);
    } else return Error(Code::kInvalidArg, "unknown option '", n, "'");
    return Status::Ok();
  });
}

uint32_t hps_shard_owner(int64_t key, uint32_t num_shards) { return num_shards ? ShardOwnerHost(key, num_shards) : 0; }

uint64_t hps_shard_bucket_workspace_bytes(uint64_t n, uint32_t num_shards) { return ShardBucketWorkspaceBytes(n, num_shards); }

int hps_shard_bucket_device(const int64_t* d_keys, uint64_t n, uint32_t num_shards, int64_t* d_keys_sorted, int32_t* d_perm,
                            uint64_t* d_totals, void* d_workspace, void* stream) {
  return Guard([&]() -> Status {
    if (n >= (1ull << 31)) return Error(Code::kUnsupported, "more than 2^31 keys per call");
    if ((n && (!d_keys || !d_keys_sorted || !d_perm)) || !d_totals || !d_workspace) return Error(Code::kInvalidArg, "null argument");
    const hipError_t e = LaunchShardBucket(d_keys, n, num_shards, d_keys_sorted, d_perm, d_totals, d_workspace, (hipStream_t)stream);
    if (e != hipSuccess) return Error(Code::kInternal, "shard bucket launch failed: ", hipGetErrorString(e));
    return Status::Ok();
  });
}

int hps_shard_unpermute_device(const float* d_rows, const int32_t* d_perm, uint64_t n, uint32_t dim, float* d_out, void* stream) {
  return Guard([&]() -> Status {
    if (n && (!d_rows || !d_perm || !d_out)) return Error(Code::kInvalidArg, "null argument");
    const hipError_t e = LaunchShardUnpermute(d_rows, d_perm, n, dim, d_out, (hipStream_t)stream);
    if (e != hipSuccess) return Error(Code::kInternal, "shard unpermute launch failed: ", hipGetErrorString(e));
    return Status::Ok();
  });
}

int hps_shard_unique_id(uint8_t* out128) {
  return Guard([&]() -> Status {
    if (!out128) return Error(Code::kInvalidArg, "null argument");
    return ShardUniqueId(out128);
  });
}

int hps_shard_session_create(hps_session_t* session, uint32_t rank, uint32_t world, const uint8_t* unique_id128, uint64_t max_local_keys,
                             hps_shard_session_t** out) {
  return Guard([&]() -> Status {
    if (!session || !unique_id128 || !out) return Error(Code::kInvalidArg, "null argument");
    if (!session->s->uses_gpu_cache()) return Error(Code::kUnsupported, "the native sharded lookup needs a GPU-cache session");
    std::unique_ptr<ShardTransport> tr;
    HPS_RETURN_IF_ERROR(MakeRcclTransport(rank, world, unique_id128, session->s->device(), &tr));
    std::unique_ptr<ShardedSession> ss;
    HPS_RETURN_IF_ERROR(ShardedSession::Create(session->s, std::move(tr), (size_t)max_local_keys, &ss));
    *out = new hps_shard_session{session->ps, std::move(ss)};
    return Status::Ok();
  });
}

int hps_shard_group_create_local(uint32_t world, hps_shard_group_t** out) {
  return Guard([&]() -> Status {
    if (!out || world == 0 || world > 64) return Error(Code::kInvalidArg, "bad argument");
    *out = new hps_shard_group{MakeLocalShardGroup(world)};
    return Status::Ok();
  });
}

void hps_shard_group_destroy(hps_shard_group_t* group) { delete group; }

int hps_shard_session_create_local(hps_session_t* session, hps_shard_group_t* group, uint32_t rank, uint64_t max_local_keys,
                                   hps_shard_session_t** out) {
  return Guard([&]() -> Status {
    if (!session || !group || !out) return Error(Code::kInvalidArg, "null argument");
    std::unique_ptr<ShardTransport> tr;
    HPS_RETURN_IF_ERROR(MakeLocalTransport(group->g, rank, &tr));
    std::unique_ptr<ShardedSession> ss;
    HPS_RETURN_IF_ERROR(ShardedSession::Create(session->s, std::move(tr), (size_t)max_local_keys, &ss));
    *out = new hps_shard_session{session->ps, std::move(ss)};
    return Status::Ok();
  });
}

int hps_shard_session_lookup(hps_shard_session_t* shard, const int64_t* d_keys, uint64_t n, float* d_out) {
  return Guard([&]() -> Status {
    if (!shard) return Error(Code::kInvalidArg, "null argument");
    return shard->s->Lookup(d_keys, (size_t)n, d_out);
  });
}

int hps_shard_session_lookup_host(hps_shard_session_t* shard, const int64_t* h_keys, uint64_t n, float* d_out) {
  return Guard([&]() -> Status {
    if (!shard) return Error(Code::kInvalidArg, "null argument");
    return shard->s->LookupHost(h_keys, (size_t)n, d_out);
  });
}

int hps_shard_session_last_timing(hps_shard_session_t* shard, float* keys_exchange_ms, float* lookup_ms, float* rows_exchange_ms,
                                  uint64_t* keys_received, int32_t* key_bytes) {
  return Guard([&]() -> Status {
    if (!shard) return Error(Code::kInvalidArg, "null argument");
    const ShardCallStats& st = shard->s->last_stats();
    if (keys_exchange_ms) *keys_exchange_ms = st.keys_exchange_ms;
    if (lookup_ms) *lookup_ms = st.lookup_ms;
    if (rows_exchange_ms) *rows_exchange_ms = st.rows_exchange_ms;
    if (keys_received) *keys_received = st.received;
    if (key_bytes) *key_bytes = st.key_bytes;
    return Status::Ok();
  });
}

int hps_shard_session_last_stats(hps_shard_session_t* shard, uint64_t* capacity, uint32_t* attempts, uint64_t* sent_per_rank, uint32_t world) {
  return Guard([&]() -> Status {
    if (!shard) return Error(Code::kInvalidArg, "null argument");
    const ShardCallStats& st = shard->s->last_stats();
    if (capacity) *capacity = st.capacity;
    if (attempts) *attempts = st.attempts;
    if (sent_per_rank) for (uint32_t p = 0; p < world && p < st.sent.size(); ++p) sent_per_rank[p] = st.sent[p];
    return Status::Ok();
  });
}

void hps_shard_session_destroy(hps_shard_session_t* shard) { delete shard; }

int hps_server_get_shard_cache(hps_server_t* sv, const char* model, uint32_t shard, hps_cache_t** out) {
  return Guard([&]() -> Status {
    if (!sv || !model || !out) return Error(Code::kInvalidArg, "null argument");
    *out = nullptr;
    auto c = sv->ps->get_shard_cache(model, shard);
    if (c) *out = new hps_cache{sv->ps, model, c->device(), c};
    return Status::Ok();
  });
}

int hps_shard_entry_create(hps_server_t* sv, const char* model, int32_t entry_device, hps_shard_entry_t** out) {
  return Guard([&]() -> Status {
    if (!sv || !model || !out) return Error(Code::kInvalidArg, "null argument");
    std::unique_ptr<ShardedEntrySession> s;
    HPS_RETURN_IF_ERROR(ShardedEntrySession::Create(sv->ps, model, entry_device, &s));
    *out = new hps_shard_entry{sv->ps, std::move(s)};
    return Status::Ok();
  });
}

void hps_shard_entry_destroy(hps_shard_entry_t* e) { delete e; }

int hps_shard_entry_lookup(hps_shard_entry_t* e, const void* const* h_keys_per_table, float* const* d_vectors_per_table,
                           const size_t* num_keys_per_table, size_t num_tables) {
  return Guard([&]() -> Status {
    if (!e || !h_keys_per_table || !d_vectors_per_table || !num_keys_per_table) return Error(Code::kInvalidArg, "null argument");
    return e->s->lookup(h_keys_per_table, d_vectors_per_table, num_keys_per_table, num_tables);
  });
}

int hps_shard_entry_lookup_device(hps_shard_entry_t* e, const int64_t* d_keys_flat, float* const* d_vectors_per_table,
                                  const size_t* num_keys_per_table, size_t num_tables) {
  return Guard([&]() -> Status {
    if (!e || !d_vectors_per_table || !num_keys_per_table) return Error(Code::kInvalidArg, "null argument");
    return e->s->lookup_from_device(d_keys_flat, d_vectors_per_table, num_keys_per_table, num_tables);
  });
}

int hps_shard_entry_last_stats(hps_shard_entry_t* e, hps_shard_entry_stats_t* out) {
  return Guard([&]() -> Status {
    if (!e || !out) return Error(Code::kInvalidArg, "null argument");
    const ShardEntryStats& st = e->s->last_stats();
    memset(out, 0, sizeof *out);
    out->keys = st.keys; out->unique_keys = st.unique_keys;
    out->misses = st.misses; out->unique_misses = st.unique_misses;
    out->bucket_ms = st.bucket_ms; out->lookup_ms = st.lookup_ms; out->expand_ms = st.expand_ms; out->key_stage_ms = st.key_stage_ms;
    out->num_shards = e->s->num_shards();
    out->key_bytes = (uint32_t)st.key_bytes;
    out->dedup_level = (uint32_t)st.dedup_level;
    out->transport = (uint32_t)st.transport;
    out->copied_bytes = st.copied_bytes;
    out->dedup_flips = st.dedup_flips;
    for (uint32_t s = 0; s < e->s->num_shards() && s < 64; ++s) {
      out->sent[s] = st.sent[s]; out->passes[s] = st.passes[s]; out->shard_ms[s] = st.shard_ms[s];
      out->copy_wait_ms[s] = st.copy_wait_ms[s];
    }
    return Status::Ok();
  });
}

int hps_shard_entry_set_option(hps_shard_entry_t* e, const char* name, int value) {
  return Guard([&]() -> Status {
    if (!e || !name) return Error(Code::kInvalidArg, "null argument");
    const std::string n(name);
    if (n == "dedup") e->s->set_dedup(value < 0 ? 0 : value > 2 ? 2 : value);
    else if (n == "timing") e->s->set_timing(value != 0);
    else if (n == "transport") return e->s->set_transport(value);
    else if (n == "copy_piece_keys") {
      if (value != 0 && value < 1024) return Error(Code::kInvalidArg, "copy_piece_keys must be 0 (automatic) or >= 1024");
      e->s->set_piece_keys((size_t)value);
    }
    else return Error(Code::kInvalidArg, "unknown option '", n, "'");
    return Status::Ok();
  });
}

uint64_t hps_shard_entry_shard_capacity(hps_shard_entry_t* e) { return e ? e->s->shard_capacity() : 0; }

uint64_t hps_shard_plan_passes(const uint32_t* counts, uint32_t num_tables, uint64_t capacity, uint64_t* out, uint64_t max_passes) {
  if (!counts || num_tables == 0) return 0;
  try {
    const std::vector<ShardPass> plan = PlanShardPasses(counts, num_tables, (size_t)capacity);
    for (size_t i = 0; i < plan.size() && i < max_passes && out; ++i) {
      uint64_t* row = out + i * (1 + (size_t)num_tables);
      row[0] = plan[i].offset;
      for (uint32_t t = 0; t < num_tables; ++t) row[1 + t] = plan[i].n[t];
    }
    return plan.size();
  } catch (...) {
    return 0;
  }
}

int hps_multi_gpu_selftest(const int32_t* devices, uint32_t n, uint64_t probe_bytes, uint32_t timeout_ms, int32_t with_rccl, char* buf,
                           uint64_t cap) {
  return Guard([&]() -> Status {
    if (!devices || n == 0 || !buf || cap == 0) return Error(Code::kInvalidArg, "null argument");
    std::vector<int> devs(devices, devices + n);
    for (uint32_t i = 0; i < n; ++i)
      for (uint32_t k = 0; k < i; ++k)
        if (devs[i] == devs[k]) return Error(Code::kInvalidArg, "device ", devs[i], " is listed twice");
    std::string json;
    const bool ok = MultiGpuSelfTest(devs, probe_bytes, timeout_ms, with_rccl != 0, &json);
    const size_t m = std::min<size_t>(json.size(), (size_t)cap - 1);
    memcpy(buf, json.data(), m);
    buf[m] = 0;
    if (!ok) return Error(Code::kUnavailable, "multi-GPU self-test did not finish within ", timeout_ms, " ms: ", json);
    return Status::Ok();
  });
}

int hps_dense_create(int device, uint32_t num_dense, uint32_t num_layers, const uint32_t* layer_dims, const float* const* weights,
                     const float* const* biases, uint32_t num_tables, uint32_t emb_dim, hps_dense_t** out) {
  return Guard([&]() -> Status {
    if (!out || !layer_dims || !weights || !biases) return Error(Code::kInvalidArg, "null argument");
    *out = nullptr;
    std::vector<uint32_t> dims(layer_dims, layer_dims + num_layers);
    std::vector<const float*> w(weights, weights + num_layers), b(biases, biases + num_layers);
    DenseInteraction* d = nullptr;
    HPS_RETURN_IF_ERROR(DenseInteraction::Create(device, num_dense, dims, w, b, num_tables, emb_dim, &d));
    *out = new hps_dense{std::unique_ptr<DenseInteraction>(d)};
    return Status::Ok();
  });
}

void hps_dense_destroy(hps_dense_t* dense) { delete dense; }

uint32_t hps_dense_out_dim(const hps_dense_t* dense) { return dense ? dense->d->out_dim() : 0; }

uint32_t hps_dense_out_stride(const hps_dense_t* dense) { return dense ? dense->d->out_stride() : 0; }

int hps_session_lookup_interact_device(hps_session_t* s, hps_dense_t* dense, const int64_t* d_keys_flat, uint64_t batch,
                                       const float* d_dense, void* d_out_f16) {
  return Guard([&]() -> Status {
    if (!s || !dense) return Error(Code::kInvalidArg, "null argument");
    return s->s->lookup_interact(dense->d.get(), d_keys_flat, batch, d_dense, d_out_f16);
  });
}

int hps_dense_forward(hps_dense_t* dense, const float* d_dense, const float* d_embeddings, uint64_t batch, void* d_out_f16,
                      void* stream) {
  return Guard([&]() -> Status {
    if (!dense) return Error(Code::kInvalidArg, "null argument");
    return dense->d->Forward(d_dense, d_embeddings, batch, d_out_f16, (hipStream_t)stream);
  });
}

}  // extern "C"
