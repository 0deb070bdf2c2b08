#include "config.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <limits>

namespace hps {

namespace {

Status Mandatory(const char* key) {
  // HPS_ARG_MANDATORY_ERROR, triton_helpers.cpp:36-40
  return Error(Code::kInvalidArg, "The parameter '", key,
               "' is mandatory. Please confirm that it has been added to the configuration file.");
}

Status BadType(const char* key, const char* want) {
  return Error(Code::kInvalidArg, "The parameter '", key, "' cannot be read as ", want, ".");
}

template <typename F>
Status Scalar(const Json& j, const char* key, bool required, const char* want, F&& conv) {
  const Json* m = j.Find(key);
  if (!m) return required ? Mandatory(key) : Status::Ok();
  if (!conv(*m)) return BadType(key, want);
  return Status::Ok();
}

std::string Normalised(std::string s, bool dash_too) {
  for (auto& c : s) {
    if (c == ' ' || (dash_too && c == '-')) c = '_';
    else c = (char)tolower((unsigned char)c);
  }
  return s;
}

template <typename E>
Status EnumField(E& v, const Json& j, const char* key, bool required, bool dash_too,
                 const std::vector<std::pair<E, std::vector<const char*>>>& table, const char* tname) {
  std::string tmp;
  HPS_RETURN_IF_ERROR(ParseField(tmp, j, key, required));
  tmp = Normalised(tmp, dash_too);
  if (tmp.empty() && !required) return Status::Ok();  // keep existing value
  for (const auto& row : table)
    for (const char* name : row.second)
      if (tmp == name) { v = row.first; return Status::Ok(); }
  return Error(Code::kInvalidArg, "Unable to map parameter '", key, "' = \"", tmp, "\" to ", tname, "!");
}

}  // namespace

const char* ToString(DatabaseType v) {
  switch (v) {
    case DatabaseType::Disabled: return "disabled";
    case DatabaseType::HashMap: return "hash_map";
    case DatabaseType::ParallelHashMap: return "parallel_hash_map";
    case DatabaseType::RedisCluster: return "redis_cluster";
    case DatabaseType::RocksDB: return "rocks_db";
  }
  return "?";
}
const char* ToString(DatabaseOverflowPolicy v) {
  switch (v) {
    case DatabaseOverflowPolicy::EvictRandom: return "evict_random";
    case DatabaseOverflowPolicy::EvictLeastUsed: return "evict_least_used";
    case DatabaseOverflowPolicy::EvictOldest: return "evict_oldest";
  }
  return "?";
}
const char* ToString(UpdateSourceType v) {
  switch (v) {
    case UpdateSourceType::Null: return "null";
    case UpdateSourceType::KafkaMessageQueue: return "kafka_message_queue";
    case UpdateSourceType::FileTail: return "file_tail";
  }
  return "?";
}
const char* ToString(EmbeddingCacheType v) {
  switch (v) {
    case EmbeddingCacheType::Dynamic: return "dynamic";
    case EmbeddingCacheType::Static: return "static";
    case EmbeddingCacheType::UVM: return "uvm";
    case EmbeddingCacheType::Stochastic: return "stochastic";
  }
  return "?";
}

Status ParseField(bool& v, const Json& j, const char* key, bool required) {
  return Scalar(j, key, required, "a boolean", [&](const Json& m) { return m.AsBool(&v); });
}
Status ParseField(double& v, const Json& j, const char* key, bool required) {
  return Scalar(j, key, required, "a number", [&](const Json& m) { return m.AsDouble(&v); });
}
Status ParseField(float& v, const Json& j, const char* key, bool required) {
  // triton_helpers.cpp:88-109: parse as double, clamp + error when out of float range.
  double tmp = v;
  HPS_RETURN_IF_ERROR(ParseField(tmp, j, key, required));
  if (tmp < std::numeric_limits<float>::lowest() || tmp > std::numeric_limits<float>::max()) {
    v = tmp < 0 ? std::numeric_limits<float>::lowest() : std::numeric_limits<float>::max();
    return Error(Code::kInvalidArg, "The parameter '", key, "' = ", tmp,
                 " was truncated because it is out of bounds!");
  }
  v = (float)tmp;
  return Status::Ok();
}
Status ParseField(int64_t& v, const Json& j, const char* key, bool required) {
  return Scalar(j, key, required, "an integer", [&](const Json& m) { return m.AsInt(&v); });
}
Status ParseField(int32_t& v, const Json& j, const char* key, bool required) {
  // triton_helpers.cpp:111-125
  int64_t tmp = v;
  HPS_RETURN_IF_ERROR(ParseField(tmp, j, key, required));
  v = (int32_t)tmp;
  if (v != tmp)
    return Error(Code::kInvalidArg, "The parameter '", key, "' = ", tmp,
                 " was truncated because it is out of bounds!");
  return Status::Ok();
}
Status ParseField(size_t& v, const Json& j, const char* key, bool required) {
  uint64_t tmp = v;
  HPS_RETURN_IF_ERROR(
      Scalar(j, key, required, "an unsigned integer", [&](const Json& m) { return m.AsUInt(&tmp); }));
  v = (size_t)tmp;
  return Status::Ok();
}
Status ParseField(std::string& v, const Json& j, const char* key, bool required) {
  return Scalar(j, key, required, "a string", [&](const Json& m) { return m.AsString(&v); });
}

Status ParseField(DatabaseType& v, const Json& j, const char* key, bool required) {
  // aliases: triton_helpers.cpp:199-242
  return EnumField<DatabaseType>(
      v, j, key, required, /*dash_too=*/true,
      {{DatabaseType::Disabled, {"disabled", "disable", "none"}},
       {DatabaseType::HashMap, {"hash_map", "hashmap", "hash", "map"}},
       {DatabaseType::ParallelHashMap,
        {"parallel_hash_map", "parallel_hashmap", "parallel_hash", "parallel_map"}},
       {DatabaseType::RedisCluster, {"redis_cluster", "redis"}},
       {DatabaseType::RocksDB, {"rocks_db", "rocksdb", "rocks"}}},
      "DatabaseType_t");
}
Status ParseField(DatabaseOverflowPolicy& v, const Json& j, const char* key, bool required) {
  // triton_helpers.cpp:267-292
  return EnumField<DatabaseOverflowPolicy>(
      v, j, key, required, false,
      {{DatabaseOverflowPolicy::EvictRandom, {"evict_random", "random"}},
       {DatabaseOverflowPolicy::EvictLeastUsed, {"evict_least_used", "least_used"}},
       {DatabaseOverflowPolicy::EvictOldest, {"evict_oldest", "oldest"}}},
      "DatabaseOverflowPolicy_t");
}
Status ParseField(UpdateSourceType& v, const Json& j, const char* key, bool required) {
  // triton_helpers.cpp:316-333
  return EnumField<UpdateSourceType>(
      v, j, key, required, false,
      {{UpdateSourceType::Null, {"null", "none"}},
       {UpdateSourceType::KafkaMessageQueue, {"kafka_message_queue", "kafka_mq", "kafka"}},
       {UpdateSourceType::FileTail, {"file_tail"}}},
      "UpdateSourceType_t");
}

namespace {
template <typename T, typename Conv>
Status ArrayField(std::vector<T>& v, const Json& j, const char* key, bool required, const char* want,
                  Conv&& conv) {
  const Json* m = j.Find(key);
  if (!m) return required ? Mandatory(key) : Status::Ok();
  if (!m->is_array()) return BadType(key, "an array");
  for (size_t i = 0; i < m->size(); ++i) {
    T e{};
    if (!conv(m->at(i), &e)) return BadType(key, want);
    v.emplace_back(e);  // appends, like the reference (callers clear() first where it matters)
  }
  return Status::Ok();
}
}  // namespace

Status ParseField(std::vector<float>& v, const Json& j, const char* key, bool required) {
  return ArrayField<float>(v, j, key, required, "an array of numbers", [](const Json& e, float* o) {
    double d;
    if (!e.AsDouble(&d)) return false;
    *o = (float)d;
    return true;
  });
}
Status ParseField(std::vector<int32_t>& v, const Json& j, const char* key, bool required) {
  return ArrayField<int32_t>(v, j, key, required, "an array of integers", [](const Json& e, int32_t* o) {
    int64_t i;
    if (!e.AsInt(&i)) return false;
    *o = (int32_t)i;
    return true;
  });
}
Status ParseField(std::vector<size_t>& v, const Json& j, const char* key, bool required) {
  return ArrayField<size_t>(v, j, key, required, "an array of integers", [](const Json& e, size_t* o) {
    int64_t i;
    if (!e.AsInt(&i)) return false;
    *o = (size_t)i;
    return true;
  });
}
Status ParseField(std::vector<std::string>& v, const Json& j, const char* key, bool required) {
  return ArrayField<std::string>(v, j, key, required, "an array of strings",
                                 [](const Json& e, std::string* o) { return e.AsString(o); });
}

Status ParseParameterServerJson(const Json& root, ParameterServerConfig* out) {
  if (!root.is_object()) return Error(Code::kInvalidArg, "ps.json: top level must be an object");
  ParameterServerConfig cfg;

  HPS_RETURN_IF_ERROR(ParseField(cfg.support_int64_key, root, "supportlonglong", true));

  if (const Json* j = root.Find("volatile_db")) {  // backend.cpp:130-216
    auto& p = cfg.volatile_db;
    HPS_RETURN_IF_ERROR(ParseField(p.type, *j, "type", false));
    HPS_RETURN_IF_ERROR(ParseField(p.address, *j, "address", false));
    HPS_RETURN_IF_ERROR(ParseField(p.user_name, *j, "user_name", false));
    HPS_RETURN_IF_ERROR(ParseField(p.password, *j, "password", false));
    HPS_RETURN_IF_ERROR(ParseField(p.num_partitions, *j, "num_partitions", false));
    HPS_RETURN_IF_ERROR(ParseField(p.allocation_rate, *j, "allocation_rate", false));
    HPS_RETURN_IF_ERROR(ParseField(p.max_batch_size, *j, "max_batch_size", false));
    HPS_RETURN_IF_ERROR(ParseField(p.overflow_margin, *j, "overflow_margin", false));
    HPS_RETURN_IF_ERROR(ParseField(p.overflow_policy, *j, "overflow_policy", false));
    HPS_RETURN_IF_ERROR(ParseField(p.overflow_resolution_target, *j, "overflow_resolution_target", false));
    HPS_RETURN_IF_ERROR(ParseField(p.initial_cache_rate, *j, "initial_cache_rate", false));
    HPS_RETURN_IF_ERROR(ParseField(p.cache_missed_embeddings, *j, "cache_missed_embeddings", false));
    if (j->Find("update_filters")) p.update_filters.clear();
    HPS_RETURN_IF_ERROR(ParseField(p.update_filters, *j, "update_filters", false));
    if (p.num_partitions == 0) return Error(Code::kInvalidArg, "volatile_db.num_partitions must be > 0");
  }
  if (const Json* j = root.Find("persistent_db")) {  // backend.cpp:220-259
    auto& p = cfg.persistent_db;
    HPS_RETURN_IF_ERROR(ParseField(p.type, *j, "type", false));
    HPS_RETURN_IF_ERROR(ParseField(p.path, *j, "path", false));
    HPS_RETURN_IF_ERROR(ParseField(p.num_threads, *j, "num_threads", false));
    HPS_RETURN_IF_ERROR(ParseField(p.read_only, *j, "read_only", false));
    HPS_RETURN_IF_ERROR(ParseField(p.max_batch_size, *j, "max_batch_size", false));
    if (j->Find("update_filters")) p.update_filters.clear();
    HPS_RETURN_IF_ERROR(ParseField(p.update_filters, *j, "update_filters", false));
  }
  if (const Json* j = root.Find("update_source")) {  // backend.cpp:263-308
    auto& p = cfg.update_source;
    HPS_RETURN_IF_ERROR(ParseField(p.type, *j, "type", false));
    HPS_RETURN_IF_ERROR(ParseField(p.brokers, *j, "brokers", false));
    HPS_RETURN_IF_ERROR(ParseField(p.receive_buffer_size, *j, "receive_buffer_size", false));
    HPS_RETURN_IF_ERROR(ParseField(p.poll_timeout_ms, *j, "poll_timeout_ms", false));
    HPS_RETURN_IF_ERROR(ParseField(p.max_batch_size, *j, "max_batch_size", false));
    HPS_RETURN_IF_ERROR(ParseField(p.failure_backoff_ms, *j, "failure_backoff_ms", false));
    HPS_RETURN_IF_ERROR(ParseField(p.max_commit_interval, *j, "max_commit_interval", false));
  }

  const Json* models = root.Find("models");  // backend.cpp:311-316 (absent/empty -> warning, not error)
  const size_t nmodels = (models && models->is_array()) ? models->size() : 0;
  for (size_t mi = 0; mi < nmodels; ++mi) {
    const Json& j = models->at(mi);
    if (!j.is_object()) return Error(Code::kInvalidArg, "ps.json: models[", mi, "] must be an object");
    InferenceParams p;
    HPS_RETURN_IF_ERROR(ParseField(p.model_name, j, "model", true));
    HPS_RETURN_IF_ERROR(ParseField(p.network_file, j, "network_file", false));
    HPS_RETURN_IF_ERROR(ParseField(p.max_batchsize, j, "max_batch_size", true));
    HPS_RETURN_IF_ERROR(ParseField(p.dense_model_file, j, "dense_file", false));
    HPS_RETURN_IF_ERROR(ParseField(p.sparse_model_files, j, "sparse_files", true));
    HPS_RETURN_IF_ERROR(ParseField(p.use_gpu_embedding_cache, j, "gpucache", true));
    HPS_RETURN_IF_ERROR(ParseField(p.hit_rate_threshold, j, "hit_rate_threshold", p.use_gpu_embedding_cache));
    HPS_RETURN_IF_ERROR(ParseField(p.cache_size_percentage, j, "gpucacheper", p.use_gpu_embedding_cache));
    p.i64_input_key = cfg.support_int64_key;
    HPS_RETURN_IF_ERROR(ParseField(p.number_of_worker_buffers_in_pool, j, "num_of_worker_buffer_in_pool", true));
    HPS_RETURN_IF_ERROR(ParseField(p.number_of_refresh_buffers_in_pool, j, "num_of_refresher_buffer_in_pool", false));
    HPS_RETURN_IF_ERROR(ParseField(p.cache_refresh_percentage_per_iteration, j,
                                   "cache_refresh_percentage_per_iteration", false));
    p.deployed_devices.clear();
    HPS_RETURN_IF_ERROR(ParseField(p.deployed_devices, j, "deployed_device_list", true));
    if (p.deployed_devices.empty())
      return Error(Code::kInvalidArg, "Model '", p.model_name, "': deployed_device_list must not be empty");
    p.device_id = p.deployed_devices.back();  // backend.cpp:422
    HPS_RETURN_IF_ERROR(ParseField(p.default_value_for_each_table, j, "default_value_for_each_table", true));
    HPS_RETURN_IF_ERROR(ParseField(p.maxnum_des_feature_per_sample, j, "maxnum_des_feature_per_sample", false));
    HPS_RETURN_IF_ERROR(ParseField(p.maxnum_catfeature_query_per_table_per_sample, j,
                                   "maxnum_catfeature_query_per_table_per_sample", true));
    HPS_RETURN_IF_ERROR(ParseField(p.embedding_vecsize_per_table, j, "embedding_vecsize_per_table", true));
    HPS_RETURN_IF_ERROR(ParseField(p.embedding_table_names, j, "embedding_table_names", false));
    HPS_RETURN_IF_ERROR(ParseField(p.label_dim, j, "label_dim", false));
    HPS_RETURN_IF_ERROR(ParseField(p.slot_num, j, "slot_num", false));
    HPS_RETURN_IF_ERROR(ParseField(p.refresh_delay, j, "refresh_delay", false));
    HPS_RETURN_IF_ERROR(ParseField(p.refresh_interval, j, "refresh_interval", false));
    std::string cache_type;
    HPS_RETURN_IF_ERROR(ParseField(cache_type, j, "embedding_cache_type", false));
    cache_type = Normalised(cache_type, false);  // backend.cpp:482 lower-cases, so "Stochastic" matches too
    if (cache_type == "static") p.embedding_cache_type = EmbeddingCacheType::Static;
    else if (cache_type == "uvm") p.embedding_cache_type = EmbeddingCacheType::UVM;
    else if (cache_type == "stochastic") p.embedding_cache_type = EmbeddingCacheType::Stochastic;
    else p.embedding_cache_type = EmbeddingCacheType::Dynamic;
    HPS_RETURN_IF_ERROR(ParseField(p.init_ec, j, "init_ec", false));
    HPS_RETURN_IF_ERROR(ParseField(p.fp8_quant, j, "fp8_quant", false));
    HPS_RETURN_IF_ERROR(ParseField(p.enable_pagelock, j, "enable_pagelock", false));
    HPS_RETURN_IF_ERROR(ParseField(p.cache_load_factor, j, "gpucache_load_factor", false));
    HPS_RETURN_IF_ERROR(ParseField(p.cache_admission, j, "gpucache_admission", false));
    HPS_RETURN_IF_ERROR(ParseField(p.small_miss_insert_interval, j, "gpucache_small_miss_insert_interval", false));
    if (p.small_miss_insert_interval < 1 || p.small_miss_insert_interval > 1024)
      return Error(Code::kInvalidArg, "Model '", p.model_name, "': gpucache_small_miss_insert_interval must be in [1, 1024]");
    HPS_RETURN_IF_ERROR(ParseField(p.ps_direct_access, j, "ps_direct_access", false));
    HPS_RETURN_IF_ERROR(ParseField(p.refresh_changed_only, j, "gpucache_refresh_changed_only", false));
    HPS_RETURN_IF_ERROR(ParseField(p.refresh_link_share, j, "gpucache_refresh_link_share", false));
    if (!(p.refresh_link_share > 0.0) || p.refresh_link_share > 1.0)
      return Error(Code::kInvalidArg, "Model '", p.model_name, "': gpucache_refresh_link_share must be in (0, 1]");
    {
      // "table_sharding": "hash" — the model's GPU caches are SHARDS, not replicas: entry s of deployed_device_list holds the
      // keys with mix64(key) mod P == s (P = the length of the list; a device may appear more than once: logical shards)
      std::string sharding;
      HPS_RETURN_IF_ERROR(ParseField(sharding, j, "table_sharding", false));
      sharding = Normalised(sharding, false);
      if (sharding == "hash") p.table_sharding = true;
      else if (!(sharding.empty() || sharding == "none" || sharding == "replicas"))
        return Error(Code::kInvalidArg, "Model '", p.model_name, "': table_sharding must be \"hash\" or \"none\", got '", sharding, "'");
      HPS_RETURN_IF_ERROR(ParseField(p.shard_capacity_factor, j, "shard_capacity_factor", false));
      HPS_RETURN_IF_ERROR(ParseField(p.shard_dedup, j, "shard_dedup", false));
      std::string transport;
      HPS_RETURN_IF_ERROR(ParseField(transport, j, "shard_transport", false));
      transport = Normalised(transport, false);
      if (transport == "staged_copy") p.shard_transport_staged = true;
      else if (!(transport.empty() || transport == "peer_store"))
        return Error(Code::kInvalidArg, "Model '", p.model_name, "': shard_transport must be \"peer_store\" or \"staged_copy\", got '", transport, "'");
      HPS_RETURN_IF_ERROR(ParseField(p.shard_copy_piece_keys, j, "shard_copy_piece_keys", false));
      if (p.shard_copy_piece_keys != 0 && (p.shard_copy_piece_keys < 1024 || p.shard_copy_piece_keys > (1u << 24)))
        return Error(Code::kInvalidArg, "Model '", p.model_name, "': shard_copy_piece_keys must be 0 (automatic) or in [1024, 16777216]");
      if (p.table_sharding) {
        if (!p.use_gpu_embedding_cache)
          return Error(Code::kInvalidArg, "Model '", p.model_name, "': table_sharding shards the GPU caches and needs gpucache = true");
        if (p.deployed_devices.empty() || p.deployed_devices.size() > 64)
          return Error(Code::kInvalidArg, "Model '", p.model_name, "': table_sharding needs 1..64 entries in deployed_device_list, got ",
                       p.deployed_devices.size());
        if (!(p.shard_capacity_factor >= 1.0))
          return Error(Code::kInvalidArg, "Model '", p.model_name, "': shard_capacity_factor must be >= 1");
      }
    }
    p.volatile_db = cfg.volatile_db;
    p.persistent_db = cfg.persistent_db;
    p.update_source = cfg.update_source;

    // --- hardening the reference leaves to the engine ---
    const size_t T = p.sparse_model_files.size();
    if (T == 0) return Error(Code::kInvalidArg, "Model '", p.model_name, "': sparse_files must not be empty");
    auto want_len = [&](size_t n, const char* key) -> Status {
      if (n != T)
        return Error(Code::kInvalidArg, "Model '", p.model_name, "': '", key, "' has ", n,
                     " entries but sparse_files has ", T);
      return Status::Ok();
    };
    HPS_RETURN_IF_ERROR(want_len(p.embedding_vecsize_per_table.size(), "embedding_vecsize_per_table"));
    HPS_RETURN_IF_ERROR(want_len(p.maxnum_catfeature_query_per_table_per_sample.size(),
                                 "maxnum_catfeature_query_per_table_per_sample"));
    HPS_RETURN_IF_ERROR(want_len(p.default_value_for_each_table.size(), "default_value_for_each_table"));
    if (p.embedding_table_names.empty()) {
      for (size_t t = 0; t < T; ++t)  // docs/architecture.md:102
        p.embedding_table_names.push_back("sparse_embedding" + std::to_string(t + 1));
    } else {
      HPS_RETURN_IF_ERROR(want_len(p.embedding_table_names.size(), "embedding_table_names"));
    }
    for (size_t d : p.embedding_vecsize_per_table)
      if (d == 0) return Error(Code::kInvalidArg, "Model '", p.model_name, "': embedding_vecsize_per_table entries must be > 0");
    if (p.use_gpu_embedding_cache) {
      if (!(p.cache_size_percentage > 0.0f) || p.cache_size_percentage > 1.0f)
        return Error(Code::kInvalidArg, "Model '", p.model_name, "': gpucacheper must be in (0,1], got ",
                     p.cache_size_percentage);
      if (!(p.cache_load_factor > 0.05) || p.cache_load_factor > 1.0)
        return Error(Code::kInvalidArg, "Model '", p.model_name, "': gpucache_load_factor must be in (0.05,1]");
    }
    if (!cfg.models.count(p.model_name)) cfg.model_order.push_back(p.model_name);
    cfg.models[p.model_name] = std::move(p);  // backend.cpp:519-522 (replace on re-parse)
  }
  *out = std::move(cfg);
  return Status::Ok();
}

Status ParseParameterServerText(const std::string& text, ParameterServerConfig* out) {
  Json root;
  std::string err;
  if (!Json::Parse(text, &root, &err)) return Error(Code::kInvalidArg, "ps.json: ", err);
  return ParseParameterServerJson(root, out);
}

Status ParseParameterServerFile(const std::string& path, ParameterServerConfig* out) {
  Json root;
  std::string err;
  if (!Json::ParseFile(path, &root, &err))
    return Error(Code::kInvalidArg, "Failed to read Parameter Server Configuration: ", err);
  return ParseParameterServerJson(root, out);
}

}  // namespace hps
