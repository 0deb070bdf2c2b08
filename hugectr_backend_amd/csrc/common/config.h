// Parameter-server configuration carriers.
//
// Field names follow the structs the reference fills from ps.json
// (HugeCTR::InferenceParams / VolatileDatabaseParams / PersistentDatabaseParams / UpdateSourceParams,
// built at /root/reference/hps_backend/src/backend.cpp:128-523).  The structs themselves live in the
// un-vendored HugeCTR headers, so only the fields the shell reads or writes are restated.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "json.h"
#include "status.h"

namespace hps {

// triton_helpers.cpp:183-248
enum class DatabaseType { Disabled, HashMap, ParallelHashMap, RedisCluster, RocksDB };
// triton_helpers.cpp:250-298
enum class DatabaseOverflowPolicy { EvictRandom, EvictLeastUsed, EvictOldest };
// triton_helpers.cpp:300-339
enum class UpdateSourceType { Null, KafkaMessageQueue, FileTail };   // FileTail: this build's transport (csrc/ps/update_source.h)
// backend.cpp:479-491
enum class EmbeddingCacheType { Dynamic, Static, UVM, Stochastic };

const char* ToString(DatabaseType v);
const char* ToString(DatabaseOverflowPolicy v);
const char* ToString(UpdateSourceType v);
const char* ToString(EmbeddingCacheType v);

struct VolatileDatabaseParams {  // backend.cpp:128-216
  DatabaseType type = DatabaseType::ParallelHashMap;
  std::string address = "127.0.0.1:7000";
  std::string user_name = "default";
  std::string password;
  size_t num_partitions = 8;  // README.md:133
  size_t allocation_rate = 256ull << 20;
  size_t max_batch_size = 64 * 1024;
  size_t overflow_margin = SIZE_MAX;
  DatabaseOverflowPolicy overflow_policy = DatabaseOverflowPolicy::EvictRandom;
  double overflow_resolution_target = 0.8;
  double initial_cache_rate = 1.0;
  bool cache_missed_embeddings = false;
  std::vector<std::string> update_filters{"^hps_.+$"};
};

struct PersistentDatabaseParams {  // backend.cpp:218-259
  DatabaseType type = DatabaseType::Disabled;
  std::string path;
  size_t num_threads = 16;
  bool read_only = false;
  size_t max_batch_size = 64 * 1024;
  std::vector<std::string> update_filters{"^hps_.+$"};
};

struct UpdateSourceParams {  // backend.cpp:261-308
  UpdateSourceType type = UpdateSourceType::Null;
  std::string brokers = "127.0.0.1:9092";
  size_t receive_buffer_size = 256 * 1024;
  size_t poll_timeout_ms = 500;
  size_t max_batch_size = 8 * 1024;
  size_t failure_backoff_ms = 50;
  size_t max_commit_interval = 32;
};

struct InferenceParams {  // backend.cpp:318-516
  std::string model_name;
  std::string network_file;
  size_t max_batchsize = 0;
  float hit_rate_threshold = 0.55f;       // backend.cpp:372
  std::string dense_model_file;
  std::vector<std::string> sparse_model_files;
  int device_id = 0;
  bool use_gpu_embedding_cache = true;
  float cache_size_percentage = 0.55f;    // backend.cpp:380
  bool i64_input_key = true;
  bool use_mixed_precision = false;       // model_state.cpp:279 (never set by ps.json)
  int number_of_worker_buffers_in_pool = 2;
  int number_of_refresh_buffers_in_pool = 1;
  float cache_refresh_percentage_per_iteration = 0.1f;
  std::vector<int> deployed_devices{0};
  std::vector<float> default_value_for_each_table;
  int maxnum_des_feature_per_sample = 26;
  float refresh_delay = 0.0f;             // model_state.cpp:327
  float refresh_interval = 0.0f;          // model_state.cpp:319
  std::vector<size_t> maxnum_catfeature_query_per_table_per_sample;
  std::vector<size_t> embedding_vecsize_per_table;
  std::vector<std::string> embedding_table_names;
  int label_dim = 1;
  int slot_num = 10;
  EmbeddingCacheType embedding_cache_type = EmbeddingCacheType::Dynamic;
  bool init_ec = true;
  bool fp8_quant = false;
  bool enable_pagelock = false;
  VolatileDatabaseParams volatile_db;
  PersistentDatabaseParams persistent_db;
  UpdateSourceParams update_source;

  // --- additions of this build (MI355X engine knobs; all optional in ps.json) ---
  double cache_load_factor = 0.75;  // "gpucache_load_factor": slots = ceil(capacity / load_factor)
  // "gpucache_admission": true (default) = a key seen once does not take a slot that was hit within the last 2.5 cache
  // turnovers (its row is served, it just is not cached on that call; one such key in 16 is let in regardless); false = every
  // missed key is inserted, evicting its bucket's least recently used key — the reference's cache
  bool cache_admission = true;
  // "gpucache_small_miss_insert_interval": n (default 4; 1 = every call): a synchronous NEAR-ALL-HIT call (at most one key in 64
  // missed; a cold or low-hit-rate cache inserts on every call) whose missed rows are FEW (they stay
  // where the host gathered them: at most in_place_kb, 1 MB) inserts them only every n-th such call of its session; the other
  // calls serve their missed rows exactly and leave them uncached (counted as `dropped`); calls whose missed rows were uploaded
  // but still fit the second stream's scatter (side_scatter_mb, 16 MB: a call at 99 % hit) insert every n/2-th time.  Near-all-hit traffic — where a
  // production cache lives (docs/architecture.md:65-67) — otherwise pays a writer window on every call: the insert kernel waits
  // for every other session's hit gather and every later probe waits for it (15-35 us of cross-queue hand-off each way,
  // profiles/round4/timeline_99.9pct_hit_after.txt: 174 us of idle GPU per pair of calls).  A key that keeps being asked for
  // enters within n calls, like the one-in-16 rule of the admission policy; with gpucache_admission = false every call inserts.
  int small_miss_insert_interval = 4;
  // "gpucache_refresh_changed_only" (default true): refresh_embedding_cache re-reads only rows that can differ from the host tier's
  // (tables reloaded or keys updated since the cache last looked); false: every resident row on every refresh, as the reference.
  // "gpucache_refresh_link_share" (default 0.15, in (0, 1]): the refresher's DUTY CYCLE while lookup sessions are serving — after a
  // piece (32,768 rows: host gather on 4 threads, upload, insert) that took t it pauses for t x (1/share - 1), so the link, the
  // serving pool and the cache's writer windows are its for at most that share of the time (measured: 1.5-2 GB/s of refresh under
  // the headline's load); with nobody serving it runs at full speed.  1.0: unpaced, 262,144-row pieces (rounds 1-5).
  bool refresh_changed_only = true;
  double refresh_link_share = 0.15;
  // "ps_direct_access": the GPU resolves missed keys through a device-resident index of the host tier and reads
  // the rows in place from pinned host memory over PCIe (no host threads, no staging copy).  Needs gpucache.
  bool ps_direct_access = false;
  // "table_sharding": "hash" — BASELINE config 3 behind the plugin boundary (csrc/cache/shard_entry.h; not in the reference,
  // which is replicas only: docs/architecture.md:11,29).  Entry s of deployed_device_list is SHARD s: its cache holds the keys
  // with mix64(key) mod P == s, gpucacheper of them.  An instance on any listed device serves whole requests: it buckets the
  // keys by owner and the owners write their rows straight into its output buffer (peer-mapped, over xGMI).
  bool table_sharding = false;
  double shard_capacity_factor = 2.0;   // one owner's lookup session holds factor x (request capacity / P) keys; a request that
                                        // sends an owner more is served in several passes
  bool shard_dedup = true;              // "shard_dedup": a key the request repeats travels to its owner once
  // "shard_transport": "peer_store" (default: the owners' kernels store rows straight into the entry GPU's output over peer
  // mappings) | "staged_copy" (owners gather pieces into local blocks, copy engines ship them — hipMemcpyPeerAsync — and a kernel
  // on the entry GPU puts the rows in place; csrc/cache/shard_entry.h).  "shard_copy_piece_keys": keys per piece; 0 (default) =
  // automatic: 131,072 for a shard on another GPU, ONE piece for a shard on the entry GPU itself (its copy is local: nothing to overlap)
  bool shard_transport_staged = false;
  size_t shard_copy_piece_keys = 0;
  size_t num_shards() const { return table_sharding ? deployed_devices.size() : 1; }
  size_t num_tables() const { return sparse_model_files.size(); }
};

struct ParameterServerConfig {
  bool support_int64_key = true;  // "supportlonglong" (backend.cpp:124-126)
  VolatileDatabaseParams volatile_db;
  PersistentDatabaseParams persistent_db;
  UpdateSourceParams update_source;
  std::vector<std::string> model_order;
  std::map<std::string, InferenceParams> models;
};

// Restatement of HPSBackend::ParseParameterServer (backend.cpp:102-526): same keys, same
// required/optional split, same defaults, same tolerant scalar conversion.  Unlike the reference it
// also validates list lengths against the table count (SURVEY.md App. C6/C9 hardening).
Status ParseParameterServerJson(const Json& root, ParameterServerConfig* out);
Status ParseParameterServerFile(const std::string& path, ParameterServerConfig* out);
Status ParseParameterServerText(const std::string& text, ParameterServerConfig* out);

// Typed, tolerant field readers = TritonJsonHelper::parse overloads (triton_helpers.cpp:42-442).
// `required`==true and key absent -> INVALID_ARG "The parameter '<key>' is mandatory...".
Status ParseField(bool& v, const Json& j, const char* key, bool required);
Status ParseField(double& v, const Json& j, const char* key, bool required);
Status ParseField(float& v, const Json& j, const char* key, bool required);
Status ParseField(int32_t& v, const Json& j, const char* key, bool required);
Status ParseField(int64_t& v, const Json& j, const char* key, bool required);
Status ParseField(size_t& v, const Json& j, const char* key, bool required);
Status ParseField(std::string& v, const Json& j, const char* key, bool required);
Status ParseField(DatabaseType& v, const Json& j, const char* key, bool required);
Status ParseField(DatabaseOverflowPolicy& v, const Json& j, const char* key, bool required);
Status ParseField(UpdateSourceType& v, const Json& j, const char* key, bool required);
Status ParseField(std::vector<float>& v, const Json& j, const char* key, bool required);
Status ParseField(std::vector<int32_t>& v, const Json& j, const char* key, bool required);
Status ParseField(std::vector<size_t>& v, const Json& j, const char* key, bool required);
Status ParseField(std::vector<std::string>& v, const Json& j, const char* key, bool required);

}  // namespace hps
