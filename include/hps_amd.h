/*
 * hps_amd.h — C ABI of the MI355X-native Hierarchical Parameter Server engine (libhps_amd.so).
 *
 * This is the drop-in boundary *under* the Triton shell: every entry point replaces one call the
 * reference backend makes into NVIDIA/HugeCTR's libhuge_ctr_hps.so (C++ ABI, not vendored in
 * /root/reference).  The reference call site each function stands for is cited (paths relative to
 * /root/reference/hps_backend).  The boundary *above* the shell — the seven TRITONBACKEND_* exports
 * of libtriton_hps.so — is declared in include/tritonbackend_hps.h.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a
 * TRITONSERVER_Error_Code-compatible positive code + 1 (see HPS_ERR_*), with the message available
 * from hps_last_error() on the calling thread.  No function throws across this boundary.
 * There is no CPU fallback for GPU-cache models: without a HIP device the cache constructors fail
 * with HPS_ERR_UNAVAILABLE.
 */
#ifndef HPS_AMD_H_
#define HPS_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPS_OK 0
#define HPS_ERR_UNKNOWN 1        /* TRITONSERVER_ERROR_UNKNOWN + 1 */
#define HPS_ERR_INTERNAL 2
#define HPS_ERR_NOT_FOUND 3
#define HPS_ERR_INVALID_ARG 4
#define HPS_ERR_UNAVAILABLE 5
#define HPS_ERR_UNSUPPORTED 6
#define HPS_ERR_ALREADY_EXISTS 7

typedef struct hps_server hps_server_t;   /* HugeCTR::HierParameterServerBase  include/backend.hpp:62 */
typedef struct hps_cache hps_cache_t;     /* HugeCTR::EmbeddingCacheBase       include/model_state.hpp:171-172 */
typedef struct hps_session hps_session_t; /* HugeCTR::LookupSessionBase        include/model_instance_state.hpp:117 */

/* Subset of HugeCTR::InferenceParams the shell reads (src/backend.cpp:390-516, src/model_state.cpp:278-366). */
typedef struct hps_model_info {
  uint64_t max_batch_size;                   /* "max_batch_size"                     backend.cpp:341-344 */
  uint32_t num_tables;                       /* len("sparse_files")                  backend.cpp:353-358 */
  int32_t use_gpu_embedding_cache;           /* "gpucache"                           backend.cpp:364-369 */
  float hit_rate_threshold;                  /* "hit_rate_threshold"                 backend.cpp:372-377 */
  float cache_size_percentage;               /* "gpucacheper"                        backend.cpp:380-385 */
  int32_t i64_input_key;                     /* "supportlonglong"                    backend.cpp:124-126,388 */
  int32_t number_of_worker_buffers_in_pool;  /* "num_of_worker_buffer_in_pool"       backend.cpp:397-402 */
  int32_t number_of_refresh_buffers_in_pool; /* "num_of_refresher_buffer_in_pool"    backend.cpp:404-409 */
  float cache_refresh_percentage_per_iteration; /*                                   backend.cpp:411-416 */
  int32_t device_id;                         /* last of "deployed_device_list"       backend.cpp:422 */
  uint32_t num_deployed_devices;
  float refresh_delay, refresh_interval;     /*                                      model_state.cpp:319,327 */
  uint64_t cat_num;                          /* sum maxnum_catfeature_query...       model_state.cpp:337-344 */
  uint64_t embedding_size;                   /* sum embedding_vecsize_per_table      model_state.cpp:352-356 */
} hps_model_info_t;

typedef struct hps_table_info {
  uint32_t embedding_vecsize;   /* embedding_vecsize_per_table[t]                      backend.cpp:454-460 */
  uint64_t maxnum_catfeature;   /* maxnum_catfeature_query_per_table_per_sample[t]     backend.cpp:443-452 */
  float default_value;          /* default_value_for_each_table[t]                     backend.cpp:427-433 */
  uint64_t rows_loaded;         /* rows currently in the host tier */
} hps_table_info_t;

typedef struct hps_cache_table_info {
  uint32_t embedding_vecsize;
  uint64_t num_buckets;     /* buckets of 14 keys (one 128-B line each) */
  uint64_t capacity_rows;   /* ceil(gpucacheper * rows) */
} hps_cache_table_info_t;

/* dropped: unique missed keys that were served but not cached — their bucket's 14 slots were all hit within the current
 * recency unit, or (admission, ps.json "gpucache_admission", default on) more recently than a new key's nominal age */
typedef struct hps_cache_counters {
  uint64_t lookups, keys, misses, unique_misses, inserted, refreshed, dropped, async_calls;
} hps_cache_counters_t;

typedef struct hps_lookup_stats {
  uint64_t misses;         /* keys of the last call that were not resident (as sent: duplicates count) */
  uint64_t unique_misses;  /* distinct (table, key) pairs among them */
  int32_t async_insert;    /* 1: answered in async-insert mode (missed keys returned the default vector) */
  float probe_gather_ms;   /* HIP-event time of the probe kernels of the call: tile dedup + probe, miss-unique
                              (option "timing"=1), else 0 */
  float phase_ms[4];       /* host wall clock of the last call: [0] until miss counts are known,
                              [1] host parameter-server gather (ps_direct_access: HIP-event time of the fetch kernel),
                              [2] H2D + scatter + insert, [3] whole call */
  float gpu_call_ms;       /* HIP-event span of the call on the session's stream, first kernel to last (option "timing"=1) */
  float hit_gather_ms;     /* HIP-event time of the hit-gather kernel (option "timing"=1) */
  uint64_t unique_keys;    /* distinct (table, key) pairs of the call — counted only when the insertion policy needs the
                              hit rate (0 < hit_rate_threshold < 1), else 0 */
  float key_stage_ms;      /* hps_session_lookup: host time spent staging the keys and enqueueing their H2D copies */
  float scatter_ms;        /* HIP-event time of the miss-scatter kernel (option "timing"=1; last staging chunk) */
  float insert_ms;         /* HIP-event time of the cache-insert kernel (option "timing"=1; last staging chunk) */
  int32_t keys_narrowed;   /* 1: the call's pageable keys all fitted 32 bits and crossed PCIe narrower than 8 bytes */
  int32_t key_bytes;       /* bytes per key that crossed PCIe in the last host-keys call: 8, 4 (uint32) or 3 (packed, keys < 2^24) */
  /* What the session's per-call switches steer by ("keys_by_kernel" 2, "probe_in_lane" 2, "interact_mode" 2): 1 while the
   * session's calls MISS MUCH.  A switch with hysteresis — up when a call's missed rows exceed side_scatter_mb, down when they
   * fall below three quarters of it, never sooner than 8 calls after the last change — so that traffic sitting on the bound does
   * not flip the arrangement call by call.  mode_flips: changes since the session was created. */
  int32_t miss_much_mode;
  int32_t interact_separate;   /* last hps_session_lookup_interact_device call: 1 = served as lookup + dense step, 0 = fused */
  uint64_t mode_flips;
} hps_lookup_stats_t;

const char* hps_last_error(void);
/* number of visible HIP devices (0 when none / no driver); never fails */
int hps_device_count(void);

/* ---- HierParameterServerBase ------------------------------------------------------------------ */
/* HierParameterServerBase::create(ps_json_config_file): parse ps.json, load every model's sparse files
 * into the host tier, build + warm GPU caches on the deployed devices.        src/backend.cpp:68-69 */
int hps_server_create(const char* ps_json_path, hps_server_t** out);
/* Same from JSON text.  load_tables=0 registers the models but loads nothing (tables are injected with
 * hps_server_load_table_*; caches are built with hps_server_create_embedding_cache_per_model). */
int hps_server_create_from_text(const char* ps_json_text, int load_tables, hps_server_t** out);
void hps_server_destroy(hps_server_t* server);

/* get_hps_model_configuration_map()                                          src/backend.cpp:70-71 */
int hps_server_model_count(hps_server_t* server);
const char* hps_server_model_name(hps_server_t* server, int index);
int hps_server_model_info(hps_server_t* server, const char* model, hps_model_info_t* out);
int hps_server_table_info(hps_server_t* server, const char* model, uint32_t table, hps_table_info_t* out);
int hps_server_deployed_device(hps_server_t* server, const char* model, uint32_t index, int32_t* device);
/* Re-read ps.json and (re)register the models found in it — HPSBackend::ParseParameterServer used for
 * online deployment of a new model/version.                                  src/hps.cc:210-219 */
int hps_server_parse_config(hps_server_t* server, const char* ps_json_path);

/* update_database_per_model(InferenceParams)                                 src/model_state.cpp:132,389 */
int hps_server_update_database_per_model(hps_server_t* server, const char* model);
/* create_embedding_cache_per_model(InferenceParams)                          src/model_state.cpp:391-392 */
int hps_server_create_embedding_cache_per_model(hps_server_t* server, const char* model);
/* destory_embedding_cache_per_model(name)  [sic]                             src/model_state.cpp:111 */
int hps_server_destroy_embedding_cache_per_model(hps_server_t* server, const char* model);
/* refresh_embedding_cache(model, device)                                     src/model_state.cpp:135,160 */
int hps_server_refresh_embedding_cache(hps_server_t* server, const char* model, int32_t device);
/* The same with a choice and an account of what it did.  By default a refresh re-reads only rows that CAN differ from the host
 * tier's: nothing for a table that was neither reloaded nor updated since the cache last looked, the resident ones among the
 * updated keys otherwise (ps.json "gpucache_refresh_changed_only": false, or full != 0 here: every resident row of every table,
 * as the reference does: docs/hierarchical_parameter_server.md:234-238).  While lookup sessions are serving, the refresher works in
 * pieces of 32,768 rows with a duty cycle of "gpucache_refresh_link_share" (default 0.15: after a piece that took t it pauses
 * for 5.7 t) — the link, the serving pool and the cache's writer windows are the sessions' for the rest of the time. */
typedef struct hps_refresh_stats {
  uint64_t tables, tables_unchanged, tables_full;   /* tables looked at / skipped as unchanged / refreshed row by row in full */
  uint64_t keys_dumped;                             /* resident keys read back from the GPU for the full passes */
  uint64_t keys_changed;                            /* change-log entries examined */
  uint64_t rows_refreshed, row_bytes;               /* rows re-read from the parameter server and uploaded, their bytes */
  double seconds;
} hps_refresh_stats_t;
int hps_server_refresh_embedding_cache_ex(hps_server_t* server, const char* model, int32_t device, int32_t full, hps_refresh_stats_t* out);
/* get_embedding_cache(model, device): *out = NULL (and HPS_OK) when there is none (unknown model, or a GPU-cache model
 * without a cache on that device), like the reference's nullptr.  A model that runs without GPU cache gets a handle too —
 * the reference hands out a cache object with use_gpu_embedding_cache = false there — so that the shell's
 * get_cache_config().num_emb_table_ and LookupSessionBase::create(params, cache) work for it.
 *                                                                            src/model_state.cpp:379,411 */
int hps_server_get_embedding_cache(hps_server_t* server, const char* model, int32_t device, hps_cache_t** out);

/* Table injection without files (tests / bench).  rows: R x D fp32, keys: R int64. */
int hps_server_load_table_arrays(hps_server_t* server, const char* model, uint32_t table, const int64_t* keys,
                                 const float* rows, uint64_t R, int borrow);
/* keys key0..key0+R-1, rows from the synthetic recipe of SURVEY.md §8d, generated in parallel. */
int hps_server_load_table_synthetic(hps_server_t* server, const char* model, uint32_t table, uint64_t seed,
                                    int64_t key0, uint64_t R);
/* Same, keeping only the keys owned by `shard` of `num_shards` (owner = mix64(key) mod num_shards, the routing
 * function of the sharded lookup, hps_shard_bucket_device): one rank's slice of a model-parallel table. */
int hps_server_load_table_synthetic_shard(hps_server_t* server, const char* model, uint32_t table, uint64_t seed,
                                          int64_t key0, uint64_t R, uint32_t shard, uint32_t num_shards);
/* Host-tier fetch of one table (the volatile-database lookup): out[i*D..] = row or default; found optional. */
int hps_server_fetch(hps_server_t* server, const char* model, uint32_t table, const int64_t* keys, uint64_t n,
                     float* out, uint8_t* found);

/* Read-only view of one table of the host tier as it sits in memory: R keys and R x D rows in file order (a key repeated
 * in the file appears more than once; the last one is the live row).  Valid until the table is reloaded or updated.
 * For tests and benchmarks that check or time against the very same rows without a second copy of a 133-GB model. */
int hps_server_table_data(hps_server_t* server, const char* model, uint32_t table, const int64_t** keys,
                          const float** rows, uint64_t* num_rows);

/* Online update of the host tier: insert-or-overwrite rows (duplicate keys: last wins).  The entry point a consumer of
 * the reference's update source would call per message batch (docs/architecture.md:104-180); GPU caches pick the
 * new rows up at their next refresh.  With a persistent database the rows are written through to the row store
 * unless persistent_db.read_only. */
int hps_server_upsert(hps_server_t* server, const char* model, uint32_t table, const int64_t* keys, const float* rows,
                      uint64_t n);

/* ---- online update source (ps.json "update_source"; reference: backend.cpp:262-308, docs/hierarchical_parameter_server.md:
 * 575-646).  "type": "file_tail" + "brokers": "<path>" starts a consumer thread that follows an append-only file of framed
 * update messages (csrc/ps/update_source.h), dispatches them in chunks of at most max_batch_size keys to the host tier and the
 * persistent store (as hps_server_upsert does), commits after at most max_commit_interval messages or poll_timeout_ms without
 * news — remembering its position in <path>.offset — and replaces the rows of updated keys that are resident in the model's
 * GPU caches.  "kafka_message_queue" is refused at start-up (no Kafka client in this build).
 *   hps_update_message_encode   one message frame for a producer to append (out = NULL: only the size)
 *   hps_server_update_source_stats   out6 = messages, keys, dispatches, commits, dispatch failures, rejected messages
 *   hps_server_update_source_drain   returns once the source has been found empty twice in a row (tests, tools); an error when
 *                                    the source is unreadable for good (a frame that is not a frame)
 *   hps_server_update_source_stop    stops the consumer thread (orderly shutdown): what was applied since the last commit is
 *                                    delivered to the GPU caches and committed, nothing more is applied; messages not applied
 *                                    yet stay uncommitted and are replayed by the next server; the statistics calls then
 *                                    report HPS_ERR_UNAVAILABLE */
int hps_update_message_encode(const char* model, uint32_t table, uint32_t dim, const int64_t* keys, const float* rows, uint64_t n,
                              void* out, uint64_t out_capacity, uint64_t* out_bytes);
int hps_server_update_source_stats(hps_server_t* server, uint64_t* out6);
/* update messages that no update_filters entry selected (volatile_db / persistent_db "update_filters": regular expressions over
 * "hps_<model>.<table name>"): skipped silently like a message on a topic nobody subscribed to — counted here, not as failures */
int hps_server_update_source_filtered(hps_server_t* server, uint64_t* out);
int hps_server_update_source_drain(hps_server_t* server, uint32_t timeout_ms);
int hps_server_update_source_stop(hps_server_t* server);

/* Host tier smaller than the table (volatile_db.overflow_margin / overflow_policy / overflow_resolution_target /
 * initial_cache_rate / cache_missed_embeddings, persistent_db.*;  docs/hierarchical_parameter_server.md:460-569). */
typedef struct hps_host_tier_stats {
  uint64_t tiered;                 /* 0: whole table in RAM (every other field except persistent_rows is 0) */
  uint64_t persistent_rows;        /* rows of the row store behind the volatile tier */
  uint64_t entries, capacity;      /* volatile tier: embeddings held now / at most (sum over partitions) */
  uint64_t max_partition_entries;  /* largest partition now (never above overflow_margin) */
  uint64_t lookups, hits;          /* volatile tier */
  uint64_t persistent_hits;        /* served from the persistent database behind it */
  uint64_t not_found;              /* answered with the default vector */
  uint64_t inserts, evictions, overflows;
} hps_host_tier_stats_t;
int hps_server_host_tier_stats(hps_server_t* server, const char* model, uint32_t table, hps_host_tier_stats_t* out);
/* Keys the volatile tier holds, ascending; *n = their number (also when cap is too small: then nothing is written). */
int hps_server_host_tier_keys(hps_server_t* server, const char* model, uint32_t table, int64_t* out, uint64_t cap,
                              uint64_t* n);

/* ---- EmbeddingCacheBase ----------------------------------------------------------------------- */
/* get_cache_config().num_emb_table_                                src/model_instance_state.cpp:107-109,169 */
int hps_cache_num_tables(hps_cache_t* cache);
/* 1: the handle stands for device tables; 0: the model runs without GPU cache (lookups go to the host tier) */
int hps_cache_on_device(hps_cache_t* cache);
int hps_cache_table_info(hps_cache_t* cache, uint32_t table, hps_cache_table_info_t* out);
int hps_cache_counters(hps_cache_t* cache, hps_cache_counters_t* out);
/* rows the refreshes of this cache have uploaded so far — live, piece by piece (hps_cache_counters' `refreshed` moves only when a
 * slice of cache_refresh_percentage_per_iteration is through): what a monitor divides by time for the refresh rate */
uint64_t hps_cache_refresh_rows_uploaded(hps_cache_t* cache);
/* residency probe without side effects: slots[i] = slot index or -1 */
int hps_cache_query(hps_cache_t* cache, uint32_t table, const int64_t* h_keys, uint64_t n, int32_t* h_slots);
/* wait for queued async insertions */
int hps_cache_wait_async(hps_cache_t* cache);
/* The SDMA copy engines of a device take one tiny copy each (both directions), once per process and device; cache creation
 * does this on its own while the model loads (csrc/cache/copy_engines.h: otherwise the HIP runtime creates an engine's queue
 * inside some request's hipMemcpyAsync, 7-12 ms during which every HIP call of the process waits).  Writes a one-line report
 * ("16 engines host->device, 16 device->host, 140 ms") into buf; returns the number of engines that took a copy. */
int hps_wake_copy_engines(int device, char* buf, uint64_t cap);
/* The NUMA node the host tier's worker pools are bound to, -1 when they are not (csrc/ps/thread_pool.h).  The first server of the
 * process decides, before the pools start: the node of the deployed GPUs when they hang off one (the pools are then sized
 * for that node's CPUs), none for a deployment without GPU caches, when the GPUs span nodes or the machine has one node; environment HPS_NUMA_NODE=<n> names
 * the node, HPS_NUMA_NODE=off switches the binding off.  Threads of the caller (Triton's instance threads) are never touched. */
int hps_pool_numa_node(void);
/* Fork-joins of the host tier's lock-free pool path in which a task ran more than once since the process started: 0 unless
 * the slot-reuse race round 5 fixed is back (csrc/ps/thread_pool.h; soak and stress drivers assert it). */
uint64_t hps_pool_fast_overruns(void);
/* The calling thread joins the worker pools' node — for the application's threads that drive lookups (libtriton_hps.so calls it for
 * every Triton instance thread at its first request).  A thread whose affinity mask already lies inside one NUMA node is left as
 * it is.  Returns 1 when the thread's affinity was changed, 0 otherwise (pools not bound, thread already placed). */
int hps_bind_calling_thread(void);
/* drop this handle's reference (the shared_ptr copy the shell holds)         src/model_instance_state.cpp:158 */
void hps_cache_release(hps_cache_t* cache);

/* ---- LookupSessionBase ------------------------------------------------------------------------ */
/* LookupSessionBase::create(InferenceParams, embedding_cache)                src/model_instance_state.cpp:170-171
 * cache may be NULL for gpucache=false models. */
int hps_session_create(hps_server_t* server, const char* model, hps_cache_t* cache, hps_session_t** out);
/* The same with the reference's two arguments: the cache handle knows its parameter server and model (InferenceParams
 * are the server's for that model).  LookupSessionBase::create(instance_params_, embedding_cache)
 *                                                                            src/model_instance_state.cpp:170-171 */
int hps_session_create_from_cache(hps_cache_t* cache, hps_session_t** out);
void hps_session_destroy(hps_session_t* session);
/* lookup(h_keys_per_table, d_vectors_per_table, num_keys_per_table): host key pointers in; device
 * (gpucache) or host (gpucache=false) vector pointers out; blocking.        src/model_instance_state.cpp:194-195
 * spec docs/architecture.md:308-323 */
int hps_session_lookup(hps_session_t* session, const void* const* h_keys_per_table, float* const* vectors_per_table,
                       const size_t* num_keys_per_table, size_t num_tables);
/* Same lookup with KEYS already resident in HBM (flat, table-major int64).  Not in the reference API: it
 * removes the host->device key copy the reference pays inside lookup() when the caller (a Triton
 * ensemble step, the benchmark) already holds the keys on the device. */
int hps_session_lookup_device(hps_session_t* session, const int64_t* d_keys_flat, float* const* d_vectors_per_table,
                              const size_t* num_keys_per_table, size_t num_tables);
int hps_session_last_stats(hps_session_t* session, hps_lookup_stats_t* out);
/* options: "timing" (0/1), "hit_rate_threshold_permille" (per-session override of the model's hit_rate_threshold:
 * 1000 = always synchronous insertion, 0 = always asynchronous), "host_gather" (0/1: on a ps_direct_access cache, serve
 * this session's misses the reference's way — host threads gather, hipMemcpyAsync ships — e.g. to compare the two tiers
 * on one deployment), "split_probe" (0/1: read the miss counts back before / after the hit gather), and the kernel A/B
 * switches "probe_variant" (1002 default; 1102: no tile-local input dedup), "xcd_walk" (0/1), "exclusive_kernels" (0/1), "fused_unique" (0/1: the
 * call-wide unique misses are found in the probe kernel's tail, default 1), "keys_pinned_check" (0/1: DMA flat
 * page-locked key arrays in place instead of staging them), "narrow_keys" (pageable keys cross PCIe at the width the
 * request needs — 0: always 8 bytes; 1 (default): 3 bytes each when every key is in [0, 2^24), uint32 when in [0, 2^32);
 * 2: uint32 only.  A width that fails is not tried again for 256 calls, doubling with every failure in a row up to
 * 65,536), "defer_insert" (0/1, default 1: a synchronous call returns when its rows are complete; the cache-insert kernel
 * of its missed rows stays enqueued behind it, ordered before every later reader of the cache by the cache's writer
 * event — hps_cache_counters and hps_cache_query see it done; 0: the call also waits for the insert kernel),
 * "in_place_kb" (default 1024: missed rows of a call up to this many KB are read by the scatter and insert kernels out of
 * the page-locked buffer the host gathered them into, next to the hit gather, instead of being uploaded first),
 * "side_scatter_mb" (default 16: missed rows up to this many MB are uploaded and scattered on the session's second stream,
 * without a turn in the kernel lane; beyond it the scatter takes its turn behind a drained stream),
 * "keys_by_kernel" (default 2: the staged keys of a big request are read out of the page-locked staging buffer by a kernel
 * while this session's calls miss much — last call's missed rows > side_scatter_mb — so that they do not queue behind the other
 * session's row copies, and go up as copy-engine copies otherwise; 1: always by kernel; 0: never),
 * "probe_in_lane" (default 2: the probe kernel runs next to another session's hit gather while this session's calls miss
 * little — last call's missed rows <= side_scatter_mb — and takes its turn in the kernel lane otherwise; 1: always in the
 * lane; 0: never), "interact_mode" (hps_session_lookup_interact_device — default 2: the fused arrangement while the session's
 * calls miss little, lookup into a buffer of the session + the dense step's kernels while they miss much; 1: always fused;
 * 0: always the separate steps) */
int hps_session_set_option(hps_session_t* session, const char* name, int value);

/* ---- table sharding across GPUs (BASELINE config 3; not in the reference, which is replicas-only) ------------
 * One table, rows partitioned by owner(key) = mix64(key) mod num_shards, one process per GPU.  The native sharded
 * session drives the whole exchange itself — RCCL (ncclSend/ncclRecv groups over xGMI) on the lookup session's stream,
 * fixed-capacity blocks, no count exchange and no device->host read-back between the two exchanges (csrc/cache/shard_session.h):
 *
 *   hps_shard_unique_id(id)                     on rank 0; distribute the 128 bytes to the other ranks (any side channel)
 *   hps_shard_session_create(session, rank, world, id, max_local_keys, &shard)     collective (ncclCommInitRank)
 *   hps_shard_session_lookup(shard, d_keys, n, d_out)                              collective, blocking
 * Every rank passes the same max_local_keys (verified at the first lookup).  The session handle may be destroyed before
 * the sharded session (the engine keeps what it needs alive).  A rank that fails inside a collective call aborts its side
 * of the transport so that in-process peers return an error instead of waiting; RCCL peers already inside a send/recv
 * kernel cannot be reached — bound the call with a watchdog.  The key INT64_MIN (the cache's reserved value) is in no
 * table: it is answered with the default vector without travelling.
 *
 * `session` is the lookup session of a ONE-table GPU-cache model that holds this rank's shard
 * (hps_server_load_table_synthetic_shard, or files holding only the rank's rows); its request capacity
 * (max_batch_size x keys per sample) divided by world bounds the keys one rank may receive from one peer.
 * hps_shard_group_* / hps_shard_session_create_local: the same session with `world` endpoints inside one process on one
 * device (device-to-device copies instead of RCCL) — for tests and single-process deployments; every endpoint runs on
 * its own thread. */
typedef struct hps_shard_session hps_shard_session_t;
typedef struct hps_shard_group hps_shard_group_t;
int hps_shard_unique_id(uint8_t* out128);
int hps_shard_session_create(hps_session_t* session, uint32_t rank, uint32_t world, const uint8_t* unique_id128,
                             uint64_t max_local_keys, hps_shard_session_t** out);
int hps_shard_group_create_local(uint32_t world, hps_shard_group_t** out);
void hps_shard_group_destroy(hps_shard_group_t* group);
int hps_shard_session_create_local(hps_session_t* session, hps_shard_group_t* group, uint32_t rank, uint64_t max_local_keys,
                                   hps_shard_session_t** out);
/* d_keys: n int64 on the session's device (this rank's keys); d_out: n x D fp32 rows in input order. */
int hps_shard_session_lookup(hps_shard_session_t* shard, const int64_t* d_keys, uint64_t n, float* d_out);
/* The reference's contract for a lookup (keys in HOST memory, docs/architecture.md:308-323): staged through page-locked memory,
 * as uint32 when every key of the request fits 32 bits, else 8 bytes each. */
int hps_shard_session_lookup_host(hps_shard_session_t* shard, const int64_t* h_keys, uint64_t n, float* d_out);
/* last call, last attempt: time of the key exchange, of the local lookup and of the row exchange (HIP events on the session's
 * stream), keys this rank's shard was asked for, bytes per key of a host request over PCIe (8 for device keys) */
int hps_shard_session_last_timing(hps_shard_session_t* shard, float* keys_exchange_ms, float* lookup_ms, float* rows_exchange_ms,
                                  uint64_t* keys_received, int32_t* key_bytes);
/* last call: keys per exchange block, attempts (2 = a block overflowed and the call was repeated with the capacity that was
 * needed — every rank learns the largest block any rank wanted from the block headers), keys sent to each rank */
int hps_shard_session_last_stats(hps_shard_session_t* shard, uint64_t* capacity, uint32_t* attempts, uint64_t* sent_per_rank,
                                 uint32_t world);
void hps_shard_session_destroy(hps_shard_session_t* shard);
/* The device-side pieces on their own (used by the host-tier variant in hugectr_backend_amd/sharded.py): */
uint32_t hps_shard_owner(int64_t key, uint32_t num_shards);
uint64_t hps_shard_bucket_workspace_bytes(uint64_t n, uint32_t num_shards);
/* Stable bucket of n device keys by owner: d_keys_sorted grouped shard 0..P-1, d_perm[j] = input index of sorted key j,
 * d_totals[P] = keys per shard (device, uint64).  `stream` is a hipStream_t (0 = default stream). */
int hps_shard_bucket_device(const int64_t* d_keys, uint64_t n, uint32_t num_shards, int64_t* d_keys_sorted,
                            int32_t* d_perm, uint64_t* d_totals, void* d_workspace, void* stream);
/* d_out[d_perm[j]*dim ..] = d_rows[j*dim ..]  for j in [0,n) */
int hps_shard_unpermute_device(const float* d_rows, const int32_t* d_perm, uint64_t n, uint32_t dim, float* d_out,
                               void* stream);

/* ---- table sharding behind ONE instance (BASELINE config 3 through the reference's own boundary) -------------------------
 * ps.json model key "table_sharding": "hash": entry s of deployed_device_list is SHARD s — its GPU cache holds gpucacheper
 * of the keys with mix64(key) mod P == s (P = the length of the list; a device may be listed more than once: logical
 * shards); the host tier stays whole.  An ENTRY session on any listed device serves whole requests the way
 * TRITONBACKEND_ModelInstanceExecute gets them (one request, one instance, blocking: src/hps.cc:353-369, 406): it buckets
 * the keys by owner on its device, drives one lookup session per shard on the shard's device from its own threads, and the
 * shards' gather / scatter kernels store the rows straight into the entry device's output over peer mappings (xGMI).  No
 * collective, no lock-step between instances (csrc/cache/shard_entry.h).  libtriton_hps.so creates one entry session per
 * model instance of such a model.  Optional keys: "shard_capacity_factor" (default 2.0: a shard session holds that many
 * times its fair share of a full request; more is served in several passes), "shard_dedup" (default true: a key the request
 * repeats travels to its owner once), "shard_transport" ("peer_store", the default: as above | "staged_copy": every owner
 * gathers pieces of at most "shard_copy_piece_keys" keys (default 0 = automatic: 131,072 for a shard on another GPU, one piece for a shard on the entry GPU itself) into local blocks with ordinary lookups, copy engines ship
 * the blocks into the entry device's receive buffer — hipMemcpyPeerAsync, SDMA over xGMI — while the next piece is gathered,
 * and a kernel on the entry device puts the delivered rows into OUTPUT0; the bucket keys travel by copy too, so no kernel
 * touches another device's memory and peer access is not needed).  Same rows either way. */
typedef struct hps_shard_entry hps_shard_entry_t;
typedef struct hps_shard_entry_stats {
  uint64_t keys, unique_keys;          /* last request: keys as sent / keys that travelled (dedup_level 2: the distinct (table, key) pairs) */
  uint64_t misses, unique_misses;      /* summed over the shards' lookups */
  float bucket_ms, lookup_ms, expand_ms, key_stage_ms;   /* wall clock of the phases of the last request */
  uint32_t num_shards;
  uint32_t key_bytes;                  /* host keys: bytes per key that crossed PCIe (8; 4 / 3 = offsets from the table's smallest key) */
  uint64_t sent[64];                   /* keys each shard was asked for */
  uint32_t passes[64];                 /* lookup calls per shard */
  float shard_ms[64];                  /* wall time of each shard's lookups */
  uint32_t dedup_level;                /* input dedup of the last request: 0 none, 1 within tiles of 1,024 keys, 2 call-wide */
  uint32_t transport;                  /* how the rows reached the entry GPU: 0 peer_store, 1 staged_copy */
  uint64_t copied_bytes;               /* staged_copy: row bytes the copy engines shipped into the entry GPU */
  float copy_wait_ms[64];              /* staged_copy: time each shard's worker waited for its copies to land */
  uint64_t dedup_flips;                /* adaptive dedup: changes of level since the session was created (a level is kept for at least 8 requests) */
} hps_shard_entry_stats_t;
/* the cache of shard s of a table-sharded model (for residency queries, counters); *out = NULL when there is none */
int hps_server_get_shard_cache(hps_server_t* server, const char* model, uint32_t shard, hps_cache_t** out);
int hps_shard_entry_create(hps_server_t* server, const char* model, int32_t entry_device, hps_shard_entry_t** out);
void hps_shard_entry_destroy(hps_shard_entry_t* entry);
/* same contracts as hps_session_lookup / hps_session_lookup_device; the vectors are on the entry device */
int hps_shard_entry_lookup(hps_shard_entry_t* entry, const void* const* h_keys_per_table, float* const* d_vectors_per_table,
                           const size_t* num_keys_per_table, size_t num_tables);
int hps_shard_entry_lookup_device(hps_shard_entry_t* entry, const int64_t* d_keys_flat, float* const* d_vectors_per_table,
                                  const size_t* num_keys_per_table, size_t num_tables);
int hps_shard_entry_last_stats(hps_shard_entry_t* entry, hps_shard_entry_stats_t* out);
/* options: "dedup" (0: every key travels as sent; 1, the model's default with shard_dedup: adaptive — repeats found within tiles
 * of 1,024 keys and call-wide, the call-wide level skipped for 31 requests after a big request of which more than 90 %
 * travelled anyway; 2: always both levels), "timing" (0/1: forwarded to the shard sessions), "transport" (0 peer_store,
 * 1 staged_copy: from the next request on; 0 is refused where peer access is not available), "copy_piece_keys" (staged_copy:
 * keys per piece, >= 1024, or 0 = automatic) */
int hps_shard_entry_set_option(hps_shard_entry_t* entry, const char* name, int value);
/* keys a shard session of this entry holds per call */
uint64_t hps_shard_entry_shard_capacity(hps_shard_entry_t* entry);
/* Pure host logic, no GPU: the passes that serve one owner's bucket of counts[t] keys per table with a session of `capacity`
 * keys.  Writes up to max_passes rows of (1 + num_tables) uint64 — [offset, n_0 .. n_{T-1}] — and returns the number of
 * passes needed (which may exceed max_passes). */
uint64_t hps_shard_plan_passes(const uint32_t* counts, uint32_t num_tables, uint64_t capacity, uint64_t* out, uint64_t max_passes);

/* First contact with a multi-GPU machine, before the table-sharded legs are trusted with it (csrc/cache/multi_gpu_probe.h): for
 * every ordered pair of the `n` DISTINCT devices — may a kernel on `from` address memory of `to` (hipDeviceCanAccessPeer), does a
 * 4-KB peer store arrive intact, GB/s of kernel stores over the peer mapping (the peer_store transport's mechanism) and of
 * hipMemcpyPeerAsync (staged_copy's), probe_bytes per transfer; then (with_rccl != 0) one RCCL all-reduce of one word with one
 * rank per device.  Runs behind a deadline on a thread of its own: a step that does not come back within timeout_ms is named in
 * the report ("timeout": true, "stuck_in": "...") and the call returns HPS_ERR_UNAVAILABLE — its thread stays behind.  Writes one
 * JSON object into buf (truncated to cap - 1 bytes). */
int hps_multi_gpu_selftest(const int32_t* devices, uint32_t n, uint64_t probe_bytes, uint32_t timeout_ms, int32_t with_rccl,
                           char* buf, uint64_t cap);

/* ---- dense step of BASELINE config 5: DLRM bottom MLP + pairwise dot interaction, consuming OUTPUT0 in place ------
 * Not in the reference backend: there the dense model is another Triton backend reached through an ensemble
 * hand-off (samples/hps-triton-ensemble); here it runs on the GPU that holds the lookup's output.
 * fp16 operands, fp32 accumulation (MFMA).  Output row of sample i (f16, stride hps_dense_out_stride elements):
 *   [ bottom(i)[0..D) | dot(z_a, z_b) for a = 1..T, b = 0..a-1 | zero padding ],  z_0 = bottom(i), z_{1+t} = row(t,i)
 * where bottom = ReLU(..ReLU(x W_0 + b_0)..W_{L-1} + b_{L-1}) and row(t,i) is OUTPUT0[(t*batch + i)*D ..]. */
typedef struct hps_dense hps_dense_t;
/* weights[l]: host fp32 [K_l][layer_dims[l]] row-major (K_0 = num_dense, K_l = layer_dims[l-1]); biases[l]: [layer_dims[l]].
 * layer widths: multiples of 32, <= 512; last width == emb_dim; emb_dim multiple of 16; num_tables <= 31. */
int hps_dense_create(int device, uint32_t num_dense, uint32_t num_layers, const uint32_t* layer_dims,
                     const float* const* weights, const float* const* biases, uint32_t num_tables, uint32_t emb_dim,
                     hps_dense_t** out);
void hps_dense_destroy(hps_dense_t* dense);
uint32_t hps_dense_out_dim(const hps_dense_t* dense);     /* emb_dim + (T+1)T/2 */
uint32_t hps_dense_out_stride(const hps_dense_t* dense);  /* out_dim rounded up to a multiple of 8 elements */
/* d_dense: [batch][num_dense] fp32; d_embeddings: table-major [T][batch][emb_dim] fp32 (the lookup's OUTPUT0);
 * d_out_f16: [batch][out_stride] f16.  All device pointers; enqueued on `stream` (a hipStream_t), not synchronised. */
int hps_dense_forward(hps_dense_t* dense, const float* d_dense, const float* d_embeddings, uint64_t batch,
                      void* d_out_f16, void* stream);
/* Lookup fused into the dense step: `batch` samples, one key per table per sample (d_keys_flat: table-major,
 * T*batch int64 on the device); the interaction reads each embedding row from the cache slot the probe found or
 * from the miss staging, OUTPUT0 is never materialised; the missed rows are inserted before the call returns.
 * Same output as hps_session_lookup_device + hps_dense_forward.  Needs a ps_direct_access model with
 * hit_rate_threshold 1.0 whose tables are all as wide as the dense step's embeddings.  Blocking. */
int hps_session_lookup_interact_device(hps_session_t* session, hps_dense_t* dense, const int64_t* d_keys_flat, uint64_t batch,
                                       const float* d_dense, void* d_out_f16);

#ifdef __cplusplus
}
#endif
#endif /* HPS_AMD_H_ */
