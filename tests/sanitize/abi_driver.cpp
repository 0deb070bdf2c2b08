// Native (no Python) stress driver over the engine C ABI (include/hps_amd.h), meant to be linked against a
// ThreadSanitizer or AddressSanitizer build of libhps_amd.so:
//   abi_driver <cpu|gpu|gpu_direct|gpu_sharded> [seconds]
// gpu_sharded: the model is table-sharded over three logical shards (ps.json "table_sharding": "hash") and every lookup thread owns an
// ENTRY session (hps_shard_entry_*: its own worker threads drive one lookup session per shard) instead of a lookup session.
// Three lookup threads (one session each) query random batches while a fourth thread reloads a table (same content)
// and, with a GPU cache, refreshes it.  Tables are synthetic (keys 0..R-1), every returned row is recomputed from the
// recipe in csrc/common/hps_hash.h and compared bit for bit; keys >= R must return the table's default value.
#include <hip/hip_runtime_api.h>
#include <hps_amd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "common/hps_hash.h"

static const uint64_t kSeed = 77;
static const int T = 3;
static const uint32_t kDims[T] = {64, 16, 3};
static const uint64_t R = 60000;
static const float kDefaults[T] = {0.5f, -2.0f, 7.0f};
static const size_t kBatch = 4096;

#define CK(call)                                                                                  \
  do {                                                                                            \
    if ((call) != 0) { fprintf(stderr, "%s failed: %s\n", #call, hps_last_error()); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "cpu";
  const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
  const bool gpu = mode != "cpu", direct = mode == "gpu_direct", sharded = mode == "gpu_sharded";
  if (gpu && hps_device_count() <= 0) { fprintf(stderr, "no HIP device\n"); return 2; }
  char json[2048];
  snprintf(json, sizeof json,
           "{\"supportlonglong\": true, \"volatile_db\": {\"type\": \"hash_map\", \"num_partitions\": 8}, \"models\": [{"
           "\"model\": \"m\", \"sparse_files\": [\"a\", \"b\", \"c\"], \"num_of_worker_buffer_in_pool\": 3,"
           "\"embedding_vecsize_per_table\": [64, 16, 3], \"maxnum_catfeature_query_per_table_per_sample\": [1, 1, 1],"
           "\"default_value_for_each_table\": [0.5, -2.0, 7.0], \"deployed_device_list\": [%s], \"max_batch_size\": %zu,"
           "\"gpucache\": %s, \"hit_rate_threshold\": 1.0, \"gpucacheper\": 0.1, \"ps_direct_access\": %s%s}]}",
           sharded ? "0, 0, 0" : "0", kBatch, gpu ? "true" : "false", direct ? "true" : "false",
           sharded ? ", \"table_sharding\": \"hash\", \"shard_capacity_factor\": 1.0" : "");
  hps_server_t* sv = nullptr;
  CK(hps_server_create_from_text(json, 0, &sv));
  for (int t = 0; t < T; ++t) CK(hps_server_load_table_synthetic(sv, "m", (uint32_t)t, kSeed, 0, R));
  hps_cache_t* cache = nullptr;
  if (gpu) {
    CK(hps_server_create_embedding_cache_per_model(sv, "m"));
    if (sharded) CK(hps_server_get_shard_cache(sv, "m", 1, &cache));   // (the monitor reads one shard's counters)
    else CK(hps_server_get_embedding_cache(sv, "m", 0, &cache));
  }
  std::atomic<bool> stop{false};
  std::atomic<long> bad{0}, calls{0};
  auto worker = [&](int id) {
    hps_session_t* s = nullptr;
    hps_shard_entry_t* e = nullptr;
    if (sharded) {
      if (hps_shard_entry_create(sv, "m", 0, &e) != 0) { bad.fetch_add(1); fprintf(stderr, "entry session: %s\n", hps_last_error()); return; }
      // round 6: one of the three entry sessions moves its rows by staged copies (owners gather pieces, copy engines ship them,
      // a kernel places them: csrc/cache/shard_entry.cpp ServeStaged), in small pieces so that several are in flight
      if (id == 1 && (hps_shard_entry_set_option(e, "transport", 1) != 0 || hps_shard_entry_set_option(e, "copy_piece_keys", 1024) != 0)) { bad.fetch_add(1); return; }
    } else if (hps_session_create(sv, "m", cache, &s) != 0) { bad.fetch_add(1); fprintf(stderr, "session: %s\n", hps_last_error()); return; }
    std::mt19937_64 rng(100 + id);
    std::vector<int64_t> keys(T * kBatch);
    size_t out_floats = 0;
    for (int t = 0; t < T; ++t) out_floats += kBatch * kDims[t];
    std::vector<float> host_out(out_floats);
    float* dev_out = nullptr;
    if (gpu && hipMalloc((void**)&dev_out, out_floats * sizeof(float)) != hipSuccess) { bad.fetch_add(1); return; }
    while (!stop.load()) {
      size_t n[T];
      const void* kp[T];
      float* vp[T];
      size_t ko = 0, vo = 0;
      for (int t = 0; t < T; ++t) {
        n[t] = rng() % (kBatch + 1);
        for (size_t i = 0; i < n[t]; ++i) {
          const uint64_t r = rng();
          keys[ko + i] = (r % 100 < 2) ? (int64_t)(R + r % 1000) : (int64_t)((r >> 8) % ((r & 1) ? R / 20 : R));
        }
        kp[t] = keys.data() + ko;
        vp[t] = (gpu ? dev_out : host_out.data()) + vo;
        ko += n[t];
        vo += n[t] * kDims[t];
      }
      if ((sharded ? hps_shard_entry_lookup(e, kp, vp, n, T) : hps_session_lookup(s, kp, vp, n, T)) != 0) {
        fprintf(stderr, "lookup: %s\n", hps_last_error());
        bad.fetch_add(1);
        break;
      }
      if (gpu && hipMemcpy(host_out.data(), dev_out, vo * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { bad.fetch_add(1); break; }
      ko = vo = 0;
      for (int t = 0; t < T; ++t) {
        const uint64_t tb = hps_synth_table_base(kSeed, (uint32_t)t);
        for (size_t i = 0; i < n[t]; i += 7) {   // sampled check
          const int64_t k = keys[ko + i];
          const float* got = host_out.data() + vo + i * kDims[t];
          for (uint32_t j = 0; j < kDims[t]; ++j) {
            uint32_t want;
            if ((uint64_t)k < R) want = hps_synth_elem_bits(hps_synth_row_base(tb, k), j);
            else memcpy(&want, &kDefaults[t], 4);
            uint32_t g;
            memcpy(&g, got + j, 4);
            if (g != want) { bad.fetch_add(1); break; }
          }
        }
        ko += n[t];
        vo += n[t] * kDims[t];
      }
      calls.fetch_add(1);
    }
    if (dev_out) (void)hipFree(dev_out);
    if (s) hps_session_destroy(s);
    if (e) hps_shard_entry_destroy(e);
  };
  std::vector<std::thread> th;
  for (int i = 0; i < 3; ++i) th.emplace_back(worker, i);
  std::thread churn([&] {
    int i = 0;
    while (!stop.load()) {
      std::this_thread::sleep_for(std::chrono::milliseconds(40));
      if (hps_server_load_table_synthetic(sv, "m", (uint32_t)(i % T), kSeed, 0, R) != 0) { bad.fetch_add(1); break; }
      // round 6: the default refresh (only what can differ: the table just reloaded), every third time the reference's full pass,
      // and an online update of a few rows (same contents: the recipe's rows) whose keys the next refreshes replay from the change log
      hps_refresh_stats_t rst;
      if (gpu && hps_server_refresh_embedding_cache_ex(sv, "m", 0, i % 3 == 2, &rst) != 0) { bad.fetch_add(1); break; }
      {
        const uint32_t t = (uint32_t)((i + 1) % T);
        std::vector<int64_t> uk(64);
        std::vector<float> ur((size_t)64 * kDims[t]);
        const uint64_t tb = hps_synth_table_base(kSeed, t);
        for (int q = 0; q < 64; ++q) {
          uk[(size_t)q] = (int64_t)(((uint64_t)i * 7919u + (uint64_t)q * 104729u) % R);
          for (uint32_t j = 0; j < kDims[t]; ++j) {
            const uint32_t bits = hps_synth_elem_bits(hps_synth_row_base(tb, uk[(size_t)q]), j);
            memcpy(&ur[(size_t)q * kDims[t] + j], &bits, 4);
          }
        }
        if (hps_server_upsert(sv, "m", t, uk.data(), ur.data(), 64) != 0) { bad.fetch_add(1); break; }
      }
      ++i;
    }
  });
  // a monitoring thread reads the cache's counters while the sessions run: since round 4 that call collects the statistics of
  // insert kernels the sessions left running behind their last calls (LookupSession::CollectDeferred, from a foreign thread)
  std::thread monitor([&] {
    uint64_t last = 0;
    while (!stop.load()) {
      std::this_thread::sleep_for(std::chrono::milliseconds(3));
      if (!cache) continue;
      hps_cache_counters_t c;
      if (hps_cache_counters(cache, &c) != 0) { fprintf(stderr, "counters: %s\n", hps_last_error()); bad.fetch_add(1); break; }
      if (c.inserted + c.refreshed + c.dropped < last) { fprintf(stderr, "counters went backwards\n"); bad.fetch_add(1); break; }
      last = c.inserted + c.refreshed + c.dropped;
    }
  });
  std::this_thread::sleep_for(std::chrono::milliseconds((long)(seconds * 1000)));
  stop.store(true);
  for (auto& t : th) t.join();
  churn.join();
  monitor.join();
  if (cache) { hps_cache_wait_async(cache); hps_cache_release(cache); }
  hps_server_destroy(sv);
  printf("abi_driver %s: %ld lookups, %ld bad -> %s\n", mode.c_str(), calls.load(), bad.load(), bad.load() ? "FAILED" : "ok");
  return bad.load() ? 1 : 0;
}
