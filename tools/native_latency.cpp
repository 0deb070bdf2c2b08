// Request latency of the engine C ABI (include/hps_amd.h) without an interpreter in the loop:
//   native_latency c1                 BASELINE config 1: one table 1,048,576 x 16, 4,096-key requests, host tier only
//   native_latency c4 [direct=1]      BASELINE config 4 shape: tables D = [1, 16], 1,024 samples x [2, 26] keys = 28,672
//                                     keys per request, GPU cache 20 %, ~99 % hit, host KEYS in, device OUTPUT0 out
// Build: clang++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tools/native_latency.cpp \
//        -Lhugectr_backend_amd/lib -lhps_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,... -o native_latency
#include <hip/hip_runtime_api.h>
#include <hps_amd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#define CK(call)                                                                                  \
  do {                                                                                            \
    if ((call) != 0) { fprintf(stderr, "%s failed: %s\n", #call, hps_last_error()); return 1; } \
  } while (0)

static double pct(std::vector<double>& v, double p) {
  std::sort(v.begin(), v.end());
  return v[(size_t)(p * (v.size() - 1))];
}

int main(int argc, char** argv) {
  const std::string which = argc > 1 ? argv[1] : "c1";
  const bool direct = argc > 2 ? atoi(argv[2]) != 0 : true;
  char json[2048];
  hps_server_t* sv = nullptr;
  hps_cache_t* cache = nullptr;
  hps_session_t* s = nullptr;
  std::mt19937_64 rng(1);
  std::vector<double> lat;
  if (which == "c1") {
    const uint64_t R = 1048576;
    const size_t n = 4096;
    snprintf(json, sizeof json,
             "{\"supportlonglong\": true, \"volatile_db\": {\"type\": \"hash_map\", \"num_partitions\": 8}, \"models\": [{"
             "\"model\": \"c1\", \"sparse_files\": [\"a\"], \"num_of_worker_buffer_in_pool\": 1, \"embedding_vecsize_per_table\": [16],"
             "\"maxnum_catfeature_query_per_table_per_sample\": [1], \"default_value_for_each_table\": [0.0],"
             "\"deployed_device_list\": [0], \"max_batch_size\": %zu, \"gpucache\": false}]}", n);
    CK(hps_server_create_from_text(json, 0, &sv));
    CK(hps_server_load_table_synthetic(sv, "c1", 0, 1, 0, R));
    CK(hps_session_create(sv, "c1", nullptr, &s));
    std::vector<int64_t> keys(n);
    std::vector<float> out(n * 16);
    for (int it = 0; it < 5200; ++it) {
      for (auto& k : keys) k = (int64_t)(rng() % R);
      const void* kp[1] = {keys.data()};
      float* vp[1] = {out.data()};
      size_t nk[1] = {n};
      const auto t0 = std::chrono::steady_clock::now();
      CK(hps_session_lookup(s, kp, vp, nk, 1));
      if (it >= 200) lat.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    double sum = 0;
    for (double v : lat) sum += v;
    printf("c1 (host tier, 1,048,576 x 16, %zu-key requests, one caller): p50 %.1f us  p99 %.1f us  mean %.1f us  = %.1f M lookups/s\n",
           n, pct(lat, 0.5), pct(lat, 0.99), sum / lat.size(), n / (sum / lat.size()));
  } else {
    const uint64_t R = 1000000;
    const size_t B = 1024, n0 = B * 2, n1 = B * 26;
    snprintf(json, sizeof json,
             "{\"supportlonglong\": true, \"volatile_db\": {\"type\": \"hash_map\", \"num_partitions\": 8}, \"models\": [{"
             "\"model\": \"c4\", \"sparse_files\": [\"a\", \"b\"], \"num_of_worker_buffer_in_pool\": 2, \"embedding_vecsize_per_table\": [1, 16],"
             "\"maxnum_catfeature_query_per_table_per_sample\": [2, 26], \"default_value_for_each_table\": [0.0, 0.0],"
             "\"deployed_device_list\": [0], \"max_batch_size\": %zu, \"gpucache\": true, \"gpucacheper\": 0.2, \"hit_rate_threshold\": 1.0,"
             "\"ps_direct_access\": %s}]}", B, direct ? "true" : "false");
    CK(hps_server_create_from_text(json, 0, &sv));
    for (uint32_t t = 0; t < 2; ++t) CK(hps_server_load_table_synthetic(sv, "c4", t, 1, 0, R));
    CK(hps_server_create_embedding_cache_per_model(sv, "c4"));
    CK(hps_server_get_embedding_cache(sv, "c4", 0, &cache));
    CK(hps_session_create(sv, "c4", cache, &s));
    std::vector<int64_t> keys(n0 + n1);
    float* d_out = nullptr;
    if (hipMalloc((void**)&d_out, (n0 * 1 + n1 * 16) * sizeof(float)) != hipSuccess) return 1;
    const uint64_t hot = (uint64_t)(0.2 * R) - 4096;
    for (double hit : {0.5, 0.9, 0.99}) {
      lat.clear();
      for (int it = 0; it < 1300; ++it) {
        for (auto& k : keys) {
          const uint64_t r = rng();
          k = ((r % 10000) < hit * 10000) ? (int64_t)((r >> 16) % hot) : (int64_t)(hot + 4096 + (r >> 16) % (R - hot - 4096));
        }
        const void* kp[2] = {keys.data(), keys.data() + n0};
        float* vp[2] = {d_out, d_out + n0};
        size_t nk[2] = {n0, n1};
        const auto t0 = std::chrono::steady_clock::now();
        CK(hps_session_lookup(s, kp, vp, nk, 2));
        if (it >= 300) lat.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
      }
      printf("c4 (%s tier, 28,672 keys/request, target hit %.2f, one session): p50 %.1f us  p99 %.1f us\n",
             direct ? "device-driven" : "host-gather", hit, pct(lat, 0.5), pct(lat, 0.99));
    }
    (void)hipFree(d_out);
  }
  hps_session_destroy(s);
  if (cache) hps_cache_release(cache);
  hps_server_destroy(sv);
  return 0;
}
