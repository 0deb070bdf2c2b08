// Native load generator for the Triton plugin boundary — what `perf_analyzer` is to the reference
// (/root/reference/.gitlab-ci.yml:70 is the reference's only performance smoke: perf_analyzer against the hps backend).
//
// Plays tritonserver through the mock core (csrc/mock_triton, TEST INFRASTRUCTURE): dlopens libtriton_mock_core.so,
// which dlopens libtriton_hps.so and calls TRITONBACKEND_Initialize / ModelInitialize / ModelInstanceInitialize; then
// `instances` threads issue TRITONBACKEND_ModelInstanceExecute back to back, each request = one Criteo-shaped batch:
//   KEYS     int64 [1, T*B]  host memory (pageable by default: the reference memcpy's them, hps.cc:586-597)
//   NUMKEYS  int32 [1, T]
//   OUTPUT0  fp32  [T*B*D]   device memory handed out by the core (TRITONBACKEND_OutputBuffer), as for a GPU instance
// The model's tables come from the "synthetic://<rows>" source (csrc/cache/parameter_server.cpp), so the 26 x 10 M x 128
// model loads through the plugin boundary without 133 GB of files.
//
// Timing follows bench.py: `blocks` timed blocks of `steps` requests (all instances together), the median block is the
// result; p50/p99 over every timed request.  One response per model is checked against the table recipe at the end.
// Prints ONE JSON line.
//
// The other BASELINE configurations run through the same driver:
//   config 1 (.gitlab-ci.yml:70: perf_analyzer against a CPU-only deployment):
//       --tables 1 --rows 1048576 --dims 16 --batch 4096 --gpucache 0 --uniform 1 --instances 1
//       gpucache=false model, KIND_CPU instance, OUTPUT0 in host memory, keys uniform over the table
//   config 4 (README.md:148-152: W&D, D = [1,16], keys per sample [2,26]; two models served side by side on one GPU):
//       --models 2 --dims 1,16 --per-sample 2,26 --rows 1000000 --batch 1024 --instances 1 --hit 0.9
//   dynamic batching (Triton hands several queued requests to ONE TRITONBACKEND_ModelInstanceExecute call; the reference then runs one
//   blocking lookup per request, hps.cc:406; this build serves them with one engine call):
//       --requests-per-execute 8   every Execute call carries 8 requests of --batch samples each; the model's max_batch_size is
//       8 x --batch (what a dynamic batcher needs to merge them); a "step" is then one Execute call
// --dims / --per-sample are comma lists (one entry per table; --tables N --dim D is the short form of N equal tables);
// --models M deploys the model M times (own tables: seed + model index) and drives all of them concurrently.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../hugectr_backend_amd/csrc/common/hps_hash.h"
#include "../hugectr_backend_amd/csrc/mock_triton/mock_core.h"

namespace {

struct Args {
  std::string lib_dir = "hugectr_backend_amd/lib";
  int tables = 26, dim = 128, instances = 2, steps = 20, warmup = 5, blocks = 10, direct = 0, pinned_keys = 0, also_pinned = 0;
  int models = 1, gpucache = 1, uniform = 0;
  int rpe = 1;      // requests per TRITONBACKEND_ModelInstanceExecute call
  int shards = 0;   // > 0: ONE table-sharded model (ps.json "table_sharding": "hash"): shard s on device s % (visible devices), instance i on device i % (those)
  std::string dims, per_sample;   // comma lists; empty: `tables` tables of `dim` floats, one key per sample
  long rows = 10000000, batch = 65536;
  double cache_frac = 0.2, hit = 0.957, zipf = 1.05, threshold = 1.0;
};

#define API(name) decltype(&::name) name = nullptr
struct Mock {
  API(mock_last_error); API(mock_server_create); API(mock_server_destroy); API(mock_model_load); API(mock_model_unload);
  API(mock_instance_create); API(mock_instance_destroy); API(mock_request_new); API(mock_request_delete);
  API(mock_request_add_input_buffer); API(mock_request_add_requested_output); API(mock_request_set_output_buffer);
  API(mock_instance_execute); API(mock_request_response_count); API(mock_request_release_count);
  API(mock_request_error_code); API(mock_request_error_message); API(mock_request_response_int_param);
  API(mock_instance_get_stats);
};
#undef API

[[noreturn]] void die(const char* what, const char* detail = "") {
  fprintf(stderr, "triton_abi_bench: %s %s\n", what, detail ? detail : "");
  exit(2);
}

uint64_t mix(uint64_t x) { return hps_mix64(x); }

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

int main(int argc, char** argv) {
  Args a;
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string k = argv[i];
    const char* v = argv[i + 1];
    if (k == "--lib-dir") a.lib_dir = v;
    else if (k == "--tables") a.tables = atoi(v);
    else if (k == "--rows") a.rows = atol(v);
    else if (k == "--dim") a.dim = atoi(v);
    else if (k == "--batch") a.batch = atol(v);
    else if (k == "--cache-frac") a.cache_frac = atof(v);
    else if (k == "--hit") a.hit = atof(v);
    else if (k == "--zipf") a.zipf = atof(v);
    else if (k == "--instances") a.instances = atoi(v);
    else if (k == "--steps") a.steps = atoi(v);
    else if (k == "--warmup") a.warmup = atoi(v);
    else if (k == "--blocks") a.blocks = atoi(v);
    else if (k == "--direct") a.direct = atoi(v);
    else if (k == "--pinned-keys") a.pinned_keys = atoi(v);
    else if (k == "--also-pinned") a.also_pinned = atoi(v);   // after the measurement: this many more blocks with KEYS in page-locked memory (TRITONSERVER_MEMORY_CPU_PINNED)
    else if (k == "--threshold") a.threshold = atof(v);
    else if (k == "--models") a.models = atoi(v);
    else if (k == "--gpucache") a.gpucache = atoi(v);
    else if (k == "--uniform") a.uniform = atoi(v);
    else if (k == "--shards") a.shards = atoi(v);
    else if (k == "--requests-per-execute") a.rpe = std::max(1, atoi(v));
    else if (k == "--dims") a.dims = v;
    else if (k == "--per-sample") a.per_sample = v;
    else die("unknown option", argv[i]);
  }
  setenv("GPU_MAX_HW_QUEUES", "8", 0);   // two instances' streams on separate hardware queues
  auto split = [](const std::string& csv) {
    std::vector<long> out;
    size_t b = 0;
    while (b < csv.size()) {
      size_t e = csv.find(',', b);
      if (e == std::string::npos) e = csv.size();
      out.push_back(atol(csv.substr(b, e - b).c_str()));
      b = e + 1;
    }
    return out;
  };
  std::vector<long> Dt = split(a.dims), Pt = split(a.per_sample);
  if (Dt.empty()) Dt.assign((size_t)a.tables, a.dim);
  const int T = (int)Dt.size();
  if (Pt.empty()) Pt.assign((size_t)T, 1);
  if ((int)Pt.size() != T) die("--dims and --per-sample must have one entry per table");
  const long R = a.rows, B = a.batch;
  const int M = std::max(1, a.models);
  std::vector<size_t> key_off((size_t)T + 1, 0), out_off((size_t)T + 1, 0);   // per request: keys / floats before table t
  for (int t = 0; t < T; ++t) {
    key_off[(size_t)t + 1] = key_off[(size_t)t] + (size_t)B * (size_t)Pt[(size_t)t];
    out_off[(size_t)t + 1] = out_off[(size_t)t] + (size_t)B * (size_t)Pt[(size_t)t] * (size_t)Dt[(size_t)t];
  }
  const size_t N = key_off[(size_t)T], OUT = out_off[(size_t)T];
  const bool gpu = a.gpucache != 0;
  // table-sharded deployment (BASELINE config 3 behind the plugin): the devices of the shards and of the instances
  int ndev = 0;
  if (gpu && hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
  std::vector<int> shard_dev, inst_dev_pool;
  if (a.shards > 0) {
    if (!gpu || ndev <= 0) die("--shards needs a GPU");
    for (int sdx = 0; sdx < a.shards; ++sdx) shard_dev.push_back(sdx % ndev);
    for (int d = 0; d < std::min(ndev, a.shards); ++d) inst_dev_pool.push_back(d);
  } else {
    inst_dev_pool.push_back(0);
  }

  void* core = dlopen((a.lib_dir + "/libtriton_mock_core.so").c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (!core) die("cannot load the mock core:", dlerror());
  Mock m;
#define LOAD(name) if (!(m.name = (decltype(m.name))dlsym(core, #name))) die("missing symbol", #name)
  LOAD(mock_last_error); LOAD(mock_server_create); LOAD(mock_server_destroy); LOAD(mock_model_load); LOAD(mock_model_unload);
  LOAD(mock_instance_create); LOAD(mock_instance_destroy); LOAD(mock_request_new); LOAD(mock_request_delete);
  LOAD(mock_request_add_input_buffer); LOAD(mock_request_add_requested_output); LOAD(mock_request_set_output_buffer);
  LOAD(mock_instance_execute); LOAD(mock_request_response_count); LOAD(mock_request_release_count);
  LOAD(mock_request_error_code); LOAD(mock_request_error_message); LOAD(mock_request_response_int_param);
  LOAD(mock_instance_get_stats);
#undef LOAD

  // ---- ps.json + model configurations (what Triton derives from config.pbtxt) ----
  const uint64_t kSeed = 20260929ull;
  std::vector<std::string> names;
  for (int mi = 0; mi < M; ++mi) names.push_back(M == 1 ? std::string("criteo_dlrm") : "model_" + std::to_string(mi));
  char tmpl[] = "/tmp/hps_abi_bench_XXXXXX";
  if (!mkdtemp(tmpl)) die("mkdtemp failed");
  const std::string ps_path = std::string(tmpl) + "/ps.json";
  {
    std::string j = "{\"supportlonglong\": true, \"volatile_db\": {\"type\": \"hash_map\", \"num_partitions\": 8}, \"models\": [";
    for (int mi = 0; mi < M; ++mi) {
      j += std::string(mi ? ", {" : "{") + "\"model\": \"" + names[(size_t)mi] + "\", \"sparse_files\": [";
      for (int t = 0; t < T; ++t)
        j += (t ? ", \"synthetic://" : "\"synthetic://") + std::to_string(R) + "?seed=" + std::to_string(kSeed + (uint64_t)mi) + "\"";
      j += "], \"num_of_worker_buffer_in_pool\": " + std::to_string(std::max(3, a.instances));   // (instances per device)
      auto list = [&](const char* key, const std::vector<long>* v, const char* all) {
        j += std::string(", \"") + key + "\": [";
        for (int t = 0; t < T; ++t) j += (t ? ", " : "") + (v ? std::to_string((*v)[(size_t)t]) : std::string(all));
        j += "]";
      };
      list("embedding_vecsize_per_table", &Dt, "");
      list("maxnum_catfeature_query_per_table_per_sample", &Pt, "");
      list("default_value_for_each_table", nullptr, "0.0");
      char buf[320];
      std::string devlist = "0";
      if (a.shards > 0) {
        devlist.clear();
        for (size_t sdx = 0; sdx < shard_dev.size(); ++sdx) devlist += (sdx ? ", " : "") + std::to_string(shard_dev[sdx]);
      }
      snprintf(buf, sizeof buf, ", \"max_batch_size\": %ld, \"gpucache\": %s, \"gpucacheper\": %.6f, "
               "\"hit_rate_threshold\": %.6f, \"ps_direct_access\": %s%s}", B * a.rpe, gpu ? "true" : "false", a.cache_frac, a.threshold,
               (a.direct && gpu) ? "true" : "false", a.shards > 0 ? ", \"table_sharding\": \"hash\", \"gpucache_load_factor\": 0.6" : "");
      j += ", \"deployed_device_list\": [" + devlist + "]";
      j += buf;
    }
    j += "]}";
    FILE* f = fopen(ps_path.c_str(), "w");
    if (!f) die("cannot write", ps_path.c_str());
    fputs(j.c_str(), f);
    fclose(f);
  }
  const std::string backend_cfg = "{\"cmdline\": {\"auto-complete-config\": \"true\", \"ps\": \"" + ps_path + "\"}}";

  const double t_load0 = now_s();
  mock_server_t* srv = nullptr;
  if (m.mock_server_create((a.lib_dir + "/libtriton_hps.so").c_str(), "hps", backend_cfg.c_str(), 0, 0, &srv) != 0)
    die("TRITONBACKEND_Initialize failed:", m.mock_last_error());
  std::vector<mock_model_t*> model((size_t)M, nullptr);
  std::vector<mock_instance_t*> inst;   // worker w serves model w / instances, instance w % instances
  std::vector<int> inst_dev;            // ... on this device
  for (int mi = 0; mi < M; ++mi) {
    const std::string model_cfg =
        "{\"name\": \"" + names[(size_t)mi] + "\", \"backend\": \"hps\", \"max_batch_size\": " + std::to_string(B * a.rpe) +
        ", \"input\": [{\"name\": \"KEYS\", \"data_type\": \"TYPE_INT64\", \"dims\": [-1]}, {\"name\": \"NUMKEYS\", \"data_type\": "
        "\"TYPE_INT32\", \"dims\": [-1]}], \"output\": [{\"name\": \"OUTPUT0\", \"data_type\": \"TYPE_FP32\", \"dims\": [-1]}], "
        "\"instance_group\": [{\"count\": " + std::to_string(a.instances) + ", \"kind\": \"" + (gpu ? "KIND_GPU" : "KIND_CPU") +
        "\", \"gpus\": [" + [&] { std::string g; for (size_t d = 0; gpu && d < inst_dev_pool.size(); ++d) g += (d ? ", " : "") + std::to_string(inst_dev_pool[d]); return g; }() + "]}]}";
    if (m.mock_model_load(srv, names[(size_t)mi].c_str(), 1, model_cfg.c_str(), &model[(size_t)mi]) != 0)
      die("ModelInitialize failed:", m.mock_last_error());
    // (sharded: `instances` per device of the pool, like instance_group { count, gpus })
    for (size_t dpi = 0; dpi < inst_dev_pool.size(); ++dpi)
      for (int i = 0; i < a.instances; ++i) {
        mock_instance_t* h = nullptr;
        const int d = inst_dev_pool[dpi];
        if (m.mock_instance_create(model[(size_t)mi], (names[(size_t)mi] + "_" + std::to_string(d) + "_" + std::to_string(i)).c_str(), gpu ? 2 : 1, d, &h) != 0)
          die("ModelInstanceInitialize failed:", m.mock_last_error());
        inst.push_back(h);
        inst_dev.push_back(d);
      }
  }
  const int W = (int)inst.size();   // workers = models x instances (x devices of a sharded model)
  const int per_model = W / M;
  auto sync_all = [&] { for (int d : inst_dev_pool) { (void)hipSetDevice(d); (void)hipDeviceSynchronize(); } };
  const double load_s = now_s() - t_load0;

  // ---- key batches: per table n_t keys; with probability `hit` a Zipf-ranked key of the warmed range [0, C), else uniform
  //      from the cold range [C, R) — bench.py's generator, on host threads (--uniform 1: uniform over [0, R)) ----
  const long C = (long)std::ceil(a.cache_frac * (double)R);
  std::vector<double> cdf;
  if (!a.uniform) {
    cdf.resize((size_t)C);
    double acc = 0;
    for (long i = 0; i < C; ++i) { acc += 1.0 / std::pow((double)(i + 1), a.zipf); cdf[(size_t)i] = acc; }
    for (long i = 0; i < C; ++i) cdf[(size_t)i] /= acc;
  }
  // (--also-pinned: its blocks get FRESH batches behind the main measurement's — a replayed batch finds its cold keys inserted)
  const int Q = a.rpe;
  const int nbatch_main = (a.warmup + a.steps * a.blocks) * Q;
  const int nbatch = nbatch_main + (a.also_pinned > 0 ? (4 + a.steps * a.also_pinned) * Q : 0);
  int64_t* keys_all = nullptr;
  const size_t key_bytes = (size_t)nbatch * N * sizeof(int64_t);
  if (a.pinned_keys) {
    if (hipHostMalloc((void**)&keys_all, key_bytes, hipHostMallocDefault) != hipSuccess) die("hipHostMalloc of the key batches failed");
  } else {
    keys_all = (int64_t*)malloc(key_bytes);
    if (!keys_all) die("out of memory for the key batches");
  }
  {
    const unsigned nth = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::atomic<long> next{0};
    std::vector<std::thread> th;
    for (unsigned w = 0; w < nth; ++w)
      th.emplace_back([&] {
        for (;;) {
          const long job = next.fetch_add(1);   // one (batch, table) slice per job
          if (job >= (long)nbatch * T) return;
          const long b = job / T, t = job % T;
          const long nt = (long)(key_off[(size_t)t + 1] - key_off[(size_t)t]);
          int64_t* dst = keys_all + (size_t)b * N + key_off[(size_t)t];
          uint64_t s = mix(0x5EEDull * 1315423911ull + (uint64_t)job);
          for (long i = 0; i < nt; ++i) {
            s = mix(s + (uint64_t)i);
            const double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);
            const uint64_t s2 = mix(s ^ 0xABCDEFull);
            const double v = (double)(s2 >> 11) * (1.0 / 9007199254740992.0);
            if (a.uniform) {
              dst[i] = std::min((long)(v * (double)R), R - 1);
            } else if (u < a.hit || R <= C) {
              const long rank = (long)(std::lower_bound(cdf.begin(), cdf.end(), v) - cdf.begin());
              dst[i] = std::min(rank, C - 1);
            } else {
              dst[i] = C + (long)(v * (double)(R - C));
            }
          }
        }
      });
    for (auto& x : th) x.join();
  }
  std::vector<int32_t> numkeys((size_t)T);
  for (int t = 0; t < T; ++t) numkeys[(size_t)t] = (int32_t)(key_off[(size_t)t + 1] - key_off[(size_t)t]);
  std::vector<float*> out_buf((size_t)W, nullptr);   // device memory for GPU instances, host memory for CPU instances
  for (int i = 0; i < W; ++i) {
    if (gpu) { if (hipSetDevice(inst_dev[(size_t)i]) != hipSuccess || hipMalloc((void**)&out_buf[(size_t)i], (size_t)Q * OUT * sizeof(float)) != hipSuccess) die("hipMalloc of OUTPUT0 failed"); }
    else if (!(out_buf[(size_t)i] = (float*)malloc((size_t)Q * OUT * sizeof(float)))) die("out of memory for OUTPUT0");
  }

  // ---- stall watchdog: a thread that does nothing but sleep 200 us at a time and notes every wake-up that comes more than
  //      2 ms late.  A late wake-up that coincides with a slow request means the whole process stood still (CPU quota of an
  //      enclosing cgroup, a frozen container, the hypervisor) — nothing the request path did. ----
  std::atomic<bool> dog_stop{false};
  std::vector<std::pair<double, double>> dog_gaps;   // (time since start [s], gap [ms])
  const double dog_t0 = now_s();
  std::thread dog([&] {
    double last = now_s();
    while (!dog_stop.load(std::memory_order_relaxed)) {
      usleep(200);
      const double t = now_s();
      if (t - last > 0.002) dog_gaps.emplace_back(last - dog_t0, (t - last) * 1e3);
      last = t;
    }
  });
  std::vector<std::pair<double, double>> slow_requests;   // (start since dog_t0 [s], duration [ms]) of requests over 5 ms
  std::mutex slow_mu;

  // ---- request loop: `count` requests shared by all workers (every model takes its share of the batches) ----
  std::atomic<long> next{0};
  std::atomic<int> failed{0};
  std::vector<std::vector<double>> lat((size_t)W);
  std::vector<long> last_batch((size_t)W, -1);
  const int64_t* keys_base = keys_all;          // where the requests' KEYS live, and what kind of memory Triton says it is
  int keys_mtype = a.pinned_keys ? 1 : 0;
  auto run = [&](long first, long count, bool record) {
    next.store(0);
    std::vector<std::thread> th;
    for (int w = 0; w < W; ++w)
      th.emplace_back([&, w] {
        if (gpu) (void)hipSetDevice(inst_dev[(size_t)w]);
        const int64_t kshape[2] = {1, (int64_t)N}, nshape[2] = {1, (int64_t)T};
        // one model: the instances share the requests (whoever is free takes the next, as Triton's scheduler hands them out);
        // several models: every model's instances get that model's own share, so that all of them are under load all the time
        const int mi = w / per_model, wi = w % per_model;
        long mine = wi;
        for (;;) {
          long i;
          if (M == 1) i = next.fetch_add(1);
          else { i = mine * M + mi; mine += per_model; }
          if (i >= count) return;
          // (Q requests per Execute call: batches (first + i) * Q .. + Q - 1, each with its own output buffer)
          const long b0 = (first + i) * Q;
          std::vector<mock_request_t*> rqs((size_t)Q, nullptr);
          for (int j = 0; j < Q; ++j) {
            const long b = b0 + j;
            mock_request_t* rq = rqs[(size_t)j] = m.mock_request_new(std::to_string(b).c_str(), 0);
            m.mock_request_add_input_buffer(rq, "KEYS", 9 /*INT64*/, kshape, 2, keys_base + (size_t)b * N, N * sizeof(int64_t), keys_mtype, 0);
            m.mock_request_add_input_buffer(rq, "NUMKEYS", 8 /*INT32*/, nshape, 2, numkeys.data(), (uint64_t)T * sizeof(int32_t), 0, 0);
            m.mock_request_add_requested_output(rq, "OUTPUT0");
            m.mock_request_set_output_buffer(rq, out_buf[(size_t)w] + (size_t)j * OUT, OUT * sizeof(float), gpu ? 2 /*GPU*/ : 0 /*CPU*/, gpu ? inst_dev[(size_t)w] : 0);
          }
          const double t0 = now_s();
          const int rc = m.mock_instance_execute(inst[(size_t)w], rqs.data(), (uint32_t)Q);
          const double dt = now_s() - t0;
          for (mock_request_t* rq : rqs) {
            if (rc != 0 || m.mock_request_error_code(rq) != -1 || m.mock_request_response_count(rq) != 1 ||
                m.mock_request_release_count(rq) != 1) {
              if (!failed.exchange(1))
                fprintf(stderr, "request of execute %ld failed: rc=%d code=%d %s\n", first + i, rc, m.mock_request_error_code(rq),
                        m.mock_request_error_message(rq) ? m.mock_request_error_message(rq) : "");
            }
          }
          if (record) lat[(size_t)w].push_back(dt * 1e3);
          if (record && dt > 0.005) { std::lock_guard<std::mutex> lk(slow_mu); slow_requests.emplace_back(t0 - dog_t0, dt * 1e3); }
          last_batch[(size_t)w] = b0 + Q - 1;     // (checked below: the LAST request of the call, in the last output buffer)
          for (mock_request_t* rq : rqs) m.mock_request_delete(rq);
        }
      });
    for (auto& x : th) x.join();
  };
  run(0, a.warmup, false);
  if (gpu) sync_all();
  std::vector<double> block_s;
  for (int blk = 0; blk < a.blocks; ++blk) {
    const double t0 = now_s();
    run(a.warmup + (long)blk * a.steps, a.steps, true);
    if (gpu) sync_all();
    block_s.push_back(now_s() - t0);
  }

  // ---- (--also-pinned n) the same requests with KEYS in page-locked host memory, as Triton hands them over out of its pinned
  //      pool when the backend asks for TRITONSERVER_MEMORY_CPU_PINNED (hps.cpp, CollectInput): the lookup DMAs them in place,
  //      8 bytes per key, no staging copy ----
  double pinned_lps = 0, pinned_p50 = 0, pinned_p99 = 0;
  if (a.also_pinned > 0 && !a.pinned_keys && gpu) {
    const long nb2 = (long)nbatch - nbatch_main;
    int64_t* pk = nullptr;
    if (hipHostMalloc((void**)&pk, (size_t)nb2 * N * sizeof(int64_t), hipHostMallocDefault) == hipSuccess) {
      memcpy(pk, keys_all + (size_t)nbatch_main * N, (size_t)nb2 * N * sizeof(int64_t));
      std::vector<std::vector<double>> lat_main;
      lat_main.swap(lat);
      lat.assign((size_t)W, {});
      keys_base = pk;
      keys_mtype = 1;
      run(0, 4, false);
      sync_all();
      std::vector<double> bs2;
      for (int blk = 0; blk < a.also_pinned && (4 + (long)(blk + 1) * a.steps) * Q <= nb2; ++blk) {
        const double t0 = now_s();
        run(4 + (long)blk * a.steps, a.steps, true);
        sync_all();
        bs2.push_back(now_s() - t0);
      }
      std::sort(bs2.begin(), bs2.end());
      if (!bs2.empty()) pinned_lps = (double)a.steps * (double)Q * (double)N / bs2[bs2.size() / 2];
      std::vector<double> all2;
      for (auto& v : lat) all2.insert(all2.end(), v.begin(), v.end());
      std::sort(all2.begin(), all2.end());
      if (!all2.empty()) { pinned_p50 = all2[all2.size() / 2]; pinned_p99 = all2[std::min(all2.size() - 1, (size_t)(0.99 * (double)all2.size()))]; }
      lat.swap(lat_main);
      keys_base = keys_all;
      keys_mtype = 0;
      for (auto& b : last_batch) if (b >= 0) b += nbatch_main;   // (the row check below reads keys_all: the pinned phase's batch b is batch nbatch_main + b there)
      (void)hipHostFree(pk);
    }
  }

  dog_stop.store(true);
  dog.join();

  // ---- the last response of every worker against the table recipe: every key of [0, R) exists, row(t, k) is
  //      a pure function of (seed of the model, t, k) ----
  long bad = 0, checked = 0;
  for (int w = 0; w < W; ++w) {
    const long b = last_batch[(size_t)w];
    const uint64_t seed = kSeed + (uint64_t)(w / per_model);
    std::vector<float> row;
    for (int s = 0; s < 2048 / W + 1 && b >= 0; ++s) {
      const size_t i = (size_t)(mix(77 + (uint64_t)s + 1000ull * (uint64_t)w) % N);
      int t = 0;
      while (i >= key_off[(size_t)t + 1]) ++t;
      const size_t D = (size_t)Dt[(size_t)t];
      const int64_t key = keys_all[(size_t)b * N + i];
      const float* src = out_buf[(size_t)w] + (size_t)(Q - 1) * OUT + out_off[(size_t)t] + (i - key_off[(size_t)t]) * D;
      row.resize(D);
      if (gpu) { if (hipMemcpy(row.data(), src, D * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { ++bad; continue; } }
      else memcpy(row.data(), src, D * sizeof(float));
      const uint64_t rb = hps_synth_row_base(hps_synth_table_base(seed, (uint32_t)t), key);
      for (size_t j = 0; j < D; ++j) {
        uint32_t got;
        memcpy(&got, &row[j], 4);
        if (got != hps_synth_elem_bits(rb, (uint32_t)j)) { ++bad; break; }
      }
      ++checked;
    }
  }
  mock_instance_stats_t st{};
  uint64_t ok_req = 0, reports = 0;
  for (int i = 0; i < W; ++i) { m.mock_instance_get_stats(inst[(size_t)i], &st); ok_req += st.success_requests; reports += st.batch_reports; }

  std::vector<double> all;
  for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
  std::sort(all.begin(), all.end());
  std::vector<double> bs = block_s;
  std::sort(bs.begin(), bs.end());
  const double med = bs.empty() ? 0 : bs[bs.size() / 2];
  auto pct = [&](double p) { return all.empty() ? 0.0 : all[std::min(all.size() - 1, (size_t)(p * (double)all.size()))]; };
  printf("{\"through\": \"TRITONBACKEND_ModelInstanceExecute (libtriton_hps.so) driven by the mock Triton core, native caller\", "
         "\"keys_memory\": \"%s\", \"output_memory\": \"%s\", \"models\": %d, \"instances\": %d, \"shards\": %d, \"visible_devices\": %d, \"tables\": %d, \"keys_per_request\": %zu, "
         "\"floats_per_response\": %zu, \"steps_per_block\": %d, \"blocks\": %d, "
         "\"lookups_per_s\": %.6g, \"requests_per_s\": %.6g, \"ms_per_step\": %.6g, \"block_ms\": [", a.pinned_keys ? "host, page-locked" : "host, pageable",
         gpu ? "device" : "host", M, W / M, a.shards, ndev, T, N, OUT, a.steps, a.blocks, med > 0 ? (double)a.steps * (double)Q * (double)N / med : 0.0,
         med > 0 ? (double)a.steps * (double)Q / med : 0.0, med / a.steps * 1e3);
  for (size_t i = 0; i < block_s.size(); ++i) printf("%s%.4g", i ? ", " : "", block_s[i] * 1e3);
  {
    uint64_t one_lookup = 0;
    for (int i = 0; i < W; ++i) { mock_instance_stats_t s2{}; m.mock_instance_get_stats(inst[(size_t)i], &s2); one_lookup += s2.last_distinct_compute_starts == 1; }
    printf("], \"requests_per_execute\": %d, \"instances_whose_last_execute_was_one_lookup\": %llu, \"p50_execute_ms\": %.5g, ", Q, (unsigned long long)one_lookup, pct(0.5));
  }
  printf("\"p50_request_ms\": %.5g, \"p99_request_ms\": %.5g, \"max_request_ms\": %.5g, \"requests_ok_reported_by_backend\": %llu, \"batch_statistics_reports\": %llu, "
         "\"failed\": %d, \"rows_checked_against_recipe\": %ld, \"rows_wrong\": %ld, \"model_load_seconds\": %.4g, \"ps_tier\": \"%s\", ",
         pct(0.5), pct(0.99), all.empty() ? 0.0 : all.back(), (unsigned long long)ok_req, (unsigned long long)reports, failed.load(), checked, bad, load_s,
         !gpu ? "CPU parameter server only (gpucache=false)" : a.direct ? "device-driven (ps_direct_access)" : "host gather");
  // requests over 5 ms, and for each the watchdog gaps that overlap it
  if (pinned_lps > 0)
    printf("\"pinned_keys\": {\"keys_memory\": \"host, page-locked (TRITONSERVER_MEMORY_CPU_PINNED): DMA in place, 8 bytes per key\", "
           "\"lookups_per_s\": %.6g, \"p50_request_ms\": %.5g, \"p99_request_ms\": %.5g, \"blocks\": %d}, ", pinned_lps, pinned_p50, pinned_p99, a.also_pinned);
  printf("\"slow_requests_ms\": [");
  int coincide = 0;
  for (size_t i = 0; i < slow_requests.size(); ++i) {
    double overlap = 0;
    for (const auto& g : dog_gaps)
      if (g.first < slow_requests[i].first + slow_requests[i].second * 1e-3 && g.first + g.second * 1e-3 > slow_requests[i].first) overlap = std::max(overlap, g.second);
    coincide += overlap > 0;
    printf("%s[%.2f, %.2f]", i ? ", " : "", slow_requests[i].second, overlap);
  }
  printf("], \"slow_requests_note\": \"[request ms, longest late wake-up (ms) of an idle watchdog thread during it]\", "
         "\"slow_requests_with_process_wide_stall\": %d, \"watchdog_late_wakeups_over_2ms\": %zu}\n", coincide, dog_gaps.size());
  fflush(stdout);
  for (auto* i : inst) m.mock_instance_destroy(i);
  for (auto* mm : model) m.mock_model_unload(mm);
  m.mock_server_destroy(srv);
  return failed.load() || bad ? 1 : 0;
}
