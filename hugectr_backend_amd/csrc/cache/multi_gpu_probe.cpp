#include "multi_gpu_probe.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <memory>
#include <mutex>
#include <thread>

#include "shard_session.h"

namespace hps {

namespace {

struct ProbeState {
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  std::string phase = "start";
  std::string json;
  std::vector<std::atomic<int>> rccl_phase;
  explicit ProbeState(size_t n) : rccl_phase(n) { for (auto& p : rccl_phase) p.store(0); }
  void Phase(const std::string& s) { std::lock_guard<std::mutex> lk(mu); phase = s; }
};

std::string Num(double v) {
  char b[32];
  snprintf(b, sizeof b, "%.2f", v);
  return b;
}

// [from][to] matrix of optional values -> JSON
template <typename F>
std::string Matrix(size_t n, F cell) {
  std::string s = "[";
  for (size_t a = 0; a < n; ++a) {
    s += a ? ",[" : "[";
    for (size_t b = 0; b < n; ++b) { if (b) s += ","; s += cell(a, b); }
    s += "]";
  }
  return s + "]";
}

double Median(std::vector<double> v) {
  if (v.empty()) return 0.0;
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

void Run(std::shared_ptr<ProbeState> st, std::vector<int> devs, uint64_t bytes, bool with_rccl) {
  const auto t0 = std::chrono::steady_clock::now();
  const size_t n = devs.size();
  std::vector<int> access(n * n, -1), ok4k(n * n, -1);
  std::vector<double> store(n * n, -1.0), copy(n * n, -1.0);
  std::string error;
  auto fail = [&](const char* what, hipError_t e) { if (error.empty()) error = std::string(what) + ": " + hipGetErrorString(e); (void)hipGetLastError(); };
  const uint64_t words = std::max<uint64_t>(bytes / 4, 1024) & ~(uint64_t)3;
  // one buffer per device
  std::vector<uint32_t*> buf(n, nullptr);
  std::vector<hipStream_t> stream(n, nullptr);
  std::vector<hipEvent_t> ev0(n, nullptr), ev1(n, nullptr);
  for (size_t a = 0; a < n && error.empty(); ++a) {
    st->Phase("allocating on device " + std::to_string(devs[a]));
    hipError_t e = hipSetDevice(devs[a]);
    if (e == hipSuccess) e = hipMalloc((void**)&buf[a], words * 4);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&stream[a], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ev0[a]);
    if (e == hipSuccess) e = hipEventCreate(&ev1[a]);
    if (e != hipSuccess) fail("setup", e);
  }
  std::vector<uint32_t> host(1024);
  for (size_t a = 0; a < n && error.empty(); ++a) {
    for (size_t b = 0; b < n && error.empty(); ++b) {
      if (a == b) continue;
      const std::string pair = std::to_string(devs[a]) + "->" + std::to_string(devs[b]);
      st->Phase("peer access " + pair);
      hipError_t e = hipSetDevice(devs[a]);
      int can = 0;
      if (e == hipSuccess) e = hipDeviceCanAccessPeer(&can, devs[a], devs[b]);
      if (e != hipSuccess) { fail("hipDeviceCanAccessPeer", e); break; }
      access[a * n + b] = can;
      if (can) {
        e = hipDeviceEnablePeerAccess(devs[b], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { fail("hipDeviceEnablePeerAccess", e); break; }
        (void)hipGetLastError();
        // 4 KB stored by a kernel on a into b's memory, read back through b
        st->Phase("4-KB peer store " + pair);
        const uint32_t seed = 0x5EED0000u + (uint32_t)(a * 64 + b);
        e = hipMemsetAsync(buf[b], 0, 4096, stream[a]);
        if (e == hipSuccess) e = LaunchProbeStore(buf[b], 1024, seed, stream[a]);
        if (e == hipSuccess) e = hipStreamSynchronize(stream[a]);
        if (e == hipSuccess) e = hipSetDevice(devs[b]);
        if (e == hipSuccess) e = hipMemcpy(host.data(), buf[b], 4096, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { fail("4-KB peer store", e); break; }
        bool good = true;
        for (uint32_t i = 0; i < 1024; ++i) good &= host[i] == seed + i;
        ok4k[a * n + b] = good ? 1 : 0;
        (void)hipSetDevice(devs[a]);
        // bandwidth of kernel stores a -> b (best of 3)
        st->Phase("peer store bandwidth " + pair);
        float best = 0.f;
        for (int rep = 0; rep < 3 && e == hipSuccess; ++rep) {
          e = hipEventRecord(ev0[a], stream[a]);
          if (e == hipSuccess) e = LaunchProbeStore(buf[b], words, seed + 7, stream[a]);
          if (e == hipSuccess) e = hipEventRecord(ev1[a], stream[a]);
          if (e == hipSuccess) e = hipStreamSynchronize(stream[a]);
          float ms = 0.f;
          if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev0[a], ev1[a]);
          if (e == hipSuccess && ms > 0.f) best = std::max(best, (float)(words * 4 / (ms * 1e-3) / 1e9));
        }
        if (e != hipSuccess) { fail("peer store bandwidth", e); break; }
        store[a * n + b] = best;
      }
      // copy engine a -> b (works with or without peer access; best of 3)
      st->Phase("peer copy bandwidth " + pair);
      float best = 0.f;
      for (int rep = 0; rep < 3 && e == hipSuccess; ++rep) {
        e = hipEventRecord(ev0[a], stream[a]);
        if (e == hipSuccess) e = hipMemcpyPeerAsync(buf[b], devs[b], buf[a], devs[a], words * 4, stream[a]);
        if (e == hipSuccess) e = hipEventRecord(ev1[a], stream[a]);
        if (e == hipSuccess) e = hipStreamSynchronize(stream[a]);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev0[a], ev1[a]);
        if (e == hipSuccess && ms > 0.f) best = std::max(best, (float)(words * 4 / (ms * 1e-3) / 1e9));
      }
      if (e != hipSuccess) { fail("peer copy bandwidth", e); break; }
      copy[a * n + b] = best;
    }
  }
  for (size_t a = 0; a < n; ++a) {
    (void)hipSetDevice(devs[a]);
    if (stream[a]) (void)hipStreamDestroy(stream[a]);
    if (ev0[a]) (void)hipEventDestroy(ev0[a]);
    if (ev1[a]) (void)hipEventDestroy(ev1[a]);
    if (buf[a]) (void)hipFree(buf[a]);
  }
  std::string rccl = "null";
  if (with_rccl && error.empty()) {
    st->Phase("RCCL all-reduce of one word over " + std::to_string(n) + " ranks");
    float ms = 0.f;
    const Status rs = RcclAllReduceSelfTest(devs, st->rccl_phase.data(), &ms);
    std::string msg = rs.message();
    for (char& c : msg) if (c == '"' || c == '\\' || c == '\n') c = ' ';
    rccl = "{\"ranks\":" + std::to_string(n) + ",\"ok\":" + (rs.ok() ? "true" : "false") + ",\"ms\":" + Num(ms) +
           ",\"error\":" + (rs.ok() ? "null" : "\"" + msg + "\"") + "}";
  }
  std::vector<double> sv, cv;
  for (size_t i = 0; i < n * n; ++i) { if (store[i] > 0) sv.push_back(store[i]); if (copy[i] > 0) cv.push_back(copy[i]); }
  auto opt = [](double v) { return v < 0 ? std::string("null") : Num(v); };
  auto opti = [](int v) { return v < 0 ? std::string("null") : std::to_string(v); };
  std::string j = "{\"devices\":[";
  for (size_t a = 0; a < n; ++a) j += (a ? "," : "") + std::to_string(devs[a]);
  j += "],\"probe_bytes\":" + std::to_string(words * 4);
  j += ",\"peer_access\":" + Matrix(n, [&](size_t a, size_t b) { return opti(access[a * n + b]); });
  j += ",\"store_4k_ok\":" + Matrix(n, [&](size_t a, size_t b) { return opti(ok4k[a * n + b]); });
  j += ",\"store_GBps\":" + Matrix(n, [&](size_t a, size_t b) { return opt(store[a * n + b]); });
  j += ",\"copy_GBps\":" + Matrix(n, [&](size_t a, size_t b) { return opt(copy[a * n + b]); });
  j += ",\"pair_GBps_min\":{\"store\":" + (sv.empty() ? "null" : Num(*std::min_element(sv.begin(), sv.end()))) +
       ",\"copy\":" + (cv.empty() ? "null" : Num(*std::min_element(cv.begin(), cv.end()))) + "}";
  j += ",\"pair_GBps_median\":{\"store\":" + (sv.empty() ? "null" : Num(Median(sv))) + ",\"copy\":" + (cv.empty() ? "null" : Num(Median(cv))) + "}";
  j += ",\"rccl_allreduce\":" + rccl;
  for (char& c : error) if (c == '"' || c == '\\' || c == '\n') c = ' ';
  j += ",\"error\":" + (error.empty() ? std::string("null") : "\"" + error + "\"");
  j += ",\"timeout\":false,\"stuck_in\":null,\"seconds\":" +
       Num(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()) + "}";
  {
    std::lock_guard<std::mutex> lk(st->mu);
    st->json = std::move(j);
    st->done = true;
  }
  st->cv.notify_all();
}

}  // namespace

bool MultiGpuSelfTest(const std::vector<int>& devices, uint64_t probe_bytes, uint32_t timeout_ms, bool with_rccl, std::string* json) {
  auto st = std::make_shared<ProbeState>(devices.size());
  // (detached: a step that hangs keeps its thread; the state it writes to is kept alive by the shared pointer)
  std::thread(Run, st, devices, probe_bytes, with_rccl).detach();
  std::unique_lock<std::mutex> lk(st->mu);
  if (st->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return st->done; })) {
    *json = st->json;
    return true;
  }
  std::string where = st->phase;
  if (where.rfind("RCCL", 0) == 0) {
    where += " (rank phases:";
    for (auto& p : st->rccl_phase) where += " " + std::to_string(p.load());
    where += "; 1 = inside ncclCommInitRank, 2 = all-reduce enqueued, 3 = result checked)";
  }
  *json = "{\"timeout\":true,\"stuck_in\":\"" + where + "\",\"seconds\":" + Num(timeout_ms / 1e3) + "}";
  return false;
}

}  // namespace hps
