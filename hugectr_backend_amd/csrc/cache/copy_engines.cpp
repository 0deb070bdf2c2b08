#include "copy_engines.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <map>
#include <mutex>
#include <sstream>

namespace hps {
namespace {

// The HSA runtime is already in the process (the HIP runtime sits on it); its entry points are taken from that copy.
struct HsaApi {
  decltype(&hsa_amd_pointer_info) pointer_info = nullptr;
  decltype(&hsa_amd_memory_copy_engine_status) engine_status = nullptr;
  decltype(&hsa_amd_memory_async_copy_on_engine) copy_on_engine = nullptr;
  decltype(&hsa_signal_create) signal_create = nullptr;
  decltype(&hsa_signal_destroy) signal_destroy = nullptr;
  decltype(&hsa_signal_wait_scacquire) signal_wait = nullptr;
  bool ok = false;
  HsaApi() {
    void* h = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("libhsa-runtime64.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) return;
    pointer_info = (decltype(pointer_info))dlsym(h, "hsa_amd_pointer_info");
    engine_status = (decltype(engine_status))dlsym(h, "hsa_amd_memory_copy_engine_status");
    copy_on_engine = (decltype(copy_on_engine))dlsym(h, "hsa_amd_memory_async_copy_on_engine");
    signal_create = (decltype(signal_create))dlsym(h, "hsa_signal_create");
    signal_destroy = (decltype(signal_destroy))dlsym(h, "hsa_signal_destroy");
    signal_wait = (decltype(signal_wait))dlsym(h, "hsa_signal_wait_scacquire");
    ok = pointer_info && engine_status && copy_on_engine && signal_create && signal_destroy && signal_wait;
    // drop our reference again: the HIP runtime keeps the library loaded for as long as these pointers are used, and an
    // extra reference changes the order in which the process unloads it at exit
    dlclose(h);
  }
};

// one pass over the engines of one direction; returns how many took a copy
// *stuck is set when a copy did not finish within the bound: the engine may still write `dst` / signal `sig`, so the caller
// must give up the scratch buffers (and this function the signal) instead of freeing memory under a DMA in flight
int WakeDirection(const HsaApi& api, void* dst, hsa_agent_t dst_agent, const void* src, hsa_agent_t src_agent, size_t bytes, bool* stuck) {
  uint32_t mask = 0;
  const hsa_status_t st = api.engine_status(dst_agent, src_agent, &mask);
  if (st != HSA_STATUS_SUCCESS && st != HSA_STATUS_ERROR_OUT_OF_RESOURCES) return 0;
  // engines that are busy right now are missing from the mask; a model that is still loading has no copies in flight
  int woken = 0;
  for (uint32_t bit = 1; bit != 0 && bit <= 0x8000u; bit <<= 1) {
    if (!(mask & bit)) continue;
    hsa_signal_t sig;
    if (api.signal_create(1, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) break;
    if (api.copy_on_engine(dst, dst_agent, src, src_agent, bytes, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)bit, false) ==
        HSA_STATUS_SUCCESS) {
      // bounded wait (10^9 ticks of the HSA system clock, seconds): a copy of a few KB that does not finish is left alone
      if (api.signal_wait(sig, HSA_SIGNAL_CONDITION_LT, 1, 1000000000ull, HSA_WAIT_STATE_BLOCKED) < 1) ++woken;
      else { *stuck = true; break; }   // signal and buffers are leaked on purpose (a few KB, once per process)
    }
    (void)api.signal_destroy(sig);
  }
  return woken;
}

}  // namespace

std::string WakeCopyEngines(int device) {
  static std::mutex mu;
  static std::map<int, std::string> done;
  std::lock_guard<std::mutex> lk(mu);
  auto it = done.find(device);
  if (it != done.end()) return it->second;
  std::string& report = done[device];
  static const HsaApi api;
  if (!api.ok) return report = "skipped: the HSA runtime's entry points were not found in the process";
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(device) != hipSuccess) return report = "skipped: hipSetDevice failed";
  const size_t bytes = 4096;
  void* d = nullptr;
  void* h = nullptr;
  if (hipMalloc(&d, bytes) != hipSuccess || hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess) {
    if (d) (void)hipFree(d);
    (void)hipGetLastError();
    (void)hipSetDevice(prev);
    return report = "skipped: no scratch memory";
  }
  const auto t0 = std::chrono::steady_clock::now();
  hsa_amd_pointer_info_t di, hi;
  di.size = hi.size = sizeof(hsa_amd_pointer_info_t);
  int up = 0, down = 0;
  bool stuck = false;
  if (api.pointer_info(d, &di, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS &&
      api.pointer_info(h, &hi, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS &&
      di.type != HSA_EXT_POINTER_TYPE_UNKNOWN && hi.type != HSA_EXT_POINTER_TYPE_UNKNOWN) {
    up = WakeDirection(api, d, di.agentOwner, h, hi.agentOwner, bytes, &stuck);
    if (!stuck) down = WakeDirection(api, h, hi.agentOwner, d, di.agentOwner, bytes, &stuck);
  }
  const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (!stuck) {
    (void)hipFree(d);
    (void)hipHostFree(h);
  }
  (void)hipSetDevice(prev);
  std::ostringstream os;
  os << up << " engines host->device, " << down << " device->host, " << ms << " ms" << (stuck ? " (a copy did not finish: scratch buffers abandoned)" : "");
  return report = os.str();
}

}  // namespace hps
