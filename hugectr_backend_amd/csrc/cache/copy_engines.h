// SDMA copy engines of a GPU, woken up once per process and device.
//
// ROCr creates the queue of an SDMA engine the first time a copy is routed to that engine, and the HIP runtime routes a
// copy to whichever engine `hsa_amd_memory_copy_engine_status` reports free at that moment.  With two lookup sessions
// uploading at the same time, the runtime keeps discovering "new" engines during the first dozens of requests, and
// each discovery blocks that `hipMemcpyAsync` — and every other HIP call of the process — for 7-12 ms
// (rocprofv3 API trace, profiles/round2/slow_api_calls_before_engine_warmup.txt: hsa_amd_memory_async_copy_on_engine
// 8.6-11.8 ms, hsa_amd_memory_copy_engine_status 6.7-8.3 ms behind the same lock).  WakeCopyEngines sends one tiny copy
// through every engine in both directions while the model is still loading, so no request pays for it.
#pragma once
#include <string>

namespace hps {

// Best effort: returns a one-line report ("12 engines host->device, 12 device->host, 38.2 ms" or why it was skipped).
// Safe to call from several threads; the work is done once per device.
std::string WakeCopyEngines(int device);

}  // namespace hps
