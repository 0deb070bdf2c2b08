// First contact with a multi-GPU machine: is every pair of GPUs reachable the two ways the table-sharded lookup uses
// (shard_entry.h: kernels storing into a peer's memory; copy engines shipping blocks), at what rate per pair, and does RCCL
// come up with one rank per GPU (shard_session.h)?  Everything runs on a thread of its own behind a deadline: a peer store or a
// collective that hangs cannot be interrupted, but it can be NAMED — the report then says which step did not come back.
// Not in the reference (replicas only, docs/architecture.md:11,29); bench.py runs it before its config-3 legs.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace hps {

// dst[i] = seed + i for i in [0, words): plain stores for the 4-KB check, nontemporal 16-byte stores for the bandwidth probe
hipError_t LaunchProbeStore(uint32_t* dst, uint64_t words, uint32_t seed, hipStream_t stream);

// JSON object (one line):
//   {"devices":[..], "peer_access":[[..]..], "store_4k_ok":[[..]..], "store_GBps":[[..]..], "copy_GBps":[[..]..],
//    "pair_GBps_min":{"store":x,"copy":y}, "pair_GBps_median":{..}, "rccl_allreduce":{"ranks":n,"ok":true,"ms":..},
//    "timeout":false, "stuck_in":null, "seconds":..}
// Matrices are [from][to] over `devices` (distinct device ids); entries for from == to are null.  probe_bytes per transfer
// (64 MB is enough to see the link rate).  Returns false when the deadline passed (the JSON then carries "timeout":true and the step).
bool MultiGpuSelfTest(const std::vector<int>& devices, uint64_t probe_bytes, uint32_t timeout_ms, bool with_rccl, std::string* json);

}  // namespace hps
