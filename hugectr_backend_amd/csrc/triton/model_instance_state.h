// Per-instance state: one lookup session (worker buffers + stream) and the staging the shell needs.
// Counterpart of the reference's ModelInstanceState (/root/reference/hps_backend/include/model_instance_state.hpp,
// src/model_instance_state.cpp).
#pragma once
#include <hip/hip_runtime_api.h>

#include <memory>
#include <string>
#include <vector>

#include "../cache/shard_entry.h"
#include "model_state.h"

namespace hps { namespace triton {

class ModelInstanceState {
 public:
  static TRITONSERVER_Error* Create(ModelState* model_state, TRITONBACKEND_ModelInstance* triton_model_instance,
                                    ModelInstanceState** state);                          // model_instance_state.cpp:44-70
  ~ModelInstanceState();

  // Fetch the model's cache for this instance's device and create the lookup session.   model_instance_state.cpp:162-174
  TRITONSERVER_Error* LoadHPSInstance();

  // One request: slice the flat KEYS / OUTPUT0 buffers per table and run the lookup.     model_instance_state.cpp:177-197
  //   keys        flat table-major int64; host memory unless keys_on_device
  //   out         fp32 output; device memory when the model uses the GPU cache and out_on_device,
  //               host memory otherwise (gpucache models then go through the instance's device buffer)
  TRITONSERVER_Error* ProcessRequest(const int64_t* keys, bool keys_on_device, const std::vector<size_t>& num_keys_per_table,
                                     float* out, bool out_on_device, size_t out_elems);

  // Several requests of ONE TRITONBACKEND_ModelInstanceExecute call served by ONE engine call (the reference runs one blocking
  // lookup per request, hps.cc:406: a dynamically batched Execute of eight small requests pays eight call overheads).  The
  // requests' per-table key slices are concatenated table by table (host staging), looked up into the instance's result buffer,
  // and every request's rows are moved to its own output buffer (one segmented device copy, or host copies).
  struct RequestSlice {
    const int64_t* keys = nullptr;              // flat table-major, HOST memory
    std::vector<size_t> num_keys_per_table;
    float* out = nullptr;
    bool out_on_device = false;
    size_t out_elems = 0;
  };
  // true when the slices can go as one call: host keys, together within the session's capacity, rows few enough that the
  // extra pass over them costs less than the calls it saves
  bool CanCoalesce(const std::vector<const RequestSlice*>& slices) const;
  TRITONSERVER_Error* ProcessCoalesced(const std::vector<const RequestSlice*>& slices);
  uint64_t CoalescedCalls() const { return coalesced_calls_; }

  const std::string& Name() const { return name_; }
  int32_t DeviceId() const { return device_id_; }
  TRITONSERVER_InstanceGroupKind Kind() const { return kind_; }
  ModelState* StateForModel() const { return model_state_; }
  TRITONBACKEND_ModelInstance* TritonModelInstance() { return triton_model_instance_; }
  const InferenceParams& Params() const { return model_state_->Params(); }
  size_t NumTables() const { return model_state_->Params().num_tables(); }
  // host staging for a NUMKEYS tensor that arrives in pieces or in device memory: one int32 per table
  std::vector<int32_t>& CountStaging(size_t tables) { if (count_staging_.size() < tables) count_staging_.resize(tables); return count_staging_; }
  // host staging for KEYS that arrive in several buffers (or in device memory for a CPU-only model)
  int64_t* KeyStaging(size_t count);
  // ... for the requests of one Execute call: reserved once for all of them (growing it would move what earlier requests staged)
  void ReserveKeyStaging(size_t count) { if (key_staging_.size() < count) key_staging_.resize(count); }
  int64_t* KeyStagingAt(size_t offset) { return key_staging_.data() + offset; }

 private:
  ModelInstanceState(ModelState* model_state, TRITONBACKEND_ModelInstance* inst, const char* name,
                     TRITONSERVER_InstanceGroupKind kind, int32_t device_id)
      : model_state_(model_state), triton_model_instance_(inst), name_(name), kind_(kind), device_id_(device_id) {}

  ModelState* model_state_;
  TRITONBACKEND_ModelInstance* triton_model_instance_;
  std::string name_;
  TRITONSERVER_InstanceGroupKind kind_;
  int32_t device_id_;
  std::shared_ptr<EmbeddingCache> embedding_cache_;
  std::unique_ptr<LookupSession> lookupsession_;
  // ps.json "table_sharding": "hash": the instance serves whole requests over the model's shards (cache/shard_entry.h)
  std::unique_ptr<ShardedEntrySession> sharded_entry_;
  std::vector<int64_t> key_staging_;
  std::vector<int32_t> count_staging_;
  float* d_result_ = nullptr;  // device result buffer, only when Triton hands out a host output buffer
  size_t d_result_elems_ = 0;
  TRITONSERVER_Error* EnsureDeviceResult(size_t elems);
  std::vector<int64_t> merged_keys_;      // coalesced call: the requests' keys, table-major over the requests
  float* h_result_ = nullptr;             // page-locked: rows of a coalesced call on their way to host output buffers
  size_t h_result_elems_ = 0;
  std::vector<float> cpu_result_;         // coalesced call of a model without GPU cache
  uint64_t coalesced_calls_ = 0;
  hipStream_t rows_stream_ = nullptr;     // the segmented copy of a coalesced call's rows (created at its first use)
};

}}  // namespace hps::triton
