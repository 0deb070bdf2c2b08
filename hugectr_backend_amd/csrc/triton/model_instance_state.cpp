#include "model_instance_state.h"

#include <hip/hip_runtime_api.h>

namespace hps { namespace triton {

TRITONSERVER_Error* ModelInstanceState::Create(ModelState* model_state,
                                               TRITONBACKEND_ModelInstance* triton_model_instance,
                                               ModelInstanceState** state) {
  const char* instance_name;
  RETURN_IF_ERROR(TRITONBACKEND_ModelInstanceName(triton_model_instance, &instance_name));
  TRITONSERVER_InstanceGroupKind instance_kind;
  RETURN_IF_ERROR(TRITONBACKEND_ModelInstanceKind(triton_model_instance, &instance_kind));
  int32_t device_id;
  RETURN_IF_ERROR(TRITONBACKEND_ModelInstanceDeviceId(triton_model_instance, &device_id));
  // The reference ends up using the LAST device of the instance group for every instance
  // (model_instance_state.cpp:65-67 passes instance_params.device_id, left at gpu_shape.back() by
  // model_state.cpp:409); the documented intent is one session per listed GPU (docs/architecture.md:11),
  // so Triton's device id is used here.
  if (!model_state->UsesGpuCache()) device_id = 0;
  *state = new ModelInstanceState(model_state, triton_model_instance, instance_name, instance_kind, device_id);
  return nullptr;
}

ModelInstanceState::~ModelInstanceState() {
  sharded_entry_.reset();
  lookupsession_.reset();
  embedding_cache_.reset();
  if (d_result_) {
    (void)hipSetDevice(device_id_);
    (void)hipFree(d_result_);
  }
}

TRITONSERVER_Error* ModelInstanceState::LoadHPSInstance() {
  if (model_state_->UsesGpuCache() && model_state_->Params().table_sharding) {
    // A table-sharded model (BASELINE config 3): Triton still hands this instance whole requests and blocks on them
    // (hps.cc:353-369, 406), so the instance is an ENTRY session: it buckets a request by owner and the shards' lookup
    // sessions write their rows into this instance's output buffer over the peer mappings.
    RETURN_IF_STATUS_ERROR(ShardedEntrySession::Create(model_state_->Server(), model_state_->Name(), device_id_, &sharded_entry_));
    HPS_TRITON_LOG(INFO, "instance ", name_, ": entry session of the table-sharded model on device ", device_id_, ", ",
                   sharded_entry_->num_shards(), " shards, ", sharded_entry_->shard_capacity(), " keys per shard and pass");
    return nullptr;
  }
  if (model_state_->UsesGpuCache()) {
    embedding_cache_ = model_state_->CacheOn(device_id_);
    if (!embedding_cache_)
      return HPS_TRITON_ERROR(INVALID_ARG, "model ", model_state_->Name(), " has no embedding cache on device ", device_id_,
                              "; list the device in the instance_group 'gpus' and in 'deployed_device_list'");
  }
  RETURN_IF_STATUS_ERROR(model_state_->Server()->create_lookup_session(model_state_->Name(), embedding_cache_,
                                                                                 &lookupsession_));
  HPS_TRITON_LOG(INFO, "instance ", name_, ": lookup session ready on device ", device_id_);
  return nullptr;
}

int64_t* ModelInstanceState::KeyStaging(size_t count) {
  if (key_staging_.size() < count) key_staging_.resize(count);
  return key_staging_.data();
}

TRITONSERVER_Error* ModelInstanceState::ProcessRequest(const int64_t* keys, bool keys_on_device,
                                                       const std::vector<size_t>& num_keys_per_table, float* out,
                                                       bool out_on_device, size_t out_elems) {
  const InferenceParams& p = model_state_->Params();
  const size_t T = num_keys_per_table.size();
  const bool gpu = model_state_->UsesGpuCache();

  float* result = out;
  if (gpu && !out_on_device) {
    // Triton gave host memory for the output of a GPU-cache model: look up into the instance's device
    // buffer, then one D2H copy (the reference always does this extra hop, hps.cc:676-691).
    if (d_result_elems_ < out_elems) {
      if (hipSetDevice(device_id_) != hipSuccess) return HPS_TRITON_ERROR(INTERNAL, "hipSetDevice(", device_id_, ") failed");
      if (d_result_) (void)hipFree(d_result_);
      d_result_ = nullptr;
      const size_t want = std::max(out_elems, (size_t)model_state_->MaxBatch() *
                                                  [&] { size_t s = 0; for (size_t t = 0; t < p.num_tables(); ++t) s += p.embedding_vecsize_per_table[t] * p.maxnum_catfeature_query_per_table_per_sample[t]; return s; }());
      if (hipMalloc((void**)&d_result_, want * sizeof(float)) != hipSuccess)
        return HPS_TRITON_ERROR(INTERNAL, "failed to allocate the lookup result buffer (", want * sizeof(float), " bytes)");
      d_result_elems_ = want;
    }
    result = d_result_;
  }
  if (!gpu && out_on_device)
    return HPS_TRITON_ERROR(UNSUPPORTED, "model ", model_state_->Name(),
                            " runs without GPU cache: its output must be in host memory");

  // keys_t = keys + sum_{u<t} n_u ; out_t = out + sum_{u<t} D_u * n_u   (model_instance_state.cpp:180-193)
  std::vector<const void*> keys_per_table(T);
  std::vector<float*> out_per_table(T);
  size_t koff = 0, ooff = 0;
  for (size_t t = 0; t < T; ++t) {
    keys_per_table[t] = keys + koff;
    out_per_table[t] = result + ooff;
    koff += num_keys_per_table[t];
    ooff += num_keys_per_table[t] * p.embedding_vecsize_per_table[t];
  }
  if (sharded_entry_) {
    if (keys_on_device) RETURN_IF_STATUS_ERROR(sharded_entry_->lookup_from_device(keys, out_per_table.data(), num_keys_per_table.data(), T));
    else RETURN_IF_STATUS_ERROR(sharded_entry_->lookup(keys_per_table.data(), out_per_table.data(), num_keys_per_table.data(), T));
  } else if (keys_on_device) {
    if (!gpu) return HPS_TRITON_ERROR(INTERNAL, "device-resident KEYS reached a host-only lookup session");
    RETURN_IF_STATUS_ERROR(lookupsession_->lookup_from_device(keys, out_per_table.data(), num_keys_per_table.data(), T));
  } else {
    RETURN_IF_STATUS_ERROR(lookupsession_->lookup(keys_per_table.data(), out_per_table.data(), num_keys_per_table.data(), T));
  }
  if (gpu && !out_on_device && out_elems) {
    if (hipSetDevice(device_id_) != hipSuccess ||
        hipMemcpy(out, d_result_, out_elems * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
      return HPS_TRITON_ERROR(INTERNAL, "failed to copy the lookup result to the host output buffer");
  }
  return nullptr;
}

}}  // namespace hps::triton
