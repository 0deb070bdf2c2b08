#include "model_instance_state.h"

#include <hip/hip_runtime_api.h>

#include <cstring>

#include "../cache/segcopy_kernels.h"

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
  if (h_result_) (void)hipHostFree(h_result_);
  if (rows_stream_) { (void)hipSetDevice(device_id_); (void)hipStreamDestroy(rows_stream_); }
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

TRITONSERVER_Error* ModelInstanceState::EnsureDeviceResult(size_t elems) {
  if (d_result_elems_ >= elems) return nullptr;
  const InferenceParams& p = model_state_->Params();
  if (hipSetDevice(device_id_) != hipSuccess) return HPS_TRITON_ERROR(INTERNAL, "hipSetDevice(", device_id_, ") failed");
  if (d_result_) (void)hipFree(d_result_);
  d_result_ = nullptr;
  d_result_elems_ = 0;
  size_t full = 0;   // a full request's output
  for (size_t t = 0; t < p.num_tables(); ++t) full += p.embedding_vecsize_per_table[t] * p.maxnum_catfeature_query_per_table_per_sample[t];
  const size_t want = std::max(elems, (size_t)model_state_->MaxBatch() * full);
  if (hipMalloc((void**)&d_result_, want * sizeof(float)) != hipSuccess)
    return HPS_TRITON_ERROR(INTERNAL, "failed to allocate the lookup result buffer (", want * sizeof(float), " bytes)");
  d_result_elems_ = want;
  return nullptr;
}

bool ModelInstanceState::CanCoalesce(const std::vector<const RequestSlice*>& slices) const {
  if (slices.size() < 2) return false;
  const size_t capacity = (size_t)model_state_->MaxBatch() * (size_t)model_state_->KeysPerSample();
  size_t keys = 0, elems = 0;
  const bool gpu = model_state_->UsesGpuCache();
  for (const RequestSlice* s : slices) {
    for (size_t n : s->num_keys_per_table) keys += n;
    elems += s->out_elems;
    if (!gpu && s->out_on_device) return false;
  }
  if (keys == 0 || keys > capacity) return false;
  // every row makes one more trip through memory (result buffer -> the request's output): worth it while that costs less
  // than the engine calls it saves (~0.1 ms each; 64 MB move in ~25 us)
  return elems * sizeof(float) <= (slices.size() - 1) * ((size_t)64 << 20);
}

TRITONSERVER_Error* ModelInstanceState::ProcessCoalesced(const std::vector<const RequestSlice*>& slices) {
  const InferenceParams& p = model_state_->Params();
  const size_t T = p.num_tables(), R = slices.size();
  const bool gpu = model_state_->UsesGpuCache();
  // ---- keys: table by table, request by request; rows come back in the same order ----
  std::vector<size_t> n(T, 0);
  size_t total_keys = 0, total_elems = 0;
  for (const RequestSlice* s : slices)
    for (size_t t = 0; t < T; ++t) { n[t] += s->num_keys_per_table[t]; total_keys += s->num_keys_per_table[t]; }
  for (size_t t = 0; t < T; ++t) total_elems += n[t] * p.embedding_vecsize_per_table[t];
  if (merged_keys_.size() < total_keys) merged_keys_.resize(total_keys);
  std::vector<const void*> keys_per_table(T);
  {
    size_t w = 0;
    std::vector<size_t> roff(R, 0);   // read offset inside each request's flat KEYS
    for (size_t t = 0; t < T; ++t) {
      keys_per_table[t] = merged_keys_.data() + w;
      for (size_t r = 0; r < R; ++r) {
        const size_t c = slices[r]->num_keys_per_table[t];
        if (c) memcpy(merged_keys_.data() + w, slices[r]->keys + roff[r], c * sizeof(int64_t));
        roff[r] += c;
        w += c;
      }
    }
  }
  // ---- one lookup into the instance's result buffer ----
  float* result = nullptr;
  if (gpu) {
    RETURN_IF_ERROR(EnsureDeviceResult(total_elems));
    result = d_result_;
  } else {
    if (cpu_result_.size() < total_elems) cpu_result_.resize(total_elems);
    result = cpu_result_.data();
  }
  std::vector<float*> out_per_table(T);
  {
    size_t o = 0;
    for (size_t t = 0; t < T; ++t) { out_per_table[t] = result + o; o += n[t] * p.embedding_vecsize_per_table[t]; }
  }
  if (sharded_entry_) RETURN_IF_STATUS_ERROR(sharded_entry_->lookup(keys_per_table.data(), out_per_table.data(), n.data(), T));
  else RETURN_IF_STATUS_ERROR(lookupsession_->lookup(keys_per_table.data(), out_per_table.data(), n.data(), T));
  ++coalesced_calls_;
  // ---- every request's rows to its own output buffer (request r: table-major, as for a call of its own) ----
  bool any_host_out = false;
  for (const RequestSlice* s : slices) any_host_out |= gpu && !s->out_on_device && s->out_elems;
  const float* host_view = result;   // where host copies read from
  if (any_host_out) {
    if (h_result_elems_ < total_elems) {
      if (h_result_) (void)hipHostFree(h_result_);
      h_result_ = nullptr;
      h_result_elems_ = 0;
      if (hipHostMalloc((void**)&h_result_, total_elems * sizeof(float), hipHostMallocDefault) != hipSuccess)
        return HPS_TRITON_ERROR(INTERNAL, "failed to allocate the page-locked result buffer (", total_elems * sizeof(float), " bytes)");
      h_result_elems_ = total_elems;
    }
    if (hipSetDevice(device_id_) != hipSuccess ||
        hipMemcpy(h_result_, d_result_, total_elems * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
      return HPS_TRITON_ERROR(INTERNAL, "failed to copy the lookup result to host memory");
    host_view = h_result_;
  }
  std::vector<const void*> seg_src;
  std::vector<void*> seg_dst;
  std::vector<uint64_t> seg_bytes;
  std::vector<size_t> woff(R, 0);     // write offset inside each request's OUTPUT0
  size_t table_base = 0;
  for (size_t t = 0; t < T; ++t) {
    const size_t D = p.embedding_vecsize_per_table[t];
    size_t in_table = 0;
    for (size_t r = 0; r < R; ++r) {
      const size_t elems = slices[r]->num_keys_per_table[t] * D;
      if (elems) {
        const size_t from = table_base + in_table;
        if (gpu && slices[r]->out_on_device) {
          seg_src.push_back(d_result_ + from);
          seg_dst.push_back(slices[r]->out + woff[r]);
          seg_bytes.push_back(elems * sizeof(float));
        } else {
          memcpy(slices[r]->out + woff[r], host_view + from, elems * sizeof(float));
        }
      }
      woff[r] += elems;
      in_table += elems;
    }
    table_base += n[t] * D;
  }
  if (!seg_src.empty()) {
    if (hipSetDevice(device_id_) != hipSuccess) return HPS_TRITON_ERROR(INTERNAL, "hipSetDevice(", device_id_, ") failed");
    // A stream of the instance's own, NOT the session's: the session's stream may still hold the insert kernel the lookup left
    // running behind it (and that kernel waits for other sessions' gathers to release the cache) — the rows in the result buffer
    // are complete (the lookup returned), nothing else has to be waited for.
    if (!rows_stream_) {
      if (hipStreamCreateWithFlags(&rows_stream_, hipStreamNonBlocking) != hipSuccess) return HPS_TRITON_ERROR(INTERNAL, "failed to create the instance's copy stream");
    }
    hipError_t e = LaunchSegmentedCopy(seg_src.data(), seg_dst.data(), seg_bytes.data(), seg_src.size(), rows_stream_);
    if (e == hipSuccess) e = hipStreamSynchronize(rows_stream_);
    if (e != hipSuccess) return HPS_TRITON_ERROR(INTERNAL, "failed to move a coalesced call's rows to the requests' output buffers: ", hipGetErrorString(e));
  }
  return nullptr;
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
    RETURN_IF_ERROR(EnsureDeviceResult(out_elems));
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
