#include "model_state.h"

#include <algorithm>

namespace hps { namespace triton {

namespace {

TRITONSERVER_Error* ParseDims(const Json& obj, const char* key, std::vector<int64_t>* out) {
  const Json* d = obj.Find(key);
  if (!d || !d->is_array()) return HPS_TRITON_ERROR(INVALID_ARG, "model config: '", key, "' must be an array");
  out->clear();
  for (size_t i = 0; i < d->size(); ++i) {
    int64_t v;
    if (!d->at(i).AsInt(&v)) return HPS_TRITON_ERROR(INVALID_ARG, "model config: '", key, "' must hold integers");
    out->push_back(v);
  }
  return nullptr;
}

std::string ShapeToString(const std::vector<int64_t>& s) {
  std::string o = "[";
  for (size_t i = 0; i < s.size(); ++i) { if (i) o += ","; o += std::to_string(s[i]); }
  return o + "]";
}

}  // namespace

ModelState::ModelState(TRITONBACKEND_Model* triton_model, const char* name, uint64_t version, uint64_t version_ps,
                       Json&& model_config, std::shared_ptr<HierParameterServer> ps, const InferenceParams& params)
    : triton_model_(triton_model), name_(name), version_(version), version_ps_(version_ps),
      model_config_(std::move(model_config)), ps_(std::move(ps)), params_(params) {}

TRITONSERVER_Error* ModelState::Create(TRITONBACKEND_Model* triton_model, ModelState** state,
                                       std::shared_ptr<HierParameterServer> ps, const InferenceParams& params,
                                       uint64_t model_ps_version) {
  TRITONSERVER_Message* config_message;
  RETURN_IF_ERROR(TRITONBACKEND_ModelConfig(triton_model, 1 /* config_version */, &config_message));
  const char* buffer;
  size_t byte_size;
  RETURN_IF_ERROR(TRITONSERVER_MessageSerializeToJson(config_message, &buffer, &byte_size));
  Json model_config;
  std::string perr;
  const bool ok = Json::Parse(std::string(buffer, byte_size), &model_config, &perr);
  RETURN_IF_ERROR(TRITONSERVER_MessageDelete(config_message));  // model_state.cpp:89: the plugin owns this message
  if (!ok) return HPS_TRITON_ERROR(INVALID_ARG, "failed to parse the model configuration: ", perr);

  const char* model_name;
  RETURN_IF_ERROR(TRITONBACKEND_ModelName(triton_model, &model_name));
  uint64_t model_version;
  RETURN_IF_ERROR(TRITONBACKEND_ModelVersion(triton_model, &model_version));
  TRITONSERVER_Server* triton_server;
  RETURN_IF_ERROR(TRITONBACKEND_ModelServer(triton_model, &triton_server));
  (void)triton_server;

  *state = new ModelState(triton_model, model_name, model_version, model_ps_version, std::move(model_config),
                          std::move(ps), params);
  return nullptr;
}

ModelState::~ModelState() {
  timer_.stop();  // joins the refresh threads before the caches go away
  if (support_gpu_cache_ && version_ps_ == version_) {
    // only the state of the latest loaded version tears the caches down (model_state.cpp:110-115)
    (void)ps_->destory_embedding_cache_per_model(name_);
    HPS_TRITON_LOG(INFO, "******Destorying Embedding Cache for model ", name_, " successfully");
  }
  embedding_cache_map_.clear();
}

std::shared_ptr<EmbeddingCache> ModelState::GetEmbeddingCache(int64_t device_id) {
  auto it = embedding_cache_map_.find(device_id);
  return it == embedding_cache_map_.end() ? nullptr : it->second;
}

void ModelState::EmbeddingCacheRefresh(const std::string& model_name, int device_id) {
  HPS_TRITON_LOG(INFO, "The model ", model_name, " is refreshing the embedding cache asynchronously on device ",
                 device_id, ".");
  if (!freeze_embedding_) {
    const Status st = ps_->update_database_per_model(params_);
    if (!st.ok()) HPS_TRITON_LOG(ERROR, "update_database_per_model failed: ", st.message());
  }
  if (support_gpu_cache_) {
    const Status st = ps_->refresh_embedding_cache(model_name, device_id);
    if (!st.ok()) HPS_TRITON_LOG(ERROR, "refresh_embedding_cache failed: ", st.message());
  }
  HPS_TRITON_LOG(INFO, "The model ", model_name,
                 " has completed the asynchronous refresh of the embedding cache on device ", device_id, ".");
}

void ModelState::Refresh_Embedding_Cache() {
  const uint64_t t0 = NowNs();
  for (int64_t dev : gpu_shape_) {
    if (!support_gpu_cache_) continue;
    HPS_TRITON_LOG(INFO, "The model ", name_, " is periodically refreshing the embedding cache asynchronously on device ", dev);
    const Status st = ps_->refresh_embedding_cache(name_, (int)dev);
    if (!st.ok()) HPS_TRITON_LOG(ERROR, "refresh_embedding_cache failed: ", st.message());
    else HPS_TRITON_LOG(INFO, "The model ", name_, " has refreshed the embedding cache asynchronously on device ", dev);
  }
  HPS_TRITON_LOG(INFO, "Refresh embedding table execution time is ", (NowNs() - t0) / 1000000, " ms");
}

TRITONSERVER_Error* ModelState::ValidateModelConfig() {
  HPS_TRITON_LOG(INFO, "Verifying model configuration: ", model_config_.Dump());
  // exactly two inputs: KEYS (TYPE_INT64) and NUMKEYS (TYPE_INT32), both with dims[0] == -1
  {
    const Json* inputs = model_config_.Find("input");
    if (!inputs || !inputs->is_array()) return HPS_TRITON_ERROR(INVALID_ARG, "model config: 'input' must be an array");
    if (inputs->size() != 2) return HPS_TRITON_ERROR(INVALID_ARG, "expect 2 input, got ", inputs->size());
    for (size_t i = 0; i < 2; ++i) {
      const Json& input = inputs->at(i);
      std::string name, data_type;
      RETURN_IF_STATUS_ERROR(ParseField(name, input, "name", true));
      if (name != "KEYS" && name != "NUMKEYS")
        return HPS_TRITON_ERROR(INVALID_ARG, "expected input name as KEYS and NUMKEYS, but got ", name);
      RETURN_IF_STATUS_ERROR(ParseField(data_type, input, "data_type", true));
      if (name == "KEYS" && data_type != "TYPE_INT64")
        return HPS_TRITON_ERROR(INVALID_ARG, "expected KEYS input datatype as TYPE_INT64, got ", data_type);
      if (name == "NUMKEYS" && data_type != "TYPE_INT32")
        return HPS_TRITON_ERROR(INVALID_ARG, "expected NUMKEYS input datatype as TYPE_INT32, got ", data_type);
      std::vector<int64_t> shape;
      RETURN_IF_ERROR(ParseDims(input, "dims", &shape));
      if (shape.empty() || shape[0] != -1)
        return HPS_TRITON_ERROR(INVALID_ARG, "expected input shape equal -1, got ", ShapeToString(shape));
    }
    std::string n0, n1;
    (void)ParseField(n0, inputs->at(0), "name", true);
    (void)ParseField(n1, inputs->at(1), "name", true);
    if (n0 == n1) return HPS_TRITON_ERROR(INVALID_ARG, "inputs must be one KEYS and one NUMKEYS, got two ", n0);
  }
  // exactly one output, TYPE_FP32, dims[0] == -1
  {
    const Json* outputs = model_config_.Find("output");
    if (!outputs || !outputs->is_array()) return HPS_TRITON_ERROR(INVALID_ARG, "model config: 'output' must be an array");
    if (outputs->size() != 1) return HPS_TRITON_ERROR(INVALID_ARG, "expect 1 output, got ", outputs->size());
    const Json& output = outputs->at(0);
    std::string data_type;
    RETURN_IF_STATUS_ERROR(ParseField(data_type, output, "data_type", true));
    if (data_type != "TYPE_FP32")
      return HPS_TRITON_ERROR(INVALID_ARG, "expected  output datatype as TYPE_FP32, got ", data_type);
    std::vector<int64_t> shape;
    RETURN_IF_ERROR(ParseDims(output, "dims", &shape));
    if (shape.empty() || shape[0] != -1)
      return HPS_TRITON_ERROR(INVALID_ARG, "expected  output shape equal -1, got ", ShapeToString(shape));
  }
  return nullptr;
}

TRITONSERVER_Error* ModelState::ParseModelConfig() {
  const Json* instance_group = model_config_.Find("instance_group");
  if (!instance_group || !instance_group->is_array() || instance_group->size() == 0)
    return HPS_TRITON_ERROR(INVALID_ARG, "expect at least one instance in instance group , got ",
                            instance_group && instance_group->is_array() ? instance_group->size() : 0);
  support_gpu_cache_ = params_.use_gpu_embedding_cache;
  gpu_shape_.clear();
  for (size_t i = 0; i < instance_group->size(); ++i) {
    const Json& instance = instance_group->at(i);
    std::string kind;
    RETURN_IF_STATUS_ERROR(ParseField(kind, instance, "kind", true));
    if (support_gpu_cache_) {
      if (kind != "KIND_GPU")
        return HPS_TRITON_ERROR(INVALID_ARG, "expect GPU kind instance in instance group , got ", kind);
      std::vector<int64_t> gpu_list;
      RETURN_IF_ERROR(ParseDims(instance, "gpus", &gpu_list));
      for (int64_t id : gpu_list)
        if (std::find(gpu_shape_.begin(), gpu_shape_.end(), id) == gpu_shape_.end()) gpu_shape_.push_back(id);
    } else if (gpu_shape_.empty()) {
      gpu_shape_.push_back(0);
    }
    int64_t count = 1;
    RETURN_IF_STATUS_ERROR(ParseField(count, instance, "count", false));
    if (count > params_.number_of_worker_buffers_in_pool)
      return HPS_TRITON_ERROR(INVALID_ARG,
                              "expect the number of instance(in instance_group) not larger than "
                              "num_of_worker_buffer_in_pool that configured in Parameter Server json file , got ",
                              count);
  }

  // per-model parameters: refresh_interval / refresh_delay / freeze_sparse (README.md:169-180)
  refresh_interval_ = params_.refresh_interval;
  refresh_delay_ = params_.refresh_delay;
  if (const Json* parameters = model_config_.Find("parameters")) {
    if (const Json* v = parameters->Find("refresh_interval"))
      RETURN_IF_STATUS_ERROR(ParseField(refresh_interval_, *v, "string_value", false));
    if (const Json* v = parameters->Find("refresh_delay"))
      RETURN_IF_STATUS_ERROR(ParseField(refresh_delay_, *v, "string_value", false));
    if (const Json* v = parameters->Find("freeze_sparse"))
      RETURN_IF_STATUS_ERROR(ParseField(freeze_embedding_, *v, "string_value", false));
  }
  HPS_TRITON_LOG(INFO, "refresh_interval = ", refresh_interval_, ", refresh_delay = ", refresh_delay_,
                 ", freeze_sparse = ", freeze_embedding_);

  cat_num_ = 0;
  for (size_t c : params_.maxnum_catfeature_query_per_table_per_sample) cat_num_ += (int64_t)c;
  if (cat_num_ <= 0) return HPS_TRITON_ERROR(INVALID_ARG, "expected at least one categorical feature, got ", cat_num_);
  embedding_size_ = 0;
  for (size_t d : params_.embedding_vecsize_per_table) embedding_size_ += (int64_t)d;

  int64_t cfg_max_batch = 0;
  RETURN_IF_STATUS_ERROR(ParseField(cfg_max_batch, model_config_, "max_batch_size", false));
  if (cfg_max_batch < 0)
    return HPS_TRITON_ERROR(INVALID_ARG, "expected max_batch_size should greater than or equal to 0 ",
                            "(the configuration should be consistent in Parameter Server json file and config.pbtxt "
                            "file), got ", cfg_max_batch);
  max_batch_size_ = (int64_t)params_.max_batchsize;  // ps.json wins (model_state.cpp:366)
  HPS_TRITON_LOG(INFO, "max_batch_size is ", max_batch_size_, " (ps.json); config.pbtxt says ", cfg_max_batch);
  return nullptr;
}

TRITONSERVER_Error* ModelState::Create_EmbeddingCache() {
  if (!support_gpu_cache_ && ps_->tables_of(name_).empty()) {
    // CPU-only model deployed online: its tables were not part of the start-up load.  (The reference only
    // reloads the database on this path for GPU-cache models, model_state.cpp:377-393.)
    HPS_TRITON_LOG(INFO, "Update Database of Parameter Server for model ", name_);
    RETURN_IF_STATUS_ERROR(ps_->update_database_per_model(params_));
  }
  if (!gpu_shape_.empty() && support_gpu_cache_) {
    if (ps_->get_embedding_cache(name_, (int)gpu_shape_[0]) == nullptr &&
        embedding_cache_map_.find(gpu_shape_[0]) == embedding_cache_map_.end()) {
      // model deployed online: its tables and caches do not exist yet (model_state.cpp:378-393)
      HPS_TRITON_LOG(INFO, "Update Database of Parameter Server for model ", name_);
      RETURN_IF_STATUS_ERROR(ps_->update_database_per_model(params_));
      HPS_TRITON_LOG(INFO, "Create embedding cache for model ", name_);
      RETURN_IF_STATUS_ERROR(ps_->create_embedding_cache_per_model(params_));
    }
  }
  for (int64_t dev : gpu_shape_) {
    if (support_gpu_cache_ &&
        std::find(params_.deployed_devices.begin(), params_.deployed_devices.end(), (int)dev) ==
            params_.deployed_devices.end())
      return HPS_TRITON_ERROR(INVALID_ARG, "Please confirm that device ", dev,
                              " is added to 'deployed_device_list' in the ps configuration file");
    if (embedding_cache_map_.find(dev) == embedding_cache_map_.end()) {
      if (support_gpu_cache_) {
        HPS_TRITON_LOG(INFO, "******Creating Embedding Cache for model ", name_, " in device ", dev);
        auto cache = ps_->get_embedding_cache(name_, (int)dev);
        if (!cache) return HPS_TRITON_ERROR(INTERNAL, "no embedding cache for model ", name_, " on device ", dev);
        embedding_cache_map_[dev] = std::move(cache);
      }
      if (version_ps_ > 0 && version_ps_ != version_) {
        // a different version of this model was serving before: refresh once, asynchronously
        timer_.startonce(refresh_delay_, [this, dev] { EmbeddingCacheRefresh(name_, (int)dev); });
      }
    }
  }
  if (refresh_interval_ > 1e-6f) timer_.start(refresh_interval_, [this] { Refresh_Embedding_Cache(); });
  HPS_TRITON_LOG(INFO, "******Creating Embedding Cache for model ", name_, " successfully");
  return nullptr;
}

}}  // namespace hps::triton
