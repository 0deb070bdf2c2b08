#include "model_state.h"

#include <algorithm>

namespace hps { namespace triton {

namespace {

// One tensor of the backend's fixed interface: what config.pbtxt has to declare for it.
struct TensorRule {
  const char* name;       // nullptr: any name (the output is addressed by position)
  const char* data_type;
};
constexpr TensorRule kInputs[] = {{"KEYS", "TYPE_INT64"}, {"NUMKEYS", "TYPE_INT32"}};
constexpr TensorRule kOutput = {nullptr, "TYPE_FP32"};

std::string DimsText(const std::vector<int64_t>& dims) {
  std::string s = "[";
  for (size_t i = 0; i < dims.size(); ++i) s += (i ? ", " : "") + std::to_string(dims[i]);
  return s + "]";
}

TRITONSERVER_Error* ReadIntList(const Json& obj, const char* key, const std::string& where, std::vector<int64_t>* out) {
  out->clear();
  const Json* list = obj.Find(key);
  if (!list || !list->is_array()) return HPS_TRITON_ERROR(INVALID_ARG, where, ": '", key, "' has to be a list of integers");
  for (size_t i = 0; i < list->size(); ++i) {
    int64_t v = 0;
    if (!list->at(i).AsInt(&v)) return HPS_TRITON_ERROR(INVALID_ARG, where, ": '", key, "' has to be a list of integers");
    out->push_back(v);
  }
  return nullptr;
}

// data type as declared + variable first dimension (requests carry any number of keys)
TRITONSERVER_Error* CheckTensor(const Json& tensor, const TensorRule& rule, const std::string& where) {
  std::string data_type;
  RETURN_IF_STATUS_ERROR(ParseField(data_type, tensor, "data_type", true));
  if (data_type != rule.data_type)
    return HPS_TRITON_ERROR(INVALID_ARG, where, " must be declared ", rule.data_type, ", the model configuration says ", data_type);
  std::vector<int64_t> dims;
  RETURN_IF_ERROR(ReadIntList(tensor, "dims", where, &dims));
  if (dims.empty() || dims[0] != -1)
    return HPS_TRITON_ERROR(INVALID_ARG, where, " must have a variable first dimension (dims: [-1]), the model configuration says ",
                            DimsText(dims));
  return nullptr;
}

}  // namespace

ModelState::ModelState(TRITONBACKEND_Model* model, std::string name, uint64_t version, uint64_t serving_version, Json&& config,
                       std::shared_ptr<HierParameterServer> server, const InferenceParams& params)
    : triton_model_(model), name_(std::move(name)), version_(version), serving_version_(serving_version),
      config_(std::move(config)), server_(std::move(server)), params_(params) {}

TRITONSERVER_Error* ModelState::Open(TRITONBACKEND_Model* model, std::shared_ptr<HierParameterServer> server,
                                     const InferenceParams& params, uint64_t serving_version, ModelState** out) {
  // The configuration message belongs to the backend once Triton has handed it over: parse, then delete it whatever
  // the parse said.
  TRITONSERVER_Message* message = nullptr;
  RETURN_IF_ERROR(TRITONBACKEND_ModelConfig(model, 1 /* config_version */, &message));
  const char* text = nullptr;
  size_t text_bytes = 0;
  TRITONSERVER_Error* err = TRITONSERVER_MessageSerializeToJson(message, &text, &text_bytes);
  Json config;
  std::string parse_error;
  const bool parsed = err == nullptr && Json::Parse(std::string(text, text_bytes), &config, &parse_error);
  TRITONSERVER_Error* del_err = TRITONSERVER_MessageDelete(message);
  if (err != nullptr) { if (del_err) TRITONSERVER_ErrorDelete(del_err); return err; }
  RETURN_IF_ERROR(del_err);
  if (!parsed) return HPS_TRITON_ERROR(INVALID_ARG, "the model configuration is not valid JSON: ", parse_error);

  const char* name = nullptr;
  RETURN_IF_ERROR(TRITONBACKEND_ModelName(model, &name));
  uint64_t version = 0;
  RETURN_IF_ERROR(TRITONBACKEND_ModelVersion(model, &version));
  TRITONSERVER_Server* triton_server = nullptr;   // asked for as the reference does; nothing here needs the handle
  RETURN_IF_ERROR(TRITONBACKEND_ModelServer(model, &triton_server));
  *out = new ModelState(model, name, version, serving_version, std::move(config), std::move(server), params);
  return nullptr;
}

ModelState::~ModelState() {
  timer_.stop();  // no refresh may be running while the caches go away
  caches_.clear();
  if (gpu_cache_ && serving_version_ == version_) {
    const Status st = server_->destory_embedding_cache_per_model(name_);
    if (st.ok()) HPS_TRITON_LOG(INFO, "model ", name_, " v", version_, ": embedding caches released");
    else HPS_TRITON_LOG(ERROR, "model ", name_, ": releasing the embedding caches failed: ", st.message());
  }
}

std::shared_ptr<EmbeddingCache> ModelState::CacheOn(int64_t device) const {
  const auto it = caches_.find(device);
  return it == caches_.end() ? nullptr : it->second;
}

void ModelState::ReloadThenRefresh(int device) {
  HPS_TRITON_LOG(INFO, "model ", name_, ": version change, bringing the cache on device ", device, " up to date");
  if (!keep_tables_) {
    const Status st = server_->update_database_per_model(params_);
    if (!st.ok()) HPS_TRITON_LOG(ERROR, "model ", name_, ": reloading the sparse files failed: ", st.message());
  }
  if (gpu_cache_) {
    const Status st = server_->refresh_embedding_cache(name_, device);
    if (!st.ok()) HPS_TRITON_LOG(ERROR, "model ", name_, ": cache refresh on device ", device, " failed: ", st.message());
  }
  HPS_TRITON_LOG(INFO, "model ", name_, ": cache on device ", device, " is up to date");
}

void ModelState::RefreshAllCaches() {
  if (!gpu_cache_) return;
  const uint64_t begin = NowNs();
  for (int64_t device : devices_) {
    const Status st = server_->refresh_embedding_cache(name_, (int)device);
    if (!st.ok()) HPS_TRITON_LOG(ERROR, "model ", name_, ": periodic cache refresh on device ", device, " failed: ", st.message());
  }
  HPS_TRITON_LOG(INFO, "model ", name_, ": periodic cache refresh of ", devices_.size(), " device(s) took ",
                 (NowNs() - begin) / 1000000, " ms");
}

TRITONSERVER_Error* ModelState::CheckTensorContract() {
  HPS_TRITON_LOG(VERBOSE, "model ", name_, ": configuration ", config_.Dump());
  const Json* inputs = config_.Find("input");
  constexpr size_t kNumInputs = sizeof(kInputs) / sizeof(kInputs[0]);
  if (!inputs || !inputs->is_array() || inputs->size() != kNumInputs)
    return HPS_TRITON_ERROR(INVALID_ARG, "model ", name_, ": the backend takes exactly the inputs KEYS and NUMKEYS; the model "
                            "configuration declares ", inputs && inputs->is_array() ? inputs->size() : 0, " input(s)");
  bool seen[kNumInputs] = {};
  for (size_t i = 0; i < inputs->size(); ++i) {
    std::string name;
    RETURN_IF_STATUS_ERROR(ParseField(name, inputs->at(i), "name", true));
    size_t r = 0;
    while (r < kNumInputs && name != kInputs[r].name) ++r;
    if (r == kNumInputs || seen[r])
      return HPS_TRITON_ERROR(INVALID_ARG, "model ", name_, ": the inputs must be one KEYS and one NUMKEYS, found '", name, "'",
                              r < kNumInputs ? " twice" : "");
    seen[r] = true;
    RETURN_IF_ERROR(CheckTensor(inputs->at(i), kInputs[r], "input " + name));
  }
  const Json* outputs = config_.Find("output");
  if (!outputs || !outputs->is_array() || outputs->size() != 1)
    return HPS_TRITON_ERROR(INVALID_ARG, "model ", name_, ": the backend produces exactly one output tensor; the model configuration "
                            "declares ", outputs && outputs->is_array() ? outputs->size() : 0);
  return CheckTensor(outputs->at(0), kOutput, "the output");
}

TRITONSERVER_Error* ModelState::ReadDeployment() {
  gpu_cache_ = params_.use_gpu_embedding_cache;
  devices_.clear();
  const Json* groups = config_.Find("instance_group");
  if (!groups || !groups->is_array() || groups->size() == 0)
    return HPS_TRITON_ERROR(INVALID_ARG, "model ", name_, ": the model configuration needs at least one instance_group entry");
  for (size_t g = 0; g < groups->size(); ++g) {
    const Json& group = groups->at(g);
    std::string kind;
    RETURN_IF_STATUS_ERROR(ParseField(kind, group, "kind", true));
    int64_t count = 1;
    RETURN_IF_STATUS_ERROR(ParseField(count, group, "count", false));
    // every instance owns one lookup session = one worker buffer set of the parameter server
    if (count > params_.number_of_worker_buffers_in_pool)
      return HPS_TRITON_ERROR(INVALID_ARG, "model ", name_, ": instance_group asks for ", count, " instances, ps.json provides "
                              "num_of_worker_buffer_in_pool = ", params_.number_of_worker_buffers_in_pool);
    if (!gpu_cache_) continue;
    if (kind != "KIND_GPU")
      return HPS_TRITON_ERROR(INVALID_ARG, "model ", name_, " uses the GPU embedding cache: its instances must be KIND_GPU, not ", kind);
    std::vector<int64_t> gpus;
    RETURN_IF_ERROR(ReadIntList(group, "gpus", "instance_group", &gpus));
    for (int64_t id : gpus)
      if (std::find(devices_.begin(), devices_.end(), id) == devices_.end()) devices_.push_back(id);
  }
  if (!gpu_cache_) devices_.assign(1, 0);   // host-only model: one pseudo device

  // refresh behaviour: ps.json defaults, overridden per model by config.pbtxt `parameters` (README.md:169-180)
  refresh_every_s_ = params_.refresh_interval;
  refresh_after_s_ = params_.refresh_delay;
  if (const Json* parameters = config_.Find("parameters")) {
    const struct { const char* key; float* f; bool* b; } knobs[] = {
        {"refresh_interval", &refresh_every_s_, nullptr}, {"refresh_delay", &refresh_after_s_, nullptr}, {"freeze_sparse", nullptr, &keep_tables_}};
    for (const auto& k : knobs) {
      const Json* entry = parameters->Find(k.key);
      if (!entry) continue;
      if (k.f) RETURN_IF_STATUS_ERROR(ParseField(*k.f, *entry, "string_value", false));
      else RETURN_IF_STATUS_ERROR(ParseField(*k.b, *entry, "string_value", false));
    }
  }

  keys_per_sample_ = 0;
  for (size_t c : params_.maxnum_catfeature_query_per_table_per_sample) keys_per_sample_ += (int64_t)c;
  if (keys_per_sample_ <= 0)
    return HPS_TRITON_ERROR(INVALID_ARG, "model ", name_, ": maxnum_catfeature_query_per_table_per_sample sums to ", keys_per_sample_);
  int64_t pbtxt_batch = 0;
  RETURN_IF_STATUS_ERROR(ParseField(pbtxt_batch, config_, "max_batch_size", false));
  if (pbtxt_batch < 0) return HPS_TRITON_ERROR(INVALID_ARG, "model ", name_, ": max_batch_size must not be negative, got ", pbtxt_batch);
  max_batch_ = (int64_t)params_.max_batchsize;   // the parameter server sized its buffers from ps.json: that value rules
  if (pbtxt_batch != 0 && pbtxt_batch != max_batch_)
    HPS_TRITON_LOG(WARN, "model ", name_, ": max_batch_size is ", pbtxt_batch, " in config.pbtxt and ", max_batch_,
                   " in ps.json; requests are limited by the latter");
  HPS_TRITON_LOG(INFO, "model ", name_, ": ", devices_.size(), " device(s), up to ", max_batch_, " samples x ", keys_per_sample_,
                 " keys per request, refresh every ", refresh_every_s_, " s, refresh ", refresh_after_s_, " s after a version change",
                 keep_tables_ ? " (sparse files frozen)" : "");
  return nullptr;
}

TRITONSERVER_Error* ModelState::AttachCaches() {
  // A model that was not in ps.json when the server started (online deployment) has neither tables nor caches yet.
  const bool tables_missing = server_->tables_of(name_).empty();
  const bool caches_missing = gpu_cache_ && !devices_.empty() && server_->get_embedding_cache(name_, (int)devices_[0]) == nullptr;
  if (tables_missing || caches_missing) {
    HPS_TRITON_LOG(INFO, "model ", name_, ": deployed online, loading its tables into the parameter server");
    RETURN_IF_STATUS_ERROR(server_->update_database_per_model(params_));
    if (caches_missing) RETURN_IF_STATUS_ERROR(server_->create_embedding_cache_per_model(params_));
  }
  const bool version_changed = serving_version_ > 0 && serving_version_ != version_;
  for (int64_t device : devices_) {
    if (gpu_cache_) {
      if (std::find(params_.deployed_devices.begin(), params_.deployed_devices.end(), (int)device) == params_.deployed_devices.end())
        return HPS_TRITON_ERROR(INVALID_ARG, "model ", name_, ": instance_group lists GPU ", device,
                                ", which is not in deployed_device_list of ps.json");
      if (caches_.count(device)) continue;
      auto cache = server_->get_embedding_cache(name_, (int)device);
      if (!cache) return HPS_TRITON_ERROR(INTERNAL, "model ", name_, ": the parameter server has no embedding cache on device ", device);
      caches_[device] = std::move(cache);
    }
    // another version of this model was serving until now: its rows may differ — refresh once, off the request path
    if (version_changed) timer_.startonce(refresh_after_s_, [this, device] { ReloadThenRefresh((int)device); });
  }
  if (refresh_every_s_ > 1e-6f) timer_.start(refresh_every_s_, [this] { RefreshAllCaches(); });
  HPS_TRITON_LOG(INFO, "model ", name_, " v", version_, ": ", caches_.size(), " embedding cache(s) attached");
  return nullptr;
}

}}  // namespace hps::triton
