// Per-model state shared by all instances of one Triton model: validated model config, the per-device
// embedding caches, cache-refresh timers.
// Counterpart of the reference's ModelState (/root/reference/hps_backend/include/model_state.hpp:45-176,
// src/model_state.cpp).
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../cache/engine.h"
#include "../common/json.h"
#include "timer.h"
#include "triton_util.h"

namespace hps { namespace triton {

class ModelState {
 public:
  static TRITONSERVER_Error* Create(TRITONBACKEND_Model* triton_model, ModelState** state,
                                    std::shared_ptr<HierParameterServer> ps, const InferenceParams& params,
                                    uint64_t model_ps_version);                                   // model_state.cpp:65-106
  ~ModelState();                                                                                  // model_state.cpp:108-122

  TRITONSERVER_Error* ValidateModelConfig();   // model_state.cpp:180-261
  TRITONSERVER_Error* ParseModelConfig();      // model_state.cpp:263-371
  TRITONSERVER_Error* Create_EmbeddingCache(); // model_state.cpp:373-432
  void SetPSModelVersion(uint64_t v) { version_ps_ = v; }                                         // model_state.cpp:58-63

  TRITONBACKEND_Model* TritonModel() { return triton_model_; }
  const std::string& Name() const { return name_; }
  uint64_t Version() const { return version_; }
  int64_t BatchSize() const { return max_batch_size_; }
  int64_t CatNum() const { return cat_num_; }
  int64_t EmbeddingSize() const { return embedding_size_; }
  bool GPUCache() const { return support_gpu_cache_; }
  const InferenceParams& ModelInferencePara() const { return params_; }
  std::shared_ptr<HierParameterServer> ParameterServer() { return ps_; }
  std::shared_ptr<EmbeddingCache> GetEmbeddingCache(int64_t device_id);                            // model_state.hpp:170-173
  const std::vector<int64_t>& DeviceList() const { return gpu_shape_; }
  // refresh every cache of this model once (used by the timers; public for tests)
  void Refresh_Embedding_Cache();                                                                  // model_state.cpp:144-178
  void EmbeddingCacheRefresh(const std::string& model_name, int device_id);                        // model_state.cpp:124-142

 private:
  ModelState(TRITONBACKEND_Model* triton_model, const char* name, uint64_t version, uint64_t version_ps,
             Json&& model_config, std::shared_ptr<HierParameterServer> ps, const InferenceParams& params);

  TRITONBACKEND_Model* triton_model_;
  std::string name_;
  uint64_t version_;
  uint64_t version_ps_;
  Json model_config_;
  std::shared_ptr<HierParameterServer> ps_;
  InferenceParams params_;

  int64_t max_batch_size_ = 64;
  int64_t cat_num_ = 0;
  int64_t embedding_size_ = 0;
  float refresh_interval_ = 0.f;
  float refresh_delay_ = 0.f;
  bool freeze_embedding_ = false;
  bool support_gpu_cache_ = true;
  std::vector<int64_t> gpu_shape_;
  std::map<int64_t, std::shared_ptr<EmbeddingCache>> embedding_cache_map_;
  Timer timer_;
};

}}  // namespace hps::triton
