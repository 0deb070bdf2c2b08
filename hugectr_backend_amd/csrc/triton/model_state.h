// What all instances of one Triton model share: the checked model configuration, the model's GPU embedding caches
// (one per device it is deployed on) and the schedule that keeps them fresh.
//
// Same lifecycle and the same accept/reject decisions as the reference's ModelState
// (/root/reference/hps_backend/src/model_state.cpp: configuration checks :180-371, cache creation and refresh scheduling
// :373-432, refresh bodies :124-178, teardown :108-122); structure, names and wording are this project's.
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
  // Reads the model configuration from Triton.  serving_version: the version of this model name the backend served
  // last (0: none) — decides whether the caches need a refresh and who tears them down.
  static TRITONSERVER_Error* Open(TRITONBACKEND_Model* model, std::shared_ptr<HierParameterServer> server,
                                  const InferenceParams& params, uint64_t serving_version, ModelState** out);
  ~ModelState();

  // The three steps of ModelInitialize after Open, in this order:
  TRITONSERVER_Error* CheckTensorContract();  // inputs KEYS int64 [-1] + NUMKEYS int32 [-1], one fp32 [-1] output
  TRITONSERVER_Error* ReadDeployment();       // instance groups -> devices, refresh parameters, batch limits
  TRITONSERVER_Error* AttachCaches();         // tables + caches for a model deployed online, refresh schedule

  // ModelFinalize / a failed ModelInitialize tell the state which version is being served now: only the state of
  // that version destroys the model's caches (an older version unloading later must leave them alone).
  void MarkServingVersion(uint64_t v) { serving_version_ = v; }

  TRITONBACKEND_Model* TritonModel() { return triton_model_; }
  const std::string& Name() const { return name_; }
  uint64_t Version() const { return version_; }
  int64_t MaxBatch() const { return max_batch_; }                // samples per request (ps.json's max_batch_size)
  int64_t KeysPerSample() const { return keys_per_sample_; }     // sum of maxnum_catfeature_query_per_table_per_sample
  bool UsesGpuCache() const { return gpu_cache_; }
  const InferenceParams& Params() const { return params_; }
  const std::shared_ptr<HierParameterServer>& Server() const { return server_; }
  std::shared_ptr<EmbeddingCache> CacheOn(int64_t device) const;
  const std::vector<int64_t>& Devices() const { return devices_; }

  // Refresh bodies (run by the timers; public for tests)
  void RefreshAllCaches();                     // periodic: every device
  void ReloadThenRefresh(int device);          // once after a version change: re-read the sparse files, then refresh

 private:
  ModelState(TRITONBACKEND_Model* model, std::string name, uint64_t version, uint64_t serving_version, Json&& config,
             std::shared_ptr<HierParameterServer> server, const InferenceParams& params);

  TRITONBACKEND_Model* triton_model_;
  std::string name_;
  uint64_t version_;
  uint64_t serving_version_;
  Json config_;
  std::shared_ptr<HierParameterServer> server_;
  InferenceParams params_;

  int64_t max_batch_ = 0;
  int64_t keys_per_sample_ = 0;
  float refresh_every_s_ = 0.f;     // 0: no periodic refresh
  float refresh_after_s_ = 0.f;     // delay of the one-shot refresh behind a version change
  bool keep_tables_ = false;        // "freeze_sparse": a version change refreshes the caches without re-reading the files
  bool gpu_cache_ = true;
  std::vector<int64_t> devices_;
  std::map<int64_t, std::shared_ptr<EmbeddingCache>> caches_;
  Timer timer_;
};

}}  // namespace hps::triton
