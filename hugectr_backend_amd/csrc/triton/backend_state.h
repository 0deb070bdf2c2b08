// Process-wide backend state: owns the single hierarchical parameter server.
// Counterpart of the reference's HPSBackend (/root/reference/hps_backend/include/backend.hpp:36-118,
// src/backend.cpp).  ps.json parsing itself lives in common/config.cpp (shared with the engine).
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>

#include "../cache/engine.h"
#include "triton_util.h"

namespace hps { namespace triton {

class HPSBackend {
 public:
  static TRITONSERVER_Error* Create(TRITONBACKEND_Backend* triton_backend, HPSBackend** backend,
                                    std::string ps_json_config_file);                       // backend.cpp:49-56
  // Creates the parameter server: parse ps.json, load tables, build + warm caches.        backend.cpp:59-78
  TRITONSERVER_Error* HPS_backend();

  TRITONBACKEND_Backend* TritonBackend() { return triton_backend_; }
  const std::string& ParameterServerJsonFile() const { return ps_json_config_file_; }
  std::shared_ptr<HierParameterServer> HierarchicalParameterServer() { return ps_; }
  // model name -> InferenceParams (get_hps_model_configuration_map)                        backend.cpp:70-71
  const std::map<std::string, InferenceParams>& HierarchicalPSConfigurationMap() const {
    return ps_->get_hps_model_configuration_map();
  }
  // Re-read ps.json to pick up a model deployed online.                                    backend.cpp:102-526
  TRITONSERVER_Error* ParseParameterServer(const std::string& path);

  uint64_t GetModelVersion(const std::string& model_name);                                  // backend.cpp:82-90
  bool UpdateModelVersion(const std::string& model_name, uint64_t version);                 // backend.cpp:93-99

 private:
  HPSBackend(TRITONBACKEND_Backend* b, std::string ps) : triton_backend_(b), ps_json_config_file_(std::move(ps)) {}
  TRITONBACKEND_Backend* triton_backend_;
  std::string ps_json_config_file_;
  std::shared_ptr<HierParameterServer> ps_;
  std::mutex version_map_mutex_;
  std::map<std::string, uint64_t> model_version_map_;
};

}}  // namespace hps::triton
