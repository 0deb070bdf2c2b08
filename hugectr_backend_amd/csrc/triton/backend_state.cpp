#include "backend_state.h"

namespace hps { namespace triton {

TRITONSERVER_Error* HPSBackend::Create(TRITONBACKEND_Backend* triton_backend, HPSBackend** backend,
                                       std::string ps_json_config_file) {
  if (ps_json_config_file.empty())
    return HPS_TRITON_ERROR(INVALID_ARG,
                            "the hps backend needs the parameter-server configuration: start tritonserver with "
                            "--backend-config=hps,ps=<path to ps.json>");
  *backend = new HPSBackend(triton_backend, std::move(ps_json_config_file));
  return nullptr;
}

TRITONSERVER_Error* HPSBackend::HPS_backend() {
  HPS_TRITON_LOG(INFO, "hps backend: building the parameter server from ", ps_json_config_file_);
  RETURN_IF_STATUS_ERROR(HierParameterServer::create(ps_json_config_file_, &ps_));
  HPS_TRITON_LOG(INFO, "hps backend: parameter server ready (host tier loaded, GPU caches warm)");
  return nullptr;
}

TRITONSERVER_Error* HPSBackend::ParseParameterServer(const std::string& path) {
  HPS_TRITON_LOG(INFO, "hps backend: reading parameter-server configuration ", path);
  RETURN_IF_STATUS_ERROR(ps_->parse_config(path));
  return nullptr;
}

uint64_t HPSBackend::GetModelVersion(const std::string& model_name) {
  std::lock_guard<std::mutex> lock(version_map_mutex_);
  auto it = model_version_map_.find(model_name);
  return it == model_version_map_.end() ? 0 : it->second;
}

bool HPSBackend::UpdateModelVersion(const std::string& model_name, uint64_t version) {
  std::lock_guard<std::mutex> lock(version_map_mutex_);
  model_version_map_[model_name] = version;
  return true;
}

}}  // namespace hps::triton
