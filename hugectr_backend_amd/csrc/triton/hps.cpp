// libtriton_hps.so — the seven TRITONBACKEND entry points of the HPS backend, MI355X engine underneath.
//
// Same surface, request/response contract and error behaviour as the reference shell
// (/root/reference/hps_backend/src/hps.cc:52-791); the file:line next to each block is the reference code
// it stands for.  Deliberate differences:
//   * per-request failures never `return` from Execute with live responses (hps.cc:450-465 does);
//   * after the "too many samples" error the request is skipped (hps.cc:577-597 goes on and copies);
//   * NUMKEYS must hold exactly one int32 per table and sum to the KEYS element count (unchecked there);
//   * KEYS split over several input buffers are concatenated (hps.cc:586-597 overwrites offset 0);
//   * KEYS handed over in GPU memory are used in place (hps.cc:587-597 asks for GPU memory, then memcpy's);
//   * the lookup writes straight into Triton's output buffer when it is device memory instead of going
//     through a private result buffer plus a second full-size copy (hps.cc:676-691);
//   * no C++ exception crosses the C ABI (CK_CUDA_THROW_ at hps.cc:677-685 does).
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

#include "backend_state.h"
#include "model_instance_state.h"
#include "model_state.h"
#include "triton_util.h"

using namespace hps;
using namespace hps::triton;

namespace {

// Load-time hint, see c_api.cpp: more HIP hardware queues than the default 4, so that the streams of two model instances do
// not share one and serialise.  No effect when tritonserver has initialised HIP before loading the backend: export
// GPU_MAX_HW_QUEUES in the launch environment then.
const int g_hw_queue_hint = (setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0), 0);

// Gather one input into a contiguous buffer the lookup can consume.
//   *data / *on_device describe the result; staging (host) is used when the input arrives in several
//   buffers or in a memory type the session cannot read directly.
TRITONSERVER_Error* CollectInput(TRITONBACKEND_Input* input, uint32_t buffer_count, uint64_t total_bytes,
                                 bool allow_device, int32_t device_id, void* host_staging, const void** data,
                                 bool* on_device) {
  *on_device = false;
  *data = nullptr;
  if (total_bytes == 0) { *data = host_staging; return nullptr; }
  if (buffer_count == 1) {
    const void* buf = nullptr;
    uint64_t bytes = 0;
    TRITONSERVER_MemoryType mt = allow_device ? TRITONSERVER_MEMORY_GPU : TRITONSERVER_MEMORY_CPU;  // preference
    int64_t mt_id = device_id;
    RETURN_IF_ERROR(TRITONBACKEND_InputBuffer(input, 0, &buf, &bytes, &mt, &mt_id));
    if (bytes != total_bytes)
      return HPS_TRITON_ERROR(INVALID_ARG, "input buffer holds ", bytes, " bytes, the tensor ", total_bytes);
    if (mt == TRITONSERVER_MEMORY_GPU) {
      if (allow_device && mt_id == device_id) { *data = buf; *on_device = true; return nullptr; }
      if (hipMemcpy(host_staging, buf, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return HPS_TRITON_ERROR(INTERNAL, "failed to copy an input tensor from device memory");
      *data = host_staging;
      return nullptr;
    }
    *data = buf;  // CPU or CPU_PINNED: read in place
    return nullptr;
  }
  uint64_t off = 0;
  for (uint32_t b = 0; b < buffer_count; ++b) {
    const void* buf = nullptr;
    uint64_t bytes = 0;
    TRITONSERVER_MemoryType mt = TRITONSERVER_MEMORY_CPU;
    int64_t mt_id = 0;
    RETURN_IF_ERROR(TRITONBACKEND_InputBuffer(input, b, &buf, &bytes, &mt, &mt_id));
    if (off + bytes > total_bytes) return HPS_TRITON_ERROR(INVALID_ARG, "input buffers exceed the tensor byte size");
    if (mt == TRITONSERVER_MEMORY_GPU) {
      if (hipMemcpy((char*)host_staging + off, buf, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return HPS_TRITON_ERROR(INTERNAL, "failed to copy an input tensor from device memory");
    } else {
      memcpy((char*)host_staging + off, buf, bytes);
    }
    off += bytes;
  }
  if (off != total_bytes) return HPS_TRITON_ERROR(INVALID_ARG, "input buffers hold ", off, " bytes, the tensor ", total_bytes);
  *data = host_staging;
  return nullptr;
}

// Everything that can fail for one request; any error becomes that request's error response.
TRITONSERVER_Error* ExecuteOne(ModelInstanceState* instance_state, ModelState* model_state, TRITONBACKEND_Request* request,
                               TRITONBACKEND_Response* response, int64_t* num_of_samples, uint64_t* exec_start_ns) {
  const char* request_id = "";
  RETURN_IF_ERROR(TRITONBACKEND_RequestId(request, &request_id));                       // hps.cc:410-412
  uint64_t correlation_id = 0;
  RETURN_IF_ERROR(TRITONBACKEND_RequestCorrelationId(request, &correlation_id));        // hps.cc:414-417
  uint32_t input_count = 0;
  RETURN_IF_ERROR(TRITONBACKEND_RequestInputCount(request, &input_count));              // hps.cc:423-425
  uint32_t requested_output_count = 0;
  RETURN_IF_ERROR(TRITONBACKEND_RequestOutputCount(request, &requested_output_count));  // hps.cc:427-430
  HPS_TRITON_LOG(VERBOSE, "request id = \"", request_id, "\", correlation_id = ", correlation_id,
                 ", input_count = ", input_count, ", requested_output_count = ", requested_output_count);

  if (input_count != 2) return HPS_TRITON_ERROR(INVALID_ARG, "expected 2 inputs (KEYS and NUMKEYS), got ", input_count);
  for (uint32_t i = 0; i < 2; ++i) {                                                    // hps.cc:446-465
    const char* input_name;
    RETURN_IF_ERROR(TRITONBACKEND_RequestInputName(request, i, &input_name));
    if (strcmp(input_name, "KEYS") != 0 && strcmp(input_name, "NUMKEYS") != 0)
      return HPS_TRITON_ERROR(INVALID_ARG, "expected input name as KEYS and NUMKEYS in request, but got ", input_name);
  }
  TRITONBACKEND_Input* catcol_input = nullptr;
  RETURN_IF_ERROR(TRITONBACKEND_RequestInput(request, "KEYS", &catcol_input));          // hps.cc:467-471
  TRITONBACKEND_Input* numkeys_input = nullptr;
  RETURN_IF_ERROR(TRITONBACKEND_RequestInput(request, "NUMKEYS", &numkeys_input));      // hps.cc:473-478
  const char* requested_output_name = nullptr;
  if (requested_output_count > 0)
    RETURN_IF_ERROR(TRITONBACKEND_RequestOutputName(request, 0, &requested_output_name));  // hps.cc:483-489

  TRITONSERVER_DataType cat_datatype, numkeys_datatype;                                 // hps.cc:517-542
  const int64_t *cat_input_shape, *num_keys_shape;
  uint32_t cat_dims_count, numkeys_dims_count, cat_input_buffer_count, numkeys_input_buffer_count;
  uint64_t cat_byte_size, numkeys_byte_size;
  RETURN_IF_ERROR(TRITONBACKEND_InputProperties(catcol_input, nullptr, &cat_datatype, &cat_input_shape, &cat_dims_count,
                                                &cat_byte_size, &cat_input_buffer_count));
  RETURN_IF_ERROR(TRITONBACKEND_InputProperties(numkeys_input, nullptr, &numkeys_datatype, &num_keys_shape,
                                                &numkeys_dims_count, &numkeys_byte_size, &numkeys_input_buffer_count));
  HPS_TRITON_LOG(VERBOSE, "\tinput KEYS: datatype = ", TRITONSERVER_DataTypeString(cat_datatype),
                 ", byte_size = ", cat_byte_size, ", buffer_count = ", cat_input_buffer_count);
  HPS_TRITON_LOG(VERBOSE, "\tinput NUMKEYS: datatype = ", TRITONSERVER_DataTypeString(numkeys_datatype),
                 ", byte_size = ", numkeys_byte_size, ", buffer_count = ", numkeys_input_buffer_count);
  if (cat_datatype != TRITONSERVER_TYPE_INT64)
    return HPS_TRITON_ERROR(INVALID_ARG, "KEYS must be TYPE_INT64, got ", TRITONSERVER_DataTypeString(cat_datatype));
  if (numkeys_datatype != TRITONSERVER_TYPE_INT32)
    return HPS_TRITON_ERROR(INVALID_ARG, "NUMKEYS must be TYPE_INT32, got ", TRITONSERVER_DataTypeString(numkeys_datatype));

  // only produce an output if one was requested (hps.cc:555)
  if (requested_output_count == 0) return nullptr;

  const size_t T = instance_state->NumTables();
  const int64_t numofcat = (int64_t)(cat_byte_size / sizeof(int64_t));                  // hps.cc:573
  *num_of_samples = numofcat / model_state->CatNum();                                   // hps.cc:575
  if (*num_of_samples > model_state->BatchSize())                                       // hps.cc:576-582
    return HPS_TRITON_ERROR(UNSUPPORTED, "The number of Input samples greater than max batch size");
  if (cat_byte_size % sizeof(int64_t) != 0)
    return HPS_TRITON_ERROR(INVALID_ARG, "KEYS byte size ", cat_byte_size, " is not a multiple of 8");
  if (numkeys_byte_size != T * sizeof(int32_t))
    return HPS_TRITON_ERROR(INVALID_ARG, "NUMKEYS must hold one int32 per embedding table (", T, "), got ",
                            numkeys_byte_size / sizeof(int32_t));

  // ---- NUMKEYS -> num_keys_per_table (hps.cc:599-618) ----
  int32_t numkeys_host[kMaxTables];
  const void* nk_data = nullptr;
  bool nk_on_device = false;
  RETURN_IF_ERROR(CollectInput(numkeys_input, numkeys_input_buffer_count, numkeys_byte_size, false,
                               instance_state->DeviceId(), numkeys_host, &nk_data, &nk_on_device));
  const int32_t* nk = reinterpret_cast<const int32_t*>(nk_data);
  std::vector<size_t> num_keys_per_table(T);
  int64_t key_total = 0;
  int64_t output_buffer_size = 0;                                                       // hps.cc:620-625
  const InferenceParams& p = instance_state->GetModelConfigutation();
  for (size_t t = 0; t < T; ++t) {
    if (nk[t] < 0) return HPS_TRITON_ERROR(INVALID_ARG, "NUMKEYS[", t, "] is negative");
    num_keys_per_table[t] = (size_t)nk[t];
    key_total += nk[t];
    output_buffer_size += (int64_t)p.embedding_vecsize_per_table[t] * nk[t];
  }
  if (key_total != numofcat)
    return HPS_TRITON_ERROR(INVALID_ARG, "sum(NUMKEYS) = ", key_total, " but KEYS holds ", numofcat, " keys");

  // ---- KEYS (hps.cc:585-597) ----
  const bool gpucache = model_state->GPUCache();
  const void* key_data = nullptr;
  bool keys_on_device = false;
  // (the staging vector is sized once per instance; it is only written when KEYS arrive in several
  //  buffers or in device memory the session cannot read in place)
  RETURN_IF_ERROR(CollectInput(catcol_input, cat_input_buffer_count, cat_byte_size, gpucache, instance_state->DeviceId(),
                               instance_state->KeyStaging((size_t)std::max<int64_t>(numofcat, 1)), &key_data,
                               &keys_on_device));

  // ---- output tensor (hps.cc:626-660) ----
  TRITONBACKEND_Output* output;
  RETURN_IF_ERROR(TRITONBACKEND_ResponseOutput(response, &output, requested_output_name, TRITONSERVER_TYPE_FP32,
                                               &output_buffer_size, 1));
  void* output_buffer = nullptr;
  TRITONSERVER_MemoryType output_memory_type = gpucache ? TRITONSERVER_MEMORY_GPU : TRITONSERVER_MEMORY_CPU;
  int64_t output_memory_type_id = gpucache ? instance_state->DeviceId() : 0;
  RETURN_IF_ERROR(TRITONBACKEND_OutputBuffer(output, &output_buffer, (uint64_t)output_buffer_size * sizeof(float),
                                             &output_memory_type, &output_memory_type_id));
  bool out_on_device = output_memory_type == TRITONSERVER_MEMORY_GPU;
  if (out_on_device && output_memory_type_id != instance_state->DeviceId())
    return HPS_TRITON_ERROR(UNSUPPORTED, "output buffer is on device ", output_memory_type_id, ", the instance on device ",
                            instance_state->DeviceId());

  // ---- lookup (hps.cc:663-691) ----
  HPS_TRITON_LOG(VERBOSE, "*****Processing request on device***** ", instance_state->DeviceId(), " for model ",
                 instance_state->Name());
  *exec_start_ns = NowNs();
  HPS_ROCTX_RANGE(roctx_process, "ProcessRequest " + instance_state->Name());           // hps.cc:671
  RETURN_IF_ERROR(instance_state->ProcessRequest(reinterpret_cast<const int64_t*>(key_data), keys_on_device,
                                                 num_keys_per_table, reinterpret_cast<float*>(output_buffer), out_on_device,
                                                 (size_t)output_buffer_size));
  HPS_TRITON_LOG(VERBOSE, "******Processing request completed!******");
  return nullptr;
}

}  // namespace

extern "C" {

TRITONSERVER_Error* TRITONBACKEND_Initialize(TRITONBACKEND_Backend* backend) {
  const char* name;
  RETURN_IF_ERROR(TRITONBACKEND_BackendName(backend, &name));
  HPS_TRITON_LOG(INFO, "TRITONBACKEND_Initialize: ", name);

  uint32_t api_version_major, api_version_minor;                                        // hps.cc:64-82
  RETURN_IF_ERROR(TRITONBACKEND_ApiVersion(&api_version_major, &api_version_minor));
  HPS_TRITON_LOG(INFO, "Triton TRITONBACKEND API version: ", api_version_major, ".", api_version_minor);
  HPS_TRITON_LOG(INFO, "'", name, "' TRITONBACKEND API version: ", TRITONBACKEND_API_VERSION_MAJOR, ".",
                 TRITONBACKEND_API_VERSION_MINOR);
  if (api_version_major != TRITONBACKEND_API_VERSION_MAJOR || api_version_minor < TRITONBACKEND_API_VERSION_MINOR)
    return HPS_TRITON_ERROR(UNSUPPORTED, "Triton backend API version does not support this backend");

  TRITONSERVER_Message* backend_config_message;                                         // hps.cc:88-90
  RETURN_IF_ERROR(TRITONBACKEND_BackendConfig(backend, &backend_config_message));
  TRITONBACKEND_ArtifactType artifact_type;
  const char* location;
  RETURN_IF_ERROR(TRITONBACKEND_BackendArtifacts(backend, &artifact_type, &location));  // hps.cc:92-98
  HPS_TRITON_LOG(INFO, "The Hierarchical Parameter Server Backend Repository location: ", location);

  // {"cmdline":{"ps":"<path to ps.json>", ...}}                                        // hps.cc:100-125
  const char* buffer;
  size_t byte_size;
  RETURN_IF_ERROR(TRITONSERVER_MessageSerializeToJson(backend_config_message, &buffer, &byte_size));
  HPS_TRITON_LOG(INFO, "The HPS configuration: ", std::string(buffer, byte_size));
  Json backend_config;
  std::string perr;
  if (!Json::Parse(std::string(buffer, byte_size), &backend_config, &perr))
    return HPS_TRITON_ERROR(INVALID_ARG, "failed to parse the backend configuration: ", perr);
  std::string ps_path;
  if (const Json* cmdline = backend_config.Find("cmdline"))
    if (const Json* ps = cmdline->Find("ps")) (void)ps->AsString(&ps_path);

  HPSBackend* hps_backend;                                                              // hps.cc:127-135
  RETURN_IF_ERROR(HPSBackend::Create(backend, &hps_backend, ps_path));
  TRITONSERVER_Error* err = hps_backend->HPS_backend();
  if (err != nullptr) { delete hps_backend; return err; }
  err = TRITONBACKEND_BackendSetState(backend, reinterpret_cast<void*>(hps_backend));
  if (err != nullptr) { delete hps_backend; return err; }
  return nullptr;
}

TRITONSERVER_Error* TRITONBACKEND_Finalize(TRITONBACKEND_Backend* backend) {             // hps.cc:142-155
  void* vstate;
  RETURN_IF_ERROR(TRITONBACKEND_BackendState(backend, &vstate));
  HPS_TRITON_LOG(INFO, "TRITONBACKEND_Backend Finalize: HPSBackend");
  delete reinterpret_cast<HPSBackend*>(vstate);
  return nullptr;
}

TRITONSERVER_Error* TRITONBACKEND_ModelInitialize(TRITONBACKEND_Model* model) {          // hps.cc:162-247
  const char* name;
  RETURN_IF_ERROR(TRITONBACKEND_ModelName(model, &name));
  uint64_t version;
  RETURN_IF_ERROR(TRITONBACKEND_ModelVersion(model, &version));
  HPS_TRITON_LOG(INFO, "TRITONBACKEND_ModelInitialize: ", name, " (version ", version, ")");
  TRITONBACKEND_ArtifactType artifact_type;
  const char* location;
  RETURN_IF_ERROR(TRITONBACKEND_ModelRepository(model, &artifact_type, &location));
  HPS_TRITON_LOG(INFO, "Repository location: ", location);
  TRITONBACKEND_Backend* backend;
  RETURN_IF_ERROR(TRITONBACKEND_ModelBackend(model, &backend));
  void* vbackendstate;
  RETURN_IF_ERROR(TRITONBACKEND_BackendState(backend, &vbackendstate));
  HPSBackend* backend_state = reinterpret_cast<HPSBackend*>(vbackendstate);
  if (backend_state == nullptr) return HPS_TRITON_ERROR(INTERNAL, "the hps backend was not initialised");

  // a model (or a new version of it) that was not in ps.json when the server started: re-read ps.json
  const uint64_t model_ps_version = backend_state->GetModelVersion(name);               // hps.cc:207-219
  auto ps = backend_state->HierarchicalParameterServer();
  InferenceParams params;
  if (!ps->model_params(name, &params) || version != model_ps_version) {
    HPS_TRITON_LOG(INFO, "Parsing the latest Parameter Server json config file for deploying model ", name, " online");
    RETURN_IF_ERROR(backend_state->ParseParameterServer(backend_state->ParameterServerJsonFile()));
  }
  if (!ps->model_params(name, &params))                                                 // hps.cc:221-223 (map.at throws there)
    return HPS_TRITON_ERROR(NOT_FOUND, "model ", name, " is not configured in the Parameter Server json file ",
                            backend_state->ParameterServerJsonFile());

  ModelState* model_state;
  RETURN_IF_ERROR(ModelState::Create(model, &model_state, ps, params, model_ps_version));
  TRITONSERVER_Error* err = TRITONBACKEND_ModelSetState(model, reinterpret_cast<void*>(model_state));
  if (err == nullptr) {
    backend_state->UpdateModelVersion(name, version);                                   // hps.cc:226
    err = model_state->ValidateModelConfig();                                           // hps.cc:232
  }
  if (err == nullptr) err = model_state->ParseModelConfig();                            // hps.cc:238
  if (err == nullptr) err = model_state->Create_EmbeddingCache();                       // hps.cc:244
  if (err != nullptr) {
    // Triton does not call ModelFinalize after a failed ModelInitialize: clean up here
    (void)TRITONBACKEND_ModelSetState(model, nullptr);
    model_state->SetPSModelVersion(std::numeric_limits<uint64_t>::max());  // never tear down another version's caches
    delete model_state;
    return err;
  }
  return nullptr;
}

TRITONSERVER_Error* TRITONBACKEND_ModelFinalize(TRITONBACKEND_Model* model) {            // hps.cc:252-274
  const char* name;
  RETURN_IF_ERROR(TRITONBACKEND_ModelName(model, &name));
  TRITONBACKEND_Backend* backend;
  RETURN_IF_ERROR(TRITONBACKEND_ModelBackend(model, &backend));
  void* vbackendstate;
  RETURN_IF_ERROR(TRITONBACKEND_BackendState(backend, &vbackendstate));
  HPSBackend* backend_state = reinterpret_cast<HPSBackend*>(vbackendstate);
  void* vstate;
  RETURN_IF_ERROR(TRITONBACKEND_ModelState(model, &vstate));
  ModelState* model_state = reinterpret_cast<ModelState*>(vstate);
  if (model_state == nullptr) return nullptr;
  if (backend_state != nullptr) model_state->SetPSModelVersion(backend_state->GetModelVersion(name));
  HPS_TRITON_LOG(INFO, "TRITONBACKEND_ModelFinalize: delete model state");
  delete model_state;
  return nullptr;
}

TRITONSERVER_Error* TRITONBACKEND_ModelInstanceInitialize(TRITONBACKEND_ModelInstance* instance) {  // hps.cc:280-325
  const char* name;
  RETURN_IF_ERROR(TRITONBACKEND_ModelInstanceName(instance, &name));
  TRITONBACKEND_Model* model;
  RETURN_IF_ERROR(TRITONBACKEND_ModelInstanceModel(instance, &model));
  void* vmodelstate;
  RETURN_IF_ERROR(TRITONBACKEND_ModelState(model, &vmodelstate));
  ModelState* model_state = reinterpret_cast<ModelState*>(vmodelstate);
  if (model_state == nullptr) return HPS_TRITON_ERROR(INTERNAL, "model state missing for instance ", name);
  int32_t device_id;
  RETURN_IF_ERROR(TRITONBACKEND_ModelInstanceDeviceId(instance, &device_id));
  HPS_TRITON_LOG(INFO, "TRITONBACKEND_ModelInstanceInitialize: ", name, " (device ", device_id, ")");

  ModelInstanceState* instance_state;
  RETURN_IF_ERROR(ModelInstanceState::Create(model_state, instance, &instance_state));
  HPS_TRITON_LOG(INFO, "******Loading HPS ******");
  TRITONSERVER_Error* err = instance_state->LoadHPSInstance();
  if (err == nullptr) err = TRITONBACKEND_ModelInstanceSetState(instance, reinterpret_cast<void*>(instance_state));
  if (err != nullptr) { delete instance_state; return err; }
  return nullptr;
}

TRITONSERVER_Error* TRITONBACKEND_ModelInstanceFinalize(TRITONBACKEND_ModelInstance* instance) {  // hps.cc:330-344
  void* vstate;
  RETURN_IF_ERROR(TRITONBACKEND_ModelInstanceState(instance, &vstate));
  HPS_TRITON_LOG(INFO, "TRITONBACKEND_ModelInstanceFinalize: delete instance state");
  delete reinterpret_cast<ModelInstanceState*>(vstate);
  return nullptr;
}

TRITONSERVER_Error* TRITONBACKEND_ModelInstanceExecute(TRITONBACKEND_ModelInstance* instance,
                                                       TRITONBACKEND_Request** requests, const uint32_t request_count) {
  // Triton never calls this concurrently for one instance, but does for different instances/models:
  // only instance-local state is touched here (hps.cc:353-369).  BLOCKING execution policy.
  ModelInstanceState* instance_state;
  RETURN_IF_ERROR(TRITONBACKEND_ModelInstanceState(instance, reinterpret_cast<void**>(&instance_state)));
  ModelState* model_state = instance_state->StateForModel();
  HPS_TRITON_LOG(VERBOSE, "model ", model_state->Name(), ", instance ", instance_state->Name(), ", executing ",
                 request_count, " requests");
  HPS_ROCTX_RANGE(roctx_execute, "ModelInstanceExecute " + instance_state->Name());     // hps.cc:375

  // one response object per request; failing here fails the whole call (hps.cc:381-390)
  std::vector<TRITONBACKEND_Response*> responses;
  responses.reserve(request_count);
  for (uint32_t r = 0; r < request_count; ++r) {
    TRITONBACKEND_Response* response;
    TRITONSERVER_Error* err = TRITONBACKEND_ResponseNew(&response, requests[r]);
    if (err != nullptr) {
      for (auto* resp : responses) LOG_IF_ERROR(TRITONBACKEND_ResponseDelete(resp), "failed to delete response");
      return err;
    }
    responses.push_back(response);
  }

  // From here on the requests are ours: exactly one final response and one release each (hps.cc:401-404).
  uint64_t min_exec_start_ns = std::numeric_limits<uint64_t>::max();
  uint64_t max_exec_end_ns = 0;
  uint64_t total_batch_size = 0;

  for (uint32_t r = 0; r < request_count; ++r) {                                         // hps.cc:406
    TRITONBACKEND_Request* request = requests[r];
    uint64_t exec_start_ns = NowNs();
    int64_t num_of_samples = 0;
    GUARDED_RESPOND_IF_ERROR(responses, r,
                             ExecuteOne(instance_state, model_state, request, responses[r], &num_of_samples, &exec_start_ns));
    if (responses[r] == nullptr) {
      HPS_TRITON_LOG(ERROR, "request ", r, ": failed, error response sent");
      continue;
    }
    min_exec_start_ns = std::min(min_exec_start_ns, exec_start_ns);
    total_batch_size += (uint64_t)num_of_samples;

    LOG_IF_ERROR(TRITONBACKEND_ResponseSetIntParameter(responses[r], "NumSample", num_of_samples),  // hps.cc:712-719
                 "failed return Number of samples");
    LOG_IF_ERROR(TRITONBACKEND_ResponseSetIntParameter(responses[r], "DeviceID", instance_state->DeviceId()),
                 "failed return device id");
    LOG_IF_ERROR(TRITONBACKEND_ResponseSend(responses[r], TRITONSERVER_RESPONSE_COMPLETE_FINAL, nullptr),  // hps.cc:726-730
                 "failed sending response");
    const uint64_t exec_end_ns = NowNs();
    max_exec_end_ns = std::max(max_exec_end_ns, exec_end_ns);
    LOG_IF_ERROR(TRITONBACKEND_ModelInstanceReportStatistics(instance_state->TritonModelInstance(), request, true,  // hps.cc:740-744
                                                             exec_start_ns, exec_start_ns, exec_end_ns, exec_end_ns),
                 "failed reporting request statistics");
    responses[r] = reinterpret_cast<TRITONBACKEND_Response*>(uintptr_t(1));  // sent OK (distinguish from failed = nullptr)
  }

  if (min_exec_start_ns == std::numeric_limits<uint64_t>::max()) min_exec_start_ns = max_exec_end_ns = NowNs();
  LOG_IF_ERROR(TRITONBACKEND_ModelInstanceReportBatchStatistics(instance_state->TritonModelInstance(), total_batch_size,  // hps.cc:756-761
                                                                min_exec_start_ns, min_exec_start_ns, max_exec_end_ns,
                                                                max_exec_end_ns),
               "failed reporting batch request statistics");

  for (uint32_t r = 0; r < request_count; ++r) {                                         // hps.cc:768-785
    TRITONBACKEND_Request* request = requests[r];
    if (responses[r] == nullptr)
      LOG_IF_ERROR(TRITONBACKEND_ModelInstanceReportStatistics(instance_state->TritonModelInstance(), request, false, 0, 0, 0, 0),
                   "failed reporting request statistics");
    LOG_IF_ERROR(TRITONBACKEND_RequestRelease(request, TRITONSERVER_REQUEST_RELEASE_ALL), "failed releasing request");
  }
  return nullptr;
}

}  // extern "C"
