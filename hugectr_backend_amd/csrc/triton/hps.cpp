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
//   * no C++ exception crosses the C ABI (CK_CUDA_THROW_ at hps.cc:677-685 does);
//   * several small requests of ONE Execute call (a dynamic batcher's hand-over) are served by one engine call instead of one
//     blocking lookup each (hps.cc:406): all requests are validated first, the valid ones' keys merged table by table, every
//     request's rows moved to its own output buffer; verdicts, parameters and statistics stay per request.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <exception>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

#include "../ps/thread_pool.h"
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
    // preference: this instance's device when the session can read the tensor there; otherwise PAGE-LOCKED host memory — the
    // lookup DMAs a flat page-locked KEYS array in place (no staging copy: 0.37 ms of host time per 13.6-MB request of 8-byte
    // keys), which Triton provides out of its pinned pool (--pinned-memory-pool-byte-size) when asked
    TRITONSERVER_MemoryType mt = allow_device ? TRITONSERVER_MEMORY_GPU : TRITONSERVER_MEMORY_CPU_PINNED;
    int64_t mt_id = allow_device ? device_id : 0;
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

// One tensor of a request as the lookup needs it: properties first, bytes later.
struct RequestTensor {
  TRITONBACKEND_Input* handle = nullptr;
  TRITONSERVER_DataType datatype = TRITONSERVER_TYPE_INVALID;
  uint64_t bytes = 0;
  uint32_t buffers = 0;
};

TRITONSERVER_Error* OpenTensor(TRITONBACKEND_Request* request, const char* name, TRITONSERVER_DataType want, RequestTensor* out) {
  RETURN_IF_ERROR(TRITONBACKEND_RequestInput(request, name, &out->handle));             // hps.cc:467-478
  const int64_t* shape = nullptr;
  uint32_t dims = 0;
  RETURN_IF_ERROR(TRITONBACKEND_InputProperties(out->handle, nullptr, &out->datatype, &shape, &dims, &out->bytes,
                                                &out->buffers));                         // hps.cc:517-542
  HPS_TRITON_LOG(VERBOSE, "\t", name, ": ", TRITONSERVER_DataTypeString(out->datatype), ", ", out->bytes, " bytes in ", out->buffers,
                 " buffer(s)");
  if (out->datatype != want)
    return HPS_TRITON_ERROR(INVALID_ARG, name, " must be ", TRITONSERVER_DataTypeString(want), ", the request carries ",
                            TRITONSERVER_DataTypeString(out->datatype));
  return nullptr;
}

// One request after validation: where its keys are, how they split over the tables, where its rows go.
struct Prepared {
  bool wants_output = false;      // false: nothing to produce (an empty success response, hps.cc:555)
  const int64_t* keys = nullptr;
  bool keys_on_device = false;
  ModelInstanceState::RequestSlice slice;   // keys (when on the host), keys per table, output buffer
  int64_t num_of_samples = 0;
};

// Everything that can fail for one request BEFORE the lookup; any error becomes that request's error response.
// staging_offset: this request's place in the instance's key staging (reserved for all requests of the call up front).
TRITONSERVER_Error* PrepareOne(ModelInstanceState* instance_state, ModelState* model_state, TRITONBACKEND_Request* request,
                               TRITONBACKEND_Response* response, size_t staging_offset, Prepared* prep) {
  int64_t* num_of_samples = &prep->num_of_samples;
  const char* request_id = "";
  RETURN_IF_ERROR(TRITONBACKEND_RequestId(request, &request_id));                       // hps.cc:410-412
  uint64_t correlation_id = 0;
  RETURN_IF_ERROR(TRITONBACKEND_RequestCorrelationId(request, &correlation_id));        // hps.cc:414-417
  uint32_t num_inputs = 0, num_outputs_wanted = 0;
  RETURN_IF_ERROR(TRITONBACKEND_RequestInputCount(request, &num_inputs));               // hps.cc:423-425
  RETURN_IF_ERROR(TRITONBACKEND_RequestOutputCount(request, &num_outputs_wanted));      // hps.cc:427-430
  HPS_TRITON_LOG(VERBOSE, "request \"", request_id, "\" (correlation ", correlation_id, "): ", num_inputs, " inputs, ",
                 num_outputs_wanted, " outputs wanted");

  // the request's inputs by name: exactly KEYS and NUMKEYS (hps.cc:446-465)
  if (num_inputs != 2) return HPS_TRITON_ERROR(INVALID_ARG, "a request carries the two inputs KEYS and NUMKEYS, this one has ", num_inputs);
  for (uint32_t i = 0; i < num_inputs; ++i) {
    const char* name = nullptr;
    RETURN_IF_ERROR(TRITONBACKEND_RequestInputName(request, i, &name));
    if (strcmp(name, "KEYS") != 0 && strcmp(name, "NUMKEYS") != 0)
      return HPS_TRITON_ERROR(INVALID_ARG, "a request carries the inputs KEYS and NUMKEYS; '", name, "' is neither");
  }
  RequestTensor keys_in, counts_in;
  RETURN_IF_ERROR(OpenTensor(request, "KEYS", TRITONSERVER_TYPE_INT64, &keys_in));
  RETURN_IF_ERROR(OpenTensor(request, "NUMKEYS", TRITONSERVER_TYPE_INT32, &counts_in));
  const char* output_name = nullptr;
  if (num_outputs_wanted > 0) RETURN_IF_ERROR(TRITONBACKEND_RequestOutputName(request, 0, &output_name));  // hps.cc:483-489
  if (num_outputs_wanted == 0) return nullptr;   // nothing to produce: an empty success response (hps.cc:555)

  const size_t T = instance_state->NumTables();
  if (keys_in.bytes % sizeof(int64_t) != 0)
    return HPS_TRITON_ERROR(INVALID_ARG, "KEYS holds ", keys_in.bytes, " bytes, not a whole number of int64 keys");
  const int64_t num_keys = (int64_t)(keys_in.bytes / sizeof(int64_t));                  // hps.cc:573
  *num_of_samples = num_keys / model_state->KeysPerSample();                            // hps.cc:575
  if (*num_of_samples > model_state->MaxBatch())                                        // hps.cc:576-582
    return HPS_TRITON_ERROR(UNSUPPORTED, "the request holds ", *num_of_samples, " samples, more than max batch size ",
                            model_state->MaxBatch());
  if (counts_in.bytes != T * sizeof(int32_t))
    return HPS_TRITON_ERROR(INVALID_ARG, "NUMKEYS must hold one int32 per embedding table (", T, "), got ",
                            counts_in.bytes / sizeof(int32_t));

  // ---- NUMKEYS -> keys per table, output size (hps.cc:599-625) ----
  std::vector<int32_t>& counts_staging = instance_state->CountStaging(T);
  const void* counts_data = nullptr;
  bool counts_on_device = false;
  RETURN_IF_ERROR(CollectInput(counts_in.handle, counts_in.buffers, counts_in.bytes, false, instance_state->DeviceId(),
                               counts_staging.data(), &counts_data, &counts_on_device));
  const int32_t* counts = reinterpret_cast<const int32_t*>(counts_data);
  const InferenceParams& p = instance_state->Params();
  std::vector<size_t> keys_per_table(T);
  int64_t counted = 0, output_elems = 0;
  for (size_t t = 0; t < T; ++t) {
    if (counts[t] < 0) return HPS_TRITON_ERROR(INVALID_ARG, "NUMKEYS[", t, "] is negative");
    keys_per_table[t] = (size_t)counts[t];
    counted += counts[t];
    output_elems += (int64_t)p.embedding_vecsize_per_table[t] * counts[t];
  }
  if (counted != num_keys) return HPS_TRITON_ERROR(INVALID_ARG, "NUMKEYS adds up to ", counted, " keys, KEYS holds ", num_keys);

  // ---- KEYS: read in place when they arrive in one buffer the session can consume (host memory, or this
  //      instance's device); otherwise gathered into the instance's staging (hps.cc:585-597) ----
  const bool gpucache = model_state->UsesGpuCache();
  const void* key_data = nullptr;
  bool keys_on_device = false;
  RETURN_IF_ERROR(CollectInput(keys_in.handle, keys_in.buffers, keys_in.bytes, gpucache, instance_state->DeviceId(),
                               instance_state->KeyStagingAt(staging_offset), &key_data, &keys_on_device));

  // ---- output tensor (hps.cc:626-660) ----
  TRITONBACKEND_Output* output = nullptr;
  RETURN_IF_ERROR(TRITONBACKEND_ResponseOutput(response, &output, output_name, TRITONSERVER_TYPE_FP32, &output_elems, 1));
  void* output_buffer = nullptr;
  TRITONSERVER_MemoryType output_memory = gpucache ? TRITONSERVER_MEMORY_GPU : TRITONSERVER_MEMORY_CPU;   // preference
  int64_t output_memory_id = gpucache ? instance_state->DeviceId() : 0;
  RETURN_IF_ERROR(TRITONBACKEND_OutputBuffer(output, &output_buffer, (uint64_t)output_elems * sizeof(float), &output_memory,
                                             &output_memory_id));
  const bool out_on_device = output_memory == TRITONSERVER_MEMORY_GPU;
  if (out_on_device && output_memory_id != instance_state->DeviceId())
    return HPS_TRITON_ERROR(UNSUPPORTED, "the output buffer is on device ", output_memory_id, ", the instance on device ",
                            instance_state->DeviceId());

  prep->wants_output = true;
  prep->keys = reinterpret_cast<const int64_t*>(key_data);
  prep->keys_on_device = keys_on_device;
  prep->slice.keys = keys_on_device ? nullptr : prep->keys;
  prep->slice.num_keys_per_table = std::move(keys_per_table);
  prep->slice.out = reinterpret_cast<float*>(output_buffer);
  prep->slice.out_on_device = out_on_device;
  prep->slice.out_elems = (size_t)output_elems;
  return nullptr;
}

// KEYS of a request in int64 elements, 0 when the request has no such input (the real check comes in PrepareOne)
size_t KeyCountOf(TRITONBACKEND_Request* request) {
  TRITONBACKEND_Input* in = nullptr;
  TRITONSERVER_Error* e = TRITONBACKEND_RequestInput(request, "KEYS", &in);
  uint64_t bytes = 0;
  if (e == nullptr) e = TRITONBACKEND_InputProperties(in, nullptr, nullptr, nullptr, nullptr, &bytes, nullptr);
  if (e != nullptr) { TRITONSERVER_ErrorDelete(e); return 0; }
  return (size_t)(bytes / sizeof(int64_t));
}

// No C++ exception may cross the C ABI (std::bad_alloc from a vector, anything a dependency throws): every entry point
// runs inside this.
template <typename F>
TRITONSERVER_Error* NoThrow(const char* entry, F&& body) {
  try {
    return body();
  } catch (const std::exception& e) {
    return HPS_TRITON_ERROR(INTERNAL, entry, ": ", e.what());
  } catch (...) {
    return HPS_TRITON_ERROR(INTERNAL, entry, ": unknown exception");
  }
}

}  // namespace

extern "C" {

TRITONSERVER_Error* TRITONBACKEND_Initialize(TRITONBACKEND_Backend* backend) {
  return NoThrow("TRITONBACKEND_Initialize", [&]() -> TRITONSERVER_Error* {
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

  TRITONSERVER_Message* cfg_msg;                                                        // hps.cc:88-90
  RETURN_IF_ERROR(TRITONBACKEND_BackendConfig(backend, &cfg_msg));
  TRITONBACKEND_ArtifactType where_kind;
  const char* where;
  RETURN_IF_ERROR(TRITONBACKEND_BackendArtifacts(backend, &where_kind, &where));        // hps.cc:92-98
  HPS_TRITON_LOG(INFO, "backend artifacts of '", name, "' are at ", where);

  // {"cmdline":{"ps":"<path to ps.json>", ...}}                                        // hps.cc:100-125
  const char* buffer;
  size_t byte_size;
  RETURN_IF_ERROR(TRITONSERVER_MessageSerializeToJson(cfg_msg, &buffer, &byte_size));
  HPS_TRITON_LOG(INFO, "backend configuration as handed over by Triton: ", std::string(buffer, byte_size));
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
  });
}

TRITONSERVER_Error* TRITONBACKEND_Finalize(TRITONBACKEND_Backend* backend) {
  return NoThrow("TRITONBACKEND_Finalize", [&]() -> TRITONSERVER_Error* {             // hps.cc:142-155
  void* vstate;
  RETURN_IF_ERROR(TRITONBACKEND_BackendState(backend, &vstate));
  HPS_TRITON_LOG(INFO, "TRITONBACKEND_Backend Finalize: HPSBackend");
  delete reinterpret_cast<HPSBackend*>(vstate);
  return nullptr;
  });
}

TRITONSERVER_Error* TRITONBACKEND_ModelInitialize(TRITONBACKEND_Model* model) {
  return NoThrow("TRITONBACKEND_ModelInitialize", [&]() -> TRITONSERVER_Error* {          // hps.cc:162-247
  const char* name;
  RETURN_IF_ERROR(TRITONBACKEND_ModelName(model, &name));
  uint64_t version;
  RETURN_IF_ERROR(TRITONBACKEND_ModelVersion(model, &version));
  HPS_TRITON_LOG(INFO, "TRITONBACKEND_ModelInitialize: ", name, " (version ", version, ")");
  TRITONBACKEND_ArtifactType repo_kind;
  const char* location;
  RETURN_IF_ERROR(TRITONBACKEND_ModelRepository(model, &repo_kind, &location));
  HPS_TRITON_LOG(INFO, "Repository location: ", location);
  TRITONBACKEND_Backend* backend;
  RETURN_IF_ERROR(TRITONBACKEND_ModelBackend(model, &backend));
  void* raw_backend_state;
  RETURN_IF_ERROR(TRITONBACKEND_BackendState(backend, &raw_backend_state));
  HPSBackend* backend_state = reinterpret_cast<HPSBackend*>(raw_backend_state);
  if (backend_state == nullptr) return HPS_TRITON_ERROR(INTERNAL, "the hps backend was not initialised");

  // a model (or a new version of it) that was not in ps.json when the server started: re-read ps.json
  const uint64_t model_ps_version = backend_state->GetModelVersion(name);               // hps.cc:207-219
  auto ps = backend_state->HierarchicalParameterServer();
  InferenceParams params;
  if (!ps->model_params(name, &params) || version != model_ps_version) {
    HPS_TRITON_LOG(INFO, "model ", name, " version ", version, " is new to the parameter server: reading ps.json again");
    RETURN_IF_ERROR(backend_state->ParseParameterServer(backend_state->ParameterServerJsonFile()));
  }
  if (!ps->model_params(name, &params))                                                 // hps.cc:221-223 (map.at throws there)
    return HPS_TRITON_ERROR(NOT_FOUND, "model ", name, " is not configured in the Parameter Server json file ",
                            backend_state->ParameterServerJsonFile());

  ModelState* model_state;
  RETURN_IF_ERROR(ModelState::Open(model, ps, params, model_ps_version, &model_state));
  TRITONSERVER_Error* err = TRITONBACKEND_ModelSetState(model, reinterpret_cast<void*>(model_state));
  if (err == nullptr) {
    backend_state->UpdateModelVersion(name, version);                                   // hps.cc:226
    err = model_state->CheckTensorContract();                                           // hps.cc:232
  }
  if (err == nullptr) err = model_state->ReadDeployment();                              // hps.cc:238
  if (err == nullptr) err = model_state->AttachCaches();                                // hps.cc:244
  if (err != nullptr) {
    // Triton does not call ModelFinalize after a failed ModelInitialize: clean up here
    (void)TRITONBACKEND_ModelSetState(model, nullptr);
    model_state->MarkServingVersion(std::numeric_limits<uint64_t>::max());  // never tear down another version's caches
    delete model_state;
    return err;
  }
  return nullptr;
  });
}

TRITONSERVER_Error* TRITONBACKEND_ModelFinalize(TRITONBACKEND_Model* model) {
  return NoThrow("TRITONBACKEND_ModelFinalize", [&]() -> TRITONSERVER_Error* {            // hps.cc:252-274
  const char* name;
  RETURN_IF_ERROR(TRITONBACKEND_ModelName(model, &name));
  TRITONBACKEND_Backend* backend;
  RETURN_IF_ERROR(TRITONBACKEND_ModelBackend(model, &backend));
  void* raw_backend_state;
  RETURN_IF_ERROR(TRITONBACKEND_BackendState(backend, &raw_backend_state));
  HPSBackend* backend_state = reinterpret_cast<HPSBackend*>(raw_backend_state);
  void* vstate;
  RETURN_IF_ERROR(TRITONBACKEND_ModelState(model, &vstate));
  ModelState* model_state = reinterpret_cast<ModelState*>(vstate);
  if (model_state == nullptr) return nullptr;
  if (backend_state != nullptr) model_state->MarkServingVersion(backend_state->GetModelVersion(name));
  HPS_TRITON_LOG(INFO, "model ", name, ": releasing the model state");
  delete model_state;
  return nullptr;
  });
}

TRITONSERVER_Error* TRITONBACKEND_ModelInstanceInitialize(TRITONBACKEND_ModelInstance* instance) {
  return NoThrow("TRITONBACKEND_ModelInstanceInitialize", [&]() -> TRITONSERVER_Error* {  // hps.cc:280-325
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
  TRITONSERVER_Error* err = instance_state->LoadHPSInstance();
  if (err == nullptr) err = TRITONBACKEND_ModelInstanceSetState(instance, reinterpret_cast<void*>(instance_state));
  if (err != nullptr) { delete instance_state; return err; }
  return nullptr;
  });
}

TRITONSERVER_Error* TRITONBACKEND_ModelInstanceFinalize(TRITONBACKEND_ModelInstance* instance) {
  return NoThrow("TRITONBACKEND_ModelInstanceFinalize", [&]() -> TRITONSERVER_Error* {  // hps.cc:330-344
  void* vstate;
  RETURN_IF_ERROR(TRITONBACKEND_ModelInstanceState(instance, &vstate));
  HPS_TRITON_LOG(INFO, "releasing a model instance (lookup session and staging buffers)");
  delete reinterpret_cast<ModelInstanceState*>(vstate);
  return nullptr;
  });
}

TRITONSERVER_Error* TRITONBACKEND_ModelInstanceExecute(TRITONBACKEND_ModelInstance* instance,
                                                       TRITONBACKEND_Request** requests, const uint32_t request_count) {
  return NoThrow("TRITONBACKEND_ModelInstanceExecute", [&]() -> TRITONSERVER_Error* {
  // Triton never calls this concurrently for one instance, but does for different instances/models:
  // only instance-local state is touched here (hps.cc:353-369).  BLOCKING execution policy.
  ModelInstanceState* instance_state;
  RETURN_IF_ERROR(TRITONBACKEND_ModelInstanceState(instance, reinterpret_cast<void**>(&instance_state)));
  ModelState* model_state = instance_state->StateForModel();
  {
    // the instance's thread joins the host tier's worker pools on the GPUs' NUMA node (thread_pool.h; HPS_NUMA_NODE=off: nobody is
    // bound; a thread that a host policy or numactl has already placed inside one node is left alone)
    thread_local bool placed = false;
    if (!placed) { placed = true; (void)hps::ThreadPool::BindCallingThread(); }
  }
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

  // ---- validate every request and find its buffers (hps.cc:406-660, per request) ----
  std::vector<Prepared> prep(request_count);
  std::vector<uint64_t> started(request_count, 0);
  {
    std::vector<size_t> staging_at(request_count, 0);
    size_t staging = 0;
    for (uint32_t r = 0; r < request_count; ++r) { staging_at[r] = staging; staging += std::max<size_t>(KeyCountOf(requests[r]), 1); }
    instance_state->ReserveKeyStaging(staging);
    for (uint32_t r = 0; r < request_count; ++r) {
      started[r] = NowNs();
      GUARDED_RESPOND_IF_ERROR(responses, r, PrepareOne(instance_state, model_state, requests[r], responses[r], staging_at[r], &prep[r]));
      if (responses[r] == nullptr) HPS_TRITON_LOG(ERROR, "request ", r, ": failed, error response sent");
    }
  }

  // ---- the lookups.  The reference runs one blocking lookup per request (hps.cc:406, 663-691).  Several SMALL requests in one
  //      Execute call (Triton's dynamic batcher hands them over together) go as ONE engine call instead — the call overhead
  //      (~0.1 ms: descriptor upload, count read-back, launches, the final synchronisation) is paid once, and the requests'
  //      keys share one probe / gather / miss path.  Requests that failed validation have their error response already and take
  //      no part; if the joint call fails, the requests are run one by one so that each gets its own verdict. ----
  std::vector<uint32_t> live;
  for (uint32_t r = 0; r < request_count; ++r)
    if (responses[r] != nullptr && prep[r].wants_output) live.push_back(r);
  bool joint = false;
  if (live.size() >= 2) {
    std::vector<const ModelInstanceState::RequestSlice*> slices;
    bool host_keys = true;
    for (uint32_t r : live) { slices.push_back(&prep[r].slice); host_keys &= !prep[r].keys_on_device; }
    if (host_keys && instance_state->CanCoalesce(slices)) {
      const uint64_t t0 = NowNs();
      HPS_ROCTX_RANGE(roctx_process, "ProcessCoalesced " + instance_state->Name());
      TRITONSERVER_Error* err = instance_state->ProcessCoalesced(slices);
      if (err == nullptr) {
        joint = true;
        for (uint32_t r : live) started[r] = t0;
      } else {
        HPS_TRITON_LOG(WARN, "one call for ", live.size(), " requests failed (", TRITONSERVER_ErrorMessage(err), "): running them one by one");
        TRITONSERVER_ErrorDelete(err);
      }
    }
  }
  if (!joint) {
    for (uint32_t r : live) {
      started[r] = NowNs();
      HPS_ROCTX_RANGE(roctx_process, "ProcessRequest " + instance_state->Name());         // hps.cc:671
      GUARDED_RESPOND_IF_ERROR(responses, r,
                               instance_state->ProcessRequest(prep[r].keys, prep[r].keys_on_device, prep[r].slice.num_keys_per_table,
                                                              prep[r].slice.out, prep[r].slice.out_on_device, prep[r].slice.out_elems));
      if (responses[r] == nullptr) HPS_TRITON_LOG(ERROR, "request ", r, ": failed, error response sent");
    }
  }

  for (uint32_t r = 0; r < request_count; ++r) {
    if (responses[r] == nullptr) continue;
    TRITONBACKEND_Request* request = requests[r];
    const uint64_t exec_start_ns = started[r];
    const int64_t num_of_samples = prep[r].num_of_samples;
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
  });
}

}  // extern "C"
