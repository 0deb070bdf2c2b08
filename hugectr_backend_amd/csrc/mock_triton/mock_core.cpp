// TEST INFRASTRUCTURE — see mock_core.h.  Implements the Triton-core side of include/tritonbackend_hps.h.
#include "mock_core.h"

#include <dlfcn.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../../include/tritonbackend_hps.h"

// ------------------------------------------------------------------------------------------------
// objects behind the opaque Triton handles
// ------------------------------------------------------------------------------------------------
struct TRITONSERVER_Error { TRITONSERVER_Error_Code code; std::string msg; };
struct TRITONSERVER_Message { std::string json; };
struct TRITONSERVER_Server { int unused; };

struct TRITONBACKEND_Backend {
  std::string name, location;
  TRITONSERVER_Message config;
  void* state = nullptr;
  uint32_t api_major = 1, api_minor = 99;
};

struct TRITONBACKEND_Model {
  std::string name, repository, config_json;
  uint64_t version = 1;
  TRITONBACKEND_Backend* backend = nullptr;
  TRITONSERVER_Server server;
  void* state = nullptr;
};

struct TRITONBACKEND_ModelInstance {
  std::string name;
  TRITONSERVER_InstanceGroupKind kind = TRITONSERVER_INSTANCEGROUPKIND_GPU;
  int32_t device_id = 0;
  TRITONBACKEND_Model* model = nullptr;
  void* state = nullptr;
  std::mutex mu;
  mock_instance_stats_t stats{0, 0, 0, 0, 0};
  std::set<uint64_t> compute_starts;   // of the successful requests since the last batch report
};

struct InputBuf { const void* ptr; uint64_t bytes; TRITONSERVER_MemoryType mt; int64_t mt_id; };
struct TRITONBACKEND_Input {
  std::string name;
  TRITONSERVER_DataType dtype;
  std::vector<int64_t> shape;
  uint64_t byte_size = 0;
  std::vector<InputBuf> buffers;
};

struct TRITONBACKEND_Response;
struct TRITONBACKEND_Output {
  TRITONBACKEND_Response* response = nullptr;
  std::string name;
  TRITONSERVER_DataType dtype;
  std::vector<int64_t> shape;
  void* buffer = nullptr;
  uint64_t bytes = 0;
  TRITONSERVER_MemoryType mt = TRITONSERVER_MEMORY_CPU;
  int64_t mt_id = 0;
  std::unique_ptr<char[]> owned;
};

struct TRITONBACKEND_Request {
  std::string id;
  uint64_t correlation_id = 0;
  std::vector<std::unique_ptr<TRITONBACKEND_Input>> inputs;
  std::vector<std::string> requested_outputs;
  // output memory provided by the test
  void* out_buffer = nullptr;
  uint64_t out_bytes = 0;
  TRITONSERVER_MemoryType out_mt = TRITONSERVER_MEMORY_CPU;
  int64_t out_mt_id = 0;
  // results
  int responses_sent = 0, releases = 0, final_flag = 0;
  int error_code = -1;
  std::string error_msg;
  std::vector<std::unique_ptr<TRITONBACKEND_Output>> outputs;
  std::map<std::string, int64_t> int_params;
};

struct TRITONBACKEND_Response {
  TRITONBACKEND_Request* request = nullptr;
  std::vector<std::unique_ptr<TRITONBACKEND_Output>> outputs;
  std::map<std::string, int64_t> int_params;
};

typedef TRITONSERVER_Error* (*BackendFn)(TRITONBACKEND_Backend*);
typedef TRITONSERVER_Error* (*ModelFn)(TRITONBACKEND_Model*);
typedef TRITONSERVER_Error* (*InstanceFn)(TRITONBACKEND_ModelInstance*);
typedef TRITONSERVER_Error* (*ExecuteFn)(TRITONBACKEND_ModelInstance*, TRITONBACKEND_Request**, const uint32_t);

struct mock_server {
  void* dl = nullptr;
  TRITONBACKEND_Backend backend;
  BackendFn init = nullptr, fini = nullptr;
  ModelFn model_init = nullptr, model_fini = nullptr;
  InstanceFn inst_init = nullptr, inst_fini = nullptr;
  ExecuteFn execute = nullptr;
  bool initialized = false;
};
struct mock_model { mock_server* server; TRITONBACKEND_Model m; };
struct mock_instance { mock_model* model; TRITONBACKEND_ModelInstance i; };
struct mock_request { TRITONBACKEND_Request r; };

namespace {
thread_local std::string g_err;
std::atomic<int> g_verbose{0};
std::atomic<uint64_t> g_log_info{0}, g_log_warn{0}, g_log_error{0};
std::atomic<uint32_t> g_api_major{TRITONBACKEND_API_VERSION_MAJOR}, g_api_minor{99};

int Consume(TRITONSERVER_Error* e) {
  if (!e) return 0;
  g_err = e->msg;
  const int c = (int)e->code + 1;
  delete e;
  return c;
}
int Fail(const std::string& m) { g_err = m; return TRITONSERVER_ERROR_INTERNAL + 1; }
TRITONSERVER_Error* Err(TRITONSERVER_Error_Code c, const std::string& m) { return new TRITONSERVER_Error{c, m}; }
#define NEED(p, what) do { if (!(p)) return Err(TRITONSERVER_ERROR_INVALID_ARG, std::string("mock core: null ") + what); } while (0)
}  // namespace

// ------------------------------------------------------------------------------------------------
// imports of the backend: TRITONSERVER_*
// ------------------------------------------------------------------------------------------------
extern "C" {

__attribute__((visibility("default"))) TRITONSERVER_Error* TRITONSERVER_ErrorNew(TRITONSERVER_Error_Code code, const char* msg) {
  return new TRITONSERVER_Error{code, msg ? msg : ""};
}
__attribute__((visibility("default"))) void TRITONSERVER_ErrorDelete(TRITONSERVER_Error* error) { delete error; }
__attribute__((visibility("default"))) TRITONSERVER_Error_Code TRITONSERVER_ErrorCode(TRITONSERVER_Error* error) { return error->code; }
__attribute__((visibility("default"))) const char* TRITONSERVER_ErrorMessage(TRITONSERVER_Error* error) { return error->msg.c_str(); }

__attribute__((visibility("default"))) TRITONSERVER_Error* TRITONSERVER_LogMessage(TRITONSERVER_LogLevel level, const char* filename,
                                                                                const int line, const char* msg) {
  switch (level) {
    case TRITONSERVER_LOG_INFO: g_log_info++; break;
    case TRITONSERVER_LOG_WARN: g_log_warn++; break;
    case TRITONSERVER_LOG_ERROR: g_log_error++; break;
    default: break;
  }
  if (g_verbose.load() || (level == TRITONSERVER_LOG_ERROR && std::getenv("MOCK_TRITON_LOG_ERRORS"))) {
    static const char* tag[] = {"I", "W", "E", "V"};
    const char* base = filename ? strrchr(filename, '/') : nullptr;
    fprintf(stderr, "[mock-triton %s %s:%d] %s\n", tag[(int)level & 3], base ? base + 1 : (filename ? filename : "?"), line,
            msg ? msg : "");
  }
  return nullptr;
}
__attribute__((visibility("default"))) bool TRITONSERVER_LogIsEnabled(TRITONSERVER_LogLevel level) {
  return level != TRITONSERVER_LOG_VERBOSE || g_verbose.load() != 0;
}
__attribute__((visibility("default"))) TRITONSERVER_Error* TRITONSERVER_MessageSerializeToJson(TRITONSERVER_Message* message,
                                                                                            const char** base, size_t* byte_size) {
  NEED(message && base && byte_size, "message");
  *base = message->json.c_str();
  *byte_size = message->json.size();
  return nullptr;
}
__attribute__((visibility("default"))) TRITONSERVER_Error* TRITONSERVER_MessageDelete(TRITONSERVER_Message* message) {
  delete message;
  return nullptr;
}
__attribute__((visibility("default"))) const char* TRITONSERVER_DataTypeString(TRITONSERVER_DataType datatype) {
  static const char* names[] = {"<invalid>", "BOOL", "UINT8", "UINT16", "UINT32", "UINT64", "INT8", "INT16",
                                "INT32", "INT64", "FP16", "FP32", "FP64", "BYTES", "BF16"};
  return (int)datatype >= 0 && (int)datatype <= 14 ? names[(int)datatype] : "<invalid>";
}

// ------------------------------------------------------------------------------------------------
// imports of the backend: TRITONBACKEND_*
// ------------------------------------------------------------------------------------------------
#define API __attribute__((visibility("default"))) TRITONSERVER_Error*

API TRITONBACKEND_ApiVersion(uint32_t* major, uint32_t* minor) {
  *major = g_api_major.load();
  *minor = g_api_minor.load();
  return nullptr;
}
API TRITONBACKEND_BackendName(TRITONBACKEND_Backend* b, const char** name) { NEED(b, "backend"); *name = b->name.c_str(); return nullptr; }
API TRITONBACKEND_BackendConfig(TRITONBACKEND_Backend* b, TRITONSERVER_Message** config) {
  NEED(b, "backend");
  *config = &b->config;  // owned by the core: the backend must NOT delete it (hps.cc:88-90 never does)
  return nullptr;
}
API TRITONBACKEND_BackendArtifacts(TRITONBACKEND_Backend* b, TRITONBACKEND_ArtifactType* type, const char** location) {
  NEED(b, "backend");
  *type = TRITONBACKEND_ARTIFACT_FILESYSTEM;
  *location = b->location.c_str();
  return nullptr;
}
API TRITONBACKEND_BackendState(TRITONBACKEND_Backend* b, void** state) { NEED(b, "backend"); *state = b->state; return nullptr; }
API TRITONBACKEND_BackendSetState(TRITONBACKEND_Backend* b, void* state) { NEED(b, "backend"); b->state = state; return nullptr; }

API TRITONBACKEND_ModelName(TRITONBACKEND_Model* m, const char** name) { NEED(m, "model"); *name = m->name.c_str(); return nullptr; }
API TRITONBACKEND_ModelVersion(TRITONBACKEND_Model* m, uint64_t* version) { NEED(m, "model"); *version = m->version; return nullptr; }
API TRITONBACKEND_ModelRepository(TRITONBACKEND_Model* m, TRITONBACKEND_ArtifactType* type, const char** location) {
  NEED(m, "model");
  *type = TRITONBACKEND_ARTIFACT_FILESYSTEM;
  *location = m->repository.c_str();
  return nullptr;
}
API TRITONBACKEND_ModelConfig(TRITONBACKEND_Model* m, const uint32_t config_version, TRITONSERVER_Message** model_config) {
  NEED(m, "model");
  if (config_version != 1) return Err(TRITONSERVER_ERROR_UNSUPPORTED, "model config version must be 1");
  *model_config = new TRITONSERVER_Message{m->config_json};  // the backend owns and deletes it (model_state.cpp:89)
  return nullptr;
}
API TRITONBACKEND_ModelServer(TRITONBACKEND_Model* m, TRITONSERVER_Server** server) { NEED(m, "model"); *server = &m->server; return nullptr; }
API TRITONBACKEND_ModelBackend(TRITONBACKEND_Model* m, TRITONBACKEND_Backend** backend) { NEED(m, "model"); *backend = m->backend; return nullptr; }
API TRITONBACKEND_ModelState(TRITONBACKEND_Model* m, void** state) { NEED(m, "model"); *state = m->state; return nullptr; }
API TRITONBACKEND_ModelSetState(TRITONBACKEND_Model* m, void* state) { NEED(m, "model"); m->state = state; return nullptr; }

API TRITONBACKEND_ModelInstanceName(TRITONBACKEND_ModelInstance* i, const char** name) { NEED(i, "instance"); *name = i->name.c_str(); return nullptr; }
API TRITONBACKEND_ModelInstanceKind(TRITONBACKEND_ModelInstance* i, TRITONSERVER_InstanceGroupKind* kind) { NEED(i, "instance"); *kind = i->kind; return nullptr; }
API TRITONBACKEND_ModelInstanceDeviceId(TRITONBACKEND_ModelInstance* i, int32_t* device_id) { NEED(i, "instance"); *device_id = i->device_id; return nullptr; }
API TRITONBACKEND_ModelInstanceModel(TRITONBACKEND_ModelInstance* i, TRITONBACKEND_Model** model) { NEED(i, "instance"); *model = i->model; return nullptr; }
API TRITONBACKEND_ModelInstanceState(TRITONBACKEND_ModelInstance* i, void** state) { NEED(i, "instance"); *state = i->state; return nullptr; }
API TRITONBACKEND_ModelInstanceSetState(TRITONBACKEND_ModelInstance* i, void* state) { NEED(i, "instance"); i->state = state; return nullptr; }
API TRITONBACKEND_ModelInstanceReportStatistics(TRITONBACKEND_ModelInstance* i, TRITONBACKEND_Request* request, const bool success,
                                                const uint64_t, const uint64_t compute_start_ns, const uint64_t, const uint64_t) {
  NEED(i && request, "instance/request");
  std::lock_guard<std::mutex> lk(i->mu);
  if (success) { i->stats.success_requests++; i->compute_starts.insert(compute_start_ns); } else i->stats.failed_requests++;
  return nullptr;
}
API TRITONBACKEND_ModelInstanceReportBatchStatistics(TRITONBACKEND_ModelInstance* i, const uint64_t batch_size, const uint64_t,
                                                     const uint64_t, const uint64_t, const uint64_t) {
  NEED(i, "instance");
  std::lock_guard<std::mutex> lk(i->mu);
  i->stats.batch_reports++;
  i->stats.last_batch_size = batch_size;
  i->stats.last_distinct_compute_starts = i->compute_starts.size();
  i->compute_starts.clear();
  return nullptr;
}

API TRITONBACKEND_RequestId(TRITONBACKEND_Request* r, const char** id) { NEED(r, "request"); *id = r->id.c_str(); return nullptr; }
API TRITONBACKEND_RequestCorrelationId(TRITONBACKEND_Request* r, uint64_t* id) { NEED(r, "request"); *id = r->correlation_id; return nullptr; }
API TRITONBACKEND_RequestInputCount(TRITONBACKEND_Request* r, uint32_t* count) { NEED(r, "request"); *count = (uint32_t)r->inputs.size(); return nullptr; }
API TRITONBACKEND_RequestInputName(TRITONBACKEND_Request* r, const uint32_t index, const char** input_name) {
  NEED(r, "request");
  if (index >= r->inputs.size()) return Err(TRITONSERVER_ERROR_INVALID_ARG, "out of bounds index " + std::to_string(index) + ": request has " + std::to_string(r->inputs.size()) + " inputs");
  *input_name = r->inputs[index]->name.c_str();
  return nullptr;
}
API TRITONBACKEND_RequestInput(TRITONBACKEND_Request* r, const char* name, TRITONBACKEND_Input** input) {
  NEED(r && name, "request");
  for (auto& in : r->inputs) if (in->name == name) { *input = in.get(); return nullptr; }
  return Err(TRITONSERVER_ERROR_INVALID_ARG, std::string("unknown request input name ") + name);
}
API TRITONBACKEND_RequestOutputCount(TRITONBACKEND_Request* r, uint32_t* count) { NEED(r, "request"); *count = (uint32_t)r->requested_outputs.size(); return nullptr; }
API TRITONBACKEND_RequestOutputName(TRITONBACKEND_Request* r, const uint32_t index, const char** output_name) {
  NEED(r, "request");
  if (index >= r->requested_outputs.size()) return Err(TRITONSERVER_ERROR_INVALID_ARG, "out of bounds requested-output index");
  *output_name = r->requested_outputs[index].c_str();
  return nullptr;
}
API TRITONBACKEND_RequestRelease(TRITONBACKEND_Request* r, const uint32_t release_flags) {
  NEED(r, "request");
  if (release_flags != TRITONSERVER_REQUEST_RELEASE_ALL) return Err(TRITONSERVER_ERROR_INVALID_ARG, "unexpected release flags");
  r->releases++;
  return nullptr;
}

API TRITONBACKEND_InputProperties(TRITONBACKEND_Input* in, const char** name, TRITONSERVER_DataType* datatype, const int64_t** shape,
                                  uint32_t* dims_count, uint64_t* byte_size, uint32_t* buffer_count) {
  NEED(in, "input");
  if (name) *name = in->name.c_str();
  if (datatype) *datatype = in->dtype;
  if (shape) *shape = in->shape.data();
  if (dims_count) *dims_count = (uint32_t)in->shape.size();
  if (byte_size) *byte_size = in->byte_size;
  if (buffer_count) *buffer_count = (uint32_t)in->buffers.size();
  return nullptr;
}
API TRITONBACKEND_InputBuffer(TRITONBACKEND_Input* in, const uint32_t index, const void** buffer, uint64_t* buffer_byte_size,
                              TRITONSERVER_MemoryType* memory_type, int64_t* memory_type_id) {
  NEED(in, "input");
  if (index >= in->buffers.size()) {
    *buffer = nullptr; *buffer_byte_size = 0;
    return Err(TRITONSERVER_ERROR_INVALID_ARG, "out of bounds index " + std::to_string(index) + ": input has " + std::to_string(in->buffers.size()) + " buffers");
  }
  const InputBuf& b = in->buffers[index];
  *buffer = b.ptr;
  *buffer_byte_size = b.bytes;
  *memory_type = b.mt;       // the core reports where the data actually is, whatever the caller preferred
  *memory_type_id = b.mt_id;
  return nullptr;
}

API TRITONBACKEND_ResponseNew(TRITONBACKEND_Response** response, TRITONBACKEND_Request* request) {
  NEED(response && request, "response/request");
  auto* resp = new TRITONBACKEND_Response();
  resp->request = request;
  *response = resp;
  return nullptr;
}
API TRITONBACKEND_ResponseDelete(TRITONBACKEND_Response* response) { delete response; return nullptr; }
API TRITONBACKEND_ResponseOutput(TRITONBACKEND_Response* response, TRITONBACKEND_Output** output, const char* name,
                                 const TRITONSERVER_DataType datatype, const int64_t* shape, const uint32_t dims_count) {
  NEED(response && output && name, "response");
  auto o = std::make_unique<TRITONBACKEND_Output>();
  o->response = response;
  o->name = name;
  o->dtype = datatype;
  o->shape.assign(shape, shape + dims_count);
  *output = o.get();
  response->outputs.push_back(std::move(o));
  return nullptr;
}
API TRITONBACKEND_OutputBuffer(TRITONBACKEND_Output* output, void** buffer, const uint64_t buffer_byte_size,
                               TRITONSERVER_MemoryType* memory_type, int64_t* memory_type_id) {
  NEED(output && buffer && memory_type && memory_type_id, "output");
  TRITONBACKEND_Request* req = output->response->request;
  if (req->out_buffer != nullptr) {
    if (buffer_byte_size > req->out_bytes) {
      *buffer = nullptr;
      return Err(TRITONSERVER_ERROR_INTERNAL, "mock core: output needs " + std::to_string(buffer_byte_size) +
                                                  " bytes, the test provided " + std::to_string(req->out_bytes));
    }
    output->buffer = req->out_buffer;
    output->mt = req->out_mt;
    output->mt_id = req->out_mt_id;
  } else {
    output->owned.reset(new char[buffer_byte_size ? buffer_byte_size : 1]);
    output->buffer = output->owned.get();
    output->mt = TRITONSERVER_MEMORY_CPU;
    output->mt_id = 0;
  }
  output->bytes = buffer_byte_size;
  *buffer = output->buffer;
  *memory_type = output->mt;  // actual placement may differ from the backend's preference
  *memory_type_id = output->mt_id;
  return nullptr;
}
API TRITONBACKEND_ResponseSetIntParameter(TRITONBACKEND_Response* response, const char* name, const int64_t value) {
  NEED(response && name, "response");
  response->int_params[name] = value;
  return nullptr;
}
API TRITONBACKEND_ResponseSend(TRITONBACKEND_Response* response, const uint32_t send_flags, TRITONSERVER_Error* error) {
  NEED(response, "response");
  TRITONBACKEND_Request* req = response->request;
  req->responses_sent++;
  req->final_flag = (send_flags & TRITONSERVER_RESPONSE_COMPLETE_FINAL) ? 1 : 0;
  if (error) { req->error_code = (int)error->code; req->error_msg = error->msg; }  // the caller keeps ownership of `error`
  else { req->error_code = -1; req->error_msg.clear(); }
  req->outputs = std::move(response->outputs);
  req->int_params = std::move(response->int_params);
  delete response;  // a sent response is consumed
  return nullptr;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// driver API
// ------------------------------------------------------------------------------------------------
extern "C" {
#define MOCK_API __attribute__((visibility("default")))

MOCK_API const char* mock_last_error(void) { return g_err.c_str(); }
MOCK_API void mock_set_verbose(int enabled) { g_verbose.store(enabled); }

MOCK_API int mock_server_create(const char* backend_lib_path, const char* backend_name, const char* backend_config_json,
                                uint32_t api_major, uint32_t api_minor, mock_server_t** out) {
  if (!backend_lib_path || !out) return Fail("null argument");
  auto s = std::make_unique<mock_server>();
  s->dl = dlopen(backend_lib_path, RTLD_NOW | RTLD_LOCAL);
  if (!s->dl) return Fail(std::string("dlopen failed: ") + dlerror());
  auto sym = [&](const char* n) { return dlsym(s->dl, n); };
  s->init = (BackendFn)sym("TRITONBACKEND_Initialize");
  s->fini = (BackendFn)sym("TRITONBACKEND_Finalize");
  s->model_init = (ModelFn)sym("TRITONBACKEND_ModelInitialize");
  s->model_fini = (ModelFn)sym("TRITONBACKEND_ModelFinalize");
  s->inst_init = (InstanceFn)sym("TRITONBACKEND_ModelInstanceInitialize");
  s->inst_fini = (InstanceFn)sym("TRITONBACKEND_ModelInstanceFinalize");
  s->execute = (ExecuteFn)sym("TRITONBACKEND_ModelInstanceExecute");
  if (!s->init || !s->fini || !s->model_init || !s->model_fini || !s->inst_init || !s->inst_fini || !s->execute) {
    dlclose(s->dl);
    return Fail("backend library does not export all seven TRITONBACKEND_* entry points");
  }
  s->backend.name = backend_name ? backend_name : "hps";
  s->backend.location = std::string(backend_lib_path).substr(0, std::string(backend_lib_path).find_last_of('/'));
  s->backend.config.json = backend_config_json ? backend_config_json : "{}";
  g_api_major.store(api_major ? api_major : TRITONBACKEND_API_VERSION_MAJOR);
  g_api_minor.store(api_minor ? api_minor : 99);
  g_log_info = 0; g_log_warn = 0; g_log_error = 0;
  const int rc = Consume(s->init(&s->backend));
  if (rc != 0) { dlclose(s->dl); return rc; }
  s->initialized = true;
  *out = s.release();
  return 0;
}

MOCK_API int mock_server_destroy(mock_server_t* s) {
  if (!s) return 0;
  int rc = 0;
  if (s->initialized) rc = Consume(s->fini(&s->backend));
  if (s->dl) dlclose(s->dl);
  delete s;
  return rc;
}

MOCK_API int mock_model_load(mock_server_t* s, const char* name, uint64_t version, const char* model_config_json, mock_model_t** out) {
  if (!s || !name || !model_config_json || !out) return Fail("null argument");
  auto m = std::make_unique<mock_model>();
  m->server = s;
  m->m.name = name;
  m->m.version = version;
  m->m.repository = "/mock/model_repository/" + std::string(name);
  m->m.config_json = model_config_json;
  m->m.backend = &s->backend;
  const int rc = Consume(s->model_init(&m->m));
  if (rc != 0) return rc;
  *out = m.release();
  return 0;
}

MOCK_API int mock_model_unload(mock_model_t* m) {
  if (!m) return 0;
  const int rc = Consume(m->server->model_fini(&m->m));
  delete m;
  return rc;
}

MOCK_API int mock_instance_create(mock_model_t* m, const char* name, int kind, int32_t device_id, mock_instance_t** out) {
  if (!m || !out) return Fail("null argument");
  auto i = std::make_unique<mock_instance>();
  i->model = m;
  i->i.name = name ? name : (m->m.name + "_0");
  i->i.kind = (TRITONSERVER_InstanceGroupKind)kind;
  i->i.device_id = device_id;
  i->i.model = &m->m;
  const int rc = Consume(m->server->inst_init(&i->i));
  if (rc != 0) return rc;
  *out = i.release();
  return 0;
}

MOCK_API int mock_instance_destroy(mock_instance_t* i) {
  if (!i) return 0;
  int rc = 0;
  if (i->i.state) rc = Consume(i->model->server->inst_fini(&i->i));
  delete i;
  return rc;
}

MOCK_API mock_request_t* mock_request_new(const char* id, uint64_t correlation_id) {
  auto* r = new mock_request();
  r->r.id = id ? id : "";
  r->r.correlation_id = correlation_id;
  return r;
}
MOCK_API void mock_request_delete(mock_request_t* r) { delete r; }

MOCK_API int mock_request_add_input_buffer(mock_request_t* r, const char* name, int datatype, const int64_t* shape, uint32_t dims,
                                           const void* buffer, uint64_t byte_size, int memory_type, int64_t memory_type_id) {
  if (!r || !name) return Fail("null argument");
  TRITONBACKEND_Input* in = nullptr;
  for (auto& i : r->r.inputs) if (i->name == name) in = i.get();
  if (!in) {
    auto n = std::make_unique<TRITONBACKEND_Input>();
    n->name = name;
    n->dtype = (TRITONSERVER_DataType)datatype;
    if (shape) n->shape.assign(shape, shape + dims);
    in = n.get();
    r->r.inputs.push_back(std::move(n));
  }
  in->buffers.push_back({buffer, byte_size, (TRITONSERVER_MemoryType)memory_type, memory_type_id});
  in->byte_size += byte_size;
  return 0;
}
MOCK_API int mock_request_add_requested_output(mock_request_t* r, const char* name) {
  if (!r || !name) return Fail("null argument");
  r->r.requested_outputs.push_back(name);
  return 0;
}
MOCK_API int mock_request_set_output_buffer(mock_request_t* r, void* buffer, uint64_t byte_size, int memory_type, int64_t memory_type_id) {
  if (!r) return Fail("null argument");
  r->r.out_buffer = buffer;
  r->r.out_bytes = byte_size;
  r->r.out_mt = (TRITONSERVER_MemoryType)memory_type;
  r->r.out_mt_id = memory_type_id;
  return 0;
}

MOCK_API int mock_instance_execute(mock_instance_t* i, mock_request_t** requests, uint32_t count) {
  if (!i || (count && !requests)) return Fail("null argument");
  std::vector<TRITONBACKEND_Request*> reqs(count);
  for (uint32_t k = 0; k < count; ++k) reqs[k] = &requests[k]->r;
  return Consume(i->model->server->execute(&i->i, reqs.data(), count));
}

MOCK_API int mock_request_response_count(mock_request_t* r) { return r ? r->r.responses_sent : 0; }
MOCK_API int mock_request_release_count(mock_request_t* r) { return r ? r->r.releases : 0; }
MOCK_API int mock_request_response_final(mock_request_t* r) { return r ? r->r.final_flag : 0; }
MOCK_API int mock_request_error_code(mock_request_t* r) { return r ? r->r.error_code : -1; }
MOCK_API const char* mock_request_error_message(mock_request_t* r) { return r ? r->r.error_msg.c_str() : ""; }
MOCK_API int mock_request_output_count(mock_request_t* r) { return r ? (int)r->r.outputs.size() : 0; }
MOCK_API const char* mock_request_output_name(mock_request_t* r, int index) {
  return (r && index >= 0 && (size_t)index < r->r.outputs.size()) ? r->r.outputs[index]->name.c_str() : nullptr;
}
MOCK_API int mock_request_output_datatype(mock_request_t* r, int index) {
  return (r && index >= 0 && (size_t)index < r->r.outputs.size()) ? (int)r->r.outputs[index]->dtype : 0;
}
MOCK_API int mock_request_output_dims(mock_request_t* r, int index, int64_t* shape, int max_dims) {
  if (!r || index < 0 || (size_t)index >= r->r.outputs.size()) return -1;
  const auto& s = r->r.outputs[index]->shape;
  for (int d = 0; d < (int)s.size() && d < max_dims; ++d) shape[d] = s[d];
  return (int)s.size();
}
MOCK_API void* mock_request_output_buffer(mock_request_t* r, int index, uint64_t* byte_size, int* memory_type, int64_t* memory_type_id) {
  if (!r || index < 0 || (size_t)index >= r->r.outputs.size()) return nullptr;
  const auto& o = r->r.outputs[index];
  if (byte_size) *byte_size = o->bytes;
  if (memory_type) *memory_type = (int)o->mt;
  if (memory_type_id) *memory_type_id = o->mt_id;
  return o->buffer;
}
MOCK_API int mock_request_response_int_param(mock_request_t* r, const char* name, int64_t* value) {
  if (!r || !name) return 1;
  auto it = r->r.int_params.find(name);
  if (it == r->r.int_params.end()) return 1;
  if (value) *value = it->second;
  return 0;
}

MOCK_API int mock_instance_get_stats(mock_instance_t* i, mock_instance_stats_t* out) {
  if (!i || !out) return Fail("null argument");
  std::lock_guard<std::mutex> lk(i->i.mu);
  *out = i->i.stats;
  return 0;
}
MOCK_API int mock_server_log_counts(mock_server_t*, uint64_t* info, uint64_t* warn, uint64_t* error) {
  if (info) *info = g_log_info.load();
  if (warn) *warn = g_log_warn.load();
  if (error) *error = g_log_error.load();
  return 0;
}

}  // extern "C"
