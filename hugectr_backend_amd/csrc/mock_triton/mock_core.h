/*
 * mock_core.h — TEST INFRASTRUCTURE: a stand-in for tritonserver's core.
 *
 * No tritonserver / perf_analyzer exists in this image or on the GPU box, so the
 * seven TRITONBACKEND_* exports of libtriton_hps.so are exercised by this library instead.  It
 *   (1) implements every TRITONSERVER_ / TRITONBACKEND_ function the shell imports
 *       (include/tritonbackend_hps.h) with the ownership rules of the real core
 *       (one FINAL response per request, RequestRelease, messages owned per Triton's contract), and
 *   (2) offers the small driver API below so a test can play Triton: load the backend with a
 *       --backend-config, load a model from its config JSON, create instances, build requests around
 *       caller-owned input/output buffers (host or device), call Execute, and inspect what came back.
 * It is not part of the product.
 */
#ifndef HPS_MOCK_CORE_H_
#define HPS_MOCK_CORE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mock_server mock_server_t;
typedef struct mock_model mock_model_t;
typedef struct mock_instance mock_instance_t;
typedef struct mock_request mock_request_t;

const char* mock_last_error(void);

/* dlopen(backend_lib) + TRITONBACKEND_Initialize.  backend_config_json e.g. {"cmdline":{"ps":"/path/ps.json"}} */
int mock_server_create(const char* backend_lib_path, const char* backend_name, const char* backend_config_json,
                       uint32_t api_major, uint32_t api_minor, mock_server_t** out);
/* TRITONBACKEND_Finalize + dlclose */
int mock_server_destroy(mock_server_t* server);

/* TRITONBACKEND_ModelInitialize with the model configuration as JSON (what Triton derives from config.pbtxt) */
int mock_model_load(mock_server_t* server, const char* name, uint64_t version, const char* model_config_json,
                    mock_model_t** out);
int mock_model_unload(mock_model_t* model); /* TRITONBACKEND_ModelFinalize */

/* kind: 1 = KIND_CPU, 2 = KIND_GPU */
int mock_instance_create(mock_model_t* model, const char* name, int kind, int32_t device_id, mock_instance_t** out);
int mock_instance_destroy(mock_instance_t* instance); /* TRITONBACKEND_ModelInstanceFinalize */

mock_request_t* mock_request_new(const char* id, uint64_t correlation_id);
void mock_request_delete(mock_request_t* request);
/* datatype: TRITONSERVER_DataType value; memory_type: 0 CPU, 1 CPU_PINNED, 2 GPU.  Appending a second buffer to
 * the same input name models a tensor delivered in several pieces. */
int mock_request_add_input_buffer(mock_request_t* request, const char* name, int datatype, const int64_t* shape,
                                  uint32_t dims, const void* buffer, uint64_t byte_size, int memory_type,
                                  int64_t memory_type_id);
int mock_request_add_requested_output(mock_request_t* request, const char* name);
/* Memory the core hands out from TRITONBACKEND_OutputBuffer for this request.  Without it the mock allocates
 * host memory itself (and reports TRITONSERVER_MEMORY_CPU whatever the backend preferred). */
int mock_request_set_output_buffer(mock_request_t* request, void* buffer, uint64_t byte_size, int memory_type,
                                   int64_t memory_type_id);

/* TRITONBACKEND_ModelInstanceExecute(instance, requests, count).  Returns the error code + 1 of the call itself
 * (0 = nullptr); per-request outcomes are read from the requests. */
int mock_instance_execute(mock_instance_t* instance, mock_request_t** requests, uint32_t count);

/* ---- what the backend did with a request ---- */
int mock_request_response_count(mock_request_t* request);    /* responses sent (must be exactly 1) */
int mock_request_release_count(mock_request_t* request);     /* TRITONBACKEND_RequestRelease calls (must be 1) */
int mock_request_response_final(mock_request_t* request);    /* 1 if sent with COMPLETE_FINAL */
int mock_request_error_code(mock_request_t* request);        /* -1: success response, else TRITONSERVER_Error_Code */
const char* mock_request_error_message(mock_request_t* request);
int mock_request_output_count(mock_request_t* request);
const char* mock_request_output_name(mock_request_t* request, int index);
int mock_request_output_datatype(mock_request_t* request, int index);
int mock_request_output_dims(mock_request_t* request, int index, int64_t* shape, int max_dims);
void* mock_request_output_buffer(mock_request_t* request, int index, uint64_t* byte_size, int* memory_type,
                                 int64_t* memory_type_id);
/* response int parameter by name; returns 0 and sets *value if present, 1 if absent */
int mock_request_response_int_param(mock_request_t* request, const char* name, int64_t* value);

/* ---- statistics the backend reported for an instance ---- */
typedef struct mock_instance_stats {
  uint64_t success_requests, failed_requests, batch_reports, last_batch_size;
  /* distinct compute-start times among the successful requests of the last Execute call: 1 when the backend served them
   * with one lookup, the number of requests when it ran one lookup each */
  uint64_t last_distinct_compute_starts;
} mock_instance_stats_t;
int mock_instance_get_stats(mock_instance_t* instance, mock_instance_stats_t* out);

/* log capture: number of messages logged at >= WARN / == ERROR since the server was created */
int mock_server_log_counts(mock_server_t* server, uint64_t* info, uint64_t* warn, uint64_t* error);
void mock_set_verbose(int enabled); /* TRITONSERVER_LogIsEnabled(VERBOSE) + echo all logs to stderr */

#ifdef __cplusplus
}
#endif
#endif
