/*
 * tritonbackend_hps.h — the Triton side of the drop-in boundary.
 *
 * libtriton_hps.so (hugectr_backend_amd/lib/) is loaded by tritonserver from <backend-dir>/hps/ exactly like
 * the reference's library of the same name (/root/reference/hps_backend/CMakeLists.txt:150-173,
 * README.md:101-109) and exports exactly the seven entry points the reference exports
 * (/root/reference/hps_backend/src/libtriton_hps.ldscript:26-30):
 *
 *     TRITONBACKEND_Initialize               hps_backend/src/hps.cc:57
 *     TRITONBACKEND_Finalize                 hps_backend/src/hps.cc:142
 *     TRITONBACKEND_ModelInitialize          hps_backend/src/hps.cc:162
 *     TRITONBACKEND_ModelFinalize            hps_backend/src/hps.cc:252
 *     TRITONBACKEND_ModelInstanceInitialize  hps_backend/src/hps.cc:280
 *     TRITONBACKEND_ModelInstanceFinalize    hps_backend/src/hps.cc:330
 *     TRITONBACKEND_ModelInstanceExecute     hps_backend/src/hps.cc:348-351
 *
 * Neither Triton's headers nor tritonserver exist in this image, so this file restates —
 * as plain C declarations — the part of Triton's public C API (tritonserver.h / tritonbackend.h) that the
 * shell imports.  Signatures and enum values follow the public API; the file:line next to each import is
 * its first use in the reference.  The mock core used by the tests (csrc/mock_triton) implements
 * exactly these imports from this same header, so shell and mock are self-consistent; against a real
 * tritonserver the symbols resolve from the server process as usual.
 */
#ifndef TRITONBACKEND_HPS_H_
#define TRITONBACKEND_HPS_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRITONBACKEND_API_VERSION_MAJOR 1
#define TRITONBACKEND_API_VERSION_MINOR 10

struct TRITONSERVER_Error;
struct TRITONSERVER_Message;
struct TRITONSERVER_Server;
struct TRITONBACKEND_Backend;
struct TRITONBACKEND_Model;
struct TRITONBACKEND_ModelInstance;
struct TRITONBACKEND_Request;
struct TRITONBACKEND_Response;
struct TRITONBACKEND_Input;
struct TRITONBACKEND_Output;
typedef struct TRITONSERVER_Error TRITONSERVER_Error;
typedef struct TRITONSERVER_Message TRITONSERVER_Message;
typedef struct TRITONSERVER_Server TRITONSERVER_Server;
typedef struct TRITONBACKEND_Backend TRITONBACKEND_Backend;
typedef struct TRITONBACKEND_Model TRITONBACKEND_Model;
typedef struct TRITONBACKEND_ModelInstance TRITONBACKEND_ModelInstance;
typedef struct TRITONBACKEND_Request TRITONBACKEND_Request;
typedef struct TRITONBACKEND_Response TRITONBACKEND_Response;
typedef struct TRITONBACKEND_Input TRITONBACKEND_Input;
typedef struct TRITONBACKEND_Output TRITONBACKEND_Output;

typedef enum TRITONSERVER_datatype_enum {
  TRITONSERVER_TYPE_INVALID = 0, TRITONSERVER_TYPE_BOOL = 1, TRITONSERVER_TYPE_UINT8 = 2,
  TRITONSERVER_TYPE_UINT16 = 3, TRITONSERVER_TYPE_UINT32 = 4, TRITONSERVER_TYPE_UINT64 = 5,
  TRITONSERVER_TYPE_INT8 = 6, TRITONSERVER_TYPE_INT16 = 7, TRITONSERVER_TYPE_INT32 = 8,
  TRITONSERVER_TYPE_INT64 = 9, TRITONSERVER_TYPE_FP16 = 10, TRITONSERVER_TYPE_FP32 = 11,
  TRITONSERVER_TYPE_FP64 = 12, TRITONSERVER_TYPE_BYTES = 13, TRITONSERVER_TYPE_BF16 = 14
} TRITONSERVER_DataType;

typedef enum TRITONSERVER_memorytype_enum {
  TRITONSERVER_MEMORY_CPU = 0, TRITONSERVER_MEMORY_CPU_PINNED = 1, TRITONSERVER_MEMORY_GPU = 2
} TRITONSERVER_MemoryType;

typedef enum TRITONSERVER_errorcode_enum {
  TRITONSERVER_ERROR_UNKNOWN = 0, TRITONSERVER_ERROR_INTERNAL = 1, TRITONSERVER_ERROR_NOT_FOUND = 2,
  TRITONSERVER_ERROR_INVALID_ARG = 3, TRITONSERVER_ERROR_UNAVAILABLE = 4, TRITONSERVER_ERROR_UNSUPPORTED = 5,
  TRITONSERVER_ERROR_ALREADY_EXISTS = 6
} TRITONSERVER_Error_Code;

typedef enum TRITONSERVER_loglevel_enum {
  TRITONSERVER_LOG_INFO = 0, TRITONSERVER_LOG_WARN = 1, TRITONSERVER_LOG_ERROR = 2, TRITONSERVER_LOG_VERBOSE = 3
} TRITONSERVER_LogLevel;

typedef enum TRITONSERVER_instancegroupkind_enum {
  TRITONSERVER_INSTANCEGROUPKIND_AUTO = 0, TRITONSERVER_INSTANCEGROUPKIND_CPU = 1,
  TRITONSERVER_INSTANCEGROUPKIND_GPU = 2, TRITONSERVER_INSTANCEGROUPKIND_MODEL = 3
} TRITONSERVER_InstanceGroupKind;

typedef enum TRITONBACKEND_artifacttype_enum { TRITONBACKEND_ARTIFACT_FILESYSTEM = 0 } TRITONBACKEND_ArtifactType;

#define TRITONSERVER_RESPONSE_COMPLETE_FINAL 1u /* hps.cc:728 */
#define TRITONSERVER_REQUEST_RELEASE_ALL 1u     /* hps.cc:783 */

/* ---------------- imports: TRITONSERVER_* ---------------- */
TRITONSERVER_Error* TRITONSERVER_ErrorNew(TRITONSERVER_Error_Code code, const char* msg);   /* include/triton_common.hpp:50 */
void TRITONSERVER_ErrorDelete(TRITONSERVER_Error* error);                                   /* include/hps_buffer.hpp:73 */
TRITONSERVER_Error_Code TRITONSERVER_ErrorCode(TRITONSERVER_Error* error);
const char* TRITONSERVER_ErrorMessage(TRITONSERVER_Error* error);
TRITONSERVER_Error* TRITONSERVER_LogMessage(TRITONSERVER_LogLevel level, const char* filename, const int line,
                                            const char* msg);                               /* include/triton_common.hpp:41 */
bool TRITONSERVER_LogIsEnabled(TRITONSERVER_LogLevel level);
TRITONSERVER_Error* TRITONSERVER_MessageSerializeToJson(TRITONSERVER_Message* message, const char** base,
                                                        size_t* byte_size);                 /* hps.cc:105 */
TRITONSERVER_Error* TRITONSERVER_MessageDelete(TRITONSERVER_Message* message);              /* model_state.cpp:89 */
const char* TRITONSERVER_DataTypeString(TRITONSERVER_DataType datatype);                    /* hps.cc:525 */

/* ---------------- imports: TRITONBACKEND_* ---------------- */
TRITONSERVER_Error* TRITONBACKEND_ApiVersion(uint32_t* major, uint32_t* minor);                                /* hps.cc:68 */
TRITONSERVER_Error* TRITONBACKEND_BackendName(TRITONBACKEND_Backend* backend, const char** name);              /* hps.cc:61 */
TRITONSERVER_Error* TRITONBACKEND_BackendConfig(TRITONBACKEND_Backend* backend, TRITONSERVER_Message** config); /* hps.cc:90 */
TRITONSERVER_Error* TRITONBACKEND_BackendArtifacts(TRITONBACKEND_Backend* backend, TRITONBACKEND_ArtifactType* type,
                                                   const char** location);                                     /* hps.cc:95 */
TRITONSERVER_Error* TRITONBACKEND_BackendState(TRITONBACKEND_Backend* backend, void** state);                  /* hps.cc:147 */
TRITONSERVER_Error* TRITONBACKEND_BackendSetState(TRITONBACKEND_Backend* backend, void* state);                /* hps.cc:132 */

TRITONSERVER_Error* TRITONBACKEND_ModelName(TRITONBACKEND_Model* model, const char** name);                    /* hps.cc:166 */
TRITONSERVER_Error* TRITONBACKEND_ModelVersion(TRITONBACKEND_Model* model, uint64_t* version);                 /* hps.cc:168 */
TRITONSERVER_Error* TRITONBACKEND_ModelRepository(TRITONBACKEND_Model* model, TRITONBACKEND_ArtifactType* type,
                                                  const char** location);                                      /* hps.cc:181 */
TRITONSERVER_Error* TRITONBACKEND_ModelConfig(TRITONBACKEND_Model* model, const uint32_t config_version,
                                              TRITONSERVER_Message** model_config);                            /* model_state.cpp:72 */
TRITONSERVER_Error* TRITONBACKEND_ModelServer(TRITONBACKEND_Model* model, TRITONSERVER_Server** server);       /* model_state.cpp:99 */
TRITONSERVER_Error* TRITONBACKEND_ModelBackend(TRITONBACKEND_Model* model, TRITONBACKEND_Backend** backend);   /* hps.cc:187 */
TRITONSERVER_Error* TRITONBACKEND_ModelState(TRITONBACKEND_Model* model, void** state);                        /* hps.cc:265 */
TRITONSERVER_Error* TRITONBACKEND_ModelSetState(TRITONBACKEND_Model* model, void* state);                      /* hps.cc:225 */

TRITONSERVER_Error* TRITONBACKEND_ModelInstanceName(TRITONBACKEND_ModelInstance* instance, const char** name); /* hps.cc:284 */
TRITONSERVER_Error* TRITONBACKEND_ModelInstanceKind(TRITONBACKEND_ModelInstance* instance,
                                                    TRITONSERVER_InstanceGroupKind* kind);         /* model_instance_state.cpp:55 */
TRITONSERVER_Error* TRITONBACKEND_ModelInstanceDeviceId(TRITONBACKEND_ModelInstance* instance, int32_t* device_id); /* hps.cc:307 */
TRITONSERVER_Error* TRITONBACKEND_ModelInstanceModel(TRITONBACKEND_ModelInstance* instance, TRITONBACKEND_Model** model); /* hps.cc:289 */
TRITONSERVER_Error* TRITONBACKEND_ModelInstanceState(TRITONBACKEND_ModelInstance* instance, void** state);     /* hps.cc:334 */
TRITONSERVER_Error* TRITONBACKEND_ModelInstanceSetState(TRITONBACKEND_ModelInstance* instance, void* state);   /* hps.cc:318 */
TRITONSERVER_Error* TRITONBACKEND_ModelInstanceReportStatistics(TRITONBACKEND_ModelInstance* instance,
                                                                TRITONBACKEND_Request* request, const bool success,
                                                                const uint64_t exec_start_ns, const uint64_t compute_start_ns,
                                                                const uint64_t compute_end_ns, const uint64_t exec_end_ns); /* hps.cc:741 */
TRITONSERVER_Error* TRITONBACKEND_ModelInstanceReportBatchStatistics(TRITONBACKEND_ModelInstance* instance,
                                                                     const uint64_t batch_size, const uint64_t exec_start_ns,
                                                                     const uint64_t compute_start_ns, const uint64_t compute_end_ns,
                                                                     const uint64_t exec_end_ns);                /* hps.cc:757 */

TRITONSERVER_Error* TRITONBACKEND_RequestId(TRITONBACKEND_Request* request, const char** id);                  /* hps.cc:412 */
TRITONSERVER_Error* TRITONBACKEND_RequestCorrelationId(TRITONBACKEND_Request* request, uint64_t* id);          /* hps.cc:417 */
TRITONSERVER_Error* TRITONBACKEND_RequestInputCount(TRITONBACKEND_Request* request, uint32_t* count);          /* hps.cc:425 */
TRITONSERVER_Error* TRITONBACKEND_RequestInputName(TRITONBACKEND_Request* request, const uint32_t index,
                                                   const char** input_name);                                   /* hps.cc:449 */
TRITONSERVER_Error* TRITONBACKEND_RequestInput(TRITONBACKEND_Request* request, const char* name,
                                               TRITONBACKEND_Input** input);                                   /* hps.cc:471 */
TRITONSERVER_Error* TRITONBACKEND_RequestOutputCount(TRITONBACKEND_Request* request, uint32_t* count);         /* hps.cc:430 */
TRITONSERVER_Error* TRITONBACKEND_RequestOutputName(TRITONBACKEND_Request* request, const uint32_t index,
                                                    const char** output_name);                                 /* hps.cc:487 */
TRITONSERVER_Error* TRITONBACKEND_RequestRelease(TRITONBACKEND_Request* request, const uint32_t release_flags); /* hps.cc:783 */

TRITONSERVER_Error* TRITONBACKEND_InputProperties(TRITONBACKEND_Input* input, const char** name,
                                                  TRITONSERVER_DataType* datatype, const int64_t** shape,
                                                  uint32_t* dims_count, uint64_t* byte_size, uint32_t* buffer_count); /* hps.cc:519 */
TRITONSERVER_Error* TRITONBACKEND_InputBuffer(TRITONBACKEND_Input* input, const uint32_t index, const void** buffer,
                                              uint64_t* buffer_byte_size, TRITONSERVER_MemoryType* memory_type,
                                              int64_t* memory_type_id);                                        /* hps.cc:592 */

TRITONSERVER_Error* TRITONBACKEND_ResponseNew(TRITONBACKEND_Response** response, TRITONBACKEND_Request* request); /* hps.cc:388 */
TRITONSERVER_Error* TRITONBACKEND_ResponseDelete(TRITONBACKEND_Response* response);
TRITONSERVER_Error* TRITONBACKEND_ResponseOutput(TRITONBACKEND_Response* response, TRITONBACKEND_Output** output,
                                                 const char* name, const TRITONSERVER_DataType datatype,
                                                 const int64_t* shape, const uint32_t dims_count);             /* hps.cc:628 */
TRITONSERVER_Error* TRITONBACKEND_OutputBuffer(TRITONBACKEND_Output* output, void** buffer,
                                               const uint64_t buffer_byte_size, TRITONSERVER_MemoryType* memory_type,
                                               int64_t* memory_type_id);                                       /* hps.cc:646 */
TRITONSERVER_Error* TRITONBACKEND_ResponseSetIntParameter(TRITONBACKEND_Response* response, const char* name,
                                                          const int64_t value);                                /* hps.cc:713 */
TRITONSERVER_Error* TRITONBACKEND_ResponseSend(TRITONBACKEND_Response* response, const uint32_t send_flags,
                                               TRITONSERVER_Error* error);                                     /* hps.cc:727 */

/* ---------------- exports of libtriton_hps.so ---------------- */
#define HPS_TRITON_EXPORT __attribute__((visibility("default")))
HPS_TRITON_EXPORT TRITONSERVER_Error* TRITONBACKEND_Initialize(TRITONBACKEND_Backend* backend);
HPS_TRITON_EXPORT TRITONSERVER_Error* TRITONBACKEND_Finalize(TRITONBACKEND_Backend* backend);
HPS_TRITON_EXPORT TRITONSERVER_Error* TRITONBACKEND_ModelInitialize(TRITONBACKEND_Model* model);
HPS_TRITON_EXPORT TRITONSERVER_Error* TRITONBACKEND_ModelFinalize(TRITONBACKEND_Model* model);
HPS_TRITON_EXPORT TRITONSERVER_Error* TRITONBACKEND_ModelInstanceInitialize(TRITONBACKEND_ModelInstance* instance);
HPS_TRITON_EXPORT TRITONSERVER_Error* TRITONBACKEND_ModelInstanceFinalize(TRITONBACKEND_ModelInstance* instance);
HPS_TRITON_EXPORT TRITONSERVER_Error* TRITONBACKEND_ModelInstanceExecute(TRITONBACKEND_ModelInstance* instance,
                                                                         TRITONBACKEND_Request** requests,
                                                                         const uint32_t request_count);

#ifdef __cplusplus
}
#endif
#endif /* TRITONBACKEND_HPS_H_ */
