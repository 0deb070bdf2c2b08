// Logging / error helpers of the backend shell.
// Counterparts of the reference's HPS_TRITON_LOG / HPS_TRITON_ERROR (include/triton_common.hpp:37-52),
// RETURN_IF_ERROR (triton backend_common.h) and GUARDED_RESPOND_IF_ERROR (include/hps_buffer.hpp:62-76).
#pragma once
#include <chrono>
#include <string>
#include <vector>

#include "../../../include/tritonbackend_hps.h"
#include "../common/status.h"

namespace hps { namespace triton {

inline TRITONSERVER_Error* MakeError(TRITONSERVER_Error_Code code, const std::string& msg) {
  return TRITONSERVER_ErrorNew(code, msg.c_str());
}

// engine Status -> TRITONSERVER_Error* (nullptr on success).  Codes map 1:1 (common/status.h).
inline TRITONSERVER_Error* ToTritonError(const Status& st) {
  if (st.ok()) return nullptr;
  const int c = (int)st.code();
  return MakeError(c < 0 ? TRITONSERVER_ERROR_UNKNOWN : (TRITONSERVER_Error_Code)c, st.message());
}

inline void LogMessage(TRITONSERVER_LogLevel level, const char* file, int line, const std::string& msg) {
  TRITONSERVER_Error* e = TRITONSERVER_LogMessage(level, file, line, msg.c_str());
  if (e) TRITONSERVER_ErrorDelete(e);
}

inline uint64_t NowNs() {
  return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
             std::chrono::steady_clock::now().time_since_epoch()).count();
}

// roctx ranges where the reference has NVTX ranges (NVTX_RANGE at hps.cc:375,671,674,701 and
// model_instance_state.cpp:179; compile-time opt-in there via TRITON_ENABLE_NVTX, CMakeLists.txt:45,102-104).
// Here: run-time opt-in — HPS_ENABLE_ROCTX=1 resolves roctxRangePushA/Pop from librocprofiler-sdk-roctx.so (or
// libroctx64.so) once; without it a range costs one predictable branch.  The ranges show up in
// `rocprofv3 --marker-trace`.
class RoctxRange {
 public:
  explicit RoctxRange(const std::string& name) {
    const Api& a = api();
    if (a.push) { a.push(name.c_str()); active_ = true; }
  }
  ~RoctxRange() { if (active_) api().pop(); }
  RoctxRange(const RoctxRange&) = delete;
  RoctxRange& operator=(const RoctxRange&) = delete;

 private:
  struct Api { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };
  static const Api& api();
  bool active_ = false;
};
#define HPS_ROCTX_RANGE(VAR, NAME) ::hps::triton::RoctxRange VAR(NAME)

#define HPS_TRITON_LOG(LEVEL, ...) \
  ::hps::triton::LogMessage(TRITONSERVER_LOG_##LEVEL, __FILE__, __LINE__, ::hps::StrCat(__VA_ARGS__))

#define HPS_TRITON_ERROR(CODE, ...) ::hps::triton::MakeError(TRITONSERVER_ERROR_##CODE, ::hps::StrCat(__VA_ARGS__))

#define RETURN_IF_ERROR(X)                 \
  do {                                     \
    TRITONSERVER_Error* rie_err__ = (X);   \
    if (rie_err__ != nullptr) return rie_err__; \
  } while (false)

#define RETURN_IF_STATUS_ERROR(X)                                   \
  do {                                                              \
    const ::hps::Status rse_st__ = (X);                             \
    if (!rse_st__.ok()) return ::hps::triton::ToTritonError(rse_st__); \
  } while (false)

#define LOG_IF_ERROR(X, MSG)                                                          \
  do {                                                                                \
    TRITONSERVER_Error* lie_err__ = (X);                                              \
    if (lie_err__ != nullptr) {                                                       \
      HPS_TRITON_LOG(ERROR, (MSG), ": ", TRITONSERVER_ErrorMessage(lie_err__));       \
      TRITONSERVER_ErrorDelete(lie_err__);                                            \
    }                                                                                 \
  } while (false)

// If X fails and the response for request IDX is still open: send the error as that request's final
// response, null the slot, free the error (hps_buffer.hpp:62-76).
#define GUARDED_RESPOND_IF_ERROR(RESPONSES, IDX, X)                                                    \
  do {                                                                                                 \
    if ((RESPONSES)[IDX] != nullptr) {                                                                 \
      TRITONSERVER_Error* gri_err__ = (X);                                                             \
      if (gri_err__ != nullptr) {                                                                      \
        LOG_IF_ERROR(TRITONBACKEND_ResponseSend((RESPONSES)[IDX], TRITONSERVER_RESPONSE_COMPLETE_FINAL, \
                                                gri_err__),                                            \
                     "failed to send error response");                                                 \
        (RESPONSES)[IDX] = nullptr;                                                                    \
        TRITONSERVER_ErrorDelete(gri_err__);                                                           \
      }                                                                                                \
    }                                                                                                  \
  } while (false)

}}  // namespace hps::triton
