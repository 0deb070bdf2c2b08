// Error carrier used inside the engine.  Codes mirror TRITONSERVER_Error_Code so the backend shell
// can hand them straight to TRITONSERVER_ErrorNew (the reference builds its errors with
// HPS_TRITON_ERROR: /root/reference/hps_backend/include/triton_common.hpp:46-52).
#pragma once
#include <sstream>
#include <string>
#include <utility>

namespace hps {

enum class Code : int {
  kOk = -1,
  kUnknown = 0,
  kInternal = 1,
  kNotFound = 2,
  kInvalidArg = 3,
  kUnavailable = 4,
  kUnsupported = 5,
  kAlreadyExists = 6,
};

class Status {
 public:
  Status() = default;
  Status(Code c, std::string m) : code_(c), msg_(std::move(m)) {}
  static Status Ok() { return Status(); }
  bool ok() const { return code_ == Code::kOk; }
  Code code() const { return code_; }
  const std::string& message() const { return msg_; }

 private:
  Code code_ = Code::kOk;
  std::string msg_;
};

template <typename... Args>
inline std::string StrCat(const Args&... args) {
  std::ostringstream os;
  (void)std::initializer_list<int>{((os << args), 0)...};
  return os.str();
}

template <typename... Args>
inline Status Error(Code c, const Args&... args) {
  return Status(c, StrCat(args...));
}

#define HPS_RETURN_IF_ERROR(expr)          \
  do {                                     \
    ::hps::Status _st = (expr);            \
    if (!_st.ok()) return _st;             \
  } while (0)

}  // namespace hps
