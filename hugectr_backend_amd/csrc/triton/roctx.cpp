// Run-time binding of the roctx marker API for the backend shell (see RoctxRange in triton_util.h).
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>

#include "triton_util.h"

namespace hps { namespace triton {

const RoctxRange::Api& RoctxRange::api() {
  static const Api a = [] {
    Api r;
    const char* e = std::getenv("HPS_ENABLE_ROCTX");
    if (!e || !*e || strcmp(e, "0") == 0) return r;
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      r.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      r.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (r.push && r.pop) return r;
      r = Api();
    }
    return r;
  }();
  return a;
}

}}  // namespace hps::triton
