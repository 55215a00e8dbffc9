// Error slot and version of the C ABI (include/feta_hip.h).
#include <cstdarg>
#include <cstdio>

#include "feta_abi_common.h"

// the one dynamic-LDS array every kernel carves (declared in feta_device.h)
namespace {
thread_local char g_err[512] = "";
}

namespace feta {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return FETA_E_LAUNCH;
  }
  return FETA_OK;
}
}  // namespace feta

extern "C" int feta_version(void) { return FETA_ABI_VERSION; }
extern "C" const char* feta_last_error(void) { return g_err; }
