// Error slot and version of the C ABI (include/feta_hip.h).
#include <cstdarg>
#include <cstdio>

#include "feta_abi_common.h"

// the one dynamic-LDS array every kernel carves (declared in feta_device.h)
namespace {
thread_local char g_err[512] = "";
thread_local char g_attr[160] = "";   // a refused hipFuncSetAttribute, reported with the launch that follows
}

namespace feta {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
void note_attr_error(const char* what) { snprintf(g_attr, sizeof(g_attr), "%s", what); }
int check_launch(const char* what) {
  // Only a failed LAUNCH is an error.  A refused hipFuncSetAttribute (note_attr_error, set by the allow_dynamic_lds
  // call in front of this launch) explains such a failure; if the launch went through anyway it is kept as a warning
  // in the error slot and the call succeeds.
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    if (g_attr[0] != 0)
      set_error("%s: %s (raising the dynamic-LDS limit of the kernel on this device failed: %s)", what,
                hipGetErrorString(e), g_attr);
    else
      set_error("%s: %s", what, hipGetErrorString(e));
    g_attr[0] = 0;
    return FETA_E_LAUNCH;
  }
  if (g_attr[0] != 0) {
    set_error("%s: warning: raising the dynamic-LDS limit failed (%s), the launch succeeded", what, g_attr);
    g_attr[0] = 0;
  }
  return FETA_OK;
}
void clear_attr_error() { g_attr[0] = 0; }
}  // namespace feta

extern "C" int feta_version(void) { return FETA_ABI_VERSION; }
extern "C" const char* feta_last_error(void) { return g_err; }
