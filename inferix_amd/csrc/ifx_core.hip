// Library identity + error plumbing of libinferix_hip (C-ABI: include/inferix_hip.h).
#include <stdarg.h>
#include <string.h>

#include "ifx_common.h"

namespace ifx {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return IFX_ELAUNCH;
  }
  return IFX_OK;
}
}  // namespace ifx

extern "C" int ifx_version(void) { return (0 << 16) | (1 << 8) | 0; }
extern "C" const char* ifx_last_error(void) { return ifx::g_err; }
extern "C" const char* ifx_arch(void) { return "gfx950"; }
