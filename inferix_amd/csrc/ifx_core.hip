// Library identity + error plumbing of libinferix_hip (C-ABI: include/inferix_hip.h).
#include <stdarg.h>
#include <stdlib.h>
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

// kernel-selection overrides (ifx_set_option); -1 = unset -> environment variable -> 0 (auto)
static int g_gemm_variant = -1, g_attn_variant = -1, g_gemm_small_split = -1;
static int opt_or_env(int& slot, const char* env) {
  if (slot < 0) {
    const char* e = getenv(env);
    slot = e ? atoi(e) : 0;
  }
  return slot;
}
int gemm_variant() { return opt_or_env(g_gemm_variant, "IFX_GEMM_VARIANT"); }
int attn_variant() { return opt_or_env(g_attn_variant, "IFX_ATTN_VARIANT"); }
int gemm_small_split() { return opt_or_env(g_gemm_small_split, "IFX_GEMM_SMALL_SPLIT"); }

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return IFX_ELAUNCH;
  }
  return IFX_OK;
}
}  // namespace ifx

extern "C" int ifx_version(void) { return (0 << 16) | (IFX_ABI_MINOR << 8) | 0; }
extern "C" const char* ifx_last_error(void) { return ifx::g_err; }
extern "C" const char* ifx_arch(void) { return "gfx950"; }
extern "C" int ifx_set_option(const char* key, int32_t value) {
  if (key && !strcmp(key, "gemm_variant") && value >= 0 && value <= 26) { ifx::g_gemm_variant = value; return IFX_OK; }
  if (key && !strcmp(key, "gemm_small_split") && (value == 0 || value == 1)) { ifx::g_gemm_small_split = value; return IFX_OK; }
  if (key && !strcmp(key, "attn_variant") && value >= 0 && value <= 7) { ifx::g_attn_variant = value; return IFX_OK; }
  ifx::set_error("ifx_set_option: unknown key or value out of range: %s = %d", key ? key : "(null)", (int)value);
  return IFX_EINVAL;
}
