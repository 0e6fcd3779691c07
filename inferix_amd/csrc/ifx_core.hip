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
static int g_gemm_variant = -1, g_attn_variant = -1, g_conv_variant = -1, g_spin_timeout_ms = -1, g_spin_fault = 0;
// gemm_small_split is a property of the CALLER (a sequence-parallel rank's block loop sets it around its own launches): per host thread,
// so that launches another thread enqueues meanwhile (a VAE decode, a text encoder) keep the row-count independent choice
static thread_local int g_gemm_small_split = -1;
static int opt_or_env(int& slot, const char* env, int dflt = 0) {
  if (slot < 0) {
    const char* e = getenv(env);
    slot = e ? atoi(e) : dflt;
  }
  return slot;
}
int gemm_variant() { return opt_or_env(g_gemm_variant, "IFX_GEMM_VARIANT"); }
int attn_variant() { return opt_or_env(g_attn_variant, "IFX_ATTN_VARIANT"); }
int conv_variant() { return opt_or_env(g_conv_variant, "IFX_CONV_VARIANT"); }
// tests: ifx_set_option("attn_debug_counters", 1) allocates (and zeroes) a device word that the ping-pong attention kernels increment once
// per (wave, key tile) that takes the rescale branch of the lazy row maximum; ifx_get_option("attn_rescale_count") synchronises the device
// and reads it; ifx_set_option("attn_debug_counters", 0) turns the counting off again.  One word per process (the current device's).
static unsigned* g_attn_dbg = nullptr;
static bool g_attn_dbg_on = false;
unsigned* attn_debug_counter() { return g_attn_dbg_on ? g_attn_dbg : nullptr; }
int gemm_small_split() { return opt_or_env(g_gemm_small_split, "IFX_GEMM_SMALL_SPLIT"); }

// ---- device-side waits are BOUNDED.  A kernel that waits for another workgroup (the split-K / stream-K hand-off of ifx_gemm_pp.hip)
// gives up after spin_timeout_ticks() of the 100 MHz wall clock, stores a code into the device error word and carries on with whatever
// the workspace holds: the launch ends, the result is garbage, and the NEXT ifx_last_error() / ifx_device_error() says why.  The word
// lives in pinned host memory mapped into every device (one word per process): the host reads it without a device synchronisation.
static unsigned* g_dev_err = nullptr;
unsigned* device_error_word() {
  if (g_dev_err == nullptr) {
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && p != nullptr) {
      memset(p, 0, 64);
      g_dev_err = (unsigned*)p;
    }
  }
  return g_dev_err;      // nullptr (no pinned memory): the kernels then only bound their waits
}
long long spin_timeout_ticks() { return (long long)max(1, opt_or_env(g_spin_timeout_ms, "IFX_SPIN_TIMEOUT_MS", 2000)) * 100000LL; }
int spin_fault() { return g_spin_fault; }
static thread_local char g_err_dev[640] = "";

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
// codes of the device error word: (kind << 24) | detail.  kind 1 = split-K / stream-K consumer of the ping-pong GEMM (detail: flag index)
extern "C" int32_t ifx_device_error(int32_t clear) {
  unsigned* w = ifx::g_dev_err;
  if (w == nullptr) return 0;
  const unsigned v = __atomic_load_n(w, __ATOMIC_ACQUIRE);
  if (v != 0 && clear) __atomic_store_n(w, 0u, __ATOMIC_RELEASE);
  return (int32_t)v;
}
extern "C" const char* ifx_last_error(void) {
  const unsigned v = (unsigned)ifx_device_error(1);
  if (v == 0) return ifx::g_err;
  const unsigned kind = v >> 24, detail = v & 0xffffffu;
  snprintf(ifx::g_err_dev, sizeof(ifx::g_err_dev),
           "device: a kernel gave up a wait (%s, flag %u) after the spin budget — the producing workgroup was not resident or did not "
           "finish; the results of that launch are invalid%s%s",
           kind == 1 ? "split-K / stream-K consumer of ifx_gemm_bf16_ws / ifx_gemm_q8_ws" : "unknown waiter", detail,
           ifx::g_err[0] ? "; last host error: " : "", ifx::g_err);
  return ifx::g_err_dev;
}
extern "C" const char* ifx_arch(void) { return "gfx950"; }
extern "C" int ifx_set_option(const char* key, int32_t value) {
  if (key && !strcmp(key, "gemm_variant") && value >= 0 && value <= 29) { ifx::g_gemm_variant = value; return IFX_OK; }
  if (key && !strcmp(key, "gemm_small_split") && (value == 0 || value == 1)) { ifx::g_gemm_small_split = value; return IFX_OK; }
  if (key && !strcmp(key, "attn_variant") && value >= 0 && value <= 7) { ifx::g_attn_variant = value; return IFX_OK; }
  if (key && !strcmp(key, "conv_variant") && value >= 0 && value <= 1) { ifx::g_conv_variant = value; return IFX_OK; }
  if (key && !strcmp(key, "attn_debug_counters") && (value == 0 || value == 1)) {
    if (value == 1) {
      if (ifx::g_attn_dbg == nullptr && hipMalloc((void**)&ifx::g_attn_dbg, 64) != hipSuccess) {
        ifx::g_attn_dbg = nullptr;
        ifx::set_error("ifx_set_option: attn_debug_counters: hipMalloc failed");
        return IFX_ELAUNCH;
      }
      (void)hipDeviceSynchronize();
      (void)hipMemset(ifx::g_attn_dbg, 0, 64);
    }
    ifx::g_attn_dbg_on = value == 1;
    return IFX_OK;
  }
  if (key && !strcmp(key, "spin_timeout_ms") && value >= 1 && value <= 600000) { ifx::g_spin_timeout_ms = value; return IFX_OK; }
  if (key && !strcmp(key, "spin_fault") && (value == 0 || value == 1)) { ifx::g_spin_fault = value; return IFX_OK; }
  ifx::set_error("ifx_set_option: unknown key or value out of range: %s = %d", key ? key : "(null)", (int)value);
  return IFX_EINVAL;
}
extern "C" int ifx_get_option(const char* key, int32_t* value) {
  if (key == nullptr || value == nullptr) { ifx::set_error("ifx_get_option: null argument"); return IFX_EINVAL; }
  if (!strcmp(key, "gemm_variant")) { *value = ifx::gemm_variant(); return IFX_OK; }
  if (!strcmp(key, "gemm_small_split")) { *value = ifx::gemm_small_split(); return IFX_OK; }
  if (!strcmp(key, "attn_variant")) { *value = ifx::attn_variant(); return IFX_OK; }
  if (!strcmp(key, "conv_variant")) { *value = ifx::conv_variant(); return IFX_OK; }
  if (!strcmp(key, "attn_debug_counters")) { *value = ifx::g_attn_dbg_on ? 1 : 0; return IFX_OK; }
  if (!strcmp(key, "attn_rescale_count")) {
    unsigned v = 0;
    if (ifx::g_attn_dbg != nullptr) {
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(&v, ifx::g_attn_dbg, sizeof(v), hipMemcpyDeviceToHost);
    }
    *value = (int32_t)(v & 0x7fffffffu);
    return IFX_OK;
  }
  if (!strcmp(key, "spin_timeout_ms")) { *value = (int32_t)(ifx::spin_timeout_ticks() / 100000LL); return IFX_OK; }
  if (!strcmp(key, "spin_fault")) { *value = ifx::spin_fault(); return IFX_OK; }
  ifx::set_error("ifx_get_option: unknown key: %s", key);
  return IFX_EINVAL;
}
