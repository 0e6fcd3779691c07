// Peer-to-peer plumbing of the sequence-parallel K/V exchange (one process per GPU, xGMI): device allocations that other
// processes can map (HIP IPC handles — dmabuf on this driver, HSA_ENABLE_IPC_MODE_LEGACY=0), and stream-ordered flags.
// Protocol per layer (inferix_amd/sequence_parallel.py, PeerStoreExchange):
//   ready : after its own cache roll, rank r stores epoch e into ready[r] of every peer       (ifx_peer_signal)
//   push  : r waits for ready[*] >= e in its own flag block (ifx_peer_wait), then stores its K/V rows into every peer's cache slots
//           (ifx_rmsnorm_rope_kv_push, ifx_norm.hip) and stores e into done[r] of every peer  (ifx_peer_signal)
//   use   : r waits for done[*] >= e, then attends to the new block.
// The flags live in fine-grained device memory (visible to a running kernel on another GPU); data rows are ordinary device memory:
// the signalling kernel starts after the push kernel has ended (its stores are released at kernel end), and the consumer's attention
// kernel starts after the wait kernel has ended (its caches are invalidated at kernel start).
#include <string.h>

#include "ifx_common.h"

using namespace ifx;

namespace ifx {

struct PeerFlagPtrs {
  int32_t* f[IFX_MAX_PEERS];
  int n;
};

__global__ void peer_signal_kernel(PeerFlagPtrs pf, int index, int value) {
  const int p = threadIdx.x;
  if (p >= pf.n) return;
  __threadfence_system();
  __hip_atomic_store(pf.f[p] + index, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// lane p waits until flags[p] >= value; gives up after `timeout_ticks` of the 100 MHz wall clock and reports through *status
__global__ void peer_wait_kernel(const int32_t* flags, int count, int value, long long timeout_ticks, int32_t* status) {
  const int p = threadIdx.x;
  if (p >= count) return;
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(flags + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > timeout_ticks) {
      if (status) atomicExch(status, 1 + p);
      return;
    }
  }
}

}  // namespace ifx

extern "C" int ifx_peer_alloc(int64_t bytes, int32_t fine_grained, void** ptr) {
  IFX_REQUIRE(ptr && bytes > 0, "ifx_peer_alloc: bad arguments");
  hipError_t e = fine_grained ? hipExtMallocWithFlags(ptr, (size_t)bytes, hipDeviceMallocFinegrained) : hipMalloc(ptr, (size_t)bytes);
  if (e != hipSuccess) {
    set_error("ifx_peer_alloc: %lld bytes: %s", (long long)bytes, hipGetErrorString(e));
    return IFX_ELAUNCH;
  }
  e = hipMemset(*ptr, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    set_error("ifx_peer_alloc: clearing the allocation: %s", hipGetErrorString(e));
    return IFX_ELAUNCH;
  }
  return IFX_OK;
}

extern "C" int ifx_peer_free(void* ptr) {
  if (!ptr) return IFX_OK;
  const hipError_t e = hipFree(ptr);
  if (e != hipSuccess) {
    set_error("ifx_peer_free: %s", hipGetErrorString(e));
    return IFX_ELAUNCH;
  }
  return IFX_OK;
}

extern "C" int ifx_peer_export(const void* ptr, uint8_t* handle, int64_t* offset) {
  IFX_REQUIRE(ptr && handle && offset, "ifx_peer_export: null argument");
  static_assert(sizeof(hipIpcMemHandle_t) == IFX_PEER_HANDLE_BYTES, "IPC handle size");
  // a handle names a whole allocation: export the base of the one `ptr` lies in (a caching allocator hands out interior pointers)
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  hipError_t e = hipMemGetAddressRange(&base, &size, const_cast<void*>(ptr));
  if (e != hipSuccess) {
    set_error("ifx_peer_export: hipMemGetAddressRange: %s", hipGetErrorString(e));
    return IFX_ELAUNCH;
  }
  hipIpcMemHandle_t h;
  e = hipIpcGetMemHandle(&h, base);
  if (e != hipSuccess) {
    set_error("ifx_peer_export: hipIpcGetMemHandle: %s (HSA_ENABLE_IPC_MODE_LEGACY must be 0 on this driver)", hipGetErrorString(e));
    return IFX_ELAUNCH;
  }
  memcpy(handle, &h, sizeof(h));
  *offset = (int64_t)((const char*)ptr - (const char*)base);
  return IFX_OK;
}

extern "C" int ifx_peer_open(const uint8_t* handle, void** ptr) {
  IFX_REQUIRE(handle && ptr, "ifx_peer_open: null argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  const hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    set_error("ifx_peer_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    return IFX_ELAUNCH;
  }
  return IFX_OK;
}

extern "C" int ifx_peer_close(void* ptr) {
  if (!ptr) return IFX_OK;
  const hipError_t e = hipIpcCloseMemHandle(ptr);
  if (e != hipSuccess) {
    set_error("ifx_peer_close: %s", hipGetErrorString(e));
    return IFX_ELAUNCH;
  }
  return IFX_OK;
}

extern "C" int ifx_peer_signal(const ifx_peer_flags* peers, int32_t index, int32_t value, void* stream) {
  IFX_REQUIRE(peers && peers->count >= 1 && peers->count <= IFX_MAX_PEERS && index >= 0, "ifx_peer_signal: bad arguments");
  PeerFlagPtrs pf{};
  pf.n = peers->count;
  for (int p = 0; p < pf.n; ++p) {
    IFX_REQUIRE(peers->flags[p], "ifx_peer_signal: flag block %d is null", p);
    pf.f[p] = peers->flags[p];
  }
  hipLaunchKernelGGL(peer_signal_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pf, index, value);
  return check_launch("ifx_peer_signal");
}

extern "C" int ifx_peer_wait(const int32_t* flags, int32_t count, int32_t value, int32_t timeout_ms, int32_t* status, void* stream) {
  IFX_REQUIRE(flags && count >= 1 && count <= 64 && timeout_ms > 0, "ifx_peer_wait: bad arguments");
  hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flags, count, value, (long long)timeout_ms * 100000LL,
                     status);
  return check_launch("ifx_peer_wait");
}
