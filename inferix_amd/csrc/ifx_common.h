// Shared device/host helpers for libinferix_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/inferix_hip.h"

namespace ifx {

constexpr int kWave = 64;  // CDNA wavefront

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// bf16 <-> f32.  bf16 -> f32 is exact; f32 -> bf16 is round-to-nearest-even
// (the __bf16 cast lowers to v_cvt_pk_bf16_f32 on gfx950), which is what
// torch's .to(bfloat16) does on both CPU and GPU.
__device__ __forceinline__ float bf2f(unsigned short u) {
  return __builtin_bit_cast(float, (unsigned int)u << 16);
}
__device__ __forceinline__ unsigned short f2bf(float f) {
  return __builtin_bit_cast(unsigned short, static_cast<__bf16>(f));
}
// round an f32 value to the nearest bf16 and come back to f32 (a bf16 "module boundary")
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// logical token -> physical slot of a paged KV view (see ifx_kv_view)
struct KvAddr {
  const int32_t* pt;
  int32_t ps;
  int32_t seg_split, seg_delta;   // no table: logical tokens >= seg_split (when > 0) live seg_delta slots further on (ifx_kv_view)
  __device__ __forceinline__ int slot(int t) const {
    if (pt == nullptr) return (seg_split > 0 && t >= seg_split) ? t + seg_delta : t;
    int pg = t / ps;
    return pt[pg] * ps + (t - pg * ps);
  }
};

// host side -----------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
int gemm_variant();   // 0 auto, 1 register-staged 128x128, 2..8 LDS-DMA tiles (see include/inferix_hip.h)
int attn_variant();   // 0 auto (= 5 for large launches), 1 four-wave kernel, 2 ping-pong, 3 three groups, 4 free-running, 5 software-pipelined

}  // namespace ifx

#define IFX_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ifx::set_error(__VA_ARGS__);        \
      return IFX_EINVAL;                  \
    }                                     \
  } while (0)
