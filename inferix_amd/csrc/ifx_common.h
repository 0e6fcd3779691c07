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

// exact-erf GELU, 0.5 x (1 + erf(x / sqrt 2)), as torch evaluates it in fp32 (F.gelu on a bf16 CPU tensor) — INCLUDING the
// cancellation of 1 + erf in the negative tail, which is part of the reference's result.  erf comes from the complementary error
// function in Chebyshev form (Numerical Recipes `erfcc`: erfc(z) = t exp(-z^2 + P(t)), t = 1 / (1 + z / 2), fractional error
// < 1.2e-7 for z >= 0): 1 - erfc rounds to the same fp32 value as libm's erff except where erff itself is off by an ulp.  Over
// all 65 280 finite bf16 inputs the bf16 result equals the erff-based expression's bit for bit (tests/test_hip_kernels.py); it
// costs two transcendentals + ~22 VALU operations per element against erff's ~60 with branches — the FFN-up epilogue of a
// 4680 x 8960 x 1536 launch went from +34 us to +6 us over the bias-only epilogue.
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fminf(fabsf(x) * 0.7071067811865476f, 12.0f);      // erfc(12) is 0 in fp32; keeps z * z finite for huge |x|
  const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, z, 1.0f));
  float p = 0.17087277f;
  p = fmaf(p, t, -0.82215223f);
  p = fmaf(p, t, 1.48851587f);
  p = fmaf(p, t, -1.13520398f);
  p = fmaf(p, t, 0.27886807f);
  p = fmaf(p, t, -0.18628806f);
  p = fmaf(p, t, 0.09678418f);
  p = fmaf(p, t, 0.37409196f);
  p = fmaf(p, t, 1.00002368f);
  p = fmaf(p, t, -1.26551223f);
  const float arg = fmaf(-z, z, p);                                  // <= 0
  // exp(arg) = exp2(a) * (1 + r ln 2) with a + r = arg * log2(e) carried in two floats: v_exp_f32 alone would lose
  // |arg| * 2^-24 of relative accuracy to the rounding of the product
  const float a = arg * 1.4426950408889634f;
  float r = fmaf(arg, 1.4426950408889634f, -a);
  r = fmaf(arg, 1.9259629911266175e-8f, r);
  const float ex = __builtin_amdgcn_exp2f(a) * fmaf(r, 0.6931471805599453f, 1.0f);
  const float erfc_z = t * ex;
  const float e = x >= 0.f ? 1.0f - erfc_z : erfc_z - 1.0f;          // erf(x / sqrt 2)
  return 0.5f * x * (1.0f + e);
}

// tanh-GELU of the FFN epilogues, x * sigmoid(2 u) with u = sqrt(2 / pi) (x + 0.044715 x^3), constants folded into the exponent:
// exp(-2u) = exp2(x (c1 + c2 x^2)) — five VALU operations and two transcendentals per element where the unfolded form (u, then
// __expf(-2u) = a multiply by log2(e) + v_exp_f32) takes nine; 128 elements per lane per 256 x 256 tile.  On every finite bf16 input the
// bf16 result is the unfolded form's, bit for bit (checked on the CPU over all 65 280 inputs; both differ from torch's fp32 tanh
// formula on the same 150, in the saturated tails).  v_rcp_f32 instead of the IEEE division sequence as before.
__device__ __forceinline__ float gelu_tanh_fast(float x) {
  constexpr float c1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, c2 = c1 * 0.044715f;
  const float a = __builtin_fmaf(x * x, c2, c1) * x;
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a));
}

// x / s for the per-token quantisers, three VALU operations per element instead of the ~10 of the IEEE sequence (v_div_scale x2,
// v_rcp, four FMAs, v_div_fmas, v_div_fixup):  q0 = x r;  e = fma(-q0, s, x) (exact residual);  q = fma(e, r, q0)  with r = rcp(s)
// computed once per row.  CORRECTLY ROUNDED for the operands this is used on — x a bf16 value, s = RN(amax / QMAX) with amax a bf16
// value, QMAX 448 or 127 — for r anywhere within one ulp of 1 / s: checked exhaustively over every (x significand, amax significand)
// pair and r in {RN(1/s), its two neighbours} (3.5 M quotients per QMAX, 0 mismatches against RN(x / s); the result does not depend
// on the exponents while everything stays normal).  Callers take the IEEE division for rows whose scale is outside [2^-60, 2^60]
// (wave-uniform test), where v_rcp_f32 or the residual could leave the normal range.
struct RowDivisor {
  float s, r;
  // `scale` is the same in every lane of the wave (it comes out of a wave reduction); readfirstlane tells the compiler so, and
  // `fast()` becomes a scalar branch the caller takes ONCE around its loops (a per-element test would be an exec-mask branch each)
  __device__ __forceinline__ explicit RowDivisor(float scale)
      : s(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scale)))), r(__builtin_amdgcn_rcpf(s)) {}
  __device__ __forceinline__ bool fast() const { return s > 8.67361738e-19f && s < 1.15292150e18f; }
  template <bool FAST>
  __device__ __forceinline__ float div(float x) const {
    if (FAST) {
      const float q0 = x * r;
      const float e = __builtin_fmaf(-q0, s, x);
      return __builtin_fmaf(e, r, q0);
    }
    return x / s;
  }
};

// Wave-wide reductions on the DPP path (no LDS traffic): hipcc lowers `__shfl_xor` to ds_bpermute_b32 + s_waitcnt lgkmcnt(0), an LDS
// round trip of ~100 cycles per step — twelve of them in a row of LayerNorm, back to back on the critical path of a kernel that lives
// for 8 us.  Steps: quad_perm xor 1, xor 2, row_half_mirror (= xor 4 once quads agree), row_mirror (= xor 8), row_bcast15 into rows
// 1 / 3, row_bcast31 into rows 2 / 3; lane 63 then holds the total, read back as a scalar.  The association tree is that of an
// ASCENDING-offset xor butterfly (1, 2, 4, 8 inside a row, then (S0 + S1) + (S2 + S3) over the four 16-lane row sums; fp add is
// commutative): bit-identical to that butterfly — NOT to the descending-offset loop (32 .. 1) the kernels used before round 4, so
// LayerNorm / RMSNorm / softmax statistics moved in their last fp32 bits at that commit (every golden is compared
// through the bf16 / ULP rules of tests/util.py, none pins fp32 statistics bit for bit).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_or<0xB1, 0xf>(0.f, v);      // quad_perm [1, 0, 3, 2]
  v += dpp_or<0x4E, 0xf>(0.f, v);      // quad_perm [2, 3, 0, 1]
  v += dpp_or<0x141, 0xf>(0.f, v);     // row_half_mirror
  v += dpp_or<0x140, 0xf>(0.f, v);     // row_mirror
  v += dpp_or<0x142, 0xa>(0.f, v);     // row_bcast15 -> rows 1, 3
  v += dpp_or<0x143, 0xc>(0.f, v);     // row_bcast31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_or<0xB1, 0xf>(v, v));
  v = fmaxf(v, dpp_or<0x4E, 0xf>(v, v));
  v = fmaxf(v, dpp_or<0x141, 0xf>(v, v));
  v = fmaxf(v, dpp_or<0x140, 0xf>(v, v));
  v = fmaxf(v, dpp_or<0x142, 0xa>(v, v));
  v = fmaxf(v, dpp_or<0x143, 0xc>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
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
int gemm_small_split();   // 1: launches of at most one workgroup per CU may split K between the wave groups of a workgroup (row-count dependent bits)
unsigned* device_error_word();    // pinned host word, mapped into every device, that a kernel raises when it gives up a wait (ifx_core.hip); may be nullptr
long long spin_timeout_ticks();   // budget of a device-side wait in ticks of the 100 MHz wall clock (option "spin_timeout_ms", default 2000 ms)
int spin_fault();                 // lab / tests: 1 = split-K producers do not raise their flag, so that the consumer's wait runs into its budget
int conv_variant();   // 0 auto (persistent ping-pong kernel where it applies), 1 the lock-step kernel of round 1
unsigned* attn_debug_counter();   // tests: device word counting lazy-maximum rescales of the ping-pong attention kernels (option "attn_debug_counters"), else nullptr
int attn_variant();   // 0 auto (= 7 for large launches), 1 four-wave kernel, 2 ping-pong, 3 three groups, 4 free-running, 5 software-pipelined, 6 its two-per-CU form, 7 its four-times-unrolled form

}  // namespace ifx

#define IFX_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ifx::set_error(__VA_ARGS__);        \
      return IFX_EINVAL;                  \
    }                                     \
  } while (0)
