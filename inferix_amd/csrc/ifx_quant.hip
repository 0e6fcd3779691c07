// Dynamic 8-bit linear layers (gfx950): per-token activation quantisation + FP8 (OCP e4m3) / INT8 MFMA GEMM
// with per-token x per-channel dequantisation and the block's fused epilogues.
//
// Replaces the DAX `quantize_dynamic` linears the reference wires in
// example/quantization/run_self_forcing_quantized.py:19-23,47-64 (dynamic per-token activation x per-channel
// weight, FP8 or INT8).  DAX is an un-vendored, unpinned dependency: its arithmetic is NOT in the reference tree,
// so the scheme is defined here (and restated in oracle/quant_oracle.py) — parity with DAX itself is unpinned:
//
//   s_a[m] = max_k |x[m,k]| / QMAX          (fp32; 1.0 for an all-zero row)      QMAX = 448 (e4m3) | 127 (int8)
//   xq[m,k] = cast( clamp(x[m,k] / s_a[m], +-QMAX) )      e4m3: round-to-nearest-even ; int8: rint
//   s_w[n], wq[n,k] : the same per OUTPUT channel of W, computed once when the module is quantised
//   y[m,n]  = epilogue( bf16( acc[m,n] * (s_a[m] * s_w[n]) + bias[n] ) )         acc: fp32 (fp8) / exact int32 (int8)
//
// Kernels:
//   quant_rows_kernel : one wavefront per token row, two passes over the row (abs-max, then scale+cast; the second
//                       pass hits L2), 16-byte loads, 8-byte stores: HBM-bound (reads 2 B, writes 1 B per element).
//   gemm_q8_kernel    : 128 x 128 x 128(bytes of K) tiles, 4 waves, 2x2 fragments of 32x32 per wave;
//                       v_mfma_f32_32x32x16_fp8_fp8 (two per 16-byte fragment read) or v_mfma_i32_32x32x32_i8;
//                       operands staged global -> registers -> LDS (double buffered), 128-byte rows with the
//                       16-byte chunk index XOR ((row >> 1) & 7): conflict-free ds_read_b128 for 32-row fragments;
//                       half the operand bytes of the bf16 kernel per FLOP.
#include <stdlib.h>

#include <type_traits>

#include "ifx_common.h"

namespace ifx {

typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) long i64x2;

// ---------------------------------------------------------------------------------------------------------
// NCH = ceil(K / 512) chunks of 8 elements per lane: the whole row is loaded ONCE with all its 16-byte loads in flight
// (NCH <= 18, K <= 9216 — the 8960-wide FFN activations); NCH == 0 is the generic two-pass loop for wider rows.
template <bool FP8, int NCH>
__global__ __launch_bounds__(256) void quant_rows_kernel(const unsigned short* __restrict__ x, int ldx,
                                                         unsigned char* __restrict__ q, int ldq,
                                                         float* __restrict__ scale, int rows, int K) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const unsigned short* xr = x + (size_t)r * ldx;
  constexpr float QMAX = FP8 ? 448.0f : 127.0f;
  auto pack = [&](const float (&v)[8]) -> u32x2 {
    if (FP8) {
      unsigned w0 = 0, w1 = 0;
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w0, false);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], w1, false);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], w1, true);
      return u32x2{w0, w1};
    }
    unsigned w[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i >> 2] |= ((unsigned)(int)rintf(v[i]) & 0xffu) << (8 * (i & 3));
    return u32x2{w[0], w[1]};
  };
  unsigned char* qr = q + (size_t)r * ldq;
  if constexpr (NCH > 0) {
    // branch-free: a lane beyond K reads the row's first chunk (a valid address) and is zeroed afterwards — with the load inside
    // `if (col < K)` hipcc waited for every chunk before requesting the next (ifx_norm.hip::load_chunks)
    u16x8 u[NCH > 0 ? NCH : 1];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 512 + lane * 8;
      u[c] = *reinterpret_cast<const u16x8*>(xr + (col < K ? col : 0));
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (c * 512 + lane * 8 >= K) u[c] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(bf2f(u[c][i])));
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax / QMAX : 1.0f;
    if (lane == 0) scale[r] = s;
    const RowDivisor rdiv(s);        // the exact three-operation x / s (ifx_common.h); 8960-wide rows: 140 quotients per lane
    auto emit = [&](auto fastc) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 512 + lane * 8;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fminf(fmaxf(rdiv.template div<decltype(fastc)::value>(bf2f(u[c][i])), -QMAX), QMAX);
        const u32x2 pk = pack(v);
        if (col < K) *reinterpret_cast<u32x2*>(qr + col) = pk;
      }
    };
    if (rdiv.fast()) emit(std::true_type{});
    else emit(std::false_type{});
    return;
  }
  float amax = 0.f;
  for (int col = lane * 8; col < K; col += 512) {
    const u16x8 u = *reinterpret_cast<const u16x8*>(xr + col);
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(bf2f(u[i])));
  }
  amax = wave_max(amax);
  const float s = amax > 0.f ? amax / QMAX : 1.0f;
  if (lane == 0) scale[r] = s;
  for (int col = lane * 8; col < K; col += 512) {
    const u16x8 u = *reinterpret_cast<const u16x8*>(xr + col);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fminf(fmaxf(bf2f(u[i]) / s, -QMAX), QMAX);
    *reinterpret_cast<u32x2*>(qr + col) = pack(v);
  }
}

// ---------------------------------------------------------------------------------------------------------
struct EpiArgsQ {
  const float* sa;
  const float* sw;
  const unsigned short* bias;
  const unsigned short* residual;
  int ld_res;
  const unsigned short* mod;
  int mod_slots, gate_slot, rows_per_group;
  const float* qdiv = nullptr;   // GELU epilogues only: y is e4m3 bytes (ldy in bytes), q = div_clamp_to(bf16 result, qdiv[n])
  int q_via_bf16 = 0;
};

// div_clamp_to (dit_module.py:367-387) of four bf16 values -> four e4m3 bytes
__device__ __forceinline__ unsigned quant4_e4m3(const float (&x)[4], const f32x4 d, int via_bf16) {
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float t = fminf(fmaxf(x[e] / d[e], -448.0f), 448.0f);
    v[e] = via_bf16 ? rbf(t) : t;
  }
  unsigned w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
}

// exact (erf) GELU as torch.nn.functional.gelu evaluates it on a bf16 tensor: fp32 math, one rounding (MAGI CustomMLP,
// inferix/models/magi/dit/dit_module.py:552).  Selected at run time inside the GELU epilogue instantiation: the epilogue's
// otherwise unused `gate_slot` field carries 1 for IFX_EPI_GELU_ERF.

template <bool FP8, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_q8_kernel(const unsigned char* __restrict__ x, int ldx,
                                                         const unsigned char* __restrict__ w,
                                                         unsigned short* __restrict__ y, int ldy, int M, int N, int K,
                                                         int tiles_m, int tiles_n, EpiArgsQ ea) {
  constexpr int BM = 128, BN = 128, BKB = 128;   // K tile = 128 one-byte elements
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  constexpr int GM = 8;
  const int total = tiles_m * tiles_n, per_xcd = (total + 7) / 8;
  const int t_id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || t_id >= total) return;
  const int grp_sz = GM * tiles_n;
  const int first_m = (t_id / grp_sz) * GM;
  const int gm = min(GM, tiles_m - first_m);
  const int rem = t_id % grp_sz;
  const int tile_m = first_m + rem % gm, tile_n = rem / gm;
  const int m_base = tile_m * BM, n_base = tile_n * BN;

  const unsigned char* gx[4];
  const unsigned char* gw[4];
  int lds_off[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int qi = tid + 256 * p, row = qi >> 3, c = qi & 7;
    gx[p] = x + (size_t)min(m_base + row, M - 1) * ldx + c * 16;
    gw[p] = w + (size_t)min(n_base + row, N - 1) * K + c * 16;
    lds_off[p] = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
  }
  u32x4 rx[4], rw[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      rx[p] = *reinterpret_cast<const u32x4*>(gx[p] + (size_t)kt * BKB);
      rw[p] = *reinterpret_cast<const u32x4*>(gw[p] + (size_t)kt * BKB);
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* bx = smem + buf * 32768;
    unsigned char* bw = bx + 16384;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *reinterpret_cast<u32x4*>(bx + lds_off[p]) = rx[p];
      *reinterpret_cast<u32x4*>(bw + lds_off[p]) = rw[p];
    }
  };

  f32x16 facc[2][2];
  i32x16 iacc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        facc[i][j][r] = 0.f;
        iacc[i][j][r] = 0;
      }

  const int KT = K / BKB;
  gload(0);
  lstore(0);
  __syncthreads();
  const int l31 = lane & 31, hi = lane >> 5;
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) gload(kt + 1);
    const unsigned char* bx = smem + buf * 32768;
    const unsigned char* bw = bx + 16384;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int c = 2 * p + hi;
      i32x4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wn * 64 + i * 32 + l31;
        a[i] = *reinterpret_cast<const i32x4*>(bw + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wm * 64 + j * 32 + l31;
        b[j] = *reinterpret_cast<const i32x4*>(bx + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (FP8) {
            // one 16-byte fragment = two K=16 steps; the k-slot relabelling is the same for both operands
            const i64x2 al = __builtin_bit_cast(i64x2, a[i]), bl = __builtin_bit_cast(i64x2, b[j]);
            facc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(al[0], bl[0], facc[i][j], 0, 0, 0);
            facc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(al[1], bl[1], facc[i][j], 0, 0, 0);
          } else {
            iacc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], iacc[i][j], 0, 0, 0);
          }
        }
    }
    if (kt + 1 < KT) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds D[n = n0 + 32i + 8g + 4hi + e][m = m0 + 32j + l31]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m_base + wm * 64 + j * 32 + l31;
    if (m >= M) continue;
    const float sa = ea.sa[m];
    const unsigned short* gate_row = nullptr;
    if (EPI == IFX_EPI_GATE_RES)
      gate_row = ea.mod + ((size_t)(m / ea.rows_per_group) * ea.mod_slots + ea.gate_slot) * N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n_base + wn * 64 + i * 32 + g * 8 + hi * 4;
        if (n >= N) continue;
        const f32x4 swv = *reinterpret_cast<const f32x4*>(ea.sw + n);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float acc = FP8 ? facc[i][j][4 * g + e] : (float)iacc[i][j][4 * g + e];
          v[e] = acc * (sa * swv[e]);
        }
        if (ea.bias) {
          const u16x4 bv = *reinterpret_cast<const u16x4*>(ea.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bf2f(bv[e]);
        }
        u16x4 o;
        if (EPI == IFX_EPI_BIAS) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        } else if (EPI == IFX_EPI_GELU_TANH) {
if (ea.gate_slot) {   // exact-erf GELU (IFX_EPI_GELU_ERF): a scalar branch around the loop, not a per-element select
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = f2bf(gelu_erf_f(rbf(v[e])));
} else {
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = f2bf(gelu_tanh_fast(rbf(v[e])));
}
        } else {
          const u16x4 rv = *reinterpret_cast<const u16x4*>(ea.residual + (size_t)m * ea.ld_res + n);
          if (EPI == IFX_EPI_RESIDUAL) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(bf2f(rv[e]) + rbf(v[e]));
          } else {
            const u16x4 gv = *reinterpret_cast<const u16x4*>(gate_row + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(bf2f(rv[e]) + rbf(rbf(v[e]) * bf2f(gv[e])));
          }
        }
        if (EPI == IFX_EPI_GELU_TANH && ea.qdiv != nullptr) {
          const float ob[4] = {bf2f(o[0]), bf2f(o[1]), bf2f(o[2]), bf2f(o[3])};
          *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(y) + (size_t)m * ldy + n) =
              quant4_e4m3(ob, *reinterpret_cast<const f32x4*>(ea.qdiv + n), ea.q_via_bf16);
        } else {
          *reinterpret_cast<u16x4*>(y + (size_t)m * ldy + n) = o;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// LDS-DMA variant for the large shapes: BM x BN tiles, K-steps of 64 one-byte elements, 8 waves, 4-stage ring of
// 64-byte rows — byte for byte the geometry of the bf16 K=32 kernel (ifx_gemm_glds.hip: gemm_big_kernel), so the DMA
// map, the chunk swizzle ((row >> 2) & 3), the fragment reads and the LDS-transposed epilogue are the same; per FLOP it
// moves half the operand bytes, which is what bounds the bf16 GEMMs (profiles/r1d_gemm_ablation.md).
//   FP8 : PP_F8F6F4 = 1 -> one v_mfma_scale_f32_32x32x64_f8f6f4 (unit scales) per 32-byte fragment pair: the 5 PFLOP/s
//         instruction; 0 -> four v_mfma_f32_32x32x16_fp8_fp8.     INT8: two v_mfma_i32_32x32x32_i8.
// The K order inside a fragment is whatever the hardware uses — both operands are read with the same lane -> byte map,
// and a dot product does not care about a common permutation of k.
#ifndef IFX_Q8_F8F6F4
#define IFX_Q8_F8F6F4 1
#endif
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((address_space(3))) void* q8_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* q8_gbl_ptr_t;

// BKB = bytes of K per operand row of a stage: 64 (four stages) or 128 (two / three stages).  LDS-DMA moves rows of >= 128 B from L2 at
// twice the rate of 64-byte rows (tools/probe_dma.hip), the same finding that moved the bf16 256x256 tile to 64-deep stages.
template <bool FP8, int BM, int BN, int WAVES_M, int EPI, int BKB = 64, int NST = 4>
__global__ __launch_bounds__(512) void gemm_q8_dma_kernel(const unsigned char* __restrict__ x, int ldx,
                                                          const unsigned char* __restrict__ w,
                                                          unsigned short* __restrict__ y, int ldy, int M, int N, int K,
                                                          int tiles_m, int total, int per_xcd, EpiArgsQ ea) {
  constexpr int CPR = BKB / 16, RPP = 64 / CPR;           // 16-byte chunks per row (4 | 8), rows per 1 KiB DMA piece (16 | 8)
  constexpr int SW_SH = BKB == 64 ? 2 : 1;                 // swizzle phase: (row >> SW_SH) & (CPR - 1)
  constexpr int STAGE = (BM + BN) * BKB;
  constexpr int A_OFF = 0, B_OFF = BM * BKB;
  constexpr int WAVES_N = 8 / WAVES_M;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TJ = WM / 32, TI = WN / 32;
  constexpr int PA = BM / (8 * RPP), PB = BN / (8 * RPP), P = PA + PB;
  static_assert((NST - 2) * P <= 63 && NST >= 2 && NST <= 4, "stages");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
  const int t_id = xcd * per_xcd + slot_i;
  if (slot_i >= per_xcd || t_id >= total) return;
  constexpr int GM = 4;
  const int tiles_n = total / tiles_m;
  const int grp_sz = GM * tiles_n;
  const int first_m = (t_id / grp_sz) * GM;
  const int gm = min(GM, tiles_m - first_m);
  const int rem = t_id % grp_sz;
  const int tile_m = first_m + rem % gm, tile_n = rem / gm;
  const int m_base = tile_m * BM, n_base = tile_n * BN;
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;

  const int r16 = lane / CPR, pc = lane % CPR;
  const unsigned char* src_a[PA];
  const unsigned char* src_b[PB];
#pragma unroll
  for (int r = 0; r < PA; ++r) {
    const int row = (r * 8 + wave) * RPP + r16;
    src_a[r] = x + (size_t)min(m_base + row, M - 1) * ldx + (pc ^ ((row >> SW_SH) & (CPR - 1))) * 16;
  }
#pragma unroll
  for (int r = 0; r < PB; ++r) {
    const int row = (r * 8 + wave) * RPP + r16;
    src_b[r] = w + (size_t)min(n_base + row, N - 1) * K + (pc ^ ((row >> SW_SH) & (CPR - 1))) * 16;
  }
  auto issue = [&](int kt) {
    unsigned char* st = smem + (kt % NST) * STAGE;
    const size_t ko = (size_t)kt * BKB;
#pragma unroll
    for (int r = 0; r < PA; ++r)
      __builtin_amdgcn_global_load_lds((q8_gbl_ptr_t)(src_a[r] + ko), (q8_lds_ptr_t)(st + A_OFF + (r * 8 + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int r = 0; r < PB; ++r)
      __builtin_amdgcn_global_load_lds((q8_gbl_ptr_t)(src_b[r] + ko), (q8_lds_ptr_t)(st + B_OFF + (r * 8 + wave) * 1024), 16, 0, 0);
  };

  f32x16 facc[TI][TJ];
  i32x16 iacc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        facc[i][j][r] = 0.f;
        iacc[i][j][r] = 0;
      }

  const int KT = K / BKB;
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < KT) issue(i);

  const int l31 = lane & 31, hi = lane >> 5;
  int a_off[TJ], b_off[TI], a_swz[TJ], b_swz[TI];
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int row = wm * WM + j * 32 + l31;
    a_off[j] = A_OFF + row * BKB;
    a_swz[j] = (row >> SW_SH) & (CPR - 1);
  }
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int row = wn * WN + i * 32 + l31;
    b_off[i] = B_OFF + row * BKB;
    b_swz[i] = (row >> SW_SH) & (CPR - 1);
  }

  for (int kt = 0; kt < KT; ++kt) {
    const int later = min(KT - 1 - kt, NST - 2);
    if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + NST - 1 < KT) issue(kt + NST - 1);
    const unsigned char* st = smem + (kt % NST) * STAGE;
#pragma unroll
    for (int kh = 0; kh < BKB / 64; ++kh) {                // 64 bytes of K per MFMA step
    i32x4 fa[2][TJ], fb[2][TI];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = 4 * kh + 2 * ks + hi;
#pragma unroll
      for (int i = 0; i < TI; ++i) fb[ks][i] = *reinterpret_cast<const i32x4*>(st + b_off[i] + ((c ^ b_swz[i]) << 4));
#pragma unroll
      for (int j = 0; j < TJ; ++j) fa[ks][j] = *reinterpret_cast<const i32x4*>(st + a_off[j] + ((c ^ a_swz[j]) << 4));
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        if (FP8) {
#if IFX_Q8_F8F6F4
          const i32x8 av = __builtin_shufflevector(fb[0][i], fb[1][i], 0, 1, 2, 3, 4, 5, 6, 7);
          const i32x8 bv = __builtin_shufflevector(fa[0][j], fa[1][j], 0, 1, 2, 3, 4, 5, 6, 7);
          facc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, facc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#else
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const i64x2 al = __builtin_bit_cast(i64x2, fb[ks][i]), bl = __builtin_bit_cast(i64x2, fa[ks][j]);
            facc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(al[0], bl[0], facc[i][j], 0, 0, 0);
            facc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(al[1], bl[1], facc[i][j], 0, 0, 0);
          }
#endif
        } else {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            iacc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[ks][i], fa[ks][j], iacc[i][j], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: v = bf16(acc * sa[m] * sw[n] + bias[n]) transposed through LDS per wave, then 16-byte row accesses
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  constexpr int RB = WN * 2, CR = RB / 16, RP = 64 / CR;
  unsigned char* tw = smem + wave * (WM * RB);
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int mrow = j * 32 + l31;
    const int m = m_base + wm * WM + mrow;
    const float sa = ea.sa[min(m, M - 1)];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = i * 32 + g * 8 + hi * 4;
        const int n = n_base + wn * WN + nl;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (n < N) {
          const f32x4 swv = *reinterpret_cast<const f32x4*>(ea.sw + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float acc = FP8 ? facc[i][j][4 * g + e] : (float)iacc[i][j][4 * g + e];
            v[e] = acc * (sa * swv[e]);
          }
          if (ea.bias) {
            const u16x4 bv = *reinterpret_cast<const u16x4*>(ea.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bf2f(bv[e]);
          }
        }
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        const int chunk = (nl >> 3) ^ (mrow & (CR - 1));
        *reinterpret_cast<u16x4*>(tw + mrow * RB + chunk * 16 + (nl & 4) * 2) = o;
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {
    const int rr = lane / CR, cc = lane % CR;
#pragma unroll
    for (int p = 0; p < WM / RP; ++p) {
      const int mrow = p * RP + rr;
      const int m = m_base + wm * WM + mrow;
      const int n = n_base + wn * WN + cc * 8;
      const u16x8 vv = *reinterpret_cast<const u16x8*>(tw + mrow * RB + ((cc ^ (mrow & (CR - 1))) << 4));
      if (m >= M || n >= N) continue;
      u16x8 o;
      if (EPI == IFX_EPI_BIAS) {
        o = vv;
      } else if (EPI == IFX_EPI_GELU_TANH) {
if (ea.gate_slot) {   // exact-erf GELU (IFX_EPI_GELU_ERF): a scalar branch around the loop, not a per-element select
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = f2bf(gelu_erf_f(bf2f(vv[e])));
} else {
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = f2bf(gelu_tanh_fast(bf2f(vv[e])));
}
      } else {
        const u16x8 rv = *reinterpret_cast<const u16x8*>(ea.residual + (size_t)m * ea.ld_res + n);
        if (EPI == IFX_EPI_RESIDUAL) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(rv[e]) + bf2f(vv[e]));
        } else {
          const u16x8 gv = *reinterpret_cast<const u16x8*>(
              ea.mod + ((size_t)(m / ea.rows_per_group) * ea.mod_slots + ea.gate_slot) * N + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(rv[e]) + rbf(bf2f(vv[e]) * bf2f(gv[e])));
        }
      }
      if (EPI == IFX_EPI_GELU_TANH && ea.qdiv != nullptr) {
        const float lo[4] = {bf2f(o[0]), bf2f(o[1]), bf2f(o[2]), bf2f(o[3])};
        const float hi4[4] = {bf2f(o[4]), bf2f(o[5]), bf2f(o[6]), bf2f(o[7])};
        const unsigned w0 = quant4_e4m3(lo, *reinterpret_cast<const f32x4*>(ea.qdiv + n), ea.q_via_bf16);
        const unsigned w1 = quant4_e4m3(hi4, *reinterpret_cast<const f32x4*>(ea.qdiv + n + 4), ea.q_via_bf16);
        *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(y) + (size_t)m * ldy + n) = u32x2{w0, w1};
      } else {
        *reinterpret_cast<u16x8*>(y + (size_t)m * ldy + n) = o;
      }
    }
  }
}

template <bool FP8, int BM, int BN, int WAVES_M, int BKB = 64, int NST = 4>
static int launch_q8_dma(const unsigned char* x, int ldx, const unsigned char* w, unsigned short* y, int ldy, int M, int N,
                         int K, int mode, const EpiArgsQ& ea, hipStream_t s) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int total = tiles_m * tiles_n, per_xcd = (total + 7) / 8;
  const dim3 grid(per_xcd * 8), block(512);
  constexpr size_t ring = (size_t)NST * (BM + BN) * BKB, epi = (size_t)BM * BN * 2;
  static_assert(ring <= 160 * 1024, "LDS");
  constexpr size_t lds = ring > epi ? ring : epi;
#define IFX_LAUNCH_Q8D(E)                                                                                            \
  do {                                                                                                               \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      (void)hipFuncSetAttribute((const void*)gemm_q8_dma_kernel<FP8, BM, BN, WAVES_M, E, BKB, NST>,                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    hipLaunchKernelGGL((gemm_q8_dma_kernel<FP8, BM, BN, WAVES_M, E, BKB, NST>), grid, block, lds, s, x, ldx, w, y, ldy, M, N, K, \
                       tiles_m, total, per_xcd, ea);                                                                 \
  } while (0)
  switch (mode) {
    case IFX_EPI_BIAS: IFX_LAUNCH_Q8D(IFX_EPI_BIAS); break;
    case IFX_EPI_GELU_TANH: IFX_LAUNCH_Q8D(IFX_EPI_GELU_TANH); break;
    case IFX_EPI_RESIDUAL: IFX_LAUNCH_Q8D(IFX_EPI_RESIDUAL); break;
    case IFX_EPI_GATE_RES: IFX_LAUNCH_Q8D(IFX_EPI_GATE_RES); break;
    default: return IFX_EINVAL;
  }
#undef IFX_LAUNCH_Q8D
  return check_launch("ifx_gemm_q8(dma)");
}

}  // namespace ifx

using namespace ifx;

extern "C" int ifx_quant_per_token(const ifx_bf16* x, int32_t ldx, void* q, int32_t ldq, float* scale, int32_t rows,
                                   int32_t K, int32_t format, void* stream) {
  IFX_REQUIRE(x && q && scale && rows >= 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldq % 8 == 0,
              "ifx_quant_per_token: bad arguments (K %d)", K);
  IFX_REQUIRE(format == IFX_Q_FP8_E4M3 || format == IFX_Q_INT8, "ifx_quant_per_token: unknown format %d", format);
  if (rows == 0) return IFX_OK;
  const dim3 grid((rows + 3) / 4), block(256);
  const int nch = (K + 511) / 512;
#define IFX_LAUNCH_QR(NC)                                                                                          \
  do {                                                                                                             \
    if (format == IFX_Q_FP8_E4M3)                                                                                  \
      hipLaunchKernelGGL((quant_rows_kernel<true, NC>), grid, block, 0, (hipStream_t)stream, x, ldx, (unsigned char*)q, \
                         ldq, scale, rows, K);                                                                     \
    else                                                                                                           \
      hipLaunchKernelGGL((quant_rows_kernel<false, NC>), grid, block, 0, (hipStream_t)stream, x, ldx, (unsigned char*)q, \
                         ldq, scale, rows, K);                                                                     \
  } while (0)
  if (nch <= 1) IFX_LAUNCH_QR(1);
  else if (nch <= 3) IFX_LAUNCH_QR(3);
  else if (nch <= 6) IFX_LAUNCH_QR(6);
  else if (nch <= 10) IFX_LAUNCH_QR(10);
  else if (nch <= 18) IFX_LAUNCH_QR(18);
  else IFX_LAUNCH_QR(0);
#undef IFX_LAUNCH_QR
  return check_launch("ifx_quant_per_token");
}

// the persistent ping-pong tile with e4m3 operands (ifx_gemm_pp.hip, Q8 instantiations)
namespace ifx {
int launch_gemm_pp(const unsigned short* x, int ldx, const unsigned short* w, unsigned short* y, int ldy, int M, int N, int K,
                   int mode, const unsigned short* bias, const unsigned short* residual, int ld_res, const unsigned short* mod,
                   int mod_slots, int gate_slot, int rows_per_group, hipStream_t s, int tj, void* workspace, const float* q8_sa,
                   const float* q8_sw, const float* q8_qdiv, int q8_via_bf16, int stream_k, int q8_int8, int force_ks,
                   unsigned short* y2, int ldy2, int split_col);
bool gemm_pp_split(int N, int K);
size_t gemm_pp_workspace_bytes(int M, int N, int K);
}

// Ping-pong tile for an FP8 launch (tokens = 64 tj), 0 = none: the model of pick_pp (ifx_gemm.hip) — a K-step moves the same bytes and
// occupies the matrix pipe for the same cycles as a bf16 one, there are half as many of them.
static int pick_pp_q8(int M, int N, int K, int mode) {
  // from 1024 rows (one MAGI chunk of a cp = 8 rank, 1519 rows: q 50 -> 33 us, proj 86 -> 55, fc2 148 -> 90, tools/bench_q8.py)
  if (M < 1024 || N % 64 != 0 || K % 128 != 0) return 0;
  static const float step_us[5] = {0.f, 0.f, 1.2f, 1.4f, 1.6f};
  const float tile_us = (mode == IFX_EPI_GELU_TANH ? 8.f : 3.f);
  int best = 0;
  float best_t = 1e30f;
  for (int tj = 4; tj >= 2; --tj) {
    const int tiles = ((M + 64 * tj - 1) / (64 * tj)) * ((N + 255) / 256);
    const int rounds = (tiles + 255) / 256;
    const float t = rounds * ((K / 128) * step_us[tj] + tile_us);
    if (t < best_t) best_t = t, best = tj;
  }
  return best;
}

// Shapes that run the 256-token tile with K split between two workgroups when the caller gives a workspace (ifx_gemm_q8_ws): the rule
// of the bf16 tiles on the K-STEP count (gemm_pp_split: N <= 2048, >= 64 steps, an even number of them) — a function of N and K only,
// taken at ANY row count so that a row's summation order does not depend on it (unless the caller opted into row-count dependent
// choices for shard-sized launches: gemm_small_split).
static bool q8_split_shape(int M, int N, int K) {
  return N % 64 == 0 && K % 128 == 0 && gemm_pp_split(N, K / 2) && gemm_pp_workspace_bytes(M, N, K / 2) > 0 && !(gemm_small_split() && M < 2048);
}

static int gemm_q8_impl(const void* xq, int32_t ldx, const float* x_scale, const void* wq, const float* w_scale,
                        const ifx_bf16* bias, ifx_bf16* y, int32_t ldy, int32_t M, int32_t N, int32_t K,
                        int32_t format, const ifx_epilogue* epi, void* stream, const float* qdiv, int q_via_bf16,
                        void* workspace = nullptr, int64_t workspace_bytes = 0) {
  IFX_REQUIRE(xq && wq && x_scale && w_scale && y && M >= 0 && N > 0 && K > 0, "ifx_gemm_q8: null/empty operand");
  IFX_REQUIRE(K % 128 == 0, "ifx_gemm_q8: K (%d) must be a multiple of 128", K);
  IFX_REQUIRE(N % 4 == 0 && ldx % 16 == 0 && ldy % 4 == 0, "ifx_gemm_q8: N %% 4, ldx %% 16, ldy %% 4 required");
  IFX_REQUIRE(format == IFX_Q_FP8_E4M3 || format == IFX_Q_INT8, "ifx_gemm_q8: unknown format %d", format);
  int mode = epi ? epi->epilogue : IFX_EPI_BIAS;
  EpiArgsQ ea{x_scale, w_scale, bias, nullptr, 0, nullptr, 1, 0, 1};
  if (qdiv != nullptr) {
    IFX_REQUIRE(mode == IFX_EPI_GELU_TANH || mode == IFX_EPI_GELU_ERF, "ifx_gemm_q8_quant_out: epilogue %d (a GELU epilogue feeds the "
                "quantised output)", mode);
    IFX_REQUIRE(ldy % 8 == 0, "ifx_gemm_q8_quant_out: ldyq (%d) must be a multiple of 8 bytes", ldy);
    ea.qdiv = qdiv;
    ea.q_via_bf16 = q_via_bf16;
  }
  if (mode == IFX_EPI_GELU_ERF) {       // the GELU instantiation with the exact-erf activation selected at run time
    mode = IFX_EPI_GELU_TANH;
    ea.gate_slot = 1;
  }
  if (mode == IFX_EPI_RESIDUAL || mode == IFX_EPI_GATE_RES) {
    IFX_REQUIRE(epi->residual && epi->ld_res % 4 == 0, "ifx_gemm_q8: residual epilogue needs residual/ld_res");
    ea.residual = epi->residual;
    ea.ld_res = epi->ld_res;
  }
  if (mode == IFX_EPI_GATE_RES) {
    IFX_REQUIRE(epi->mod && epi->rows_per_group > 0 && epi->gate_slot >= 0 && epi->gate_slot < epi->mod_slots,
                "ifx_gemm_q8: gate epilogue needs mod/mod_slots/gate_slot/rows_per_group");
    ea.mod = epi->mod;
    ea.mod_slots = epi->mod_slots;
    ea.gate_slot = epi->gate_slot;
    ea.rows_per_group = epi->rows_per_group;
  }
  IFX_REQUIRE(epi == nullptr || epi->y2 == nullptr, "ifx_gemm_q8: the second destination (ifx_epilogue.y2) is built for the bf16 launches only");
  if (M == 0) return IFX_OK;
  hipStream_t s = (hipStream_t)stream;
  const unsigned char* xp = (const unsigned char*)xq;
  const unsigned char* wp = (const unsigned char*)wq;
  // large shapes: LDS-DMA tiles (256x256 with >= 2 rounds of tiles, else 256x128 when it fills the chip);
  // variant override through ifx_set_option("gemm_variant"): 1 = always the register-staged 128x128 kernel
  const bool wide_ok = N % 8 == 0 && ldy % 8 == 0 && (ea.residual == nullptr || ea.ld_res % 8 == 0) && K % 64 == 0;
  // FP8 / INT8 launches of >= 1024 rows: the ping-pong tile (gemm_variant 22 / 23 / 24 force its 256 / 192 / 128-token form, 3 = never)
  if (wide_ok && gemm_variant() != 1 && gemm_variant() != 2 && gemm_variant() != 3) {
    const int v = gemm_variant();
    const bool split_ok = qdiv == nullptr && v != 25 && workspace != nullptr && q8_split_shape(M, N, K) &&
                          workspace_bytes >= (int64_t)gemm_pp_workspace_bytes(M, N, K / 2) &&
                          (mode != IFX_EPI_GATE_RES || ea.rows_per_group >= 128);
    int tj = (v == 22 || v == 25 || (v == 0 && split_ok)) ? 4 : v == 23 ? 3 : v == 24 ? 2 : pick_pp_q8(M, N, K, mode);
    if (mode == IFX_EPI_GATE_RES && ea.rows_per_group < 32 * tj) tj = ea.rows_per_group >= 64 ? 2 : 0;
    const bool aligned = ((uintptr_t)bias & 7) == 0 && ((uintptr_t)ea.residual & 15) == 0 && ((uintptr_t)ea.mod & 15) == 0 &&
                         ((uintptr_t)w_scale & 15) == 0 && ((uintptr_t)qdiv & 15) == 0 && ldx % 16 == 0 && N % 64 == 0 && K % 128 == 0;
    if (tj != 0 && aligned)
      return launch_gemm_pp((const unsigned short*)xp, ldx, (const unsigned short*)wp, y, ldy, M, N, K,
                            mode, ea.bias, ea.residual, ea.ld_res, ea.mod, ea.mod_slots, ea.gate_slot, ea.rows_per_group, s, tj,
                            split_ok && tj == 4 ? workspace : nullptr, x_scale, w_scale, qdiv, q_via_bf16, 0, format == IFX_Q_INT8 ? 1 : 0, 0, nullptr, 0, 0);
  }
  if (wide_ok && gemm_variant() != 1) {
    auto wgs = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    const int t = wgs(256, 256) >= 512 ? 2 : (wgs(256, 128) >= 224 ? 1 : 0);
    // 128-byte operand rows (two / three stages) unless gemm_variant 2 asks for the 64-byte four-stage ring they replaced
    const bool rows128 = gemm_variant() != 2;
    // (a 128x128 eight-wave tile at two workgroups per CU for the 1536- and 4608-wide outputs measured the same as 256x128)
    if (t == 2 && rows128)
      return format == IFX_Q_FP8_E4M3 ? launch_q8_dma<true, 256, 256, 2, 128, 2>(xp, ldx, wp, y, ldy, M, N, K, mode, ea, s)
                                      : launch_q8_dma<false, 256, 256, 2, 128, 2>(xp, ldx, wp, y, ldy, M, N, K, mode, ea, s);
    if (t == 1 && rows128)
      return format == IFX_Q_FP8_E4M3 ? launch_q8_dma<true, 256, 128, 4, 128, 3>(xp, ldx, wp, y, ldy, M, N, K, mode, ea, s)
                                      : launch_q8_dma<false, 256, 128, 4, 128, 3>(xp, ldx, wp, y, ldy, M, N, K, mode, ea, s);
    if (t == 2)
      return format == IFX_Q_FP8_E4M3 ? launch_q8_dma<true, 256, 256, 2>(xp, ldx, wp, y, ldy, M, N, K, mode, ea, s)
                                      : launch_q8_dma<false, 256, 256, 2>(xp, ldx, wp, y, ldy, M, N, K, mode, ea, s);
    if (t == 1)
      return format == IFX_Q_FP8_E4M3 ? launch_q8_dma<true, 256, 128, 4>(xp, ldx, wp, y, ldy, M, N, K, mode, ea, s)
                                      : launch_q8_dma<false, 256, 128, 4>(xp, ldx, wp, y, ldy, M, N, K, mode, ea, s);
  }
  const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128;
  const dim3 grid(((tiles_m * tiles_n + 7) / 8) * 8), block(256);
  const size_t lds = 65536;
#define IFX_LAUNCH_Q8(F, E)                                                                                      \
  do {                                                                                                           \
    static bool attr_set = false;                                                                                \
    if (!attr_set) {                                                                                             \
      (void)hipFuncSetAttribute((const void*)gemm_q8_kernel<F, E>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                (int)lds);                                                                       \
      attr_set = true;                                                                                           \
    }                                                                                                            \
    hipLaunchKernelGGL((gemm_q8_kernel<F, E>), grid, block, lds, s, xp, ldx, wp, y, ldy, M, N, K, tiles_m, tiles_n, \
                       ea);                                                                                      \
  } while (0)
#define IFX_SWITCH_Q8(F)                                                          \
  switch (mode) {                                                                 \
    case IFX_EPI_BIAS: IFX_LAUNCH_Q8(F, IFX_EPI_BIAS); break;                     \
    case IFX_EPI_GELU_TANH: IFX_LAUNCH_Q8(F, IFX_EPI_GELU_TANH); break;           \
    case IFX_EPI_RESIDUAL: IFX_LAUNCH_Q8(F, IFX_EPI_RESIDUAL); break;             \
    case IFX_EPI_GATE_RES: IFX_LAUNCH_Q8(F, IFX_EPI_GATE_RES); break;             \
    default: set_error("ifx_gemm_q8: unknown epilogue %d", mode); return IFX_EINVAL; \
  }
  if (format == IFX_Q_FP8_E4M3) {
    IFX_SWITCH_Q8(true)
  } else {
    IFX_SWITCH_Q8(false)
  }
#undef IFX_SWITCH_Q8
#undef IFX_LAUNCH_Q8
  return check_launch("ifx_gemm_q8");
}

extern "C" int ifx_gemm_q8(const void* xq, int32_t ldx, const float* x_scale, const void* wq, const float* w_scale,
                           const ifx_bf16* bias, ifx_bf16* y, int32_t ldy, int32_t M, int32_t N, int32_t K,
                           int32_t format, const ifx_epilogue* epi, void* stream) {
  return gemm_q8_impl(xq, ldx, x_scale, wq, w_scale, bias, y, ldy, M, N, K, format, epi, stream, nullptr, 0);
}

extern "C" int64_t ifx_gemm_q8_workspace_bytes(int32_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0 || (gemm_variant() != 0 && gemm_variant() != 22)) return 0;
  return q8_split_shape(M, N, K) ? (int64_t)gemm_pp_workspace_bytes(M, N, K / 2) : 0;
}

extern "C" int ifx_gemm_q8_ws(const void* xq, int32_t ldx, const float* x_scale, const void* wq, const float* w_scale,
                              const ifx_bf16* bias, ifx_bf16* y, int32_t ldy, int32_t M, int32_t N, int32_t K, int32_t format,
                              const ifx_epilogue* epi, void* workspace, int64_t workspace_bytes, void* stream) {
  return gemm_q8_impl(xq, ldx, x_scale, wq, w_scale, bias, y, ldy, M, N, K, format, epi, stream, nullptr, 0, workspace, workspace_bytes);
}

extern "C" int ifx_gemm_q8_quant_out(const void* xq, int32_t ldx, const float* x_scale, const void* wq, const float* w_scale,
                                     const ifx_bf16* bias, void* yq, int32_t ldyq, int32_t M, int32_t N, int32_t K, int32_t format,
                                     const ifx_epilogue* epi, const float* out_divisor, int32_t via_bf16, void* stream) {
  IFX_REQUIRE(out_divisor && epi && N % 8 == 0, "ifx_gemm_q8_quant_out: out_divisor / epilogue missing or N %% 8 != 0");
  return gemm_q8_impl(xq, ldx, x_scale, wq, w_scale, bias, (ifx_bf16*)yq, ldyq, M, N, K, format, epi, stream, out_divisor, via_bf16);
}
