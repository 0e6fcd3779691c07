// Row kernels of the MAGI transformer layer (BASELINE config 5; gfx950).  All HBM-bound: one pass over the row, 16-byte accesses,
// statistics by wavefront shuffles, every elementwise neighbour fused in.
//
//   magi_head_prep_kernel   : what FullyParallelAttention.get_q / get_k / get_v / get_xqkv do between the projections and the
//                             attention calls (inferix/models/magi/dit/dit_module.py:902-970): per-HEAD LayerNorm over 128 channels
//                             (fp32 module + rotary for q / k, bf16 module for the cross-attention query / key), non-interleaved
//                             rotary embedding, and the scatter of k / v rows to where attention reads them (the in-place cache,
//                             or the staging buffer of the Ulysses all-to-all).  One 16-lane group per head, four heads per wave.
//   magi_gate_norm_kernel   : bias_modulate_add (dit_module.py:295-313) = the Triton range_mod kernel (:204-292) + post-norm +
//                             residual: y = bf16( LN_fp32( float(x) * float(gate[map[row]]) ) + float(residual) ).  The gate cannot
//                             ride in the producing GEMM's epilogue here: the LayerNorm that follows it needs the statistics of
//                             the whole 3072-wide row, a GEMM tile sees 128-256 columns of it.
//   act_rows_kernel         : SiLU / softcap(tanh) of AdaModulateLayer + gating_and_mlp (:196-198,:363-364,:1300-1303).
#include <type_traits>

#include "ifx_common.h"

namespace ifx {

// head types of magi_head_prep_kernel
enum { HT_Q = 0, HT_QX = 1, HT_K = 2, HT_V = 3, HT_KX = 4 };

struct HeadPrepArgs {
  const unsigned short* in;      // [rows, ld_in]
  int ld_in, rows, n_heads;      // n_heads 128-wide head slots per row
  int layout;                    // 0: [HQ q | HQ qx | HK k | HK v]      1: HK x (kx | v) interleaved (linear_kv_xattn output)
  int hq, hk;
  const float* rope;             // [rows, 2 * rope_half] fp32 = (sin | cos) per token (rotary_pos_emb, dit_module.py:1097)
  int rope_half;                 // rotary pairs per head: channel e < rope_half pairs with e + rope_half, channels >= 2 rope_half pass
  const float* qn_w;             // fp32 [128] q_layernorm / k_layernorm (high-precision modules, dit_model.py:620-637)
  const float* qn_b;
  const float* kn_w;
  const float* kn_b;
  const unsigned short* xn_w;    // bf16 [128] q_layernorm_xattn (layout 0) or k_layernorm_xattn (layout 1)
  const unsigned short* xn_b;
  float eps;
  int one_p;                     // apply_layernorm_1p: weight + 1 (in the parameter's dtype)
  unsigned short* q_out;         // [rows, ld_q]   head h at column h * 128
  int ld_q;
  unsigned short* qx_out;        // [rows, ld_qx]
  int ld_qx;
  unsigned short* k_out;         // k / v rows: row r goes to dest(r) * ld_kv + head * kv_head_stride
  unsigned short* v_out;
  int ld_kv, kv_head_stride;
  int row0, split, row1;         // dest(r) = r < split ? row0 + r : row1 + (r - split)
  float q_scale;                 // self-attention q only: applied in fp32 before the rounding to bf16 (1 = the reference's q)
  int q_group;                   // 0: head h of q at column h * 128 of row r; > 0: at (h / q_group) * q_group_stride + r * ld_q + (h % q_group) * 128
  long long q_group_stride;      //    (the head -> rank all-to-all's send order)
};

__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

__global__ __launch_bounds__(256) void magi_head_prep_kernel(HeadPrepArgs A) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int chunks = (A.n_heads + 3) >> 2;                 // four heads (512 channels) per wave
  const int r = wave / chunks, chunk = wave - r * chunks;
  if (r >= A.rows) return;
  const int head = chunk * 4 + (lane >> 4);
  if (head >= A.n_heads) return;
  const int e0 = (lane & 15) * 8;                          // this lane's 8 channels inside the head
  int type, hidx;
  if (A.layout == 0) {
    if (head < A.hq) type = HT_Q, hidx = head;
    else if (head < 2 * A.hq) type = HT_QX, hidx = head - A.hq;
    else if (head < 2 * A.hq + A.hk) type = HT_K, hidx = head - 2 * A.hq;
    else type = HT_V, hidx = head - 2 * A.hq - A.hk;
  } else {
    type = (head & 1) ? HT_V : HT_KX;
    hidx = head >> 1;
  }
  const u16x8 u = *reinterpret_cast<const u16x8*>(A.in + (size_t)r * A.ld_in + head * 128 + e0);
  const int dest = r < A.split ? A.row0 + r : A.row1 + (r - A.split);
  if (type == HT_V) {                                      // get_v: a copy into the cache / staging layout
    *reinterpret_cast<u16x8*>(A.v_out + (size_t)dest * A.ld_kv + hidx * A.kv_head_stride + e0) = u;
    return;
  }
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = bf2f(u[i]);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  const float mean = group16_sum(s) * (1.0f / 128.0f);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float d = v[i] - mean;
    ss += d * d;
  }
  const float rstd = 1.0f / sqrtf(group16_sum(ss) * (1.0f / 128.0f) + A.eps);
  u16x8 o;
  if (type == HT_QX || type == HT_KX) {
    // FusedLayerNorm as a bf16 module on a bf16 tensor: weight + 1 evaluated in bf16, fp32 math inside, one rounding
    const u16x8 w = *reinterpret_cast<const u16x8*>(A.xn_w + e0);
    const u16x8 b = *reinterpret_cast<const u16x8*>(A.xn_b + e0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float wi = A.one_p ? rbf(bf2f(w[i]) + 1.0f) : bf2f(w[i]);
      o[i] = f2bf((v[i] - mean) * rstd * wi + bf2f(b[i]));
    }
    if (type == HT_QX) *reinterpret_cast<u16x8*>(A.qx_out + (size_t)r * A.ld_qx + hidx * 128 + e0) = o;
    else *reinterpret_cast<u16x8*>(A.k_out + (size_t)dest * A.ld_kv + hidx * A.kv_head_stride + e0) = o;
    return;
  }
  // q / k: fp32 LayerNorm (weights fp32), then the non-interleaved rotary in fp32, one rounding to bf16
  const float* wp = type == HT_Q ? A.qn_w : A.kn_w;
  const float* bp = type == HT_Q ? A.qn_b : A.kn_b;
  const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp + e0), w1 = *reinterpret_cast<const f32x4*>(wp + e0 + 4);
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp + e0), b1 = *reinterpret_cast<const f32x4*>(bp + e0 + 4);
  const float one = A.one_p ? 1.0f : 0.0f;
  float y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float wi = (i < 4 ? w0[i] : w1[i - 4]) + one;
    const float bi = i < 4 ? b0[i] : b1[i - 4];
    y[i] = (v[i] - mean) * rstd * wi + bi;
  }
  // rotary over the first 2 * rope_half channels of the head (flash-attn's non-interleaved form: x1 = [0, half), x2 = [half, 2 half);
  // MAGI's table has 3 axes x head_dim / 8 bands = 96 of 128 channels, dit_module.py:673-720): channel e pairs with e +- half = the
  // lane half / 8 further inside this 16-lane head group; the channels behind the rotary width pass through
  const int half = A.rope_half;
  const bool rot = e0 < 2 * half, lo = e0 < half;
  const int j0 = rot ? (lo ? e0 : e0 - half) : 0;          // index into sin / cos
  const float* rp = A.rope + (size_t)r * (2 * half);
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(rp + j0), s1 = *reinterpret_cast<const f32x4*>(rp + j0 + 4);
  const f32x4 c0 = *reinterpret_cast<const f32x4*>(rp + half + j0), c1 = *reinterpret_cast<const f32x4*>(rp + half + j0 + 4);
  const int partner = rot ? lane + (lo ? (half >> 3) : -(half >> 3)) : lane;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float p = __shfl(y[i], partner, 64);
    const float sn = i < 4 ? s0[i] : s1[i - 4], cs = i < 4 ? c0[i] : c1[i - 4];
    // out1 = x1 * cos - x2 * sin ; out2 = x1 * sin + x2 * cos   (products rounded separately, as the elementwise torch ops do)
    const float out = !rot ? y[i] : (lo ? __fmul_rn(y[i], cs) - __fmul_rn(p, sn) : __fmul_rn(p, sn) + __fmul_rn(y[i], cs));
    o[i] = f2bf(type == HT_Q ? out * A.q_scale : out);
  }
  if (type == HT_Q) {
    const size_t qcol = A.q_group > 0 ? (size_t)(hidx / A.q_group) * (size_t)A.q_group_stride + (size_t)(hidx % A.q_group) * 128
                                      : (size_t)hidx * 128;
    *reinterpret_cast<u16x8*>(A.q_out + (size_t)r * A.ld_q + qcol + e0) = o;
  } else {
    *reinterpret_cast<u16x8*>(A.k_out + (size_t)dest * A.ld_kv + hidx * A.kv_head_stride + e0) = o;
  }
}

// ---------------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void magi_gate_norm_kernel(const unsigned short* __restrict__ x, int ldx,
                                                             const unsigned short* __restrict__ residual, int ldr,
                                                             const int32_t* __restrict__ map,
                                                             const unsigned short* __restrict__ gate, int ld_gate,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             int one_p, unsigned short* __restrict__ y, int ldy, int rows,
                                                             int dim, float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const unsigned short* xr = x + (size_t)r * ldx;
  const unsigned short* rr = residual + (size_t)r * ldr;
  const unsigned short* gr = gate + (size_t)map[r] * ld_gate;
  // every request of the row up front and branch-free (x, gate, residual and the fp32 norm weights, which do not depend on the
  // statistics): with the loads inside `if (col < dim)` hipcc waited for each 512-channel chunk before requesting the next — six
  // memory latencies per 3072-wide row (round 4; same rewrite as ifx_norm.hip::load_chunks)
  float v[NCH][8];
  u16x8 xu[NCH], gu[NCH], res[NCH];
  f32x4 w0[NCH], w1[NCH], b0[NCH], b1[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 512 + lane * 8;
    const int cc = col < dim ? col : 0;
    xu[c] = *reinterpret_cast<const u16x8*>(xr + cc);
    gu[c] = *reinterpret_cast<const u16x8*>(gr + cc);
    res[c] = *reinterpret_cast<const u16x8*>(rr + cc);
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 512 + lane * 8;
    const int cc = col < dim ? col : 0;
    w0[c] = *reinterpret_cast<const f32x4*>(w + cc), w1[c] = *reinterpret_cast<const f32x4*>(w + cc + 4);
    b0[c] = *reinterpret_cast<const f32x4*>(b + cc), b1[c] = *reinterpret_cast<const f32x4*>(b + cc + 4);
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const bool ok = c * 512 + lane * 8 < dim;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[c][i] = ok ? bf2f(xu[c][i]) * bf2f(gu[c][i]) : 0.f;       // range_mod in fp32
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[c][i];
  const float inv_n = 1.0f / (float)dim;
  const float mean = wave_sum(s) * inv_n;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (c * 512 + lane * 8 < dim) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = v[c][i] - mean;
        ss += d * d;
      }
    }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) * inv_n + eps);
  const float one = one_p ? 1.0f : 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 512 + lane * 8;
    u16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float wi = (i < 4 ? w0[c][i] : w1[c][i - 4]) + one;
      const float bi = i < 4 ? b0[c][i] : b1[c][i - 4];
      const float n = (v[c][i] - mean) * rstd * wi + bi;                   // post_norm in fp32
      o[i] = f2bf(n + bf2f(res[c][i]));                                    // + residual.float(), one rounding
    }
    if (col < dim) *reinterpret_cast<u16x8*>(y + (size_t)r * ldy + col) = o;
  }
}

// ---------------------------------------------------------------------------------------------------------
// mode 0: SiLU as torch evaluates it on a bf16 tensor (fp32 x / (1 + exp(-x)), one rounding)
// mode 1: softcap with cap 1: bf16(tanh(float(x)))
__global__ __launch_bounds__(256) void act_rows_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                       long n, int mode) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = bf2f(x[i]);
  y[i] = f2bf(mode == 0 ? v / (1.0f + expf(-v)) : tanhf(v));
}

// ---------------------------------------------------------------------------------------------------------
// Static-scale / per-tensor quantisers (see include/inferix_hip.h, ifx_quant_static / ifx_quant_per_tensor)
__global__ __launch_bounds__(256) void amax_kernel(const unsigned short* __restrict__ x, int ldx, int rows, int K,
                                                   unsigned* __restrict__ amax_bits) {
  float m = 0.f;
  const int chunks_per_row = K / 8;
  const long total = (long)rows * chunks_per_row;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / chunks_per_row), c = (int)(i - (long)r * chunks_per_row);
    const u16x8 u = *reinterpret_cast<const u16x8*>(x + (size_t)r * ldx + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(bf2f(u[e])));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax_bits, __builtin_bit_cast(unsigned, m));   // non-negative floats order as uints
}

// SCALE: 0 = divisor vector [K] (per input channel), 1 = one divisor at scale[0], 2 = amax_bits[0] / QMAX (1 when amax == 0)
template <bool FP8>
__global__ __launch_bounds__(256) void quant_static_kernel(const unsigned short* __restrict__ x, int ldx,
                                                           unsigned char* __restrict__ q, int ldq,
                                                           const float* __restrict__ scale, int scale_mode,
                                                           const unsigned* __restrict__ amax_bits, float* __restrict__ row_scale,
                                                           int rows, int K, int via_bf16) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  constexpr float QMAX = FP8 ? 448.0f : 127.0f;
  float s1 = 1.0f;
  if (scale_mode == 1) s1 = scale[0];
  if (scale_mode == 2) {
    const float amax = __builtin_bit_cast(float, amax_bits[0]);
    s1 = amax > 0.f ? amax / QMAX : 1.0f;
  }
  if (row_scale != nullptr && lane == 0) row_scale[r] = s1;
  // four 512-column chunks per pass, every load of the pass (16 B of x, 32 B of divisors per lane and chunk) issued before the first
  // use: one wave per row has nothing else to hide the latency with (56 -> ~20 us on 6075 x 3072)
  constexpr int UN = 4;
  for (int col0 = lane * 8; col0 < K; col0 += 512 * UN) {
    u16x8 u[UN];
    f32x4 d0[UN], d1[UN];
#pragma unroll
    for (int c = 0; c < UN; ++c) {
      const int col = col0 + 512 * c;
      if (col < K) {
        u[c] = *reinterpret_cast<const u16x8*>(x + (size_t)r * ldx + col);
        if (scale_mode == 0) {
          d0[c] = *reinterpret_cast<const f32x4*>(scale + col);
          d1[c] = *reinterpret_cast<const f32x4*>(scale + col + 4);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < UN; ++c) {
      const int col = col0 + 512 * c;
      if (col >= K) continue;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = scale_mode == 0 ? (i < 4 ? d0[c][i] : d1[c][i - 4]) : s1;
        float t = fminf(fmaxf(bf2f(u[c][i]) / d, -QMAX), QMAX);
        if (via_bf16) t = rbf(t);                      // div_clamp_to rounds to bf16 before the e4m3 cast (dit_module.py:379-384)
        v[i] = t;
      }
      u32x2 pk;
      if (FP8) {
        unsigned w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], w1, true);
        pk = u32x2{w0, w1};
      } else {
        unsigned w[2] = {0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i >> 2] |= ((unsigned)(int)rintf(v[i]) & 0xffu) << (8 * (i & 3));
        pk = u32x2{w[0], w[1]};
      }
      *reinterpret_cast<u32x2*>(q + (size_t)r * ldq + col) = pk;
    }
  }
}

template <typename F>
static int dispatch_nch_magi(int dim, F&& f) {
  const int nch = (dim + 511) / 512;
  if (nch <= 1) return f(std::integral_constant<int, 1>{});
  if (nch <= 2) return f(std::integral_constant<int, 2>{});
  if (nch <= 4) return f(std::integral_constant<int, 4>{});
  if (nch <= 6) return f(std::integral_constant<int, 6>{});
  if (nch <= 8) return f(std::integral_constant<int, 8>{});
  if (nch <= 12) return f(std::integral_constant<int, 12>{});
  set_error("ifx_magi_gate_norm_residual: dim <= 6144 supported (got %d)", dim);
  return IFX_EUNSUP;
}

}  // namespace ifx

using namespace ifx;

extern "C" int ifx_magi_head_prep(const ifx_magi_head_prep_desc* d, void* stream) {
  IFX_REQUIRE(d && d->in && d->rows >= 0 && d->ld_in % 8 == 0, "ifx_magi_head_prep: bad input");
  IFX_REQUIRE(d->head_dim == 128, "ifx_magi_head_prep: head_dim %d not built (128 only)", d->head_dim);
  IFX_REQUIRE(d->layout == 0 || d->layout == 1, "ifx_magi_head_prep: layout %d", d->layout);
  HeadPrepArgs a{};
  a.in = d->in;
  a.ld_in = d->ld_in;
  a.rows = d->rows;
  a.layout = d->layout;
  a.hq = d->q_heads;
  a.hk = d->kv_heads;
  IFX_REQUIRE(a.hk > 0, "ifx_magi_head_prep: kv_heads must be > 0");
  if (d->layout == 0) {
    IFX_REQUIRE(a.hq > 0 && d->rope && d->qn_w && d->qn_b && d->kn_w && d->kn_b && d->xn_w && d->xn_b && d->q_out && d->qx_out,
                "ifx_magi_head_prep: layout 0 needs q/qx outputs, rope and the three norms");
    IFX_REQUIRE(d->q_group >= 0 && (d->q_group == 0 || (a.hq % d->q_group == 0 && d->q_group_stride >= (int64_t)d->rows * d->ld_q && d->q_group_stride % 8 == 0)),
                "ifx_magi_head_prep: q_group %d / q_group_stride %lld", d->q_group, (long long)d->q_group_stride);
    IFX_REQUIRE(d->ld_q >= (d->q_group > 0 ? d->q_group : a.hq) * 128 && d->ld_qx >= a.hq * 128 && d->ld_q % 8 == 0 && d->ld_qx % 8 == 0,
                "ifx_magi_head_prep: q row strides too small");
    a.n_heads = 2 * a.hq + 2 * a.hk;
  } else {
    IFX_REQUIRE(d->xn_w && d->xn_b, "ifx_magi_head_prep: layout 1 needs k_layernorm_xattn");
    a.n_heads = 2 * a.hk;
  }
  IFX_REQUIRE(d->ld_in >= a.n_heads * 128, "ifx_magi_head_prep: ld_in %d < %d heads x 128", d->ld_in, a.n_heads);
  IFX_REQUIRE(d->k_out && d->v_out && d->ld_kv % 8 == 0 && d->kv_head_stride % 8 == 0 && d->kv_head_stride >= 128,
              "ifx_magi_head_prep: k/v destination");
  IFX_REQUIRE(d->split >= 0 && d->row0 >= 0 && d->row1 >= 0, "ifx_magi_head_prep: negative destination rows");
  a.rope = d->rope;
  a.rope_half = d->rope_half > 0 ? d->rope_half : 64;
  IFX_REQUIRE(a.rope_half % 8 == 0 && a.rope_half <= 64, "ifx_magi_head_prep: rope_half (%d) must be a multiple of 8, at most 64", a.rope_half);
  a.qn_w = d->qn_w, a.qn_b = d->qn_b, a.kn_w = d->kn_w, a.kn_b = d->kn_b;
  a.xn_w = d->xn_w, a.xn_b = d->xn_b;
  a.eps = d->eps;
  a.one_p = d->layernorm_1p ? 1 : 0;
  a.q_out = d->q_out, a.ld_q = d->ld_q, a.qx_out = d->qx_out, a.ld_qx = d->ld_qx;
  IFX_REQUIRE(d->q_scale >= 0.f && d->q_scale == d->q_scale, "ifx_magi_head_prep: q_scale must be >= 0 (0 = 1)");
  a.q_scale = d->q_scale > 0.f ? d->q_scale : 1.0f;
  a.q_group = d->q_group, a.q_group_stride = d->q_group_stride;
  a.k_out = d->k_out, a.v_out = d->v_out, a.ld_kv = d->ld_kv, a.kv_head_stride = d->kv_head_stride;
  a.row0 = d->row0, a.split = d->split, a.row1 = d->row1;
  if (d->rows == 0) return IFX_OK;
  const long waves = (long)d->rows * ((a.n_heads + 3) / 4);
  hipLaunchKernelGGL(magi_head_prep_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("ifx_magi_head_prep");
}

extern "C" int ifx_magi_gate_norm_residual(const ifx_bf16* x, int32_t ldx, const ifx_bf16* residual, int32_t ld_res,
                                           const int32_t* condition_map, const ifx_bf16* gate, int32_t ld_gate,
                                           const float* norm_w, const float* norm_b, int32_t layernorm_1p, ifx_bf16* y,
                                           int32_t ldy, int32_t rows, int32_t dim, float eps, void* stream) {
  IFX_REQUIRE(x && residual && condition_map && gate && norm_w && norm_b && y && rows >= 0 && dim > 0 && dim % 8 == 0,
              "ifx_magi_gate_norm_residual: bad arguments (dim %d)", dim);
  IFX_REQUIRE(ldx % 8 == 0 && ld_res % 8 == 0 && ld_gate % 8 == 0 && ldy % 8 == 0 && ldx >= dim && ld_res >= dim && ldy >= dim,
              "ifx_magi_gate_norm_residual: row strides must be >= dim and multiples of 8");
  if (rows == 0) return IFX_OK;
  return dispatch_nch_magi(dim, [&](auto nch) {
    hipLaunchKernelGGL((magi_gate_norm_kernel<decltype(nch)::value>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, residual, ld_res, condition_map, gate, ld_gate, norm_w, norm_b, layernorm_1p ? 1 : 0, y, ldy,
                       rows, dim, eps);
    return check_launch("ifx_magi_gate_norm_residual");
  });
}

// K | V rows of an all-to-all message -> the cache planes, with the store rule's two destination runs (MagiKVCacheManager): row r of
// `kv` [n, heads, 2 * 128] goes to cache row dest(r) = r < split ? row0 + r : row1 + (r - split); K = the first 128 channels of a head,
// V the second.  One launch for what four strided torch copies did (k / v x stored run / scratch run).
namespace ifx {
__global__ __launch_bounds__(256) void kv_split_rows_kernel(const unsigned short* __restrict__ kv, unsigned short* __restrict__ kc,
                                                            unsigned short* __restrict__ vc, long n_chunks, int heads, int row0,
                                                            int split, int row1) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;          // one 16-byte chunk of K and the matching one of V per thread
  if (c >= n_chunks) return;
  const int ch = (int)(c & 15);                                 // chunk inside the 128-channel head
  const long rh = c >> 4;                                       // (row, head)
  const int h = (int)(rh % heads);
  const long r = rh / heads;
  const long dest = r < split ? (long)row0 + r : (long)row1 + (r - split);
  const u16x8 k = *reinterpret_cast<const u16x8*>(kv + (rh * 2) * 128 + ch * 8);
  const u16x8 v = *reinterpret_cast<const u16x8*>(kv + (rh * 2 + 1) * 128 + ch * 8);
  *reinterpret_cast<u16x8*>(kc + (dest * heads + h) * 128 + ch * 8) = k;
  *reinterpret_cast<u16x8*>(vc + (dest * heads + h) * 128 + ch * 8) = v;
}
}  // namespace ifx

extern "C" int ifx_kv_split_rows(const ifx_bf16* kv, ifx_bf16* k_cache, ifx_bf16* v_cache, int32_t rows, int32_t heads, int32_t row0,
                                 int32_t split, int32_t row1, void* stream) {
  IFX_REQUIRE(kv && k_cache && v_cache && rows >= 0 && heads > 0 && row0 >= 0 && row1 >= 0 && split >= 0 && split <= rows,
              "ifx_kv_split_rows: bad arguments (rows %d, heads %d, split %d)", rows, heads, split);
  IFX_REQUIRE(!((uintptr_t)kv & 15) && !((uintptr_t)k_cache & 15) && !((uintptr_t)v_cache & 15), "ifx_kv_split_rows: 16-byte aligned tensors");
  if (rows == 0) return IFX_OK;
  const long n_chunks = (long)rows * heads * 16;
  hipLaunchKernelGGL(ifx::kv_split_rows_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kv, k_cache,
                     v_cache, n_chunks, heads, row0, split, row1);
  return check_launch("ifx_kv_split_rows");
}

extern "C" int ifx_act_rows(const ifx_bf16* x, ifx_bf16* y, int64_t n, int32_t mode, void* stream) {
  IFX_REQUIRE(x && y && n >= 0 && (mode == IFX_ACT_SILU || mode == IFX_ACT_TANH), "ifx_act_rows: bad arguments (mode %d)", mode);
  if (n == 0) return IFX_OK;
  hipLaunchKernelGGL(act_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, mode);
  return check_launch("ifx_act_rows");
}

extern "C" int ifx_quant_static(const ifx_bf16* x, int32_t ldx, void* q, int32_t ldq, const float* divisor, int32_t divisor_len,
                                float* row_scale, int32_t rows, int32_t K, int32_t format, int32_t via_bf16, void* stream) {
  IFX_REQUIRE(x && q && divisor && rows >= 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldq % 8 == 0,
              "ifx_quant_static: bad arguments (K %d)", K);
  IFX_REQUIRE(divisor_len == 1 || divisor_len == K, "ifx_quant_static: divisor_len %d must be 1 or K (%d)", divisor_len, K);
  IFX_REQUIRE(format == IFX_Q_FP8_E4M3 || format == IFX_Q_INT8, "ifx_quant_static: unknown format %d", format);
  if (rows == 0) return IFX_OK;
  const dim3 grid((rows + 3) / 4), block(256);
  const int sm = divisor_len == 1 ? 1 : 0;
  if (format == IFX_Q_FP8_E4M3)
    hipLaunchKernelGGL((quant_static_kernel<true>), grid, block, 0, (hipStream_t)stream, x, ldx, (unsigned char*)q, ldq, divisor, sm,
                       (const unsigned*)nullptr, row_scale, rows, K, via_bf16 ? 1 : 0);
  else
    hipLaunchKernelGGL((quant_static_kernel<false>), grid, block, 0, (hipStream_t)stream, x, ldx, (unsigned char*)q, ldq, divisor,
                       sm, (const unsigned*)nullptr, row_scale, rows, K, via_bf16 ? 1 : 0);
  return check_launch("ifx_quant_static");
}

extern "C" int ifx_quant_per_tensor(const ifx_bf16* x, int32_t ldx, void* q, int32_t ldq, float* row_scale, void* amax_workspace,
                                    int32_t rows, int32_t K, int32_t format, void* stream) {
  IFX_REQUIRE(x && q && row_scale && amax_workspace && rows >= 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldq % 8 == 0,
              "ifx_quant_per_tensor: bad arguments (K %d)", K);
  IFX_REQUIRE(format == IFX_Q_FP8_E4M3 || format == IFX_Q_INT8, "ifx_quant_per_tensor: unknown format %d", format);
  if (rows == 0) return IFX_OK;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(amax_workspace, 0, 4, s) != hipSuccess) {
    set_error("ifx_quant_per_tensor: hipMemsetAsync failed");
    return IFX_ELAUNCH;
  }
  const long chunks = (long)rows * (K / 8);
  const int blocks = (int)((chunks + 255) / 256 < 1024 ? (chunks + 255) / 256 : 1024);
  hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(256), 0, s, x, ldx, rows, K, (unsigned*)amax_workspace);
  const dim3 grid((rows + 3) / 4), block(256);
  if (format == IFX_Q_FP8_E4M3)
    hipLaunchKernelGGL((quant_static_kernel<true>), grid, block, 0, s, x, ldx, (unsigned char*)q, ldq, (const float*)nullptr, 2,
                       (const unsigned*)amax_workspace, row_scale, rows, K, 0);
  else
    hipLaunchKernelGGL((quant_static_kernel<false>), grid, block, 0, s, x, ldx, (unsigned char*)q, ldq, (const float*)nullptr, 2,
                       (const unsigned*)amax_workspace, row_scale, rows, K, 0);
  return check_launch("ifx_quant_per_tensor");
}
