// HBM-bound row kernels of the denoising step (gfx950):
//   * LayerNorm / AdaLN-modulated LayerNorm           (ifx_layernorm)
//   * WanRMSNorm                                       (ifx_rmsnorm)
//   * fused QK-RMSNorm + 3-axis RoPE + paged KV append (ifx_rmsnorm_rope_kv_append)
//
// Design: one 64-lane wavefront owns one token row.  The row lives in registers
// (8 bf16 = 16 B per lane per 512-channel chunk, fully coalesced 1 KiB per wave
// load), statistics are reduced with wavefront shuffles (no LDS, no barriers),
// and every elementwise consumer of the normalised row is fused behind it so the
// row is read once and written once.  4 waves per workgroup, grid = rows / 4, which
// is >> 256 workgroups for the 4680-row blocks of the 480p path.
//
// Rounding points reproduce the reference's bf16 module boundaries (see
// include/inferix_hip.h); statistics are fp32, the rotation is fp64 like the
// reference's complex128 multiply (causal_model.py:33-61).
#include <type_traits>

#include "ifx_common.h"

namespace ifx {

// 16-byte chunks of a row, requested WITHOUT a per-lane branch: a lane whose columns lie beyond `dim` reads the row's first chunk
// instead (a valid address) and its values are zeroed at the conversion.  With the load inside `if (col < dim)` hipcc gave every chunk
// its own basic block — load, s_waitcnt vmcnt(0), convert — so a wave had ONE 1 KiB request in flight at a time and paid the memory
// latency once per chunk (round 4: layernorm 3.1 TB/s, rmsnorm + RoPE + append 3.6 TB/s with three-chunk rows).
template <int NCH>
__device__ __forceinline__ void load_chunks(u16x8 (&u)[NCH], const unsigned short* p, int dim, int lane) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 512 + lane * 8;
    u[c] = *reinterpret_cast<const u16x8*>(p + (col < dim ? col : 0));
  }
}

template <int NCH>
struct Row {
  float v[NCH][8];
  __device__ __forceinline__ void from(const u16x8 (&u)[NCH], int dim, int lane) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const bool ok = c * 512 + lane * 8 < dim;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[c][i] = ok ? bf2f(u[c][i]) : 0.f;
    }
  }
  __device__ __forceinline__ void load(const unsigned short* p, int dim, int lane) {
    u16x8 u[NCH];
    load_chunks<NCH>(u, p, dim, lane);
    from(u, dim, lane);
  }
  __device__ __forceinline__ float sum() const {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[c][i];
    return s;
  }
  __device__ __forceinline__ float sumsq() const {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[c][i] * v[c][i];
    return s;
  }
};

// ---------------------------------------------------------------------------
// QUANT: 0 = bf16 row out; 1 / 2 = the per-token e4m3 / int8 quantiser of the 8-bit linears (ifx_quant_per_token) applied to the
// bf16-rounded row while it is still in registers: amax -> scale -> bytes, op for op the standalone quantiser, so the 8-bit GEMM
// sees the same bytes and scales and the bf16 row never goes to HBM (7.7 us per 4680 x 1536 launch, three per layer).
template <int NCH, int QUANT = 0>
__global__ __launch_bounds__(256) void layernorm_kernel(
    const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int rows, int dim, float eps,
    int mode, const unsigned short* __restrict__ gamma, const unsigned short* __restrict__ beta,
    const unsigned short* __restrict__ mod, int mod_slots, int shift_slot, int scale_slot,
    int rows_per_group, unsigned char* __restrict__ q = nullptr, int ldq = 0, float* __restrict__ qscale = nullptr,
    int n_out = 0, int via_bf16 = 0) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  // the row and the per-channel vectors of the epilogue (which do not depend on the statistics) are requested together, branch-free:
  // a wave is one chain load -> two wave reductions -> store
  u16x8 xr[NCH], pa[NCH], pb[NCH];
  load_chunks<NCH>(xr, x + (size_t)r * dim, dim, lane);
  {
    const unsigned short* pa_p = nullptr;
    const unsigned short* pb_p = nullptr;
    if (mode == IFX_LN_MODULATE) {
      const size_t g0 = (size_t)(r / rows_per_group) * mod_slots;
      pa_p = mod + (g0 + scale_slot) * dim;
      pb_p = mod + (g0 + shift_slot) * dim;
    } else if (mode == IFX_LN_AFFINE) {
      pa_p = gamma;
      pb_p = beta;
    }
    if (pa_p != nullptr) {          // wave-uniform
      load_chunks<NCH>(pa, pa_p, dim, lane);
      load_chunks<NCH>(pb, pb_p, dim, lane);
    } else {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        pa[c] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
        pb[c] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
  }
  Row<NCH> row;
  row.from(xr, dim, lane);
  const float inv_n = 1.0f / (float)dim;
  const float mean = wave_sum(row.sum()) * inv_n;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c * 512 + lane * 8 < dim) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float d = row.v[c][i] - mean;
        ss += d * d;
      }
    }
  }
  const float var = wave_sum(ss) * inv_n;
  const float rstd = 1.0f / sqrtf(var + eps);

  u16x8 qrow[QUANT != 0 ? NCH : 1];
  if (QUANT != 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) qrow[c] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 512 + lane * 8;
    if (col >= dim) continue;
    u16x8 o;
    if (mode == IFX_LN_MODULATE) {
      const u16x8 sc = pa[c], sh = pb[c];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float t = rbf((row.v[c][i] - mean) * rstd);   // norm output is a bf16 tensor
        float s1 = rbf(1.0f + bf2f(sc[i]));           // (1 + e) evaluated in bf16
        t = rbf(t * s1);
        o[i] = f2bf(t + bf2f(sh[i]));
      }
    } else if (mode == IFX_LN_AFFINE) {
      const u16x8 g = pa[c], b = pb[c];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        o[i] = f2bf((row.v[c][i] - mean) * rstd * bf2f(g[i]) + bf2f(b[i]));
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = f2bf((row.v[c][i] - mean) * rstd);
    }
    if (QUANT == 0) *reinterpret_cast<u16x8*>(y + (size_t)r * dim + col) = o;
    else qrow[c] = o;
  }
  if (QUANT == 3) {
    // static-scale e4m3 (MAGI's PerTensorQuantizedFp8Linear inputs): output j = div_clamp_to(row, divisor_j) at byte column j * dim of
    // q; the q / qx / k / v linears quantise the same normalised row with their own input_scale vectors, so one read feeds all four
    const float* divs = qscale;
    for (int j = 0; j < n_out; ++j) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 512 + lane * 8;
        if (col >= dim) continue;
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(divs + (size_t)j * dim + col);
        const f32x4 d1 = *reinterpret_cast<const f32x4*>(divs + (size_t)j * dim + col + 4);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float t = fminf(fmaxf(bf2f(qrow[c][i]) / (i < 4 ? d0[i] : d1[i - 4]), -448.0f), 448.0f);
          if (via_bf16) t = rbf(t);
          v[i] = t;
        }
        unsigned w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], w1, true);
        *reinterpret_cast<u32x2*>(q + (size_t)r * ldq + (size_t)j * dim + col) = u32x2{w0, w1};
      }
    }
  } else if (QUANT != 0) {
    constexpr float QMAX = QUANT == 1 ? 448.0f : 127.0f;
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(bf2f(qrow[c][i])));
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / QMAX : 1.0f;
    if (lane == 0) qscale[r] = sc;
    const RowDivisor rdiv(sc);       // the exact three-operation x / sc (ifx_common.h)
    auto emit = [&](auto fastc) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 512 + lane * 8;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fminf(fmaxf(rdiv.template div<decltype(fastc)::value>(bf2f(qrow[c][i])), -QMAX), QMAX);
        u32x2 pk;
        if (QUANT == 1) {
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
        if (col < dim) *reinterpret_cast<u32x2*>(q + (size_t)r * ldq + col) = pk;
      }
    };
    if (rdiv.fast()) emit(std::true_type{});
    else emit(std::false_type{});
  }
}

// ---------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const unsigned short* __restrict__ x, int ldx,
                                                      unsigned short* __restrict__ y, int ldy,
                                                      const unsigned short* __restrict__ w, int rows,
                                                      int dim, float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  u16x8 xr[NCH], wv[NCH];
  load_chunks<NCH>(xr, x + (size_t)r * ldx, dim, lane);
  load_chunks<NCH>(wv, w, dim, lane);            // the weight does not depend on the statistics: requested with the row
  Row<NCH> row;
  row.from(xr, dim, lane);
  const float rs = 1.0f / sqrtf(wave_sum(row.sumsq()) / (float)dim + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 512 + lane * 8;
    u16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f2bf(rbf(row.v[c][i] * rs) * bf2f(wv[c][i]));
    if (col < dim) *reinterpret_cast<u16x8*>(y + (size_t)r * ldy + col) = o;
  }
}

// ---------------------------------------------------------------------------
struct RopeArgs {
  const double* freqs;
  int max_pos, start_frame, height, width, hw_offset, hw_local;
  float q_scale = 1.0f;   // applied to the rotated q in fp32 before its ONE rounding to bf16 (ifx_rope_grid.q_scale)
};

// rotate the 4 adjacent-channel pairs held in t[0..7]; pair index jp0..jp0+3 within the head
__device__ __forceinline__ void rope4(float (&t)[8], int jp0, const RopeArgs& ra, int half, int n_t,
                                      int n_h, int pos_t, int pos_h, int pos_w) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int j = jp0 + p;
    const int pos = (j < n_t) ? pos_t : ((j < n_t + n_h) ? pos_h : pos_w);
    const double2 cs = *reinterpret_cast<const double2*>(ra.freqs + ((size_t)pos * half + j) * 2);
    const double a = (double)t[2 * p], b = (double)t[2 * p + 1];
    // complex multiply exactly as (a+ib)(c+is) evaluates in complex128:
    // re = a*c - b*s ; im = a*s + b*c  (each product and sum rounded in fp64)
    const double re = __dmul_rn(a, cs.x) - __dmul_rn(b, cs.y);
    const double im = __dmul_rn(a, cs.y) + __dmul_rn(b, cs.x);
    t[2 * p] = (float)re;       // torch's double->bf16 goes through float
    t[2 * p + 1] = (float)im;
  }
}

// rotation with the four (cos, sin) pairs of this lane already in registers: a lane's 8 channels sit at the same
// offset inside their head in every 512-channel chunk (512 % head_dim == 0), and q and k use the same positions, so
// one set of table reads serves the whole token (was re-read per chunk and per q / k: 24 loads instead of 4)
__device__ __forceinline__ void rope4_cs(float (&t)[8], const double2 (&cs)[4]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const double a = (double)t[2 * p], b = (double)t[2 * p + 1];
    const double re = __dmul_rn(a, cs[p].x) - __dmul_rn(b, cs[p].y);
    const double im = __dmul_rn(a, cs[p].y) + __dmul_rn(b, cs[p].x);
    t[2 * p] = (float)re;
    t[2 * p + 1] = (float)im;
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_rope_append_kernel(
    const unsigned short* __restrict__ qkv, int ld, unsigned short* __restrict__ q_out,
    const unsigned short* __restrict__ wq, const unsigned short* __restrict__ wk, RopeArgs ra, int has_rope,
    unsigned short* __restrict__ kc, unsigned short* __restrict__ vc, KvAddr ka, int local_start, int rows,
    int dim, int head_dim, float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const unsigned short* base = qkv + (size_t)r * ld;

  int pos_t = 0, pos_h = 0, pos_w = 0;
  const int half = head_dim >> 1;
  const int n_h = half / 3, n_t = half - 2 * n_h;
  if (has_rope) {
    const int f = r / ra.hw_local;
    const int p = ra.hw_offset + (r - f * ra.hw_local);
    pos_t = ra.start_frame + f;
    pos_h = p / ra.width;
    pos_w = p - pos_h * ra.width;
  }
  size_t slot_off = 0;
  if (kc != nullptr) slot_off = (size_t)ka.slot(local_start + r) * dim;
  const bool shared_cs = has_rope && (512 % head_dim) == 0;
  double2 cs4[4];
  if (shared_cs) {
    const int jp0 = ((lane * 8) % head_dim) >> 1;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int j = jp0 + p;
      const int pos = (j < n_t) ? pos_t : ((j < n_t + n_h) ? pos_h : pos_w);
      cs4[p] = *reinterpret_cast<const double2*>(ra.freqs + ((size_t)pos * half + j) * 2);
    }
  }

  // everything the token needs is requested up front and branch-free — the q, k and raw v chunks, both norm weights and the (cos, sin)
  // pairs above: 15 + 4 loads in flight per lane.  The kernel is a chain load -> wave reduction -> store per row; with the loads
  // inside per-chunk `if (col < dim)` blocks hipcc waited for each one before issuing the next (round 4, 23.4 us per 4680-row launch).
  const bool has_kv = kc != nullptr;               // wave-uniform
  const bool copy_v = vc != nullptr;               // wave-uniform: nullptr = the projection wrote the V rows into the cache itself
  u16x8 qraw[NCH], kraw[NCH], vraw[NCH], wqv[NCH], wkv[NCH];
  load_chunks<NCH>(qraw, base, dim, lane);
  load_chunks<NCH>(wqv, wq, dim, lane);
  if (has_kv) {
    load_chunks<NCH>(kraw, base + dim, dim, lane);
    if (copy_v) load_chunks<NCH>(vraw, base + 2 * dim, dim, lane);
    load_chunks<NCH>(wkv, wk, dim, lane);
  }
  // ---- v (raw copy into the cache slot): out first, nothing depends on it ----
  if (has_kv && copy_v) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 512 + lane * 8;
      if (col < dim) *reinterpret_cast<u16x8*>(vc + slot_off + col) = vraw[c];
    }
  }
  Row<NCH> row;
  // ---- q ----
  {
    row.from(qraw, dim, lane);
    const float rs = 1.0f / sqrtf(wave_sum(row.sumsq()) / (float)dim + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 512 + lane * 8;
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = rbf(rbf(row.v[c][i] * rs) * bf2f(wqv[c][i]));
      if (shared_cs) rope4_cs(t, cs4);
      else if (has_rope && col < dim) rope4(t, (col % head_dim) >> 1, ra, half, n_t, n_h, pos_t, pos_h, pos_w);
      u16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = f2bf(t[i] * ra.q_scale);
      if (col < dim) *reinterpret_cast<u16x8*>(q_out + (size_t)r * dim + col) = o;
    }
  }
  if (!has_kv) return;
  // ---- k ----
  {
    row.from(kraw, dim, lane);
    const float rs = 1.0f / sqrtf(wave_sum(row.sumsq()) / (float)dim + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 512 + lane * 8;
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = rbf(rbf(row.v[c][i] * rs) * bf2f(wkv[c][i]));
      if (shared_cs) rope4_cs(t, cs4);
      else if (has_rope && col < dim) rope4(t, (col % head_dim) >> 1, ra, half, n_t, n_h, pos_t, pos_h, pos_w);
      u16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = f2bf(t[i]);
      if (col < dim) *reinterpret_cast<u16x8*>(kc + slot_off + col) = o;
    }
  }
}

// ---------------------------------------------------------------------------
// KV roll: physical realisation of the eviction shift (causal_model.py:287-292)
__global__ __launch_bounds__(256) void kv_copy_rows_kernel(const unsigned short* __restrict__ src,
                                                           unsigned short* __restrict__ dst, KvAddr ka,
                                                           int src_tok0, int dst_tok0, int ntok, int row_elems,
                                                           int src_is_paged, int dst_is_paged) {
  const int chunks = row_elems / 8;
  const size_t total = (size_t)ntok * chunks;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i / chunks), c = (int)(i - (size_t)t * chunks);
    const size_t s = (size_t)(src_is_paged ? ka.slot(src_tok0 + t) : (src_tok0 + t)) * row_elems + c * 8;
    const size_t d = (size_t)(dst_is_paged ? ka.slot(dst_tok0 + t) : (dst_tok0 + t)) * row_elems + c * 8;
    *reinterpret_cast<u16x8*>(dst + d) = *reinterpret_cast<const u16x8*>(src + s);
  }
}

// Sequence-parallel cache write: rank-major all-gathered K/V rows -> cache slots in the single-GPU (frame, hw) order.
//   gathered [world][2][frames*hw_local][row_elems]   (per rank: K rows then V rows of its shard)
//   row (r, f, i) -> logical token local_start + f*frame_tokens + r*hw_local + i
__global__ __launch_bounds__(256) void kv_scatter_shards_kernel(const unsigned short* __restrict__ gathered,
                                                                unsigned short* __restrict__ kc,
                                                                unsigned short* __restrict__ vc, KvAddr ka, int world,
                                                                int frames, int hw_local, int frame_tokens,
                                                                int local_start, int row_elems) {
  const int n_local = frames * hw_local;
  const int chunks = row_elems / 8;
  const size_t total = (size_t)world * n_local * chunks;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / chunks), c = (int)(i - (size_t)row * chunks);
    const int r = row / n_local, j = row - r * n_local;
    const int f = j / hw_local, w = j - f * hw_local;
    const int tok = local_start + f * frame_tokens + r * hw_local + w;
    const size_t d = (size_t)ka.slot(tok) * row_elems + c * 8;
    const unsigned short* ks = gathered + ((size_t)(2 * r) * n_local + j) * row_elems + c * 8;
    *reinterpret_cast<u16x8*>(kc + d) = *reinterpret_cast<const u16x8*>(ks);
    *reinterpret_cast<u16x8*>(vc + d) = *reinterpret_cast<const u16x8*>(ks + (size_t)n_local * row_elems);
  }
}

template <typename F>
static int dispatch_nch(int dim, F&& f) {
  const int nch = (dim + 511) / 512;
  if (nch <= 1) return f(std::integral_constant<int, 1>{});
  if (nch <= 2) return f(std::integral_constant<int, 2>{});
  if (nch <= 3) return f(std::integral_constant<int, 3>{});
  if (nch <= 4) return f(std::integral_constant<int, 4>{});
  if (nch <= 6) return f(std::integral_constant<int, 6>{});
  if (nch <= 8) return f(std::integral_constant<int, 8>{});
  if (nch <= 10) return f(std::integral_constant<int, 10>{});
  set_error("row kernels support dim <= 5120 (got %d)", dim);
  return IFX_EUNSUP;
}

}  // namespace ifx

using namespace ifx;

extern "C" int ifx_layernorm(const ifx_bf16* x, ifx_bf16* y, int32_t rows, int32_t dim, float eps, int32_t mode,
                             const ifx_bf16* gamma, const ifx_bf16* beta, const ifx_bf16* mod, int32_t mod_slots,
                             int32_t shift_slot, int32_t scale_slot, int32_t rows_per_group, void* stream) {
  IFX_REQUIRE(x && y && rows >= 0 && dim > 0 && dim % 8 == 0, "ifx_layernorm: bad x/y/rows/dim(%d)", dim);
  IFX_REQUIRE(mode >= IFX_LN_PLAIN && mode <= IFX_LN_MODULATE, "ifx_layernorm: bad mode %d", mode);
  if (mode == IFX_LN_AFFINE) IFX_REQUIRE(gamma && beta, "ifx_layernorm: affine mode needs gamma/beta");
  if (mode == IFX_LN_MODULATE)
    IFX_REQUIRE(mod && rows_per_group > 0 && mod_slots > 0 && shift_slot >= 0 && shift_slot < mod_slots &&
                    scale_slot >= 0 && scale_slot < mod_slots,
                "ifx_layernorm: modulate mode needs mod/slots/rows_per_group");
  if (rows == 0) return IFX_OK;
  return dispatch_nch(dim, [&](auto nch) {
    hipLaunchKernelGGL((layernorm_kernel<decltype(nch)::value, 0>), dim3((rows + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, x, y, rows, dim, eps, mode, gamma, beta, mod, mod_slots, shift_slot,
                       scale_slot, rows_per_group > 0 ? rows_per_group : 1, (unsigned char*)nullptr, 0, (float*)nullptr);
    return check_launch("ifx_layernorm");
  });
}

extern "C" int ifx_layernorm_quant(const ifx_bf16* x, void* q, int32_t ldq, float* scale, int32_t rows, int32_t dim, float eps,
                                   int32_t mode, const ifx_bf16* gamma, const ifx_bf16* beta, const ifx_bf16* mod,
                                   int32_t mod_slots, int32_t shift_slot, int32_t scale_slot, int32_t rows_per_group,
                                   int32_t format, void* stream) {
  IFX_REQUIRE(x && q && scale && rows >= 0 && dim > 0 && dim % 8 == 0 && ldq >= dim && ldq % 8 == 0,
              "ifx_layernorm_quant: bad x/q/scale/rows/dim(%d)/ldq(%d)", dim, ldq);
  IFX_REQUIRE(mode >= IFX_LN_PLAIN && mode <= IFX_LN_MODULATE, "ifx_layernorm_quant: bad mode %d", mode);
  IFX_REQUIRE(format == IFX_Q_FP8_E4M3 || format == IFX_Q_INT8, "ifx_layernorm_quant: unknown format %d", format);
  if (mode == IFX_LN_AFFINE) IFX_REQUIRE(gamma && beta, "ifx_layernorm_quant: affine mode needs gamma/beta");
  if (mode == IFX_LN_MODULATE)
    IFX_REQUIRE(mod && rows_per_group > 0 && mod_slots > 0 && shift_slot >= 0 && shift_slot < mod_slots &&
                    scale_slot >= 0 && scale_slot < mod_slots,
                "ifx_layernorm_quant: modulate mode needs mod/slots/rows_per_group");
  if (rows == 0) return IFX_OK;
  return dispatch_nch(dim, [&](auto nch) {
    constexpr int NC = decltype(nch)::value;
    const dim3 grid((rows + 3) / 4), block(256);
    const int rpg = rows_per_group > 0 ? rows_per_group : 1;
    if (format == IFX_Q_FP8_E4M3)
      hipLaunchKernelGGL((layernorm_kernel<NC, 1>), grid, block, 0, (hipStream_t)stream, x, (unsigned short*)nullptr, rows, dim, eps,
                         mode, gamma, beta, mod, mod_slots, shift_slot, scale_slot, rpg, (unsigned char*)q, ldq, scale);
    else
      hipLaunchKernelGGL((layernorm_kernel<NC, 2>), grid, block, 0, (hipStream_t)stream, x, (unsigned short*)nullptr, rows, dim, eps,
                         mode, gamma, beta, mod, mod_slots, shift_slot, scale_slot, rpg, (unsigned char*)q, ldq, scale);
    return check_launch("ifx_layernorm_quant");
  });
}

extern "C" int ifx_layernorm_quant_static(const ifx_bf16* x, void* q, int32_t ldq, const float* divisors, int32_t n_out, int32_t rows,
                                          int32_t dim, float eps, int32_t mode, const ifx_bf16* gamma, const ifx_bf16* beta,
                                          int32_t via_bf16, void* stream) {
  IFX_REQUIRE(x && q && divisors && rows >= 0 && dim > 0 && dim % 8 == 0 && n_out >= 1 && n_out <= 8 && ldq >= n_out * dim && ldq % 8 == 0,
              "ifx_layernorm_quant_static: bad x/q/divisors/rows/dim(%d)/n_out(%d)/ldq(%d)", dim, n_out, ldq);
  IFX_REQUIRE(mode == IFX_LN_PLAIN || mode == IFX_LN_AFFINE, "ifx_layernorm_quant_static: mode %d (plain or affine)", mode);
  if (mode == IFX_LN_AFFINE) IFX_REQUIRE(gamma && beta, "ifx_layernorm_quant_static: affine mode needs gamma/beta");
  if (rows == 0) return IFX_OK;
  return dispatch_nch(dim, [&](auto nch) {
    constexpr int NC = decltype(nch)::value;
    hipLaunchKernelGGL((layernorm_kernel<NC, 3>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)nullptr,
                       rows, dim, eps, mode, gamma, beta, (const unsigned short*)nullptr, 0, 0, 0, 1, (unsigned char*)q, ldq,
                       const_cast<float*>(divisors), n_out, via_bf16);
    return check_launch("ifx_layernorm_quant_static");
  });
}

extern "C" int ifx_rmsnorm(const ifx_bf16* x, int32_t ldx, ifx_bf16* y, int32_t ldy, const ifx_bf16* w,
                           int32_t rows, int32_t dim, float eps, void* stream) {
  IFX_REQUIRE(x && y && w && rows >= 0 && dim > 0 && dim % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0,
              "ifx_rmsnorm: bad arguments (dim %d)", dim);
  if (rows == 0) return IFX_OK;
  return dispatch_nch(dim, [&](auto nch) {
    hipLaunchKernelGGL((rmsnorm_kernel<decltype(nch)::value>), dim3((rows + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, y, ldy, w, rows, dim, eps);
    return check_launch("ifx_rmsnorm");
  });
}

extern "C" int ifx_rmsnorm_rope_kv_append(const ifx_bf16* qkv, int32_t ld, ifx_bf16* q_out, const ifx_bf16* wq,
                                          const ifx_bf16* wk, const ifx_rope_grid* rope, const ifx_kv_view* kv,
                                          int32_t local_start, int32_t rows, int32_t dim, float eps,
                                          void* stream) {
  IFX_REQUIRE(qkv && q_out && wq && rows >= 0 && dim > 0 && dim % 8 == 0 && ld % 8 == 0,
              "ifx_rmsnorm_rope_kv_append: bad arguments");
  int head_dim = 128;
  KvAddr ka{nullptr, 1};
  unsigned short *kc = nullptr, *vc = nullptr;
  if (kv) {
    IFX_REQUIRE(wk && kv->k && kv->v && kv->kv_heads * kv->head_dim == dim,
                "ifx_rmsnorm_rope_kv_append: kv view (heads %d x head_dim %d) does not match dim %d",
                kv->kv_heads, kv->head_dim, dim);
    IFX_REQUIRE(local_start >= 0 && local_start + rows <= kv->num_slots,
                "ifx_rmsnorm_rope_kv_append: append [%d, %d) exceeds cache capacity %d", local_start,
                local_start + rows, kv->num_slots);
    if (kv->page_table) IFX_REQUIRE(kv->page_size > 0, "ifx_rmsnorm_rope_kv_append: page_size must be > 0");
    head_dim = kv->head_dim;
    kc = kv->k;
    vc = kv->v;
    ka = KvAddr{kv->page_table, kv->page_size};
  }
  RopeArgs ra{};
  if (rope) {
    IFX_REQUIRE(rope->freqs && rope->hw_local > 0 && rope->width > 0 && rope->height > 0,
                "ifx_rmsnorm_rope_kv_append: bad rope grid");
    IFX_REQUIRE(head_dim % 16 == 0 && dim % head_dim == 0, "ifx_rmsnorm_rope_kv_append: head_dim %d", head_dim);
    const int frames = (rows + rope->hw_local - 1) / rope->hw_local;
    IFX_REQUIRE(rope->start_frame + frames <= rope->max_pos && rope->height <= rope->max_pos &&
                    rope->width <= rope->max_pos,
                "ifx_rmsnorm_rope_kv_append: positions exceed rope table (%d)", rope->max_pos);
    ra = RopeArgs{rope->freqs, rope->max_pos, rope->start_frame, rope->height,
                  rope->width, rope->hw_offset, rope->hw_local};
    IFX_REQUIRE(rope->q_scale >= 0.f && rope->q_scale == rope->q_scale, "ifx_rmsnorm_rope_kv_append: q_scale must be >= 0 (0 = 1)");
    ra.q_scale = rope->q_scale > 0.f ? rope->q_scale : 1.0f;
    if (rope->flags & 1) {            // the V rows are in their slots already (ifx_epilogue.y2): q and K only
      IFX_REQUIRE(kv != nullptr, "ifx_rmsnorm_rope_kv_append: flags bit 0 (V in place) needs a cache view");
      vc = nullptr;
    }
  }
  if (rows == 0) return IFX_OK;
  return dispatch_nch(dim, [&](auto nch) {
    hipLaunchKernelGGL((rmsnorm_rope_append_kernel<decltype(nch)::value>), dim3((rows + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, qkv, ld, q_out, wq, wk, ra, rope ? 1 : 0, kc, vc, ka, local_start,
                       rows, dim, head_dim, eps);
    return check_launch("ifx_rmsnorm_rope_kv_append");
  });
}

// K / V of this rank's rows of the new block, stored straight into the cache slots of EVERY peer (sequence-parallel exchange without a
// collective: the destinations are the peers' KV caches opened through IPC handles — xGMI stores; `n_dest` = 1 with the rank's own
// staging buffer gives the K/V-only form of rmsnorm_rope_append_kernel).  kv row = [k (dim) | v (dim)]; row r = local token
// (f = r / hw_local, i = r % hw_local) -> logical token local_start + f * frame_tokens + hw_offset + i (the single-GPU order).
struct PeerDest {
  unsigned short* k[IFX_MAX_PEERS];
  unsigned short* v[IFX_MAX_PEERS];
  int n;
};

template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_rope_kv_push_kernel(
    const unsigned short* __restrict__ kv, int ld, const unsigned short* __restrict__ wk, RopeArgs ra, int has_rope, PeerDest pd,
    KvAddr ka, int local_start, int frame_tokens, int slot_hw_local, int slot_hw_offset, int rows, int dim, int head_dim, float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const unsigned short* base = kv + (size_t)r * ld;
  int pos_t = 0, pos_h = 0, pos_w = 0;
  const int half = head_dim >> 1;
  const int n_h = half / 3, n_t = half - 2 * n_h;
  if (has_rope) {
    const int f = r / ra.hw_local;
    const int p = ra.hw_offset + (r - f * ra.hw_local);
    pos_t = ra.start_frame + f;
    pos_h = p / ra.width;
    pos_w = p - pos_h * ra.width;
  }
  const int sf = r / slot_hw_local;
  const size_t slot_off = (size_t)ka.slot(local_start + sf * frame_tokens + slot_hw_offset + (r - sf * slot_hw_local)) * dim;
  // as in rmsnorm_rope_append_kernel: the (cos, sin) pairs once per token, every load requested up front and branch-free
  const bool shared_cs = has_rope && (512 % head_dim) == 0;
  double2 cs4[4];
  if (shared_cs) {
    const int jp0 = ((lane * 8) % head_dim) >> 1;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int j = jp0 + p;
      const int pos = (j < n_t) ? pos_t : ((j < n_t + n_h) ? pos_h : pos_w);
      cs4[p] = *reinterpret_cast<const double2*>(ra.freqs + ((size_t)pos * half + j) * 2);
    }
  }
  u16x8 kraw[NCH], vraw[NCH], wkv[NCH];
  load_chunks<NCH>(kraw, base, dim, lane);
  load_chunks<NCH>(vraw, base + dim, dim, lane);
  load_chunks<NCH>(wkv, wk, dim, lane);
  Row<NCH> rowk;
  rowk.from(kraw, dim, lane);
  const float rs = 1.0f / sqrtf(wave_sum(rowk.sumsq()) / (float)dim + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 512 + lane * 8;
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = rbf(rbf(rowk.v[c][i] * rs) * bf2f(wkv[c][i]));
    if (shared_cs) rope4_cs(t, cs4);
    else if (has_rope && col < dim) rope4(t, (col % head_dim) >> 1, ra, half, n_t, n_h, pos_t, pos_h, pos_w);
    u16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f2bf(t[i]);
    if (col < dim) {
      for (int p = 0; p < pd.n; ++p) {
        *reinterpret_cast<u16x8*>(pd.k[p] + slot_off + col) = o;
        *reinterpret_cast<u16x8*>(pd.v[p] + slot_off + col) = vraw[c];
      }
    }
  }
}

extern "C" int ifx_rmsnorm_rope_kv_push(const ifx_bf16* kv_rows, int32_t ld, const ifx_bf16* wk, const ifx_rope_grid* rope,
                                        const ifx_peer_caches* peers, const ifx_kv_view* geometry, int32_t local_start,
                                        int32_t frame_tokens, int32_t slot_hw_local, int32_t slot_hw_offset, int32_t rows, int32_t dim,
                                        float eps, void* stream) {
  IFX_REQUIRE(kv_rows && wk && peers && geometry && rows >= 0 && dim > 0 && dim % 8 == 0 && ld % 8 == 0 && ld >= 2 * dim,
              "ifx_rmsnorm_rope_kv_push: bad arguments");
  IFX_REQUIRE(peers->count >= 1 && peers->count <= IFX_MAX_PEERS, "ifx_rmsnorm_rope_kv_push: %d destinations (1..%d)", peers->count,
              IFX_MAX_PEERS);
  IFX_REQUIRE(geometry->kv_heads * geometry->head_dim == dim, "ifx_rmsnorm_rope_kv_push: cache rows (heads %d x head_dim %d) != dim %d",
              geometry->kv_heads, geometry->head_dim, dim);
  IFX_REQUIRE(slot_hw_local > 0 && frame_tokens >= slot_hw_local && slot_hw_offset >= 0 && slot_hw_offset + slot_hw_local <= frame_tokens,
              "ifx_rmsnorm_rope_kv_push: bad shard geometry (frame %d tokens, shard [%d, %d))", frame_tokens, slot_hw_offset,
              slot_hw_offset + slot_hw_local);
  if (rows == 0) return IFX_OK;
  const int last_f = (rows - 1) / slot_hw_local;
  const int last_token = local_start + last_f * frame_tokens + slot_hw_offset + (rows - 1 - last_f * slot_hw_local);
  IFX_REQUIRE(local_start >= 0 && last_token < geometry->num_slots, "ifx_rmsnorm_rope_kv_push: token %d exceeds the cache capacity %d",
              last_token, geometry->num_slots);
  if (geometry->page_table) IFX_REQUIRE(geometry->page_size > 0, "ifx_rmsnorm_rope_kv_push: page_size must be > 0");
  PeerDest pd{};
  pd.n = peers->count;
  for (int p = 0; p < pd.n; ++p) {
    IFX_REQUIRE(peers->k[p] && peers->v[p], "ifx_rmsnorm_rope_kv_push: destination %d is null", p);
    pd.k[p] = peers->k[p];
    pd.v[p] = peers->v[p];
  }
  const int head_dim = geometry->head_dim;
  RopeArgs ra{};
  if (rope) {
    IFX_REQUIRE(rope->freqs && rope->hw_local > 0 && rope->width > 0 && rope->height > 0, "ifx_rmsnorm_rope_kv_push: bad rope grid");
    IFX_REQUIRE(head_dim % 16 == 0 && dim % head_dim == 0, "ifx_rmsnorm_rope_kv_push: head_dim %d", head_dim);
    const int frames = (rows + rope->hw_local - 1) / rope->hw_local;
    IFX_REQUIRE(rope->start_frame + frames <= rope->max_pos && rope->height <= rope->max_pos && rope->width <= rope->max_pos,
                "ifx_rmsnorm_rope_kv_push: positions exceed rope table (%d)", rope->max_pos);
    ra = RopeArgs{rope->freqs, rope->max_pos, rope->start_frame, rope->height, rope->width, rope->hw_offset, rope->hw_local};
  }
  const KvAddr ka{geometry->page_table, geometry->page_size, geometry->page_table ? 0 : geometry->seg_split,
                  geometry->page_table ? 0 : geometry->seg_delta};
  return dispatch_nch(dim, [&](auto nch) {
    hipLaunchKernelGGL((rmsnorm_rope_kv_push_kernel<decltype(nch)::value>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       kv_rows, ld, wk, ra, rope ? 1 : 0, pd, ka, local_start, frame_tokens, slot_hw_local, slot_hw_offset, rows, dim,
                       head_dim, eps);
    return check_launch("ifx_rmsnorm_rope_kv_push");
  });
}


extern "C" int ifx_kv_roll(const ifx_kv_view* kv, int32_t sink_tokens, int32_t evicted, int32_t rolled,
                           ifx_bf16* scratch, void* stream) {
  IFX_REQUIRE(kv && kv->k && kv->v && scratch, "ifx_kv_roll: null argument");
  IFX_REQUIRE(sink_tokens >= 0 && evicted >= 0 && rolled >= 0 &&
                  sink_tokens + evicted + rolled <= kv->num_slots,
              "ifx_kv_roll: span out of range");
  if (rolled == 0 || evicted == 0) return IFX_OK;
  const int row_elems = kv->kv_heads * kv->head_dim;
  KvAddr ka{kv->page_table, kv->page_size};
  const int blocks = 1024;
  // two-pass through scratch: source and destination spans may overlap
  for (int which = 0; which < 2; ++which) {
    unsigned short* c = which == 0 ? kv->k : kv->v;
    hipLaunchKernelGGL(kv_copy_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, c, scratch, ka,
                       sink_tokens + evicted, 0, rolled, row_elems, 1, 0);
    hipLaunchKernelGGL(kv_copy_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, scratch, c, ka, 0,
                       sink_tokens, rolled, row_elems, 0, 1);
  }
  return check_launch("ifx_kv_roll");
}

extern "C" int ifx_kv_scatter_shards(const ifx_bf16* gathered, int32_t world, int32_t frames, int32_t hw_local,
                                     int32_t frame_tokens, int32_t local_start, const ifx_kv_view* kv, void* stream) {
  IFX_REQUIRE(gathered && kv && kv->k && kv->v, "ifx_kv_scatter_shards: null argument");
  IFX_REQUIRE(world > 0 && frames > 0 && hw_local > 0 && world * hw_local <= frame_tokens && local_start >= 0,
              "ifx_kv_scatter_shards: bad shard geometry (world %d, frames %d, hw_local %d, frame_tokens %d)", world,
              frames, hw_local, frame_tokens);
  IFX_REQUIRE(local_start + (frames - 1) * frame_tokens + world * hw_local <= kv->num_slots,
              "ifx_kv_scatter_shards: tokens [%d, %d) exceed the cache capacity %d", local_start,
              local_start + (frames - 1) * frame_tokens + world * hw_local, kv->num_slots);
  if (kv->page_table) IFX_REQUIRE(kv->page_size > 0, "ifx_kv_scatter_shards: page_size must be > 0");
  const int row_elems = kv->kv_heads * kv->head_dim;
  IFX_REQUIRE(row_elems % 8 == 0, "ifx_kv_scatter_shards: row of %d elements is not 16-byte granular", row_elems);
  const size_t total = (size_t)world * frames * hw_local * (row_elems / 8);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(kv_scatter_shards_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gathered, kv->k, kv->v,
                     KvAddr{kv->page_table, kv->page_size}, world, frames, hw_local, frame_tokens, local_start, row_elems);
  return check_launch("ifx_kv_scatter_shards");
}
