// Block-causal flash-attention forward over a paged KV cache (gfx950, head_dim 128).
//
// The dominant kernel of the denoising step: N query tokens of the current block
// against the whole cached prefix (L = 4680 ... 32760 keys), no mask — causality
// is what is in the cache (reference: causal_model.py:307-315).  MFMA-bound
// (AI ~ 2-4 kFLOP/B), so the design goal is to keep the matrix pipe busy:
//
//  * workgroup = 4 waves x 32 query rows (128-row Q tile) of one head; grid =
//    heads x ceil(N/128) workgroups (444 for 480p), mapped so that each XCD gets a
//    contiguous head-major range of tiles: workgroups streaming the same head's
//    K/V share one XCD's L2.  64 KiB LDS + <=256 VGPR -> 2 workgroups per CU.
//  * K/V tiles of 64 keys are read IN PLACE from the cache pages (token-major rows,
//    256 B per (token, head)), staged global -> registers -> LDS, double buffered,
//    one barrier per tile; the loads of tile t+1 are issued before the MFMAs of tile t.
//  * S^T = K·Q^T with v_mfma_f32_32x32x16_bf16 (A = K rows from LDS via
//    ds_read_b128, 16-byte chunk index XOR (row & 15): conflict free; B = Q^T held
//    in registers for the whole kernel).  Swapped operands put ONE query per lane
//    column, so the online-softmax row reductions are in-register plus a single
//    cross-half exchange, and the running max / rescale factor are lane-local
//    scalars shared with the O^T accumulators.
//  * O^T += V^T·P^T: P is used straight from the S^T accumulator registers as the
//    B operand (no LDS round trip, no lane shuffles): the reduction index of the
//    second MFMA is RELABELLED so that k-slot (half, j) of step s means key
//    32*blk + 16*s + 8*(j>>2) + 4*half + (j&3) — exactly the keys this lane
//    already owns — and the V^T A-operand is fetched for those same keys with the
//    LDS transpose read ds_read_b64_tr_b16 (two per fragment; V rows keep their
//    256-byte layout, 64-byte chunk index XOR (row & 3): conflict free).
//  * fp32 softmax with exp2 and the log2(e)-folded scale; P rounded to bf16 for PV
//    (same as FA2 / SDPA); O normalised once at the end; optional LSE for split-KV merges.
#include "ifx_common.h"

namespace ifx {

constexpr int QT = 128;   // query rows per workgroup
constexpr int KT = 64;    // keys per tile
constexpr int HD = 128;   // head_dim

struct AttnArgs {
  const unsigned short* q;
  unsigned short* out;
  float* lse;
  const unsigned short* k;
  const unsigned short* v;
  KvAddr ka;
  int q_rows, heads, kv_start, kv_len, q_tiles, per_xcd, total;   // keys [kv_start, kv_len)
  float scale, scale_log2;
};

// SHORT = cross-attention specialisation (kv_len <= 1024: the 512 cached text keys).  Same algorithm today;
// a separate instantiation so that profiles separate it from the block-causal self-attention launches.
template <bool PAGED, bool SHORT>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs A) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];   // [buf][K 16K | V 16K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;

  // XCD-aware work mapping (speed only): XCD x owns work items [x*per, (x+1)*per), head-major
  const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
  const int wi = xcd * A.per_xcd + slot_i;
  if (slot_i >= A.per_xcd || wi >= A.total) return;
  const int head = wi / A.q_tiles, qt = wi - head * A.q_tiles;
  const int row_stride = A.heads * HD;   // elements between consecutive tokens

  // ---- Q fragments (B operand of S^T = K Q^T): Q[q = l31][16*ks + 8*hi + 0..7]
  const int qrow = qt * QT + wave * 32 + l31;
  const int qrow_c = min(qrow, A.q_rows - 1);
  bf16x8 qf[8];
  {
    const unsigned short* qp = A.q + (size_t)qrow_c * row_stride + head * HD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }

  // ---- staging map: 4 x 16-byte chunks of K and of V per thread per tile
  const int st_row = tid >> 4, st_c = tid & 15;        // rows st_row + 16p, chunk st_c
  int k_off[4], v_off[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = st_row + 16 * p;
    k_off[p] = row * 256 + ((st_c ^ (row & 15)) << 4);
    v_off[p] = row * 256 + ((((st_c >> 2) ^ (row & 3)) << 6) | ((st_c & 3) << 4));
  }
  const unsigned short* kbase = A.k + head * HD + st_c * 8;
  const unsigned short* vbase = A.v + head * HD + st_c * 8;
  u32x4 rk[4], rv[4];
  auto gload = [&](int t) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int key = A.kv_start + t * KT + st_row + 16 * p;
      if (key < A.kv_len) {
        const size_t off = (size_t)(PAGED ? A.ka.slot(key) : key) * row_stride;
        rk[p] = *reinterpret_cast<const u32x4*>(kbase + off);
        rv[p] = *reinterpret_cast<const u32x4*>(vbase + off);
      } else {
        rk[p] = u32x4{0, 0, 0, 0};
        rv[p] = u32x4{0, 0, 0, 0};
      }
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* kb = smem + buf * 32768;
    unsigned char* vb = kb + 16384;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *reinterpret_cast<u32x4*>(kb + k_off[p]) = rk[p];
      *reinterpret_cast<u32x4*>(vb + v_off[p]) = rv[p];
    }
  };

  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c2 = A.scale_log2;

  // per-lane LDS read bases
  //  K A-operand: row = 32*blk + l31, 16-byte chunk (2*ks + hi) ^ (row & 15)   [(32*blk + l31) & 15 == l31 & 15]
  const int kswz = l31 & 15;
  //  V^T A-operand (transpose read): in 16-lane group g = lane>>4, lane i = lane&15 supplies
  //  row key0 + (i>>2), cols 32*db + 16*(g&1) + 4*(i&3);  key0 = 32*blk + 16*s + 4*hi (+8)
  const int vi = lane & 15, vg1 = (lane >> 4) & 1;
  const int v_rowq = vi >> 2;                                  // == (row & 3) since key0 % 4 == 0
  const int v_in = (vg1 << 5) | ((vi & 3) << 3);               // byte offset inside the 64-byte chunk

  const int nkeys = A.kv_len - A.kv_start;
  const int NT = (nkeys + KT - 1) / KT;
  gload(0);
  lstore(0);
  __syncthreads();

  for (int t = 0; t < NT; ++t) {
    const int buf = t & 1;
    if (t + 1 < NT) gload(t + 1);
    const unsigned char* kb = smem + buf * 32768;
    const unsigned char* vb = kb + 16384;

    // ---------------- S^T = K Q^T ----------------
    f32x16 s[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[b][r] = 0.f;
      const unsigned char* krow = kb + (32 * b + l31) * 256;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(krow + (((2 * ks + hi) ^ kswz) << 4));
        s[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s[b], 0, 0, 0);
      }
    }
    if (t == NT - 1 && (nkeys & (KT - 1))) {   // ragged last tile: keys >= kv_len get -inf
      const int kbase_idx = t * KT + 4 * hi;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbase_idx + 32 * b + (r & 3) + 8 * (r >> 2) >= nkeys) s[b][r] = -INFINITY;
    }

    // ---------------- online softmax (one query per lane column) ----------------
    float mx = s[0][0];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[b][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
    const float mc = m_new * c2;
    m_run = m_new;
    float psum = 0.f;
    bf16x8 pb[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[b][r] * c2 - mc);
        psum += p;
        pb[b][r >> 3][r & 7] = static_cast<__bf16>(p);
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

    // ---------------- O^T += V^T P^T ----------------
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int key0 = 32 * b + 16 * s2 + 4 * hi;
        const unsigned char* vr0 = vb + (key0 + v_rowq) * 256 + v_in;
        const unsigned char* vr1 = vr0 + 8 * 256;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int ch = (d ^ v_rowq) << 6;
          const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (bf16x4 __attribute__((address_space(3)))*)(vr0 + ch));
          const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (bf16x4 __attribute__((address_space(3)))*)(vr1 + ch));
          const bf16x8 a = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb[b][s2], o[d], 0, 0, 0);
        }
      }

    if (t + 1 < NT) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (qrow < A.q_rows) {
    unsigned short* op = A.out + (size_t)qrow * row_stride + head * HD + 4 * hi;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = f2bf(o[d][4 * g + e] * inv);
        *reinterpret_cast<u16x4*>(op + 32 * d + 8 * g) = w;
      }
    if (A.lse != nullptr && hi == 0) A.lse[(size_t)head * A.q_rows + qrow] = m_run * A.scale + __logf(l_tot);
  }
}

__global__ __launch_bounds__(256) void lse_merge_kernel(unsigned short* __restrict__ oa, float* __restrict__ la,
                                                        const unsigned short* __restrict__ ob,
                                                        const float* __restrict__ lb, int rows, int heads) {
  // one thread per (row, head, 8 channels)
  const size_t total = (size_t)rows * heads * 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i & 15);
    const size_t rh = i >> 4;
    const int h = (int)(rh % heads);
    const size_t r = rh / heads;
    const float a = la[(size_t)h * rows + r], b = lb[(size_t)h * rows + r];
    // out = out_a - sigmoid(lse_b - lse_a) * (out_a - out_b) ; lse = lse_a - logsigmoid(lse_a - lse_b)
    const float sg = 1.0f / (1.0f + __expf(a - b));
    u16x8 ua = *reinterpret_cast<const u16x8*>(oa + rh * HD + c * 8);
    const u16x8 ub = *reinterpret_cast<const u16x8*>(ob + rh * HD + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = bf2f(ua[e]), y = bf2f(ub[e]);
      ua[e] = f2bf(x - sg * (x - y));
    }
    *reinterpret_cast<u16x8*>(oa + rh * HD + c * 8) = ua;
    if (c == 0) {
      const float mxv = fmaxf(a, b);
      la[(size_t)h * rows + r] = mxv + __logf(__expf(a - mxv) + __expf(b - mxv));
    }
  }
}

}  // namespace ifx

using namespace ifx;

extern "C" int ifx_attn_fwd_paged(const ifx_bf16* q, ifx_bf16* out, float* lse, const ifx_kv_view* kv,
                                  int32_t q_rows, int32_t heads, int32_t kv_start, int32_t kv_len, float scale,
                                  void* stream) {
  IFX_REQUIRE(q && out && kv && kv->k && kv->v, "ifx_attn_fwd_paged: null argument");
  IFX_REQUIRE(kv->head_dim == HD, "ifx_attn_fwd_paged: head_dim %d not built (128 only)", kv->head_dim);
  IFX_REQUIRE(heads > 0 && kv->kv_heads == heads, "ifx_attn_fwd_paged: heads %d vs kv_heads %d", heads,
              kv->kv_heads);
  IFX_REQUIRE(q_rows >= 0 && kv_start >= 0 && kv_len > kv_start && kv_len <= kv->num_slots,
              "ifx_attn_fwd_paged: key range [%d, %d) out of range (capacity %d)", kv_start, kv_len, kv->num_slots);
  if (kv->page_table) IFX_REQUIRE(kv->page_size > 0, "ifx_attn_fwd_paged: page_size must be > 0");
  if (q_rows == 0) return IFX_OK;
  AttnArgs a;
  a.q = q;
  a.out = out;
  a.lse = lse;
  a.k = kv->k;
  a.v = kv->v;
  a.ka = KvAddr{kv->page_table, kv->page_size};
  a.q_rows = q_rows;
  a.heads = heads;
  a.kv_start = kv_start;
  a.kv_len = kv_len;
  a.q_tiles = (q_rows + QT - 1) / QT;
  a.total = a.q_tiles * heads;
  a.per_xcd = (a.total + 7) / 8;
  a.scale = scale > 0.f ? scale : 0.08838834764831845f;   // 1/sqrt(128)
  a.scale_log2 = a.scale * 1.4426950408889634f;
  const dim3 grid(a.per_xcd * 8), block(256);
  const bool short_kv = kv_len - kv_start <= 1024;
  if (kv->page_table) {
    if (short_kv) hipLaunchKernelGGL((attn_fwd_kernel<true, true>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<true, false>), grid, block, 0, (hipStream_t)stream, a);
  } else {
    if (short_kv) hipLaunchKernelGGL((attn_fwd_kernel<false, true>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, false>), grid, block, 0, (hipStream_t)stream, a);
  }
  return check_launch("ifx_attn_fwd_paged");
}

extern "C" int ifx_lse_merge(ifx_bf16* out_a, float* lse_a, const ifx_bf16* out_b, const float* lse_b,
                             int32_t rows, int32_t heads, void* stream) {
  IFX_REQUIRE(out_a && lse_a && out_b && lse_b && rows >= 0 && heads > 0, "ifx_lse_merge: bad arguments");
  if (rows == 0) return IFX_OK;
  const size_t total = (size_t)rows * heads * 16;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(lse_merge_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out_a, lse_a, out_b,
                     lse_b, rows, heads);
  return check_launch("ifx_lse_merge");
}
