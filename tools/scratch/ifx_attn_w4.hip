// NOT BUILT INTO libinferix_hip.so: a measured experiment kept for the record (757 vs 1050 TFLOP/s for the shipped schedule).
// To try it: add it to inferix_amd/csrc/Makefile SRCS and call launch_attn_w4 from attn_dispatch (ifx_attn.hip).
// Block-causal flash-attention forward, FOUR-wave schedule with 64 queries per wave (gfx950, head_dim 128).
//
// Why (tools/probe_overlap.hip): ONE wave that interleaves its own MFMAs with its own softmax VALU work keeps the
// matrix pipe 77 % busy at the softmax's instruction mix (two such waves per SIMD: 95 %), whereas the eight-wave kernels of
// ifx_attn_pp.hip — two waves of 32 queries per SIMD coupled by a workgroup barrier per tile — measure 59 %: the younger wave of
// every SIMD loses issue arbitration, the older one waits for it at the barrier.  Here a workgroup is four waves, one per SIMD,
// each owning 64 queries: per 32-key step a wave issues
//     PV(h-1): O^T += V^T P^T      16 MFMAs (4 d-blocks x 2 key slots x 2 query blocks, V^T fragments shared by the query blocks)
//     QK(h+1): S^T  = K Q^T        16 MFMAs (8 d-steps x 2 query blocks, K fragments shared by the query blocks)
//     softmax(h)                   32 scores per lane: exp2, row sums, bf16 pack  (VALU, in the MFMAs' shadows)
// three mutually independent streams (S and P are double buffered by step parity), pinned with sched_group_barrier.  A wave reads
// each K / V^T fragment once for 64 queries: half the LDS fragment traffic per FLOP of the 32-query waves.
// K/V tiles of 64 keys arrive by LDS-DMA through buffer descriptors exactly as in ifx_attn_pp.hip (same LDS images and swizzles:
// K chunk ^ (row & 15), V 64-byte chunk ^ (row & 3)); K ring 4 tiles, V ring 5 (144 KiB), prefetch distance 3, one counted
// s_waitcnt + one workgroup barrier per 64-key tile.  Lazy reference maximum as in the other kernels (exact; redo when a row sum
// outgrows 2^20).  Unsplit launches only (the large self-attention launches of the clip); everything else stays on ifx_attn_pp.hip.
#include <type_traits>

#include "ifx_common.h"

namespace ifx {

struct AttnArgsW4 {
  const unsigned short* q;
  unsigned short* out;
  float* lse;
  const unsigned short* k;
  const unsigned short* v;
  KvAddr ka;
  int q_rows, heads, kv_start, kv_len, num_slots, q_tiles, per_xcd, total;
  int ldq, ldo;
  int kv_heads, q_per_kv;
  float scale, scale_log2;
};

namespace aw4 {
constexpr int KT = 64, HD = 128, QT = 256;
constexpr int RK = 4, RV = 5, PD = 3;
constexpr int V_RING = RK * 16384;
constexpr int LDS_BYTES = (RK + RV) * 16384;      // 147456

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(v4i rsrc, unsigned lds, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ v4i make_rsrc(const void* base, unsigned num_bytes) {
  const unsigned long long a = (unsigned long long)base;
  v4i r;
  r[0] = (int)(unsigned)a;
  r[1] = (int)((unsigned)(a >> 32) & 0xffffu);
  r[2] = (int)num_bytes;
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float half_max(float x) {      // max over lane and lane ^ 32
  float a = x, b = x, r;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %2, %0, %1" : "+v"(a), "+v"(b), "=v"(r));
  return r;
}
}  // namespace aw4

#define AW4_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

template <bool PAGED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_w4_kernel(AttnArgsW4 A) {
  using namespace aw4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
  const int wi = xcd * A.per_xcd + slot_i;
  if (slot_i >= A.per_xcd || wi >= A.total) return;
  const int head = wi / A.q_tiles, qt = wi - head * A.q_tiles;
  const int kv_s = A.kv_start, kv_e = A.kv_len;
  const int nkeys = kv_e - kv_s, NT = (nkeys + KT - 1) / KT, NH = (nkeys + 31) / 32;   // 64-key tiles, 32-key steps

  // ---- Q fragments (B operand of S^T = K Q^T): query block qb = rows q0 + 32 qb + l31, chunk 16 ks + 8 hi
  const int q0 = qt * QT + wave * 64;
  bf16x8 qf[2][8];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qr = min(q0 + 32 * qb + l31, A.q_rows - 1);
    const unsigned short* qp = A.q + (size_t)qr * A.ldq + head * HD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[qb][ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): retire the Q loads before the asm-issued DMA shares the counter

  // ---- LDS-DMA: a tile = 16 K pieces + 16 V pieces of 1 KiB (4 key rows x 256 B); wave w moves pieces w, w + 4 (rows 0..31) and
  //      the same 32 rows further down
  const int row_bytes = A.kv_heads * HD * 2;
  const int kvh = head / A.q_per_kv;
  const unsigned valid_rows = PAGED ? (unsigned)A.num_slots : (unsigned)kv_e;
  const unsigned nrec = (valid_rows - 1) * (unsigned)row_bytes + 256u;
  const v4i krs = make_rsrc(A.k + kvh * HD, nrec), vrs = make_rsrc(A.v + kvh * HD, nrec);
  const unsigned lds00 = (unsigned)(unsigned long long)(lds_ptr_t)smem;
  const int last_key = kv_e - 1;
  auto issue = [&](int t_req) {
    const int t = min(t_req, NT - 1);          // past the end: re-read the last tile into a slot nobody reads (keeps vmcnt uniform)
    const unsigned kslot = lds00 + (t_req % RK) * 16384, vslot = lds00 + V_RING + (t_req % RV) * 16384;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int wv = wave + 4 * half;
      const int d_pc = lane & 15, d_row0 = wv * 4 + (lane >> 4);
      const int k_voff = d_row0 * row_bytes + ((d_pc ^ (d_row0 & 15)) << 4);
      const int v_voff = d_row0 * row_bytes + (((((d_pc >> 2) ^ (d_row0 & 3)) << 2) | (d_pc & 3)) << 4);
      const unsigned kb = kslot + wv * 1024, vb = vslot + wv * 1024;
      if (!PAGED) {
        const int soff = (kv_s + t * KT) * row_bytes;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          dma16(krs, kb + r * 8192, k_voff, soff + r * 32 * row_bytes);
          dma16(vrs, vb + r * 8192, v_voff, soff + r * 32 * row_bytes);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int key = min(kv_s + t * KT + d_row0 + 32 * r, last_key);
          const int delta = (A.ka.slot(key) - d_row0) * row_bytes;
          dma16(krs, kb + r * 8192, k_voff + delta, 0);
          dma16(vrs, vb + r * 8192, v_voff + delta, 0);
        }
      }
    }
  };

  // ---- state
  f32x16 o[4][2];                    // O^T blocks [d-block][query block]
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][qb][r] = 0.f;
  f32x16 s0[2], s1[2];               // S^T of the even / odd step, per query block
  bf16x8 p0[2][2], p1[2][2];         // P^T of the even / odd step: [query block][key slot]
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  const float c2 = A.scale_log2;
  const int kswz = l31 & 15;
  const int vi = lane & 15, vg1 = (lane >> 4) & 1;
  const int v_rowq = vi >> 2;
  const int v_in = (vg1 << 5) | ((vi & 3) << 3);

  // K fragments of key block b (32 keys) of the tile in `kb`, d-steps 4 kh .. 4 kh + 3
  auto ldK = [&](bf16x8(&f)[4], const unsigned char* kb, int b, int kh) {
    const unsigned char* krow = kb + (32 * b + l31) * 256;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) f[k4] = *reinterpret_cast<const bf16x8*>(krow + (((2 * (4 * kh + k4) + hi) ^ kswz) << 4));
  };
  // V^T fragments (4 d-blocks) of key slot s2 (16 keys in the order the S accumulators hold them) of key block b
  auto ldV = [&](bf16x8(&f)[4], const unsigned char* vb, int b, int s2) {
    const unsigned char* vr0 = vb + (32 * b + 16 * s2 + 4 * hi + v_rowq) * 256 + v_in;
    const unsigned char* vr1 = vr0 + 8 * 256;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int ch = (d ^ v_rowq) << 6;
      const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vr0 + ch));
      const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vr1 + ch));
      f[d] = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  // QK of one 32-key block for both query blocks: 16 MFMAs, 8 fragment reads
  auto qk_block = [&](f32x16(&s)[2], const unsigned char* kb, int b) {
    bf16x8 ka[4], kc[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) s[0][r] = 0.f, s[1][r] = 0.f;
    ldK(ka, kb, b, 0);
    ldK(kc, kb, b, 1);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[k4], qf[0][k4], s[0], 0, 0, 0);
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[k4], qf[1][k4], s[1], 0, 0, 0);
    }
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[k4], qf[0][4 + k4], s[0], 0, 0, 0);
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[k4], qf[1][4 + k4], s[1], 0, 0, 0);
    }
  };
  // PV of one 32-key block for both query blocks: 16 MFMAs, 8 fragments (16 transposing reads)
  auto pv_block = [&](const bf16x8(&p)[2][2], const unsigned char* vb, int b) {
    bf16x8 va[4], vc[4];
    ldV(va, vb, b, 0);
    ldV(vc, vb, b, 1);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      o[d][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[d], p[0][0], o[d][0], 0, 0, 0);
      o[d][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[d], p[1][0], o[d][1], 0, 0, 0);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      o[d][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc[d], p[0][1], o[d][0], 0, 0, 0);
      o[d][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc[d], p[1][1], o[d][1], 0, 0, 0);
    }
  };
  auto block_max = [&](const f32x16& sb, float m) -> float {
    float mx = __builtin_fmaxf(sb[0], sb[1]);
    mx = max3(mx, sb[2], m);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = max3(mx, sb[r], sb[r + 1]);
    mx = max3(mx, sb[15], mx);
    return half_max(mx);
  };
  auto exp_block = [&](const f32x16& sb, bf16x8(&pb)[2], float m) -> float {
    const float mc = m * c2;
    const f32x2 c2v = {c2, c2}, mcv = {mc, mc};
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x2 sv = {sb[2 * i], sb[2 * i + 1]};
      const f32x2 e = sv * c2v - mcv;
      const float e0 = __builtin_amdgcn_exp2f(e[0]), e1 = __builtin_amdgcn_exp2f(e[1]);
      a0 += e0;
      a1 += e1;
      pb[i >> 2][(2 * i) & 7] = static_cast<__bf16>(e0);
      pb[i >> 2][(2 * i + 1) & 7] = static_cast<__bf16>(e1);
    }
    return a0 + a1;
  };
  // softmax of step h (32 keys) for both query blocks; alpha[qb] = the factor O must be rescaled by (1 normally).  Both blocks'
  // exponentials come first, straight-line, so that they sit in the basic block of the step's MFMAs; one test covers both.
  auto softmax_step = [&](f32x16(&s)[2], bf16x8(&p)[2][2], int h, float (&alpha)[2], auto first_tag, auto ragged_tag) {
    constexpr bool FIRST = decltype(first_tag)::value, RAGGED = decltype(ragged_tag)::value;
    if (RAGGED) {
      const int kidx = 32 * h + 4 * hi;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kidx + (r & 3) + 8 * (r >> 2) >= nkeys) s[qb][r] = -INFINITY;
    }
    if (FIRST) m_run[0] = block_max(s[0], m_run[0]), m_run[1] = block_max(s[1], m_run[1]);
    float ps0 = exp_block(s[0], p[0], m_run[0]);
    float ps1 = exp_block(s[1], p[1], m_run[1]);
    alpha[0] = alpha[1] = 1.0f;
    if (__any(!(ps0 < 1048576.f) || !(ps1 < 1048576.f))) {            // rare: redo against the true maxima (inf / NaN land here)
      const float n0 = block_max(s[0], m_run[0]), n1 = block_max(s[1], m_run[1]);
      alpha[0] = __builtin_amdgcn_exp2f((m_run[0] - n0) * c2);
      alpha[1] = __builtin_amdgcn_exp2f((m_run[1] - n1) * c2);
      m_run[0] = n0, m_run[1] = n1;
      ps0 = exp_block(s[0], p[0], n0);
      ps1 = exp_block(s[1], p[1], n1);
      l_run[0] *= alpha[0];
      l_run[1] *= alpha[1];
    }
    l_run[0] += ps0;
    l_run[1] += ps1;
  };
  auto rescale = [&](const float (&alpha)[2]) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float a = alpha[qb];
      if (__any(a != 1.0f)) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[d][qb][r] *= a;
      }
    }
  };
  // pin one step: 32 MFMAs, 24 LDS reads, the softmax VALU work spread under them
#ifndef AW4_QA
#define AW4_QA 1
#endif
#ifndef AW4_PIN
#define AW4_PIN 1
#endif
#if AW4_PIN == 0
#define AW4_PIN_STEP() __builtin_amdgcn_sched_barrier(0)
#elif AW4_PIN == 1
#define AW4_PIN_STEP()                                   \
  do {                                                   \
    _Pragma("unroll") for (int _n = 0; _n < 8; ++_n) {   \
      AW4_SG(0x008, 1); AW4_SG(0x100, 1); AW4_SG(0x002, 4); \
      AW4_SG(0x008, 1); AW4_SG(0x100, 1); AW4_SG(0x002, 4); \
      AW4_SG(0x008, 1); AW4_SG(0x100, 1); AW4_SG(0x002, 4); \
      AW4_SG(0x008, 1); AW4_SG(0x002, 4);                \
    }                                                    \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)
#elif AW4_PIN == 2
#define AW4_PIN_STEP()                                   \
  do {                                                   \
    AW4_SG(0x100, 4); AW4_SG(0x002, 6);                  \
    _Pragma("unroll") for (int _n = 0; _n < 10; ++_n) {  \
      AW4_SG(0x008, 2); AW4_SG(0x100, 2); AW4_SG(0x002, 8); \
    }                                                    \
    _Pragma("unroll") for (int _n = 0; _n < 6; ++_n) {   \
      AW4_SG(0x008, 2); AW4_SG(0x002, 8);                \
    }                                                    \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)
#endif

  // ---- prologue: PD tiles in flight, tile 0 (and 1) landed, S(0) computed
#pragma unroll
  for (int t = 0; t < PD; ++t) issue(t);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // tiles 0 and 1 of this wave landed (tile 2 may be in flight)
  __builtin_amdgcn_s_barrier();
  qk_block(s0, smem, 0);
  __builtin_amdgcn_sched_barrier(0);

  float alpha[2];
  auto tile_iter = [&](int t, auto first_tag, auto last_tag) {
    constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
    const unsigned char* kt0 = smem + (t % RK) * 16384;
    const unsigned char* kt1 = smem + ((t + 1) % RK) * 16384;
    const unsigned char* vtm = smem + V_RING + ((t + RV - 1) % RV) * 16384;
    const unsigned char* vt0 = smem + V_RING + (t % RV) * 16384;
    if (!FIRST) {
      // K(t+1), V(t) needed below: at most the youngest tile (t + 2) may still be in flight
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // everyone done with K(t-1), V(t-2): their slots take tile t + 3
    }
    issue(t + PD);
#if AW4_QA
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+a"(qf[qb][ks]));     // keep the Q fragments in AGPRs (MFMA reads them there)
#endif
    __builtin_amdgcn_sched_barrier(0);
    // ---- step 2t: PV(2t-1) || QK(2t+1) || softmax(2t)
    qk_block(s1, kt0, 1);
    if (!FIRST) pv_block(p1, vtm, 1);
    softmax_step(s0, p0, 2 * t, alpha, first_tag, last_tag);
    AW4_PIN_STEP();
    rescale(alpha);
    // ---- step 2t+1: PV(2t) || QK(2t+2) || softmax(2t+1)
    if (!LAST) qk_block(s0, kt1, 0);
    pv_block(p0, vt0, 0);
    if (!LAST || 2 * t + 1 < NH) softmax_step(s1, p1, 2 * t + 1, alpha, std::false_type{}, last_tag);
    else {
      alpha[0] = alpha[1] = 1.0f;
      p1[0][0] = p1[0][1] = p1[1][0] = p1[1][1] = bf16x8{};
    }
    AW4_PIN_STEP();
    rescale(alpha);
  };
  if (NT == 1) tile_iter(0, std::true_type{}, std::true_type{});
  else {
    tile_iter(0, std::true_type{}, std::false_type{});
    for (int t = 1; t < NT - 1; ++t) tile_iter(t, std::false_type{}, std::false_type{});
    tile_iter(NT - 1, std::false_type{}, std::true_type{});
  }
  // ---- drain: PV of the last step
  {
    const unsigned char* vl = smem + V_RING + ((NT - 1) % RV) * 16384;
    pv_block(p1, vl, 1);
  }

  // ---- epilogue
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + 32 * qb + l31;
    if (qrow < A.q_rows) {
      unsigned short* op = A.out + (size_t)qrow * A.ldo + head * HD + 4 * hi;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u16x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = f2bf(o[d][qb][4 * g + e] * inv);
          *reinterpret_cast<u16x4*>(op + 32 * d + 8 * g) = w;
        }
      if (A.lse != nullptr && hi == 0) A.lse[(size_t)head * A.q_rows + qrow] = m_run[qb] * A.scale + __logf(l_tot);
    }
  }
}

int launch_attn_w4(const unsigned short* q, unsigned short* out, float* lse, const ifx_kv_view* kv, int q_rows, int heads, int kv_start,
                   int kv_len, float scale, hipStream_t stream, int ldq, int ldo) {
  using namespace aw4;
  AttnArgsW4 a;
  a.q = q, a.out = out, a.lse = lse, a.k = kv->k, a.v = kv->v;
  a.ka = KvAddr{kv->page_table, kv->page_size, kv->page_table ? 0 : kv->seg_split, kv->page_table ? 0 : kv->seg_delta};
  a.q_rows = q_rows, a.heads = heads, a.kv_start = kv_start, a.kv_len = kv_len, a.num_slots = kv->num_slots;
  a.ldq = ldq > 0 ? ldq : heads * HD;
  a.ldo = ldo > 0 ? ldo : heads * HD;
  a.kv_heads = kv->kv_heads, a.q_per_kv = heads / kv->kv_heads;
  a.q_tiles = (q_rows + QT - 1) / QT;
  a.total = a.q_tiles * heads;
  a.per_xcd = (a.total + 7) / 8;
  a.scale = scale > 0.f ? scale : 0.08838834764831845f;
  a.scale_log2 = a.scale * 1.4426950408889634f;
  const bool paged = kv->page_table != nullptr || kv->seg_split > 0;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_w4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)attn_w4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  const dim3 grid(a.per_xcd * 8), block(256);
  if (paged) hipLaunchKernelGGL((attn_w4_kernel<true>), grid, block, LDS_BYTES, stream, a);
  else hipLaunchKernelGGL((attn_w4_kernel<false>), grid, block, LDS_BYTES, stream, a);
  return check_launch("ifx_attn_fwd_paged(w4)");
}

}  // namespace ifx
