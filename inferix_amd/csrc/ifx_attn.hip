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
#include <stdlib.h>
#include <type_traits>

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
  int ldq, ldo;             // elements between consecutive rows of q / out
  float last_key_bias;      // added to the RAW score of the last key (ln(multiplicity) / scale): that key stands for `multiplicity` identical ones
  int kv_heads, q_per_kv;   // grouped-query attention: query head h reads kv head h / q_per_kv
  float scale, scale_log2;
};

typedef __attribute__((address_space(3))) void* attn_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* attn_gbl_ptr_t;

// One LDS-DMA instruction issued from inline asm (64 lanes x 16 B, lane-linear at LDS byte address `lds`).  hipcc tracks
// the LDS-DMA builtin as an LDS store and drains vmcnt in front of the next LDS read that may alias it — here the
// V^T fragment reads of the CURRENT tile — which made every tile wait for the prefetch of the NEXT one
// (profiles/r1d_attention_step_trace.md).  Ordering is by the explicit vmcnt(0) + barrier at the end of each tile.
__device__ __forceinline__ void attn_dma16(const unsigned short* src, unsigned lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lds), "v"(src) : "memory");
}

// single-instruction 3-input max (plain fmaxf on MFMA outputs makes hipcc emit a canonicalising v_max per input)
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// max over the two half-waves that share a query column (lane, lane^32) without touching LDS:
// v_permlane32_swap exchanges vdst[32..63] with src[0..31] (verified by tools/probe_layouts).  Done in asm:
// the builtin called with two copies of one value is folded to a no-op by the optimiser.
__device__ __forceinline__ float half_swap_max(float x) {
  float a = x, b = x, r;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %2, %0, %1"
               : "+v"(a), "+v"(b), "=v"(r));
  return r;
}

// SHORT = cross-attention specialisation (kv_len <= 1024: the 512 cached text keys).  Same algorithm today;
// a separate instantiation so that profiles separate it from the block-causal self-attention launches.
//
// v2 structure (round 1, second pass — v1 measured 34 % MFMA utilisation, profiles/r1_pmc_attn_gemm.md):
//  * K/V tiles go global -> LDS by LDS-DMA (global_load_lds, 16 B/lane, 8 per wave per tile): no staging
//    VGPRs, no ds_write pass; the XOR swizzles are applied on the per-lane SOURCE address (the DMA writes
//    LDS lane-linearly) and on the fragment reads.  Tile t+1 is in flight during all of tile t.
//  * the ragged last tile is peeled out of the main loop (no masking selects in the steady state);
//  * O is rescaled only in tiles where some row's running max actually grows (exact: alpha == 1 otherwise);
//  * K fragments of a 32-key block are fetched as one batch before its 8 MFMAs, V^T fragments 8 reads per
//    4 MFMAs, so the LDS latency is paid once per batch and the MFMAs of a batch issue back to back
//    (s_setprio 1 around them so the partner wave's VALU work does not starve the matrix pipe).
template <bool PAGED, bool SHORT>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs A) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];   // [buf][K 16K | V 16K]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;

  // XCD-aware work mapping (speed only): XCD x owns work items [x*per, (x+1)*per), head-major
  const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
  const int wi = xcd * A.per_xcd + slot_i;
  if (slot_i >= A.per_xcd || wi >= A.total) return;
  const int head = wi / A.q_tiles, qt = wi - head * A.q_tiles;
  const int kv_stride = A.kv_heads * HD;  // ... of the cache rows
  const int kvh = head / A.q_per_kv;

  // ---- Q fragments (B operand of S^T = K Q^T): Q[q = l31][16*ks + 8*hi + 0..7]
  const int qrow = qt * QT + wave * 32 + l31;
  const int qrow_c = min(qrow, A.q_rows - 1);
  bf16x8 qf[8];
  {
    const unsigned short* qp = A.q + (size_t)qrow_c * A.ldq + head * HD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }
  // retire the Q loads with a wait the compiler can see, or it guards every use of qf in the loop with vmcnt waits
  // that (sharing the counter with the asm-issued DMA) drain the prefetch
  __builtin_amdgcn_s_waitcnt(0x0F70);

  // ---- LDS-DMA map: one wave instruction = 1 KiB = 4 key rows x 256 B; lane -> row (lane >> 4) of the
  //      group, PHYSICAL 16-byte chunk pc = lane & 15.  K: physical chunk pc holds logical chunk
  //      pc ^ (row & 15);  V: 64-byte chunk (pc >> 2) holds logical 64-byte chunk (pc >> 2) ^ (row & 3).
  const int d_row = lane >> 4, d_pc = lane & 15;
  int k_src_c[4], v_src_c[4], d_rowi[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = (r * 4 + wave) * 4 + d_row;           // 0..63 within the tile
    d_rowi[r] = row;
    k_src_c[r] = (d_pc ^ (row & 15)) * 8;                  // element offset inside the 128-element head row
    v_src_c[r] = ((((d_pc >> 2) ^ (row & 3)) << 2) | (d_pc & 3)) * 8;
  }
  const unsigned short* kbase = A.k + kvh * HD;
  const unsigned short* vbase = A.v + kvh * HD;
  const int last_key = A.kv_len - 1;
  const unsigned lds0 = (unsigned)(unsigned long long)(attn_lds_ptr_t)smem + wave * 1024;
  // issue<CLAMP=false>: steady-state tiles (all 64 keys valid) use one 64-bit add per piece
  auto issue = [&](int t, int buf, auto clamp_tag) {
    constexpr bool CLAMP = decltype(clamp_tag)::value;
    const unsigned kb = lds0 + buf * 32768;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int key = A.kv_start + t * KT + d_rowi[r];
      if (CLAMP) key = min(key, last_key);                                 // ragged tail: re-read a valid row
      const size_t off = (size_t)(PAGED ? A.ka.slot(key) : key) * kv_stride;
      attn_dma16(kbase + off + k_src_c[r], kb + r * 4096);
      attn_dma16(vbase + off + v_src_c[r], kb + 16384 + r * 4096);
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
  const int kswz = l31 & 15;
  const int vi = lane & 15, vg1 = (lane >> 4) & 1;
  const int v_rowq = vi >> 2;                                  // == (row & 3) since key0 % 4 == 0
  const int v_in = (vg1 << 5) | ((vi & 3) << 3);               // byte offset inside the 64-byte chunk

  const int nkeys = A.kv_len - A.kv_start;
  const int NT = (nkeys + KT - 1) / KT;

  // ------------------------------------------------------------------------------------------------
  // One 64-key tile = two 32-key blocks, software pipelined so that in every stage the wave has MFMAs
  // AND independent VALU / LDS work to issue in their shadow:
  //   stage A : S0 = K0 Q^T                       (K fragment batches double buffered: reads of batch i+1
  //                                                 are in flight while the MFMAs of batch i run)
  //   stage B : S1 = K1 Q^T   ||  softmax(block 0) + V^T fragment reads of block 0
  //   stage C : O += V0^T P0^T || softmax(block 1) + V^T fragment reads of block 1
  //   stage D : O += V1^T P1^T
  // Online softmax runs per 32-key block; O is rescaled (in place) only when some row's max grew.
  auto tile = [&](int t, int buf, auto ragged_tag) {
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    const unsigned char* kb = smem + buf * 32768;
    const unsigned char* vb = kb + 16384;
    bf16x8 kA[4], kB[4], vA[4], vB[4];
    auto ldK = [&](bf16x8(&f)[4], int b, int kh) {
      const unsigned char* krow = kb + (32 * b + l31) * 256;
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4)
        f[k4] = *reinterpret_cast<const bf16x8*>(krow + (((2 * (4 * kh + k4) + hi) ^ kswz) << 4));
    };
    auto mmaK = [&](f32x16& acc, bf16x8(&f)[4], int kh) {
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[k4], qf[4 * kh + k4], acc, 0, 0, 0);
    };
    auto ldV = [&](bf16x8(&f)[4], int b, int s2) {
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
    auto mmaV = [&](bf16x8(&f)[4], const bf16x8& p) {
#pragma unroll
      for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[d], p, o[d], 0, 0, 0);
    };
    // softmax of one 32-key block against the LAZY reference maximum m_run (see ifx_attn_pp.hip): the true row
    // maximum is taken for the very first block only; afterwards p = exp2(c2*(s - m_run)) is formed directly and only
    // the block's row sums are checked — a row that outgrew the reference by 2^20 (or overflowed) makes the wave redo
    // the block against its true maximum.  Returns the factor O and l must be rescaled by (1 in the common case).
    auto block_max = [&](f32x16& sb) -> float {
      // the FIRST read of the S accumulators is a compiler-visible instruction: hipcc inserts the MFMA -> VALU wait
      // states for it, which it does not do for operands of inline asm
      float mx = __builtin_fmaxf(sb[0], sb[1]);
      mx = max3f(mx, sb[2], m_run);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = max3f(mx, sb[r], sb[r + 1]);
      mx = max3f(mx, sb[15], mx);
      return half_swap_max(mx);                        // >= m_run, identical in lane and lane^32
    };
    auto exp_block = [&](f32x16& sb, bf16x8(&pb)[2]) -> float {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const float mc = m_run * c2;
      const f32x2 c2v = {c2, c2}, mcv = {mc, mc};
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x2 sv = {sb[2 * i], sb[2 * i + 1]};
        const f32x2 e = sv * c2v - mcv;
        const float p0 = __builtin_amdgcn_exp2f(e[0]), p1 = __builtin_amdgcn_exp2f(e[1]);
        a0 += p0;
        a1 += p1;
        pb[i >> 2][(2 * i) & 7] = static_cast<__bf16>(p0);
        pb[i >> 2][(2 * i + 1) & 7] = static_cast<__bf16>(p1);
      }
      return a0 + a1;
    };
    auto softmax_block = [&](f32x16& sb, int b, bf16x8(&pb)[2]) -> float {
      if (RAGGED) {
        const int kidx = t * KT + 32 * b + 4 * hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kk = kidx + (r & 3) + 8 * (r >> 2);
          if (kk >= nkeys) sb[r] = -INFINITY;
          else if (kk == nkeys - 1) sb[r] += A.last_key_bias;     // a key that stands for several identical ones (0 otherwise)
        }
      }
      if (t == 0 && b == 0) m_run = block_max(sb);       // O and l are still zero: nothing to rescale
      float ps = exp_block(sb, pb);
      float alpha = 1.0f;
      if (__any(!(ps < 1048576.f))) {                    // rare: redo against the true maximum (inf / NaN land here)
        const float m_new = block_max(sb);
        alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
        m_run = m_new;
        ps = exp_block(sb, pb);
        l_run *= alpha;
      }
      l_run += ps;
      return alpha;
    };
    auto rescale_o = [&](float alpha) {
      asm volatile("" : "+v"(alpha));       // pin the (rare) branch at the stage boundary: the test must not
                                            // be hoisted into the stage and split its MFMA || VALU region
      if (__any(alpha != 1.0f)) {          // wave-uniform branch; in-place so the common path moves nothing
        // inline-asm operands are invisible to the hazard recogniser: pad the MFMA -> VALU read (O accumulators
        // of the PV MFMAs just issued) and the VALU write -> MFMA SrcC read by hand; the branch is rare
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(o[d][r]) : "v"(alpha));
        asm volatile("s_nop 3" ::: "memory");
      }
    };

    f32x16 s0, s1;
    bf16x8 p0[2], p1[2];
    // ---- stage A
    ldK(kA, 0, 0);
    ldK(kB, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[r] = 0.f;
    __builtin_amdgcn_s_setprio(1);
    mmaK(s0, kA, 0);
    ldK(kA, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mmaK(s0, kB, 1);
    ldK(kB, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage B: QK(block 1) || softmax(block 0) ; V fragments of block 0 fetched underneath
#pragma unroll
    for (int r = 0; r < 16; ++r) s1[r] = 0.f;
    mmaK(s1, kA, 0);
    ldV(vA, 0, 0);
    mmaK(s1, kB, 1);
    ldV(vB, 0, 1);
    const float a0 = softmax_block(s0, 0, p0);
    // shape the stage: each MFMA is followed by the VALU / LDS work that fits in its 32-cycle shadow
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
      __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);  // 10 VALU
    }
    __builtin_amdgcn_sched_barrier(0);
    rescale_o(a0);
    // ---- stage C: PV(block 0) || softmax(block 1) ; V fragments of block 1
    mmaV(vA, p0[0]);
    ldV(vA, 1, 0);
    mmaV(vB, p0[1]);
    ldV(vB, 1, 1);
    const float a1 = softmax_block(s1, 1, p1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
      __builtin_amdgcn_sched_group_barrier(0x002, 10, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    rescale_o(a1);
    // ---- stage D
    mmaV(vA, p1[0]);
    mmaV(vB, p1[1]);
    __builtin_amdgcn_s_setprio(0);
  };

  issue(0, 0, std::true_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int t = 0; t < NT - 1; ++t) {
    const int buf = t & 1;
    if (t + 2 < NT) issue(t + 1, buf ^ 1, std::false_type{});   // slot buf^1 was last read in iteration t-1
    else issue(t + 1, buf ^ 1, std::true_type{});               // the last tile may be ragged
    tile(t, buf, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if ((nkeys & (KT - 1)) || A.last_key_bias != 0.f) tile(NT - 1, (NT - 1) & 1, std::true_type{});
  else tile(NT - 1, (NT - 1) & 1, std::false_type{});

  // ---------------- epilogue ----------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (qrow < A.q_rows) {
    unsigned short* op = A.out + (size_t)qrow * A.ldo + head * HD + 4 * hi;
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

int launch_attn_pp(const unsigned short* q, unsigned short* out, float* lse, const ifx_kv_view* kv, int q_rows,
                   int heads, int kv_start, int kv_len, float scale, int splits, void* workspace, int groups,
                   hipStream_t stream, int slot_base = 0, int slot_cap = 0, int* slots_used = nullptr, int ldq = 0, int ldo = 0,
                   int n_ranges = 0, const int* q_ranges = nullptr, const int* k_ranges = nullptr);
int launch_attn_merge(const float* workspace, int slot_cap, int slots_used, unsigned short* out, float* lse, int q_rows,
                      int heads, hipStream_t stream, int ldo = 0);
size_t attn_pp_workspace_bytes(int q_rows, int heads, int splits);
int attn_pp_split_heuristic(int q_rows, int heads, int nkeys, int qt, int slots);
// schedule of the multi-wave kernel for a launch (the `groups` code of launch_attn_pp): 2 phase-locked ping-pong, 3 three groups
// (384-row tiles), 4 free-running, 5 software-pipelined, 6 software-pipelined in four-wave workgroups (128-row tiles, two per CU),
// 7 software-pipelined, unrolled four times over constant LDS slots (what auto takes for 256-row tiles).
// Auto (variant 0): 7, except for launches whose rows fill 128-row tiles markedly better than 256-row tiles — one rank's 585 rows
// of an eight-way sequence-parallel shard: 5 x 128 (91 %) against 3 x 256 (76 %); one rank's clip 348 -> 332 ms
static int attn_groups(int variant, int q_rows, int heads = 0) {
  if (variant == 3 || variant == 4 || variant == 2 || variant == 5 || variant == 6 || variant == 7) return variant;
  if (variant == 0 && q_rows > 0) {
    const int t256 = (q_rows + 255) / 256, t128 = (q_rows + 127) / 128;
    const float u256 = (float)q_rows / (256.f * t256), u128 = (float)q_rows / (128.f * t128);
    if (u128 > 1.1f * u256) return 6;
    // Launches of more than one round of workgroups: whole rounds are what costs.  256-row tiles run one per CU (256 slots), the
    // 128-row four-wave tiles two per CU (512 slots) at 0.953 of the rate (1061 vs 1113 TFLOP/s).  CausVid 720p: 10800
    // rows x 12 heads = 516 tiles = 2.02 rounds -> THREE rounds of 256-row tiles, but 1020 / 512 = 1.99 -> two of 128-row tiles
    // (measured 778 -> see profiles/r2_*); the 480p block (228 tiles, one round either way) stays on the 256-row schedule.
    if (heads > 0 && t256 * heads > 256) {
      const float c5 = (float)((t256 * heads + 255) / 256);
      const float c6 = (float)((t128 * heads + 511) / 512) / 0.953f;       // 1061 vs 1113 TFLOP/s at L = 32760 (both with constant LDS slots)
      if (c6 < 0.95f * c5) return 6;
    }
  }
  return 7;
}
static int attn_qt(int groups) { return groups == 3 ? 384 : (groups == 6 ? 128 : 256); }
static int attn_slots(int groups) { return groups == 6 ? 512 : 256; }

}  // namespace ifx

using namespace ifx;

static int attn_dispatch(const ifx_bf16* q, ifx_bf16* out, float* lse, const ifx_kv_view* kv, int32_t q_rows,
                         int32_t heads, int32_t kv_start, int32_t kv_len, float scale, int32_t splits, void* workspace,
                         int64_t workspace_bytes, void* stream, int32_t ldq = 0, int32_t ldo = 0, int32_t last_key_multiplicity = 1) {
  IFX_REQUIRE(q && out && kv && kv->k && kv->v, "ifx_attn_fwd_paged: null argument");
  if (ldq <= 0) ldq = heads * HD;
  if (ldo <= 0) ldo = heads * HD;
  IFX_REQUIRE(ldq >= heads * HD && ldo >= heads * HD && ldq % 8 == 0 && ldo % 8 == 0,
              "ifx_attn_fwd_paged: row strides (%d, %d) must be >= heads * 128 and multiples of 8", ldq, ldo);
  IFX_REQUIRE(kv->head_dim == HD, "ifx_attn_fwd_paged: head_dim %d not built (128 only)", kv->head_dim);
  IFX_REQUIRE(heads > 0 && kv->kv_heads > 0 && heads % kv->kv_heads == 0,
              "ifx_attn_fwd_paged: heads %d is not a multiple of kv_heads %d", heads, kv->kv_heads);
  IFX_REQUIRE(q_rows >= 0 && kv_start >= 0 && kv_len > kv_start && kv_len <= kv->num_slots,
              "ifx_attn_fwd_paged: key range [%d, %d) out of range (capacity %d)", kv_start, kv_len, kv->num_slots);
  if (kv->page_table) IFX_REQUIRE(kv->page_size > 0, "ifx_attn_fwd_paged: page_size must be > 0");
  if (q_rows == 0) return IFX_OK;
  const int variant = attn_variant();
  if (last_key_multiplicity > 1) {
    IFX_REQUIRE(splits <= 1 && kv_len - kv_start <= 1024 && lse == nullptr,
                "ifx_attn_fwd_dedup: the multiplicity form is built for short key ranges (<= 1024 keys), unsplit, without LSE");
  }
  if (splits > 1) {
    IFX_REQUIRE(workspace && workspace_bytes >= (int64_t)attn_pp_workspace_bytes(q_rows, heads, splits),
                "ifx_attn_fwd_paged_split: workspace of %lld B too small for %d splits (need %lld B)",
                (long long)workspace_bytes, splits, (long long)attn_pp_workspace_bytes(q_rows, heads, splits));
    return launch_attn_pp(q, out, lse, kv, q_rows, heads, kv_start, kv_len, scale, splits, workspace, attn_groups(variant, q_rows, heads),
                          (hipStream_t)stream, 0, 0, nullptr, ldq, ldo);
  }
  if (last_key_multiplicity <= 1 && (variant >= 2 || (variant == 0 && q_rows >= 1024 && kv_len - kv_start > 1024)))
    return launch_attn_pp(q, out, lse, kv, q_rows, heads, kv_start, kv_len, scale, 1, nullptr, attn_groups(variant, q_rows, heads),
                          (hipStream_t)stream, 0, 0, nullptr, ldq, ldo);
  AttnArgs a;
  a.q = q;
  a.out = out;
  a.lse = lse;
  a.k = kv->k;
  a.v = kv->v;
  a.ka = KvAddr{kv->page_table, kv->page_size, kv->page_table ? 0 : kv->seg_split, kv->page_table ? 0 : kv->seg_delta};
  a.q_rows = q_rows;
  a.heads = heads;
  a.ldq = ldq;
  a.ldo = ldo;
  a.kv_start = kv_start;
  a.kv_len = kv_len;
  a.kv_heads = kv->kv_heads;
  a.q_per_kv = heads / kv->kv_heads;
  a.q_tiles = (q_rows + QT - 1) / QT;
  a.total = a.q_tiles * heads;
  a.per_xcd = (a.total + 7) / 8;
  a.scale = scale > 0.f ? scale : 0.08838834764831845f;   // 1/sqrt(128)
  a.scale_log2 = a.scale * 1.4426950408889634f;
  a.last_key_bias = last_key_multiplicity > 1 ? logf((float)last_key_multiplicity) / a.scale : 0.f;

  const dim3 grid(a.per_xcd * 8), block(256);
  const bool short_kv = kv_len - kv_start <= 1024;
  if (kv->page_table || kv->seg_split > 0) {
    if (short_kv) hipLaunchKernelGGL((attn_fwd_kernel<true, true>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<true, false>), grid, block, 0, (hipStream_t)stream, a);
  } else {
    if (short_kv) hipLaunchKernelGGL((attn_fwd_kernel<false, true>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, false>), grid, block, 0, (hipStream_t)stream, a);
  }
  return check_launch("ifx_attn_fwd_paged");
}

extern "C" int ifx_attn_fwd_paged(const ifx_bf16* q, ifx_bf16* out, float* lse, const ifx_kv_view* kv,
                                  int32_t q_rows, int32_t heads, int32_t kv_start, int32_t kv_len, float scale,
                                  void* stream) {
  return attn_dispatch(q, out, lse, kv, q_rows, heads, kv_start, kv_len, scale, 1, nullptr, 0, stream);
}

extern "C" int ifx_attn_fwd_paged_ld(const ifx_bf16* q, int32_t ldq, ifx_bf16* out, int32_t ldo, float* lse, const ifx_kv_view* kv,
                                     int32_t q_rows, int32_t heads, int32_t kv_start, int32_t kv_len, float scale,
                                     int32_t num_splits, void* workspace, int64_t workspace_bytes, void* stream) {
  IFX_REQUIRE(num_splits >= 1 && num_splits <= 64, "ifx_attn_fwd_paged_ld: num_splits %d outside [1, 64]", num_splits);
  IFX_REQUIRE(ldq > 0 && ldo > 0, "ifx_attn_fwd_paged_ld: row strides must be given");
  return attn_dispatch(q, out, lse, kv, q_rows, heads, kv_start, kv_len, scale, num_splits, workspace, workspace_bytes, stream,
                       ldq, ldo);
}

extern "C" int ifx_attn_fwd_dedup(const ifx_bf16* q, ifx_bf16* out, const ifx_kv_view* kv, int32_t q_rows, int32_t heads,
                                  int32_t kv_len, int32_t last_key_multiplicity, float scale, void* stream) {
  IFX_REQUIRE(last_key_multiplicity >= 1, "ifx_attn_fwd_dedup: multiplicity %d", last_key_multiplicity);
  return attn_dispatch(q, out, nullptr, kv, q_rows, heads, 0, kv_len, scale, 1, nullptr, 0, stream, 0, 0, last_key_multiplicity);
}

extern "C" int ifx_attn_fwd_ranges(const ifx_bf16* q, int32_t ldq, ifx_bf16* out, int32_t ldo, const ifx_kv_view* kv, int32_t q_rows,
                                   int32_t heads, int32_t n_ranges, const int32_t* q_ranges, const int32_t* k_ranges, float scale,
                                   void* stream) {
  IFX_REQUIRE(q && out && kv && kv->k && kv->v && q_ranges && k_ranges, "ifx_attn_fwd_ranges: null argument");
  IFX_REQUIRE(kv->head_dim == HD, "ifx_attn_fwd_ranges: head_dim %d not built (128 only)", kv->head_dim);
  IFX_REQUIRE(heads > 0 && kv->kv_heads > 0 && heads % kv->kv_heads == 0, "ifx_attn_fwd_ranges: heads %d / kv_heads %d", heads,
              kv->kv_heads);
  IFX_REQUIRE(n_ranges >= 1 && n_ranges <= 8, "ifx_attn_fwd_ranges: 1..8 ranges per launch (got %d)", n_ranges);
  if (ldq <= 0) ldq = heads * HD;
  if (ldo <= 0) ldo = heads * HD;
  IFX_REQUIRE(ldq >= heads * HD && ldo >= heads * HD && ldq % 8 == 0 && ldo % 8 == 0, "ifx_attn_fwd_ranges: row strides (%d, %d)", ldq, ldo);
  if (kv->page_table) IFX_REQUIRE(kv->page_size > 0, "ifx_attn_fwd_ranges: page_size must be > 0");
  int kmin = 0x7fffffff, kmax = 0, rows = 0;
  for (int i = 0; i < n_ranges; ++i) {
    const int q0 = q_ranges[2 * i], q1 = q_ranges[2 * i + 1], k0 = k_ranges[2 * i], k1 = k_ranges[2 * i + 1];
    IFX_REQUIRE(q0 >= 0 && q1 > q0 && q1 <= q_rows, "ifx_attn_fwd_ranges: query range %d = [%d, %d) outside [0, %d)", i, q0, q1, q_rows);
    IFX_REQUIRE(k0 >= 0 && k1 > k0 && k1 <= kv->num_slots, "ifx_attn_fwd_ranges: key range %d = [%d, %d) out of range (capacity %d)",
                i, k0, k1, kv->num_slots);
    kmin = k0 < kmin ? k0 : kmin;
    kmax = k1 > kmax ? k1 : kmax;
    rows += q1 - q0;
  }
  const int groups = attn_groups(attn_variant(), rows / n_ranges, heads * n_ranges);   // every range tiles separately
  return launch_attn_pp(q, out, nullptr, kv, q_rows, heads, kmin, kmax, scale, 1, nullptr, groups == 5 || groups == 6 || groups == 7 ? groups : 7,
                        (hipStream_t)stream, 0, 0, nullptr, ldq, ldo, n_ranges, q_ranges, k_ranges);
}

extern "C" int32_t ifx_attn_split_plan(int32_t q_rows, int32_t heads, int32_t kv_start, int32_t kv_len,
                                       int64_t* workspace_bytes) {
  int splits = 1;
  if (q_rows > 0 && heads > 0 && kv_len > kv_start)
    splits = attn_pp_split_heuristic(q_rows, heads, kv_len - kv_start, attn_qt(attn_groups(attn_variant(), q_rows, heads)),
                                     attn_slots(attn_groups(attn_variant(), q_rows, heads)));
  if (workspace_bytes) *workspace_bytes = (int64_t)attn_pp_workspace_bytes(q_rows, heads, splits);
  return splits;
}

extern "C" int ifx_attn_fwd_paged_split(const ifx_bf16* q, ifx_bf16* out, float* lse, const ifx_kv_view* kv,
                                        int32_t q_rows, int32_t heads, int32_t kv_start, int32_t kv_len, float scale,
                                        int32_t num_splits, void* workspace, int64_t workspace_bytes, void* stream) {
  IFX_REQUIRE(num_splits >= 1 && num_splits <= 64, "ifx_attn_fwd_paged_split: num_splits %d outside [1, 64]", num_splits);
  return attn_dispatch(q, out, lse, kv, q_rows, heads, kv_start, kv_len, scale, num_splits, workspace, workspace_bytes,
                       stream);
}

extern "C" int ifx_attn_fwd_partial(const ifx_bf16* q, const ifx_kv_view* kv, int32_t q_rows, int32_t heads,
                                    int32_t kv_start, int32_t kv_len, float scale, int32_t num_splits, void* workspace,
                                    int64_t workspace_bytes, int32_t slot_base, int32_t slot_cap, int32_t* slots_used,
                                    void* stream) {
  IFX_REQUIRE(q && kv && kv->k && kv->v && workspace, "ifx_attn_fwd_partial: null argument");
  IFX_REQUIRE(kv->head_dim == HD, "ifx_attn_fwd_partial: head_dim %d not built (128 only)", kv->head_dim);
  IFX_REQUIRE(heads > 0 && kv->kv_heads > 0 && heads % kv->kv_heads == 0,
              "ifx_attn_fwd_partial: heads %d is not a multiple of kv_heads %d", heads, kv->kv_heads);
  IFX_REQUIRE(q_rows > 0 && kv_start >= 0 && kv_len > kv_start && kv_len <= kv->num_slots,
              "ifx_attn_fwd_partial: key range [%d, %d) out of range (capacity %d)", kv_start, kv_len, kv->num_slots);
  IFX_REQUIRE(num_splits >= 1 && slot_base >= 0 && slot_cap >= 1 && slot_cap <= 128,
              "ifx_attn_fwd_partial: bad split / slot arguments (%d splits, base %d, cap %d)", num_splits, slot_base, slot_cap);
  IFX_REQUIRE(workspace_bytes >= (int64_t)slot_cap * q_rows * heads * 129 * (int64_t)sizeof(float),
              "ifx_attn_fwd_partial: workspace of %lld B too small for %d slots", (long long)workspace_bytes, slot_cap);
  if (kv->page_table) IFX_REQUIRE(kv->page_size > 0, "ifx_attn_fwd_partial: page_size must be > 0");
  return launch_attn_pp(q, nullptr, nullptr, kv, q_rows, heads, kv_start, kv_len, scale, num_splits, workspace,
                        attn_groups(attn_variant(), q_rows, heads), (hipStream_t)stream, slot_base, slot_cap, slots_used);
}

extern "C" int ifx_attn_merge_partials(const void* workspace, int32_t slot_cap, int32_t slots_used, ifx_bf16* out,
                                       float* lse, int32_t q_rows, int32_t heads, void* stream) {
  IFX_REQUIRE(workspace && out && q_rows > 0 && heads > 0 && slots_used >= 1 && slots_used <= slot_cap,
              "ifx_attn_merge_partials: bad arguments (%d of %d slots)", slots_used, slot_cap);
  return launch_attn_merge((const float*)workspace, slot_cap, slots_used, out, lse, q_rows, heads, (hipStream_t)stream);
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
