// Block-causal flash-attention forward, PING-PONG structure (gfx950, head_dim 128).
//
// Why: profiles of the 4-wave kernel (ifx_attn.hip) show MFMA utilisation 34-40 % although its loop is lean
// (32 MFMA + ~200 VALU + 48 LDS reads per 64-key tile per wave): per tile a wave needs ~1024 cycles of matrix
// pipe and ~1000 cycles of VALU issue, the two waves that share a SIMD belong to unrelated workgroups, and
// nothing makes one of them sit in its VALU part while the other owns the matrix pipe.
//
// Here the two waves of a SIMD are in the SAME 8-wave workgroup and alternate roles under workgroup barriers:
//
//   step M(t) : O^T += V(t-1)^T P(t-1)^T   then   S(t)^T = K(t) Q^T        32 MFMAs, 48 LDS fragment reads,
//                                                                           DMA issue of tile t+2 in the shadows
//   step V(t) : online softmax of S(t) -> P(t) (bf16), running max / sum, conditional in-place rescale of O
//
//   barrier index :   2t            2t+1          2t+2          2t+3
//   waves 0-3 (G0):   M(t)          V(t)          M(t+1)        V(t+1)
//   waves 4-7 (G1):   V(t-1)        M(t)          V(t)          M(t+1)
//
// so in every phase each SIMD has exactly one wave feeding the matrix pipe and one wave doing VALU work.
// K/V tiles (64 keys) arrive by LDS-DMA into rings (K: 3 slots, V: 4 slots, 16 KiB each = 112 KiB): tile t is
// issued two tiles ahead; one counted s_waitcnt vmcnt(4) per wave per tile retires exactly the tile that the
// next barrier publishes.  Layouts, swizzles, MFMA operand mapping and the k-slot relabelling that feeds P
// straight from the S accumulators are those of ifx_attn.hip.
#include <stdlib.h>
#include <type_traits>

#include "ifx_common.h"

// compile-time ablation for bottleneck studies (tools/ablate_attn.sh): 1 no DMA, 2 no softmax step,
// 4 no LDS fragment reads, 8 no MFMA.  0 in the shipped library.
#ifndef PP_ABLATE
#define PP_ABLATE 0
#endif
// PP_TRACE=1 (tools/trace_attn.sh): workgroup 0 stamps the cycle counter at the step boundaries of its first 64 tiles
// into the LSE buffer (as int64 [tile][wave][4]: M start, M end, V start, V end).  0 in the shipped library.
#ifndef PP_TRACE
#define PP_TRACE 0
#endif
#ifndef PP_PACKED
#define PP_PACKED 1    // 1: v_pk_fma_f32 for the exponent arguments (measured +5 % over plain v_fma_f32), 0: plain
#endif
#ifndef PP_PRIO
#define PP_PRIO 2      // 1: raised priority around the MFMA step, 2: around the softmax step (measured best)
#endif
#ifndef PP_RING
#define PP_RING 6      // fragment-ring slots of the MFMA steps with two waves per SIMD (reads PP_RING - 1 batches ahead)
#endif
// software-pipelined schedule (FR = 2) knobs
#ifndef PP_SWP_RING
#define PP_SWP_RING 4
#endif
#ifndef PP_SWP_BALANCE
#define PP_SWP_BALANCE 1
#endif
#ifndef PP_SWP_YIELD
#define PP_SWP_YIELD 0
#endif
#ifndef PP_SWP_PACKED
#define PP_SWP_PACKED 0
#endif
#ifndef PP_SWP_RECOMPUTE
#define PP_SWP_RECOMPUTE 1   // 1: lane-derived LDS address terms recomputed every tile; 0 keeps them live: 256 VGPRs + 170 B of spills, 1078 -> 783 TFLOP/s
#endif
#ifndef PP_GROUP
#define PP_GROUP 0     // 0: groups = waves 0-3 / 4-7 (waves w, w+4 share a SIMD); 1: even / odd waves
#endif

namespace ifx {

namespace pp {
constexpr int KT = 64;
constexpr int HD = 128;
#ifndef PP_PD
#define PP_PD 2        // DMA prefetch distance in tiles (issued from the softmax step); 3 measured slower
#endif
constexpr int PD = PP_PD;
constexpr int RK = PD + 1, RV = PD + 2;
constexpr int K_OFF = 0;
constexpr int LDS_BYTES = (RK + RV) * 16384;   // 114688
constexpr int LDS_SWP = 8 * 16384;              // software-pipelined schedule (FR = 2): 3 K tiles + 5 V tiles
constexpr int LDS_ALLOC = PP_TRACE ? LDS_SWP + 24576 : (LDS_BYTES > LDS_SWP ? LDS_BYTES : LDS_SWP);   // trace: 24 KiB of stamps behind the rings
}  // namespace pp

struct AttnArgsPP {
  const unsigned short* q;
  unsigned short* out;
  float* lse;
  const unsigned short* k;
  const unsigned short* v;
  KvAddr ka;
  int q_rows, heads, kv_start, kv_len, num_slots, q_tiles, per_xcd, total;
  int ldq, ldo;             // elements between consecutive rows of q / out (heads * 128 unless the caller strides them)
  float scale, scale_log2;
  // split-KV (SPLIT kernels only): `splits` key chunks of `chunk_tiles` 64-key tiles each; chunk sp of (head, q tile)
  // writes a normalised fp32 partial O to part_o[sp][row][head][128] and its LSE to part_lse[sp][head][row]
  int splits, chunk_tiles;
  int kv_heads, q_per_kv;   // grouped-query attention: query head h reads kv head h / q_per_kv
  float* part_o;
  float* part_lse;
  // multi-range launch (n_ranges > 0, unsplit kernels only): query rows [rq0[r], rq1[r]) attend keys [rk0[r], rk1[r]); q tile ids
  // [rt0[r], rt0[r + 1]) of a head belong to range r (MAGI: the denoising chunks of one forward in ONE launch)
  int n_ranges;
  int rq0[8], rq1[8], rk0[8], rk1[8], rt0[9];
  // paged views (PAGED = 1 instantiations): key -> page by a multiply-high with `ps_magic` = floor(2^32 / page_size) + 1 (exact while
  // key * page_size < 2^32), on the SCALAR side, once per page (see the request lambda).  History: round 5 divided and loaded the table
  // entry per LANE in front of every K/V request (0.54x the contiguous kernel over 32760 keys), early round 6 read an LDS copy of the
  // table per lane (0.86x: the reads and their waits sit in the software-pipelined loop, the kernel was twice the contiguous one's size).
  unsigned ps_magic;
  // tests (option "attn_debug_counters"): a device word that counts the (wave, tile) pairs that took the rescale branch of the lazy
  // row maximum; nullptr in every normal launch (the increment sits inside the rare branch only)
  unsigned* dbg_rescales;
};

typedef __attribute__((address_space(3))) void* pp_lds_ptr_t;
typedef int pp_v4i __attribute__((ext_vector_type(4)));
typedef float pp_f32x2 __attribute__((ext_vector_type(2)));

// One LDS-DMA instruction (64 lanes x 16 B, lane-linear at LDS byte address `lds`), issued from inline asm ON PURPOSE:
// hipcc tracks the LDS-DMA builtins as LDS stores and puts `s_waitcnt vmcnt(0)` in front of the next ds_read that
// might alias them — here every fragment read of the MFMA step — which drains the whole prefetch queue once per tile.
// The ring discipline below (counted vmcnt + workgroup barrier before a slot is read) is what orders DMA and reads.
__device__ __forceinline__ void pp_dma16(pp_v4i rsrc, unsigned lds, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");      // m0 is a reserved register: hipcc sets it itself right before each of its own uses
}
__device__ __forceinline__ pp_v4i pp_make_rsrc(const void* base, unsigned num_bytes) {
  const unsigned long long a = (unsigned long long)base;
  pp_v4i r;
  r[0] = (int)(unsigned)a;
  r[1] = (int)((unsigned)(a >> 32) & 0xffffu);      // stride 0
  r[2] = (int)num_bytes;
  r[3] = 0x00020000;
  return r;
}
typedef const __attribute__((address_space(1))) void* pp_gbl_ptr_t;

__device__ __forceinline__ float pp_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float pp_half_max(float x) {   // max over lane and lane^32 (see ifx_attn.hip)
  float a = x, b = x, r;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %2, %0, %1" : "+v"(a), "+v"(b), "=v"(r));
  return r;
}

// wait until at most `tiles` DMA tile-groups (4 pieces each) of this wave are still in flight
__device__ __forceinline__ void pp_wait_tiles(int tiles) {
  if (tiles <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (tiles == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (tiles == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (tiles == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}

// NG = number of wave groups (4 waves each, one per SIMD): 2 = ping-pong (M | V), 3 = three-phase (M | V1 | V2).
// Why three: while one wave streams MFMAs, a SINGLE other wave on that SIMD gets a VALU instruction issued only every
// ~15 cycles, TWO other waves get one every ~8 (tools/probe_overlap.hip).  The softmax of a tile is ~110 VALU
// instructions against 32 MFMAs (1024 cycles): with one softmax wave per SIMD the step takes ~2000 cycles and the
// matrix pipe idles half the time (step trace, tools/trace_attn.sh); with two it fits in two MFMA steps.
// FR = 1: free-running schedule (attn_variant 4): every wave runs QK(t) -> softmax(t) -> PV(t) for its own 32 queries with ONE
// workgroup barrier per tile and no phase assignment — the two waves of a SIMD drift apart by themselves, the older one takes
// the matrix pipe first and its softmax then overlaps with the younger wave's MFMAs (tools/probe_roles.hip).
// PAGED = 1: the page (or segment) that holds key `kk` -> its key range [lo, hi) and the row offset physical - logical.  All scalar;
// the table entry comes through the scalar cache (s_load_dword from inline asm with its own wait: a compiler-visible load would
// put lgkmcnt / vmcnt waits for it into the software-pipelined loop around the call).
__device__ __forceinline__ void pp_page_range(const KvAddr& ka, unsigned ps_magic, int kk, int& lo, int& hi, int& off) {
  if (ka.pt == nullptr) {
    const bool up = ka.seg_split > 0 && kk >= ka.seg_split;
    lo = up ? ka.seg_split : 0;
    hi = (up || ka.seg_split <= 0) ? 0x7fffffff : ka.seg_split;
    off = up ? ka.seg_delta : 0;
    return;
  }
  const int pg = (int)__umulhi((unsigned)kk, ps_magic);
  int entry;
  asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(entry) : "s"(ka.pt), "s"(pg * 4) : "memory");
  lo = pg * ka.ps;
  hi = lo + ka.ps;
  off = entry * ka.ps - lo;
}

// PAGED: 0 = contiguous cache (logical row = physical row); 1 = page table with pages of at least 3 rows, or a two-segment view —
// wave-uniform translation (a 4-key request piece spans at most two pages); 2 = one- and two-row pages, tables whose multiply-high
// page index would not be exact — per-lane translation (KvAddr::slot: a division and a table load per request), instantiated for the
// plain two-group schedule only (the launcher routes such views there).
template <int PAGED, bool SPLIT, int NG, int FR = 0>
__global__ __launch_bounds__(NG * 256, NG == 1 ? 2 : 1) void attn_fwd_pp_kernel(AttnArgsPP A) {
  using namespace pp;
  constexpr int QT = 128 * NG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // DMA prefetch distance and LDS rings of this schedule (FR = 2 reads K one tile ahead and V one tile behind)
  // FR = 3 (attn_variant 6): the software-pipelined loop in FOUR-wave workgroups of 128 queries, TWO of them per CU (80 KiB of
  // LDS each: K ring 2, V ring 3).  The two waves of a SIMD then belong to different workgroups: no barrier couples them, so
  // the older wave no longer waits ~600 cycles per tile for the younger one; the price is that each workgroup streams K/V itself.
  // FR = 5 (attn_variant 7): the software-pipelined loop unrolled FOUR times over rings of 4 K + 4 V tiles, so that every LDS slot is
  // a compile-time constant: a fragment read is `ds_read v_term offset:imm` with 12 lane terms computed once per kernel, where the
  // two-times-unrolled loop re-derives its addresses every tile (56 of the 177 non-MFMA VALU instructions of a tile; the loop is
  // bound by VALU issue).  V ring first (imm offsets reach 64 KiB), K ring behind it; K is requested three tiles ahead,
  // V two (it is consumed two iterations later), which is what lets four V slots do.
  // FR = 6: the same treatment for the two-per-CU form (rings of 2 K + 3 V tiles: unrolled six times).
  constexpr bool SWP = FR >= 2, DUAL = FR == 3 || FR == 6 || FR == 8, U4 = FR == 5 || FR == 7, U6 = FR == 6 || FR == 8, CS = U4 || U6;   // CS: constant LDS slots
  // PRE (FR = 7): the caller's scale * log2(e) is exactly 1 (q was scaled where it was produced): scores ARE exponents, and the softmax
  // reference -m enters as the C operand of a score block's first MFMA, so exp2 is applied straight to the accumulator
  constexpr bool PRE = FR == 7 || FR == 8;           // 8: the two-per-CU form (FR = 6) in its exponent form
  constexpr int PD = (FR == 2 || U4) ? 3 : pp::PD, RK = DUAL ? 2 : (U4 ? 4 : (FR == 2 ? 3 : pp::RK)),
                RV = DUAL ? 3 : (U4 ? 4 : (FR == 2 ? 5 : pp::RV));
  constexpr int K_OFF = CS ? RV * 16384 : 0, V_OFF = CS ? 0 : RK * 16384;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = (NG == 2 && PP_GROUP) ? (wave & 1) : (wave >> 2);
  const bool loader = wave < 8;          // 32 DMA pieces per tile = 8 waves x 4 (a third group only computes)
  const int hi = lane >> 5, l31 = lane & 31;

  const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
  // Multi-range launches hand tiles of DIFFERENT cost (key ranges of 2 .. 5 chunks) out in launch order, most expensive first: the
  // contiguous per-XCD blocks of the uniform case would give one XCD all the long tiles (measured: 780 -> 477 TFLOP/s on the MAGI
  // rank shape), and with grouped-query heads sharing one K/V stream there is no per-head L2 locality to protect.
  const int wi = (!SPLIT && A.n_ranges > 0) ? (int)blockIdx.x : xcd * A.per_xcd + slot_i;
  if ((A.n_ranges == 0 && slot_i >= A.per_xcd) || wi >= A.total) return;
  // work order (head, key chunk, q tile): the q tiles that stream the same K/V chunk sit on one XCD's L2
  int head, qt, sp = 0;
  if (SPLIT) {
    const int per_head = A.splits * A.q_tiles;
    head = wi / per_head;
    const int rem = wi - head * per_head;
    sp = rem / A.q_tiles;
    qt = rem - sp * A.q_tiles;
  } else if (A.n_ranges > 0) {
    qt = wi / A.heads;                     // tile-major: the longest range's tiles of every head first
    head = wi - qt * A.heads;
  } else {
    head = wi / A.q_tiles;
    qt = wi - head * A.q_tiles;
  }
  int kv_s = SPLIT ? A.kv_start + sp * A.chunk_tiles * KT : A.kv_start;
  int kv_e = SPLIT ? min(A.kv_len, kv_s + A.chunk_tiles * KT) : A.kv_len;
  int q_base = qt * QT, q_lim = A.q_rows;
  if (!SPLIT && A.n_ranges > 0) {
    int r = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
      if (i < A.n_ranges && qt >= A.rt0[i]) r = i;
    q_base = A.rq0[r] + (qt - A.rt0[r]) * QT;
    q_lim = A.rq1[r];
    kv_s = A.rk0[r];
    kv_e = A.rk1[r];
  }
  const int row_stride = A.heads * HD;

  const int qrow = q_base + wave * 32 + l31;
  const int qrow_c = min(qrow, q_lim - 1);
  bf16x8 qf[8];
  {
    const unsigned short* qp = A.q + (size_t)qrow_c * A.ldq + head * HD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }
  // Retire the Q loads HERE with a wait the compiler can see (vmcnt(0), expcnt/lgkmcnt untouched).  Otherwise it
  // carries them as "maybe pending" around the loop back-edge and protects every first use of qf in the loop with
  // vmcnt(7..0) — which, since the asm-issued DMA below shares the counter, drains the K/V prefetch queue every tile.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // ---- LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... lds): a tile is 16 K pieces + 16 V pieces of
  //      1 KiB (4 key rows x 256 B); wave w moves pieces w and w+8.  Per lane only a 32-bit voffset per piece is
  //      live (row * row_bytes + swizzled source chunk); the tile offset is a scalar soffset.  Rows beyond the
  //      descriptor's range (keys >= kv_len in the ragged last tile) read as ZERO by the hardware bounds check.
  const int row_bytes = A.kv_heads * HD * 2;     // cache row pitch
  const int kvh = head / A.q_per_kv;
  // piece r = 1 is 32 key rows below piece 0 (same swizzle phase): one voffset per matrix, +32 rows on the scalar side
  const int nkeys = kv_e - kv_s;
  const int NT = (nkeys + KT - 1) / KT;
  // identity page map: descriptor covers logical tokens [0, kv_len); paged: whole cache, keys clamped by hand
  const unsigned valid_rows = PAGED ? (unsigned)A.num_slots : (unsigned)kv_e;
  const unsigned nrec = (valid_rows - 1) * (unsigned)row_bytes + 256u;
  const pp_v4i krs = pp_make_rsrc(A.k + kvh * HD, nrec);
  const pp_v4i vrs = pp_make_rsrc(A.v + kvh * HD, nrec);
  const unsigned lds00 = (unsigned)(unsigned long long)(pp_lds_ptr_t)smem;
  const int last_key = kv_e - 1;
  int pc_lo = 0, pc_hi = 0, pc_off = 0;              // PAGED = 1: key range and (physical - logical) row offset of the page of the last request
  // The two lane offsets of a request (K and V rows of this wave's pieces: row * row_bytes + swizzled 16-byte chunk) in ONE register kept
  // across the loop: row_bytes is a multiple of 256, so bits 0-7 hold the K chunk, bits 8-27 the row term and bits 28-31 the V chunk index.
  // A call unpacks its offset with one (K) or three (V) VALU instructions instead of re-deriving it from the lane id with seven or eight —
  // the older wave of every SIMD issues eight requests per tile and the loop is bound by instruction issue.  Pieces of wave `wv` != wave
  // (wave + 4: sixteen rows on) have the same swizzle phase: the same lane offsets.
  int lane_pack;
  {
    const int ln0 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int pc0 = ln0 & 15, row0 = wave * 4 + (ln0 >> 4);          // the row term holds the lane's row INSIDE the piece (0 .. 3) only:
    lane_pack = (ln0 >> 4) * row_bytes + ((pc0 ^ (row0 & 15)) << 4) +   // the piece's first row is the scalar side's business
                (int)((unsigned)((((pc0 >> 2) ^ (row0 & 3)) << 2) | (pc0 & 3)) << 28);
  }
  auto issue_w = [&](int t, int wv, int what = 3) {  // the pieces that belong to wave `wv` (rows 4 wv .. 4 wv + 3 of each half); what: 1 K, 2 V
    // (opaque: hipcc otherwise precomputes every (ring slot, half, wave) destination address of the unrolled loop — 32 loop-invariant
    //  scalars, spilled to lanes and fetched with a v_readlane in front of each request; one s_add per request is cheaper in a loop
    //  that is bound by VALU issue)
    unsigned lds0 = lds00 + wv * 1024;
    asm volatile("" : "+s"(lds0));
    int ln = 0;                                     // lane id: only the boundary / per-lane paths of the paged forms use it
    if constexpr (PAGED != 0) {
      ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(ln));                  // recomputed, not kept live (see stepM)
    }
    int lp = lane_pack;
    asm volatile("" : "+v"(lp));                    // (unpacked at every call: two derived registers kept live would spill)
    const int k_lane = lp & 0x0fffffff;
    const int v_lane = (lp & 0x0fffff00) | (int)(((unsigned)lp >> 28) << 4);
    const unsigned kb = lds0 + K_OFF + (t % RK) * 16384;
    const unsigned vb = lds0 + V_OFF + (t % RV) * 16384;
    if (!PAGED) {
      int rb = row_bytes;
      asm volatile("" : "+s"(rb));                  // (opaque for the same reason: one s_mul per request instead of 16 spilled row offsets)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int soff = (kv_s + t * KT + wv * 4 + 32 * r) * rb;     // first row of the piece
        if (what & 1) pp_dma16(krs, kb + r * 8192, k_lane, soff);
        if (what & 2) pp_dma16(vrs, vb + r * 8192, v_lane, soff);
      }
    } else if constexpr (PAGED == 1) {
      // A piece is four CONSECUTIVE keys, nearly always inside ONE page: the translation is wave-uniform.  The physical row of
      // the piece's first key goes into the scalar offset, the lane offsets are the contiguous cache's; the page of the last request is
      // remembered in three scalars (a page is ~24 tiles) and looked up again — one scalar load from the table, inline asm so that the
      // compiler's wait counters never see it — only when a request leaves it.  The common path has no memory operation and two VALU
      // instructions per request.  Keys behind the last one read the last key's row (as the per-lane form did: finite values under
      // the mask) by a lane-row clamp, and a piece that straddles a page boundary adds the next page's offset to its upper lanes:
      // both in one scalar branch that only boundary pieces take.
      const int key00 = __builtin_amdgcn_readfirstlane(kv_s + t * KT + wv * 4);
      if (key00 >= pc_lo && key00 + 35 < pc_hi && key00 + 35 <= last_key) {
        // both pieces of the wave (keys key00 .. +3 and key00 + 32 .. +35) inside the remembered page and the key range: the
        // contiguous kernel's requests with another scalar offset (the loop is bound by instruction issue: every scalar
        // instruction per request shows)
        const int soff = (key00 + pc_off) * row_bytes;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          if (what & 1) pp_dma16(krs, kb + r * 8192, k_lane, soff + r * 32 * row_bytes);
          if (what & 2) pp_dma16(vrs, vb + r * 8192, v_lane, soff + r * 32 * row_bytes);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int key0 = key00 + 32 * r;
          const int kk = min(key0, last_key);
          if (kk < pc_lo || kk >= pc_hi) pp_page_range(A.ka, A.ps_magic, kk, pc_lo, pc_hi, pc_off);
          int soff = (kk + pc_off) * row_bytes;
          int adj = 0;
          if (kk + 3 > last_key || kk + 3 >= pc_hi) {        // rare: the ragged last piece of the key range, or a piece that runs into the next page
            const int j = min(ln >> 4, last_key - kk);
            adj += ((ln >> 4) - j) * row_bytes;
            if (kk + min(3, last_key - kk) >= pc_hi) {      // its rows from pc_hi on live in the NEXT page (a piece spans at most two: page_size >= 3)
              int lo2, hi2, off2;
              pp_page_range(A.ka, A.ps_magic, pc_hi, lo2, hi2, off2);
              // whole physical row per lane (the next page may lie anywhere, and a lane offset cannot go negative); scalar offset 0
              adj -= (kk + ((kk + j >= pc_hi) ? off2 : pc_off)) * row_bytes;
              soff = 0;
            }
          }
          if (what & 1) pp_dma16(krs, kb + r * 8192, k_lane - adj, soff);
          if (what & 2) pp_dma16(vrs, vb + r * 8192, v_lane - adj, soff);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int key = min(kv_s + t * KT + wv * 4 + (ln >> 4) + 32 * r, last_key);
        const int delta = (A.ka.slot(key) - (ln >> 4)) * row_bytes;     // physical row instead of the row inside the piece
        if (what & 1) pp_dma16(krs, kb + r * 8192, k_lane + delta, 0);
        if (what & 2) pp_dma16(vrs, vb + r * 8192, v_lane + delta, 0);
      }
    }
  };
  auto issue = [&](int t) { issue_w(t, wave); };

  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c2 = (FR == 7 || FR == 8) ? 1.0f : A.scale_log2;     // FR = 7 / 8 are only launched when the product is 1 to rounding
  const int kswz = l31 & 15;
  const int vi = lane & 15, vg1 = (lane >> 4) & 1;
  const int v_rowq = vi >> 2;
  const int v_in = (vg1 << 5) | ((vi & 3) << 3);

  f32x16 s[2];
  bf16x8 pb[2][2];
  bf16x8 fA[4], fB[4];     // drain only
  // operand-fragment ring of the MFMA steps: RD slots of two fragments, LDS reads issued LA = RD - 1 batches (2 LA MFMAs) ahead.
  // Three slots in the phase-locked schedules (deeper measured slower there: 992 -> 946 TFLOP/s); six in the free-running one,
  // whose step trace showed every MFMA waiting for its own LDS round trip with reads only two batches ahead.
  constexpr int RD = FR == 1 ? PP_RING : 3, LA = RD - 1;
  bf16x8 fr[RD][2];
  pb[0][0] = pb[0][1] = pb[1][0] = pb[1][1] = bf16x8{};      // P(-1) = 0 for the unconditional PV of tile 0
  if (PP_ABLATE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fA[i] = fB[i] = bf16x8{};
    pb[0][0] = pb[0][1] = pb[1][0] = pb[1][1] = bf16x8{};
    s[0] = s[1] = f32x16{};
  }

  auto ldV = [&](bf16x8(&f)[4], const unsigned char* vb, int b, int s2) {
    if (PP_ABLATE & 4) return;
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
    if (PP_ABLATE & 8) return;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[d], p, o[d], 0, 0, 0);
  };
#define PP_SB() __builtin_amdgcn_sched_barrier(0)

  // ---- step M(t): PV(t-1) then QK(t) = 16 batches of two MFMAs.  Operand fragments rotate through a three-slot
  //      ring of two fragments (24 VGPRs): the LDS reads of batch j+2 are issued before the MFMAs of batch j, so a
  //      read has two batches (~128 matrix-pipe cycles) to land.  (Four-fragment double buffering cost 8 VGPRs more,
  //      which is what did not fit next to O (64) + Q (32) + S (32) at three waves per SIMD.)
  //        PV batch j = (block b = j>>2, k-slot s2 = (j>>1)&1, d pair dh = j&1)     j = 0..7
  //        QK batch j = (block b = j>>2, k-steps 2*(j&3), 2*(j&3)+1)                 j = 0..7
  // phase: 0 = everything, 1 = only the first LA fragment reads (issued ahead of time), 2 = the rest (reads already issued)
  auto stepJ = [&](int t, int vt, auto j0c, auto j1c, auto phasec) {   // batches [J0, J1) of the list below, K from tile t, V from tile vt
    constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value, PHASE = decltype(phasec)::value;
    const unsigned char* kb = smem + K_OFF + (t % RK) * 16384;
    const unsigned char* vb = smem + V_OFF + (vt % RV) * 16384;
    // lane-derived address terms are RECOMPUTED here from an opaque lane id (a handful of VALU ops): kept live across
    // the loop they were spilled at three waves per SIMD, and a scratch reload drains the DMA queue (shared vmcnt)
    int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(ln));
    const int hi = ln >> 5, l31 = ln & 31, kswz = ln & 15, v_rowq = (ln >> 2) & 3;
    const int v_in = (((ln >> 4) & 1) << 5) | ((ln & 3) << 3);
    auto load = [&](int j) {
      if (PP_ABLATE & 4) return;
      bf16x8(&f)[2] = fr[j % RD];
      if (j < 8) {
        const int bb = j >> 2, s2 = (j >> 1) & 1, dh = j & 1;
        const unsigned char* vr0 = vb + (32 * bb + 16 * s2 + 4 * hi + v_rowq) * 256 + v_in;
        const unsigned char* vr1 = vr0 + 8 * 256;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int ch = ((2 * dh + e) ^ v_rowq) << 6;
          const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vr0 + ch));
          const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vr1 + ch));
          f[e] = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      } else {
        const int q = j - 8, bb = q >> 2, ks0 = 2 * (q & 3);
        const unsigned char* krow = kb + (32 * bb + l31) * 256;
#pragma unroll
        for (int e = 0; e < 2; ++e)
          f[e] = *reinterpret_cast<const bf16x8*>(krow + (((2 * (ks0 + e) + hi) ^ kswz) << 4));
      }
    };
    auto mma = [&](int j) {
      if (PP_ABLATE & 8) return;
      bf16x8(&f)[2] = fr[j % RD];
      if (j < 8) {
        const int bb = j >> 2, s2 = (j >> 1) & 1, dh = j & 1;
#pragma unroll
        for (int e = 0; e < 2; ++e)
          o[2 * dh + e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[e], pb[bb][s2], o[2 * dh + e], 0, 0, 0);
      } else {
        const int q = j - 8, bb = q >> 2, ks0 = 2 * (q & 3);
        if ((q & 3) == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[bb][r] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) s[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[e], qf[ks0 + e], s[bb], 0, 0, 0);
      }
    };
    if (PHASE != 2) {
#pragma unroll
      for (int j = J0; j < J0 + LA && j < J1; ++j) load(j);
    }
    if (PHASE == 1) return;
#pragma unroll
    for (int j = J0; j < J1; ++j) {
      PP_SB();
      if (j + LA < J1) load(j + LA);
      mma(j);
    }
  };
  // PV(t-1) + QK(t).  At t == 0 there is no previous tile: P is all zero and the V fragments are read from tile 0, whose
  // DMA has landed — O += V^T * 0.  Unconditional on purpose: a branch here made the O accumulators a phi and cost
  // 32 v_mov_b64 per tile.
  using ic0 = std::integral_constant<int, 0>;
  using ic1 = std::integral_constant<int, 1>;
  using ic2 = std::integral_constant<int, 2>;
  using ic8 = std::integral_constant<int, 8>;
  using ic16 = std::integral_constant<int, 16>;
  auto stepM = [&](int t) { stepJ(t, t > 0 ? t - 1 : 0, ic0{}, ic16{}, ic0{}); };
  auto stepQK = [&](int t) { stepJ(t, 0, ic8{}, ic16{}, ic0{}); };
  auto prePV = [&](int t) { stepJ(t, t, ic0{}, ic8{}, ic1{}); };        // first V fragments, issued before the softmax
  auto stepPV = [&](int t) { stepJ(t, t, ic0{}, ic8{}, ic2{}); };

  // ---- step V(t): online softmax of the 64-key tile; P -> pb (bf16), in-place rescale of O when a max grew.
  //      Two halves (one per 32-key block of exponentials) so that the three-group schedule can put a phase
  //      boundary between them; with two groups they run back to back.
  pp_f32x2 v_mcv = {0.f, 0.f}, v_acc = {0.f, 0.f};
  const pp_f32x2 c2v = {c2, c2};
  auto exp_block = [&](int b) {
    // exponent arguments two at a time (v_pk_fma_f32); row sums in two plain chains
    float a0 = 0.f, a1 = 0.f;
#if !PP_PACKED
    const float mc = v_mcv[0];
#endif
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#if PP_PACKED
      const pp_f32x2 sv = {s[b][2 * i], s[b][2 * i + 1]};
      const pp_f32x2 e = sv * c2v - v_mcv;
      const float p0 = __builtin_amdgcn_exp2f(e[0]), p1 = __builtin_amdgcn_exp2f(e[1]);
#else
      const float p0 = __builtin_amdgcn_exp2f(s[b][2 * i] * c2 - mc);
      const float p1 = __builtin_amdgcn_exp2f(s[b][2 * i + 1] * c2 - mc);
#endif
      a0 += p0;
      a1 += p1;
      pb[b][i >> 2][(2 * i) & 7] = static_cast<__bf16>(p0);
      pb[b][i >> 2][(2 * i + 1) & 7] = static_cast<__bf16>(p1);
    }
    v_acc += pp_f32x2{a0, a1};
  };
  // Lazy reference maximum.  softmax is invariant to the exponent reference: p = exp2(c2*(s - M)) with ANY finite M
  // gives the same O / l as long as nothing overflows.  The true row maximum is computed for the first tile only;
  // later tiles reuse M (no 16-deep v_max3 chain, no cross-half exchange, no rescale factor — a third of the step's
  // VALU time, tools/trace_attn.sh) and only check the tile's row sums: if some row would exceed 2^20 (or overflowed
  // to inf), the tile is redone against its true maximum and O / l are rescaled.  m_run holds M; LSE = M*scale + ln l.
  constexpr float kLazyLimit = 1048576.f;
  auto tile_max = [&]() -> float {
    // first read of the S accumulators through a compiler-visible instruction (MFMA -> VALU wait states are not
    // inserted for inline-asm operands; a wave that arrives last at the barrier starts this step at once)
    float mx = pp_max3(__builtin_fmaxf(s[0][0], s[1][0]), m_run, m_run);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = pp_max3(mx, s[0][r], s[1][r]);
    return pp_half_max(mx);
  };
  auto rescale_o = [&](float alpha) {
    asm volatile("" : "+v"(alpha));
    if (__any(alpha != 1.0f)) {
      asm volatile("s_nop 7" ::: "memory");          // hazards around inline-asm operands are padded by hand (rare path)
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(o[d][r]) : "v"(alpha));
      asm volatile("s_nop 3" ::: "memory");
    }
  };
  auto stepV1 = [&](int t) {
    // DMA of tile t+PD is issued HERE: an LDS-DMA instruction costs its issuing wave ~100+ cycles, which a
    // VALU step can afford and the MFMA step cannot
    if (!FR && loader && t + PD < NT && !(PP_ABLATE & 1)) issue(t + PD);
    if (t == NT - 1 && (nkeys & (KT - 1))) {     // ragged last tile (wave-uniform, executed once)
      const int kidx = t * KT + 4 * hi;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kidx + 32 * b + (r & 3) + 8 * (r >> 2) >= nkeys) s[b][r] = -INFINITY;
    }
    if (t == 0) m_run = tile_max();               // O and l are still zero: nothing to rescale
    const float mc = m_run * c2;
    v_mcv = pp_f32x2{mc, mc};
    v_acc = pp_f32x2{0.f, 0.f};
    exp_block(0);
  };
  auto stepV2 = [&](int t) {
    exp_block(1);
    float tile_sum = v_acc[0] + v_acc[1];
    if (__any(!(tile_sum < kLazyLimit))) {         // rare: this tile outgrew the reference (inf / NaN land here too)
      if (A.dbg_rescales != nullptr && lane == 0) atomicAdd(A.dbg_rescales, 1u);
      const float m_new = tile_max();              // >= m_run, identical in lane and lane^32
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
      m_run = m_new;
      const float mc = m_new * c2;
      v_mcv = pp_f32x2{mc, mc};
      v_acc = pp_f32x2{0.f, 0.f};
      exp_block(0);
      exp_block(1);
      tile_sum = v_acc[0] + v_acc[1];
      l_run *= alpha;
      rescale_o(alpha);
    }
    l_run += tile_sum;
  };
  auto phase_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: tiles 0 .. PD-1 in flight
  if (!SWP) {
#pragma unroll
    for (int i = 0; i < PD; ++i)
      if (loader && i < NT) issue(i);
  }

  if (SWP) {
    // ================= software-pipelined schedule (attn_variant 5; 6 = two four-wave workgroups per CU) =================================================
    // One wave keeps BOTH pipes busy by itself: iteration t interleaves, instruction by instruction, the 16 MFMAs of
    // PV(t-1), the 16 MFMAs of QK(t+1) and the ~110 VALU instructions of softmax(t) — three mutually independent streams
    // (P(t-1) and S(t) were finished last iteration, S(t+1) and P(t) are for the next one).  An MFMA holds the matrix pipe
    // for 32 cycles after issue; the wave uses them for the 3-4 VALU instructions placed behind it, so the wave-age
    // arbitration between the two waves of a SIMD (tools/probe_overlap.hip) no longer decides whether softmax and MFMA
    // overlap.  Costs: two S and two P register sets (parity-unrolled), K ring read one tile ahead, V ring one behind.
    f32x16 s2[2];
    bf16x8 pb2[2][2];
    pb2[0][0] = pb2[0][1] = pb2[1][0] = pb2[1][1] = bf16x8{};          // P(-1) = 0
    auto qk_plain = [&](int t, f32x16(&sx)[2]) {
      const unsigned char* kb = smem + K_OFF + (t % RK) * 16384;
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sx[bb][r] = 0.f;
        const unsigned char* krow = kb + (32 * bb + l31) * 256;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
          sx[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(krow + (((2 * ks + hi) ^ kswz) << 4)),
                                                           qf[ks], sx[bb], 0, 0, 0);
      }
    };
    auto mask_ragged = [&](int t, f32x16(&sx)[2]) {
      // (opaque on purpose: the 32 comparisons below are loop-invariant lane masks — hipcc hoisted them out of the tile loop and kept
      //  64 scalar registers alive for a branch that one tile per work item takes; the scalars the loop does use — LDS slot addresses,
      //  row offsets of the K / V requests — were spilled to lanes and read back with v_readlane in front of every request)
      int kidx = t * KT + 4 * hi;
      asm volatile("" : "+v"(kidx));
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kidx + 32 * b + (r & 3) + 8 * (r >> 2) >= nkeys) sx[b][r] = -INFINITY;
    };
    auto true_max = [&](f32x16(&sx)[2], float floor_v) -> float {
      float mx = pp_max3(__builtin_fmaxf(sx[0][0], sx[1][0]), floor_v, floor_v);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = pp_max3(mx, sx[0][r], sx[1][r]);
      return pp_half_max(mx);
    };
    auto exp_plain = [&](f32x16(&sx)[2], bf16x8(&px)[2][2], float mc) -> float {      // non-interleaved (tile 0 redo path)
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float p0 = __builtin_amdgcn_exp2f(sx[b][2 * i] * c2 - mc);
          const float p1 = __builtin_amdgcn_exp2f(sx[b][2 * i + 1] * c2 - mc);
          a0 += p0;
          a1 += p1;
          px[b][i >> 2][(2 * i) & 7] = static_cast<__bf16>(p0);
          px[b][i >> 2][(2 * i + 1) & 7] = static_cast<__bf16>(p1);
        }
      return a0 + a1;
    };
    // iteration t: sC = S(t) -> pC = P(t) (softmax), pP = P(t-1) -> O (PV), sN = S(t+1) (QK)
    // U4: lane terms of the fragment reads, once per kernel (K_OFF folded into the K terms; V_OFF = 0)
    f32x16 negm;                                       // PRE: -m in every element (MFMA C operand)
    auto set_negm = [&](float m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) negm[r] = -m;
      asm volatile("" : "+v"(negm));
    };
    if (PRE) set_negm(0.f);
    int kterm[8], vterm[4];
    if (CS) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        kterm[ks] = K_OFF + l31 * 256 + (((2 * ks + hi) ^ kswz) << 4);
        asm volatile("" : "+v"(kterm[ks]));
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        vterm[d] = (4 * hi + v_rowq) * 256 + v_in + ((d ^ v_rowq) << 6);
        asm volatile("" : "+v"(vterm[d]));
      }
    }
    auto body = [&](int t, f32x16(&sC)[2], f32x16(&sN)[2], bf16x8(&pC)[2][2], bf16x8(&pP)[2][2], auto kslot_c, auto vslot_c) -> float {
      constexpr int KSLOT = decltype(kslot_c)::value, VSLOT = decltype(vslot_c)::value;     // constant-slot forms only (-1 otherwise)
      const unsigned char* kb = smem + K_OFF + ((t + 1) % RK) * 16384;
      const unsigned char* vb = smem + V_OFF + ((t > 0 ? t - 1 : 0) % RV) * 16384;
#if PP_SWP_RECOMPUTE
      int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(ln));
      const int hi = ln >> 5, l31 = ln & 31, kswz = ln & 15, v_rowq = (ln >> 2) & 3;
      const int v_in = (((ln >> 4) & 1) << 5) | ((ln & 3) << 3);
#endif
      constexpr int RDs = PRE ? 2 : PP_SWP_RING, LAs = RDs - 1;  // fragment ring of this schedule (registers are tight; PRE spends 16 on negm)
      bf16x8 fs[RDs][2];
      auto load = [&](int j) {
        bf16x8(&f)[2] = fs[j % RDs];
        if (CS && j < 8) {
          const int bb = j >> 2, sl = (j >> 1) & 1, dh = j & 1;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const unsigned char* a0 = smem + vterm[2 * dh + e] + (VSLOT * 16384 + (32 * bb + 16 * sl) * 256);
            const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0));
            const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0 + 2048));
            f[e] = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
          }
        } else if (CS) {
          const int q = j - 8, bb = q >> 2, ks0 = 2 * (q & 3);
#pragma unroll
          for (int e = 0; e < 2; ++e)
            f[e] = *reinterpret_cast<const bf16x8*>(smem + kterm[ks0 + e] + (KSLOT * 16384 + 32 * bb * 256));
        } else if (j < 8) {
          const int bb = j >> 2, sl = (j >> 1) & 1, dh = j & 1;
          const unsigned char* vr0 = vb + (32 * bb + 16 * sl + 4 * hi + v_rowq) * 256 + v_in;
          const unsigned char* vr1 = vr0 + 8 * 256;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int ch = ((2 * dh + e) ^ v_rowq) << 6;
            const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vr0 + ch));
            const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vr1 + ch));
            f[e] = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
          }
        } else {
          const int q = j - 8, bb = q >> 2, ks0 = 2 * (q & 3);
          const unsigned char* krow = kb + (32 * bb + l31) * 256;
#pragma unroll
          for (int e = 0; e < 2; ++e) f[e] = *reinterpret_cast<const bf16x8*>(krow + (((2 * (ks0 + e) + hi) ^ kswz) << 4));
        }
      };
      auto mma1 = [&](int j, int e) {
        bf16x8(&f)[2] = fs[j % RDs];
        if (j < 8) {
          const int bb = j >> 2, sl = (j >> 1) & 1, dh = j & 1;
          o[2 * dh + e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[e], pP[bb][sl], o[2 * dh + e], 0, 0, 0);
        } else {
          const int q = j - 8, bb = q >> 2, ks0 = 2 * (q & 3);
          if (PRE && (q & 3) == 0 && e == 0) {
            sN[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[e], qf[ks0 + e], negm, 0, 0, 0);     // S - m
          } else {
            if ((q & 3) == 0 && e == 0) {
#pragma unroll
              for (int r = 0; r < 16; ++r) sN[bb][r] = 0.f;
            }
            sN[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[e], qf[ks0 + e], sN[bb], 0, 0, 0);
          }
        }
      };
      float p0 = 0.f, p1 = 0.f, e1 = 0.f;
      float acc4[4] = {0.f, 0.f, 0.f, 0.f};
      pp_f32x2 accp[2] = {pp_f32x2{0.f, 0.f}, pp_f32x2{0.f, 0.f}};
      u32x4 pw[2][2];                                  // P(t) as packed bf16 pairs: pw[b][half][k] = keys 2k, 2k+1 of that half
      const float mc = v_mcv[0];
#pragma unroll
      for (int j = 0; j < LAs; ++j) load(j);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int b = j >> 3, i = j & 7;               // softmax piece j = pair i of block b
        PP_SB();
#if PP_SWP_YIELD
        if ((j & 3) == 3 && wave < 4) __builtin_amdgcn_s_sleep(PP_SWP_YIELD);   // the older wave yields issue slots (experiment)
#endif
        if (j + LAs < 16 && !(PP_ABLATE & 4)) load(j + LAs);
        if (!(PP_ABLATE & 8)) mma1(j, 0);
        if (PP_ABLATE & 16) {                          // timing experiments: no transcendental
          p0 = __builtin_fmaf(sC[b][2 * i], c2, -mc);
          p1 = __builtin_fmaf(sC[b][2 * i + 1], c2, -mc);
        } else if (PP_SWP_PACKED) {                    // one v_pk_fma_f32 for the two exponent arguments
          const pp_f32x2 e2 = pp_f32x2{sC[b][2 * i], sC[b][2 * i + 1]} * c2v - v_mcv;
          p0 = __builtin_amdgcn_exp2f(e2[0]);
          p1 = __builtin_amdgcn_exp2f(e2[1]);
        } else if (PRE) {                              // the score is the exponent
          p0 = __builtin_amdgcn_exp2f(sC[b][2 * i]);
          e1 = sC[b][2 * i + 1];
        } else if (PP_SWP_BALANCE) {
          // the VALU work of a pair is split evenly around the second MFMA: two scale-FMAs + one exponential (24 cycles) here, the
          // other exponential + two adds + the pack (28 cycles) behind it — each piece fits under the 32 matrix-pipe cycles of the
          // OTHER wave's MFMA, where 40 + 12 left a bubble every pair
          p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sC[b][2 * i], c2, -mc));
          e1 = __builtin_fmaf(sC[b][2 * i + 1], c2, -mc);
          asm volatile("" : "+v"(e1));
        } else {
          p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sC[b][2 * i], c2, -mc));
          p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sC[b][2 * i + 1], c2, -mc));
        }
        PP_SB();
        if (!(PP_ABLATE & 8)) mma1(j, 1);
        if ((PRE || PP_SWP_BALANCE) && !PP_SWP_PACKED && !(PP_ABLATE & 16)) p1 = __builtin_amdgcn_exp2f(e1);
        if (!(PP_ABLATE & 64)) {                       // four independent row-sum chains, pinned behind this MFMA
          if (PP_SWP_PACKED) {
            pp_f32x2& a2 = (j & 1) ? accp[1] : accp[0];
            const pp_f32x2 pv = {p0, p1};
            asm volatile("s_nop 0\n\tv_pk_add_f32 %0, %0, %1" : "+v"(a2) : "v"(pv));
          } else {
            // p0 / p1 come straight out of v_exp_f32: a VALU instruction that reads a transcendental's result needs one wait state
            // behind it, and the hazard recogniser does not look into inline asm.  Latent until a recompile puts the exponential
            // directly in front of this add: seen as -inf row sums in lanes 0-3 of every 8 in a build with a loop around the kernel body.
            asm volatile("s_nop 0\n\tv_add_f32 %0, %0, %1" : "+v"(acc4[(2 * j) & 3]) : "v"(p0));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc4[(2 * j + 1) & 3]) : "v"(p1));
          }
        }
        if (!(PP_ABLATE & 32)) {
          unsigned pk;
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(p0), "v"(p1));
          pw[b][i >> 2][i & 3] = pk;
        } else if (j == 0) {
          pw[0][0][0] = __builtin_bit_cast(unsigned, p0);
        }
      }
      PP_SB();
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) pC[b][h2] = __builtin_bit_cast(bf16x8, pw[b][h2]);
      if (PP_SWP_PACKED) return (accp[0][0] + accp[0][1]) + (accp[1][0] + accp[1][1]);
      return (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
    };
    auto iteration = [&](int t, f32x16(&sC)[2], f32x16(&sN)[2], bf16x8(&pC)[2][2], bf16x8(&pP)[2][2], auto kslot_c, auto vslot_c) {
      long long tr[6] = {0, 0, 0, 0, 0, 0};
      if (PP_TRACE) tr[0] = __builtin_readcyclecounter();
      if (DUAL) {
        // issue order per iteration: K(t+2) then V(t+1) (4 pieces each per wave).  Here K(t+1) must have landed; the V(t) pieces
        // issued after it may stay in flight (PV(t) is next iteration's) — except at t = 0, where K(1) was the last thing issued
        if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        // own pieces of tile t+1 landed (t+2 may stay in flight); only waves 0-3 have any (8 per tile, see below)
        if (t + 2 < NT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (PP_TRACE) tr[1] = __builtin_readcyclecounter();
      __builtin_amdgcn_s_barrier();                    // tile t+1 complete for everyone; K(t) and V(t-2) are free
      if (PP_TRACE) tr[2] = __builtin_readcyclecounter();
      // (the four pieces were also tried behind MFMAs 2 / 6 / 10 / 14 of the body: 1045 -> 991 TFLOP/s — an LDS-DMA
      //  instruction stalls its wave for ~100 cycles and inside the body that stall also holds back the wave's MFMAs)
      // LDS-DMA issue stalls a wave for ~80 cycles per instruction: the OLDER wave of each SIMD issues the pieces of both
      // (8 per tile) while the younger one starts the matrix pipe at once — which also staggers the two waves' bodies
      if (DUAL) {
        if (t + 2 < NT) {
          issue_w(t + 2, wave, 1);
          issue_w(t + 2, wave + 4, 1);
        }
        if (t + 1 < NT) {
          issue_w(t + 1, wave, 2);
          issue_w(t + 1, wave + 4, 2);
        }
      } else if (U4) {
        if (wave < 4) {                                  // K three tiles ahead, V two: four slots each
          if (t + 3 < NT) {
            issue_w(t + 3, wave, 1);
            issue_w(t + 3, wave + 4, 1);
          }
          if (t + 2 < NT) {
            issue_w(t + 2, wave, 2);
            issue_w(t + 2, wave + 4, 2);
          }
        }
      } else if (wave < 4 && t + PD < NT && !(PP_ABLATE & 1)) {
        issue_w(t + PD, wave);
        issue_w(t + PD, wave + 4);
      }
      if (PP_TRACE) tr[3] = __builtin_readcyclecounter();
      if (t == NT - 1 && (nkeys & (KT - 1))) mask_ragged(t, sC);
      const float mc0 = m_run * c2;
      v_mcv = pp_f32x2{mc0, mc0};
      float tile_sum = body(t, sC, sN, pC, pP, kslot_c, vslot_c);
      if (PP_TRACE) {
        asm volatile("" ::"v"(tile_sum), "v"(sN[1][15]), "v"(o[3][15]));
        tr[4] = __builtin_readcyclecounter();
      }
      if (PRE) {
        if (__any(!(tile_sum < kLazyLimit))) {         // rare: sC holds S(t) - m_run, sN already holds S(t+1) - m_run
          if (A.dbg_rescales != nullptr && lane == 0) atomicAdd(A.dbg_rescales, 1u);
          const float d = true_max(sC, 0.f);           // >= 0: how far this tile's maximum lies above the reference
          const float alpha = __builtin_amdgcn_exp2f(-d);
          tile_sum = exp_plain(sC, pC, d);
          m_run += d;
          l_run *= alpha;
          rescale_o(alpha);
          set_negm(m_run);
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) sN[b][r] -= d;
        }
      } else if (__any(!(tile_sum < kLazyLimit))) {    // rare: this tile outgrew the reference maximum
        if (A.dbg_rescales != nullptr && lane == 0) atomicAdd(A.dbg_rescales, 1u);
        const float m_new = true_max(sC, m_run);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
        m_run = m_new;
        tile_sum = exp_plain(sC, pC, m_new * c2);
        l_run *= alpha;
        rescale_o(alpha);
      }
      l_run += tile_sum;
      if (PP_TRACE && t < 48 && lane == 0) {           // stamps of the first 48 tiles, staged in LDS behind the rings
        long long* tp = reinterpret_cast<long long*>(smem + LDS_SWP) + (t * 8 + wave) * 8;
        tr[5] = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 6; ++i) tp[i] = tr[i];
      }
    };
    if (U6) {                                          // V slot 2 is read (times P(-1) = 0) by the PV of iteration 0: clear it (256 threads x 64 B)
#pragma unroll
      for (int z = 0; z < 4; ++z) *reinterpret_cast<u32x4*>(smem + V_OFF + 2 * 16384 + tid * 64 + z * 16) = u32x4{0, 0, 0, 0};
    }
    if (DUAL) {                                        // K(0), V(0), K(1); K(0) and V(0) must land before QK(0) / the P = 0 pass over V(0)
      issue_w(0, wave, 3);
      issue_w(0, wave + 4, 3);
      if (NT > 1) {
        issue_w(1, wave, 1);
        issue_w(1, wave + 4, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    if (U4) {
      // V slot 3 is read (times P(-1) = 0) by the PV of iteration 0 before anything was copied there: LDS starts with arbitrary bits
      *reinterpret_cast<u32x4*>(smem + V_OFF + 3 * 16384 + tid * 32) = u32x4{0, 0, 0, 0};
      *reinterpret_cast<u32x4*>(smem + V_OFF + 3 * 16384 + tid * 32 + 16) = u32x4{0, 0, 0, 0};
      if (wave < 4) {                                   // K(0), V(0), K(1), V(1), K(2): everything but K(2) has to land now
        issue_w(0, wave, 3);
        issue_w(0, wave + 4, 3);
        if (NT > 1) {
          issue_w(1, wave, 3);
          issue_w(1, wave + 4, 3);
        }
        if (NT > 2) {
          issue_w(2, wave, 1);
          issue_w(2, wave + 4, 1);
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
    }
#pragma unroll
    for (int i = 0; i < PD; ++i)
      if (!DUAL && !U4 && i < NT && wave < 4) {
        issue_w(i, wave);
        issue_w(i, wave + 4);
      }
    if (!DUAL && !U4) {
      const int fl = min(NT, PD) - 1;                  // tiles that may stay in flight behind tile 0 (8 pieces each)
      if (fl <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (fl == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    qk_plain(0, s);
    if (NT == 1 && (nkeys & (KT - 1))) mask_ragged(0, s);
    m_run = true_max(s, m_run);
    if (PRE) {                                         // the loop's scores come out of the MFMA as S - m: bring S(0) to the same form
      set_negm(m_run);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[b][r] -= m_run;
    }
    using ic = std::integral_constant<int, -1>;
    if (U4) {                                          // iteration t reads K slot (t + 1) % 4 and V slot (t - 1) % 4
      for (int t = 0; t < NT; t += 4) {
        iteration(t, s, s2, pb, pb2, std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{});
        if (t + 1 < NT) iteration(t + 1, s2, s, pb2, pb, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
        if (t + 2 < NT) iteration(t + 2, s, s2, pb, pb2, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
        if (t + 3 < NT) iteration(t + 3, s2, s, pb2, pb, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
      }
    } else if (U6) {                                   // K slot (t + 1) % 2, V slot (t + 2) % 3
      for (int t = 0; t < NT; t += 6) {
        iteration(t, s, s2, pb, pb2, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
        if (t + 1 < NT) iteration(t + 1, s2, s, pb2, pb, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        if (t + 2 < NT) iteration(t + 2, s, s2, pb, pb2, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        if (t + 3 < NT) iteration(t + 3, s2, s, pb2, pb, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
        if (t + 4 < NT) iteration(t + 4, s, s2, pb, pb2, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        if (t + 5 < NT) iteration(t + 5, s2, s, pb2, pb, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
      }
    } else {
      for (int t = 0; t < NT; t += 2) {
        iteration(t, s, s2, pb, pb2, ic{}, ic{});
        if (t + 1 < NT) iteration(t + 1, s2, s, pb2, pb, ic{}, ic{});
      }
    }
    if (NT & 1) {                                      // P(NT-1) lives in pb when the last iteration had even parity
    } else {
      pb[0][0] = pb2[0][0];
      pb[0][1] = pb2[0][1];
      pb[1][0] = pb2[1][0];
      pb[1][1] = pb2[1][1];
    }
    if (DUAL) {                                        // V(NT-1) was allowed to stay in flight; everyone's pieces must be visible
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    {   // drain: PV(NT-1)
      const unsigned char* vb = smem + V_OFF + ((NT - 1) % RV) * 16384;
      ldV(fA, vb, 0, 0);
      ldV(fB, vb, 0, 1);
      PP_SB();
      mmaV(fA, pb[0][0]);
      ldV(fA, vb, 1, 0);
      PP_SB();
      mmaV(fB, pb[0][1]);
      ldV(fB, vb, 1, 1);
      PP_SB();
      mmaV(fA, pb[1][0]);
      mmaV(fB, pb[1][1]);
    }
  } else if (FR) {
    for (int t = 0; t < NT; ++t) {
      long long tr[7] = {0, 0, 0, 0, 0, 0, 0};
      if (PP_TRACE) tr[0] = __builtin_readcyclecounter();
      pp_wait_tiles(min(NT - 1 - t, PD - 1));          // own pieces of tile t landed (t+1 may stay in flight)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (PP_TRACE) tr[1] = __builtin_readcyclecounter();
      __builtin_amdgcn_s_barrier();                    // tile t complete for everyone; everyone is done with tile t-1
      if (PP_TRACE) tr[2] = __builtin_readcyclecounter();
      // an LDS-DMA instruction stalls its issuing wave for ~100 cycles: the older wave of each SIMD issues its pieces now,
      // while the younger one starts the matrix pipe, and the younger one after its QK, while the older one computes
      if (wave < 4 && t + PD < NT && !(PP_ABLATE & 1)) issue(t + PD);
      if (PP_TRACE) tr[3] = __builtin_readcyclecounter();
      stepQK(t);
      if (wave >= 4 && t + PD < NT && !(PP_ABLATE & 1)) issue(t + PD);
      prePV(t);
      if (PP_TRACE) {
        asm volatile("" ::"v"(s[1][15]));
        tr[4] = __builtin_readcyclecounter();
      }
      if (!(PP_ABLATE & 2)) {
        stepV1(t);
        stepV2(t);
      }
      if (PP_TRACE) {
        asm volatile("" ::"v"(l_run), "v"(pb[1][1]));
        tr[5] = __builtin_readcyclecounter();
      }
      stepPV(t);
      if (PP_TRACE && t < 64 && lane == 0) {
        asm volatile("" ::"v"(o[3][15]));
        long long* tp = reinterpret_cast<long long*>(smem + LDS_BYTES) + (t * 8 + wave) * 8;
        tr[6] = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 7; ++i) tp[i] = tr[i];
      }
    }
  } else if (NG == 2) {
    // barrier index :   2t            2t+1          2t+2
    // G0            :   M(t)          V(t)          M(t+1)
    // G1            :   V(t-1)        M(t)          V(t)
    if (grp == 1) {                                   // phase shift: G1 runs one barrier behind G0
      pp_wait_tiles(min(NT, PD) - 1);
      __builtin_amdgcn_s_barrier();
    }
    for (int t = 0; t < NT; ++t) {
      long long tr[7] = {0, 0, 0, 0, 0, 0, 0};
      if (PP_TRACE) tr[0] = __builtin_readcyclecounter();
      if (grp == 0) pp_wait_tiles(min(NT - 1 - t, PD - 1));   // own pieces of tile t landed (t+1 may stay in flight)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (PP_TRACE) tr[1] = __builtin_readcyclecounter();
      __builtin_amdgcn_s_barrier();
      if (PP_PRIO == 1) __builtin_amdgcn_s_setprio(1);
      if (PP_TRACE) tr[2] = __builtin_readcyclecounter();
      stepM(t);
      if (PP_TRACE) tr[3] = __builtin_readcyclecounter();
      if (PP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
      if (grp == 1 && t + 1 < NT) pp_wait_tiles(max(min(NT - 2 - t, PD - 2), 0));   // tile t+1 before G0's M(t+1)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (PP_TRACE) tr[4] = __builtin_readcyclecounter();
      __builtin_amdgcn_s_barrier();
      if (PP_TRACE) tr[5] = __builtin_readcyclecounter();
      if (PP_PRIO == 2) __builtin_amdgcn_s_setprio(3);
      if (!(PP_ABLATE & 2)) {
        stepV1(t);
        stepV2(t);
      }
      if (PP_PRIO == 2) __builtin_amdgcn_s_setprio(0);
      if (PP_TRACE && t < 64 && lane == 0) {           // stamps go to LDS (no VMEM traffic inside the loop)
        long long* tp = reinterpret_cast<long long*>(smem + LDS_BYTES) + (t * 8 + wave) * 8;
        tr[6] = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 7; ++i) tp[i] = tr[i];
      }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();        // balance the phase shift
  } else {
    // Three groups, one phase apart; phase p is opened by barrier B_p, group g runs its local phase p - g:
    //   B index :  3t       3t+1     3t+2     3t+3     3t+4
    //   G0      :  M(t)     V1(t)    V2(t)    M(t+1)   V1(t+1)
    //   G1      :  V2(t-1)  M(t)     V1(t)    V2(t)    M(t+1)
    //   G2      :  V1(t-1)  V2(t-1)  M(t)     V1(t)    V2(t)
    // -> every phase has exactly one group on the matrix pipe and two groups sharing the VALU.
    // Tile t (K for QK(t), V for PV(t) one tile later) must be complete at B_3t, the barrier before G0's M(t): the
    // loader waves of G0 retire their pieces just before it at the top of their iteration t, those of G1 before
    // the same barrier between their V1(t-1) and V2(t-1).  Ring slot of K(t+PD) = slot of K(t-1), last read by G2's
    // M(t-1) in phase 3t-1; it is refilled by G0 in phase 3t+1 and by G1 in phase 3t+2.  (V: slot of V(t-2), same.)
    if (grp >= 1) {
      if (loader) pp_wait_tiles(min(NT, PD) - 1);      // B_0 publishes tile 0
      __builtin_amdgcn_s_barrier();
    }
    if (grp == 2) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < NT; ++t) {
      long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (PP_TRACE) tr[0] = __builtin_readcyclecounter();
      if (grp == 0) pp_wait_tiles(min(NT - 1 - t, PD - 1));
      phase_barrier();
      if (PP_PRIO == 1) __builtin_amdgcn_s_setprio(2);
      if (PP_TRACE) tr[1] = __builtin_readcyclecounter();
      stepM(t);
      if (PP_TRACE) tr[2] = __builtin_readcyclecounter();
      if (PP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
      phase_barrier();
      if (PP_PRIO == 2) __builtin_amdgcn_s_setprio(3);
      if (PP_TRACE) tr[3] = __builtin_readcyclecounter();
      if (!(PP_ABLATE & 2)) stepV1(t);
      if (grp == 1 && t + 1 < NT) pp_wait_tiles(min(NT - 2 - t, PD - 1));   // own pieces of tile t+1 (B_3(t+1) is next)
      if (PP_TRACE) {
        asm volatile("" ::"v"(v_acc));
        tr[4] = __builtin_readcyclecounter();
      }
      phase_barrier();
      if (PP_TRACE) tr[5] = __builtin_readcyclecounter();
      if (!(PP_ABLATE & 2)) stepV2(t);
      if (PP_PRIO == 2) __builtin_amdgcn_s_setprio(0);
      if (PP_TRACE && t < 64 && lane == 0) {
        asm volatile("" ::"v"(l_run), "v"(pb[1][1]));
        long long* tp = reinterpret_cast<long long*>(smem + LDS_BYTES) + (t * 12 + wave) * 8;
        tr[6] = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 7; ++i) tp[i] = tr[i];
      }
    }
    if (grp <= 1) __builtin_amdgcn_s_barrier();        // balance the phase shifts
    if (grp == 0) __builtin_amdgcn_s_barrier();
  }
  if (PP_TRACE && wi == 0 && A.lse != nullptr) {
    __syncthreads();
    const long long* tp = reinterpret_cast<const long long*>(smem + (FR >= 2 ? LDS_SWP : LDS_BYTES));
    for (int i = tid; i < (FR >= 2 ? 48 : 64) * 4 * NG * 8; i += 256 * NG) reinterpret_cast<long long*>(A.lse)[i] = tp[i];
  }
  // ---- drain: PV of the last tile
  if (FR == 0) {
    const unsigned char* vb = smem + V_OFF + ((NT - 1) % RV) * 16384;
    ldV(fA, vb, 0, 0);
    ldV(fB, vb, 0, 1);
    PP_SB();
    mmaV(fA, pb[0][0]);
    ldV(fA, vb, 1, 0);
    PP_SB();
    mmaV(fB, pb[0][1]);
    ldV(fB, vb, 1, 1);
    PP_SB();
    mmaV(fA, pb[1][0]);
    mmaV(fB, pb[1][1]);
  }
#undef PP_SB

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (SPLIT) {
    if (qrow < A.q_rows) {
      float* op = A.part_o + ((size_t)sp * A.q_rows + qrow) * row_stride + head * HD + 4 * hi;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = o[d][4 * g + e] * inv;
          *reinterpret_cast<f32x4*>(op + 32 * d + 8 * g) = w;
        }
      if (hi == 0) A.part_lse[((size_t)sp * A.heads + head) * A.q_rows + qrow] = m_run * A.scale + __logf(l_tot);
    }
    return;
  }
  if (qrow < q_lim) {
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
    if (A.lse != nullptr && hi == 0 && !PP_TRACE) A.lse[(size_t)head * A.q_rows + qrow] = m_run * A.scale + __logf(l_tot);
  }
}

// out[row][head][:] = sum_s w_s * part_o[s][row][head][:],  w_s = exp(lse_s - LSE) ; LSE = log sum_s exp(lse_s)
// (the same algebra as ifx_lse_merge, over `splits` fp32 partials, rounded to bf16 once)
__global__ __launch_bounds__(256) void attn_split_merge_kernel(const float* __restrict__ part_o,
                                                               const float* __restrict__ part_lse,
                                                               unsigned short* __restrict__ out, float* __restrict__ lse,
                                                               int q_rows, int heads, int splits, int ldo) {
  const int pair = blockIdx.x * 8 + (threadIdx.x >> 5);      // (row, head) pair, 32 lanes x 4 channels
  if (pair >= q_rows * heads) return;
  const int row = pair / heads, head = pair - row * heads, c = (threadIdx.x & 31) * 4;
  // four slots per pass: their LSE values and partial rows are requested together (a run-time loop of dependent load -> use steps pays
  // one memory latency per slot: 12.9 us for 585 rows x 12 heads x 5 slots); the accumulation order stays slot 0, 1, 2, ...
  float mx = -INFINITY;
  for (int s0 = 0; s0 < splits; s0 += 4) {
    float l[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) l[u] = s0 + u < splits ? part_lse[((size_t)(s0 + u) * heads + head) * q_rows + row] : -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u) mx = fmaxf(mx, l[u]);
  }
  float den = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < splits; s0 += 4) {
    float l[4];
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = min(s0 + u, splits - 1);
      l[u] = part_lse[((size_t)s * heads + head) * q_rows + row];
      v[u] = *reinterpret_cast<const f32x4*>(part_o + (((size_t)s * q_rows + row) * heads + head) * 128 + c);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (s0 + u < splits) {
        const float w = __expf(l[u] - mx);
        den += w;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += w * v[u][e];
      }
    }
  }
  const float inv = 1.0f / den;
  u16x4 w4;
#pragma unroll
  for (int e = 0; e < 4; ++e) w4[e] = f2bf(acc[e] * inv);
  *reinterpret_cast<u16x4*>(out + (size_t)row * ldo + head * 128 + c) = w4;
  if (lse != nullptr && c == 0) lse[(size_t)head * q_rows + row] = mx + __logf(den);
}

size_t attn_pp_workspace_bytes(int q_rows, int heads, int splits) {
  return splits <= 1 ? 0 : (size_t)splits * q_rows * heads * (128 + 1) * sizeof(float);
}

// number of key chunks that fills the chip (one workgroup per CU, 256 CUs) for a query tile of `qt` rows
int attn_pp_split_heuristic(int q_rows, int heads, int nkeys, int qt, int slots) {
  using namespace pp;
  const int tiles = ((q_rows + qt - 1) / qt) * heads;
  const int nt = (nkeys + KT - 1) / KT;
  if (tiles * 16 >= slots * 13 || nt < 16) return 1;   // >= ~80 % of the workgroup slots already busy / nothing to split
  int best = 1;
  float best_eff = (float)tiles / ((float)slots * ((tiles + slots - 1) / slots));
  // per-chunk penalty: every chunk writes one fp32 partial per (row, head) and the merge reads it back — (128 + 1) x 4 bytes each.  1 % per
  // chunk at the 3.6 MB of a sequence-parallel rank's launch (585 rows x 12 heads: where the constant was fitted), scaled with the
  // partial's size: MAGI's range launches (12150 rows x 3 heads = 18.8 MB per chunk) then take 5 chunks instead of 7 — rank-clip 8.50 ->
  // 8.36 s (round 5, `IFX_ATTN_SPLIT_PENALTY` sweep: 0.005 / 0.01 / 0.02 / 0.03 / 0.05 -> 8.45 / 8.50 / 8.49 / 8.37 / 8.36 s; the
  // sharded rank is best at 0.01: 268 vs 289 / 273 ms at 0.03 / 0.05)
  static float base = -1.f;
  if (base < 0.f) {
    const char* e = getenv("IFX_ATTN_SPLIT_PENALTY");  // lab: the base of the per-chunk cost
    base = e ? (float)atof(e) : 0.01f;
  }
  const float part_mb = (float)q_rows * (float)heads * 516.f / 3.6e6f;
  const float pen = base * (part_mb > 1.f ? part_mb : 1.f);
  for (int s = 2; s <= 32 && s * 8 <= nt; ++s) {        // chunks of >= 8 tiles (512 keys)
    const int wg = tiles * s, rounds = (wg + slots - 1) / slots;
    const float eff = (float)wg / ((float)slots * rounds) * (1.f - pen * s);   // per-chunk prologue / partial / merge penalty
    if (eff > best_eff + 1e-3f) best_eff = eff, best = s;
  }
  return best;
}

#ifndef PP_DUAL_FR
#define PP_DUAL_FR 6     // 6: six-times unrolled over constant LDS slots; 3: the two-times unrolled form it replaced
#endif
// `paged`: 0 contiguous, 1 wave-uniform page translation (PAGED = 1 instantiations); launch_attn_pp sends every other geometry to
// launch_pp_generic before it gets here
template <int NG, int FR>
static void launch_pp_any(const AttnArgsPP& a, int paged, bool split, dim3 grid, int lds, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_pp_kernel<1, false, NG, FR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)attn_fwd_pp_kernel<0, false, NG, FR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)attn_fwd_pp_kernel<1, true, NG, FR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)attn_fwd_pp_kernel<0, true, NG, FR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const dim3 block(NG * 256);
  if (split) {
    if (paged) hipLaunchKernelGGL((attn_fwd_pp_kernel<1, true, NG, FR>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((attn_fwd_pp_kernel<0, true, NG, FR>), grid, block, lds, stream, a);
  } else {
    if (paged) hipLaunchKernelGGL((attn_fwd_pp_kernel<1, false, NG, FR>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((attn_fwd_pp_kernel<0, false, NG, FR>), grid, block, lds, stream, a);
  }
}
template <int DFR>
static void launch_pp_dual(const AttnArgsPP& a, int paged, bool split, dim3 grid, hipStream_t stream) {
  launch_pp_any<1, DFR>(a, paged, split, grid, 5 * 16384, stream);      // 80 KiB: two workgroups per CU
}
template <int FR>
static void launch_pp_fr(const AttnArgsPP& a, int paged, bool split, dim3 grid, hipStream_t stream) {
  launch_pp_any<2, FR>(a, paged, split, grid, pp::LDS_ALLOC, stream);
}
template <int NG>
static void launch_pp_ng(const AttnArgsPP& a, int paged, bool split, dim3 grid, hipStream_t stream) {
  launch_pp_any<NG, 0>(a, paged, split, grid, pp::LDS_ALLOC, stream);
}
// page geometries whose request pieces may straddle pages (page_size % 4 != 0, unaligned range starts, one-row pages ...): per-lane
// translation on the plain two-group schedule
static void launch_pp_generic(const AttnArgsPP& a, bool split, dim3 grid, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_pp_kernel<2, false, 2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, pp::LDS_ALLOC);
    (void)hipFuncSetAttribute((const void*)attn_fwd_pp_kernel<2, true, 2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, pp::LDS_ALLOC);
    attr_set = true;
  }
  if (split) hipLaunchKernelGGL((attn_fwd_pp_kernel<2, true, 2, 0>), grid, dim3(512), pp::LDS_ALLOC, stream, a);
  else hipLaunchKernelGGL((attn_fwd_pp_kernel<2, false, 2, 0>), grid, dim3(512), pp::LDS_ALLOC, stream, a);
}

// groups: 2 = ping-pong (256 query rows per workgroup), 3 = three-phase (384 rows), 4 = free-running (256 rows).
// slot_cap == 0: self-contained launch (partials in slots [0, splits) of `workspace`, merged here when splits > 1).
// slot_cap  > 0: PARTIAL launch for a workspace laid out for slot_cap slots: always writes fp32 partials, into slots
//                [slot_base, slot_base + splits), no merge; *slots_used reports how many chunks were written.
int launch_attn_pp(const unsigned short* q, unsigned short* out, float* lse, const ifx_kv_view* kv, int q_rows,
                   int heads, int kv_start, int kv_len, float scale, int splits, void* workspace, int groups,
                   hipStream_t stream, int slot_base = 0, int slot_cap = 0, int* slots_used = nullptr, int ldq = 0, int ldo = 0,
                   int n_ranges = 0, const int* q_ranges = nullptr, const int* k_ranges = nullptr) {
  using namespace pp;
  int fr_mode = groups == 4 ? 1 : (groups == 5 ? 2 : (groups == 6 ? 3 : (groups == 7 ? 5 : 0)));   // attn_variant 4 / 5 / 6 / 7
  // page geometry: 0 contiguous; 1 a page table with pages of >= 3 rows (multiply-high page index exact) or a two-segment view — every
  // schedule; 2 anything else — per-lane translation, the plain two-group schedule only
  const unsigned ps_magic = (kv->page_table && kv->page_size >= 2 && (long long)kv->num_slots * kv->page_size < (1ll << 32))
                                ? (unsigned)((1ull << 32) / (unsigned)kv->page_size) + 1u : 0u;
  int paged = 0;
  if (kv->page_table != nullptr) paged = (kv->page_size >= 3 && ps_magic != 0) ? 1 : 2;
  else if (kv->seg_split > 0) paged = 1;
  if (paged == 2) fr_mode = 0, groups = 2;
  if (fr_mode) groups = fr_mode == 3 ? 1 : 2;
  const int QT = 128 * groups;
  AttnArgsPP a;
  a.q = q;
  a.out = out;
  a.lse = lse;
  a.k = kv->k;
  a.v = kv->v;
  a.ka = KvAddr{kv->page_table, kv->page_size, kv->page_table ? 0 : kv->seg_split, kv->page_table ? 0 : kv->seg_delta};
  a.q_rows = q_rows;
  a.heads = heads;
  a.ldq = ldq > 0 ? ldq : heads * 128;
  a.ldo = ldo > 0 ? ldo : heads * 128;
  a.kv_start = kv_start;
  a.kv_len = kv_len;
  a.num_slots = kv->num_slots;
  a.kv_heads = kv->kv_heads;
  a.q_per_kv = heads / kv->kv_heads;
  a.q_tiles = (q_rows + QT - 1) / QT;
  a.dbg_rescales = attn_debug_counter();
  a.ps_magic = ps_magic;
  a.n_ranges = 0;
  if (n_ranges > 0) {
    // longest key ranges first: tile ids are handed out in order, so the expensive tiles must not be the tail of the launch
    int order[8];
    for (int i = 0; i < n_ranges; ++i) order[i] = i;
    for (int i = 1; i < n_ranges; ++i)
      for (int j = i; j > 0 && k_ranges[2 * order[j] + 1] - k_ranges[2 * order[j]] > k_ranges[2 * order[j - 1] + 1] - k_ranges[2 * order[j - 1]]; --j) {
        const int t = order[j];
        order[j] = order[j - 1];
        order[j - 1] = t;
      }
    a.n_ranges = n_ranges;
    a.rt0[0] = 0;
    for (int i = 0; i < n_ranges; ++i) {
      const int r = order[i];
      a.rq0[i] = q_ranges[2 * r], a.rq1[i] = q_ranges[2 * r + 1], a.rk0[i] = k_ranges[2 * r], a.rk1[i] = k_ranges[2 * r + 1];
      a.rt0[i + 1] = a.rt0[i] + (a.rq1[i] - a.rq0[i] + QT - 1) / QT;
    }
    a.q_tiles = a.rt0[n_ranges];
    splits = 1;
  }
  const int nt = (kv_len - kv_start + KT - 1) / KT;
  splits = max(1, min(splits, nt));
  a.chunk_tiles = (nt + splits - 1) / splits;
  a.splits = (nt + a.chunk_tiles - 1) / a.chunk_tiles;   // no empty chunk
  const bool partial = slot_cap > 0;
  if (partial && slot_base + a.splits > slot_cap) {
    set_error("ifx_attn_fwd_partial: slots [%d, %d) exceed the workspace's %d", slot_base, slot_base + a.splits, slot_cap);
    return IFX_EINVAL;
  }
  const int cap = partial ? slot_cap : a.splits;
  float* ws = (float*)workspace;
  a.part_o = ws ? ws + (size_t)slot_base * q_rows * heads * 128 : nullptr;
  a.part_lse = ws ? ws + (size_t)cap * q_rows * heads * 128 + (size_t)slot_base * heads * q_rows : nullptr;
  a.total = a.q_tiles * heads * a.splits;
  a.per_xcd = (a.total + 7) / 8;
  a.scale = scale > 0.f ? scale : 0.08838834764831845f;
  a.scale_log2 = a.scale * 1.4426950408889634f;
  const dim3 grid(a.per_xcd * 8);
  const bool write_partials = partial || a.splits > 1;
  if (write_partials && workspace == nullptr) {
    set_error("ifx_attn_fwd_paged_split: split / partial launches need a workspace");
    return IFX_EINVAL;
  }
  const bool pre = fabsf(a.scale_log2 - 1.0f) <= 2.5e-7f;   // q carries scale * log2(e) already (ifx_rope_grid.q_scale): scores are exponents
  if (paged == 2) launch_pp_generic(a, write_partials, grid, stream);
  else if (fr_mode == 3 && pre && PP_DUAL_FR == 6) launch_pp_dual<8>(a, paged, write_partials, grid, stream);
  else if (fr_mode == 3) launch_pp_dual<PP_DUAL_FR>(a, paged, write_partials, grid, stream);
  else if (fr_mode == 2) launch_pp_fr<2>(a, paged, write_partials, grid, stream);
  else if (fr_mode == 5 && pre) launch_pp_fr<7>(a, paged, write_partials, grid, stream);
  else if (fr_mode == 5) launch_pp_fr<5>(a, paged, write_partials, grid, stream);
  else if (fr_mode == 1) launch_pp_fr<1>(a, paged, write_partials, grid, stream);
  else if (groups == 3) launch_pp_ng<3>(a, paged, write_partials, grid, stream);
  else launch_pp_ng<2>(a, paged, write_partials, grid, stream);
  if (slots_used) *slots_used = a.splits;
  if (!partial && a.splits > 1) {
    const int pairs = q_rows * heads;
    hipLaunchKernelGGL(attn_split_merge_kernel, dim3((pairs + 7) / 8), dim3(256), 0, stream, a.part_o, a.part_lse, out,
                       lse, q_rows, heads, a.splits, a.ldo);
  }
  return check_launch("ifx_attn_fwd_paged(pp)");
}

int launch_attn_merge(const float* workspace, int slot_cap, int slots_used, unsigned short* out, float* lse, int q_rows,
                      int heads, hipStream_t stream, int ldo = 0) {
  const float* part_o = workspace;
  const float* part_lse = workspace + (size_t)slot_cap * q_rows * heads * 128;
  const int pairs = q_rows * heads;
  hipLaunchKernelGGL(attn_split_merge_kernel, dim3((pairs + 7) / 8), dim3(256), 0, stream, part_o, part_lse, out, lse,
                     q_rows, heads, slots_used, ldo > 0 ? ldo : heads * 128);
  return check_launch("ifx_attn_merge_partials");
}

}  // namespace ifx
