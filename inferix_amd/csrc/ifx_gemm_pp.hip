// bf16 linear layer, PERSISTENT PING-PONG tile (gfx950): one 8-wave workgroup per CU walks its list of 64*TJ (tokens) x 256 (channels)
// output tiles; the two waves of every SIMD alternate roles every half K-step, so the matrix pipe always has a wave issuing MFMAs
// out of registers while its partner issues the LDS-DMA of later K-steps and reads its own fragments of the next one.
//
//   y[M,N] = epilogue( x[M,K] @ W[N,K]^T + bias[N] )            (same contract and epilogues as ifx_gemm_glds.hip)
//
// Why (round 3; profiles/r3_gemm_pp.md).  The eight-wave 256 x 256 x 64 tile of ifx_gemm_glds.hip runs its two waves per SIMD in LOCK
// STEP: after the one barrier per K-step both issue their 8 LDS-DMA pieces (~60 cycles each, serialised per SIMD: ~1000 cycles with
// nothing on the matrix pipe), then both read fragments and wait for LDS (four times per K-step), then both want the matrix pipe.
// ISA of that loop: vmcnt(0) / barrier / 8 x DMA / 4 x (6 ds_read_b128, lgkmcnt(0), 8 MFMA) = ~3660 cycles per K-step for 2048 cycles
// of MFMA (SQ_VALU_MFMA_BUSY 0.45-0.57).  Here, per SIMD and per K-step g (two phases, one barrier each):
//
//   phase 2g   : group 0 (waves 0-3)  32 MFMAs of K-step g from registers      | group 1 (waves 4-7)  DMA issue, fragment reads of g
//   phase 2g+1 : group 0  DMA issue, fragment reads of K-step g+1              | group 1  32 MFMAs of K-step g
//
//   * wave tile 32*TJ (tokens) x 64 (channels); group 0 owns the upper token half of the tile, group 1 the lower one, the four waves
//     of a group the four channel quarters.  ALL fragments of a K-step (16 x + 8 W ds_read_b128 = 96 VGPRs) are read in the wave's
//     loader phase: the MFMA phase contains nothing but MFMAs and an LDS slot is free one phase after its last reader.
//   * LDS = two K-step stages of four 16 KiB slots (x-upper, W-lo, W-hi, x-lower) + 2 x 16 KiB of transpose scratch for the epilogues.
//     Group 0 requests W (lo, hi of K-step g+1 in phase 2g-1), group 1 requests x (x-lower of g+1 and x-upper of g+2 in phase 2g): eight
//     1 KiB pieces per wave and phase, every piece two to three phases (>= 2 x 1024 matrix-pipe cycles) ahead of its first reader,
//     64-96 KiB in flight per CU; waits are COUNTED on the requesting wave (vmcnt(2 TJ) / vmcnt(TJ) / vmcnt(0) = "my pieces of the
//     slot the next phase reads").  DMA from inline asm (hipcc would fence every later ds_read with vmcnt(0), see ifx_attn_pp.hip).
//   * persistent: the request stream runs across output tiles (the first K-steps of the next tile are in flight while the last ones
//     of this tile are multiplied).  BOTH groups run the epilogue of a tile in the phase of group 1's last MFMAs (group 0 under those
//     MFMAs, group 1 right behind them), each through its own scratch; bias / residual / gate are fetched by asm loads with counted
//     waits, so the only compiler-visible vector-memory operations in the kernel are the output stores (hipcc's own vmcnt waits
//     would otherwise drain the request stream: 6000 cycles per epilogue in the first version, profiles/r3_gemm_pp.md).
//   * layout, swizzle (16-byte chunk XOR ((row >> 1) & 7)), transposed MFMA tile (A = W rows, B = x rows: a lane owns 4 consecutive
//     channels of one token), K order and the LDS-transposed epilogue are those of the other tiles: outputs are bit-identical.
#include <stdlib.h>

#include <type_traits>

#include "ifx_common.h"

#ifndef IFX_PP_TRACE
#define IFX_PP_TRACE 0      // 1: s_memtime segment sums of workgroup 0 (tools/gemm_lab.cpp -t reads them back)
#endif
#ifndef IFX_PP_LAB
#define IFX_PP_LAB IFX_PP_TRACE   // 1: the IFX_PP_DEBUG lab switches (bits 1 / 2 / 4 / 8 / 16) are live; 0: they are compile-time zeros — their run-time
#endif                            // tests (a K rotation select per request block, a flag test per phase) sat in front of every LDS-DMA request block


namespace ifx {

struct EpiArgsP {
  const unsigned short* bias;
  const unsigned short* residual;
  int ld_res;
  const unsigned short* mod;
  int mod_slots, gate_slot, rows_per_group;
  // 8-bit operands (Q8 instantiations): per-token / per-channel dequantisation scales, and (GELU epilogues) an optional static quantiser
  // of the result for the next linear (ifx_gemm_q8_quant_out): y then holds e4m3 bytes, ldy in bytes
  const float* sa = nullptr;
  const float* sw = nullptr;
  const float* qdiv = nullptr;
  int q_via_bf16 = 0;
  // second destination (ifx_epilogue.y2): column tiles from split_col on are stored to y2 (row stride ldy2) at column n - split_col
  unsigned short* y2 = nullptr;
  int ldy2 = 0, split_col = 0;
};

namespace gpp {
constexpr int BN = 256, BK = 64;
constexpr int SLOT = 16384, STAGE = 4 * SLOT, SCRATCH = 2 * STAGE;      // two 64 KiB stages + 2 x 16 KiB epilogue scratch = 160 KiB
constexpr int LDS_BYTES = SCRATCH + 32768;
constexpr int X_UP = 0, W_LO = SLOT, W_HI = 2 * SLOT, X_DN = 3 * SLOT;  // slots of a stage
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(v4i rsrc, unsigned lds, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}
__device__ __forceinline__ v4i make_rsrc(const void* base, unsigned num_bytes) {
  const unsigned long long a = (unsigned long long)base;
  v4i r;
  r[0] = (int)(unsigned)a;
  r[1] = (int)((unsigned)(a >> 32) & 0xffffu);      // stride 0: raw buffer, byte offsets, reads past num_bytes return 0
  r[2] = (int)num_bytes;
  r[3] = 0x00020000;
  return r;
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// lgkmcnt(0) through the builtin (simm16: vmcnt 63, expcnt 7, lgkmcnt 0), so that hipcc's own scoreboard knows the fragment reads have
// returned: after an asm wait it keeps protecting their registers with lgkmcnt(14 .. 0) in the middle of the next batch of reads
__device__ __forceinline__ void wait_lds() { __builtin_amdgcn_s_waitcnt(0xC07F); }

// one request cursor: K-step kt of tile i (the g-th K-step of this workgroup's stream), descriptor of that tile's operand rows
struct Cursor {
  int i, kt, g, rot, k0, len;     // k0 / len: first K-step and number of K-steps of the item inside the tile's K range (split-K, stream-K)
  int kb;                         // byte offset of the cursor's K-step in an operand row, (k0 + kt) * 128: advanced, not recomputed per request block
  unsigned stg;                   // LDS byte offset of the cursor's stage, (g & 1) * STAGE: toggled
  v4i rs;
};
}  // namespace gpp

// Q8: e4m3 operands (one byte per element: a K-step is 128 elements, the same 128-byte rows), two v_mfma_scale_f32_32x32x64_f8f6f4 with
// unit block scales per accumulator and K-step in place of four v_mfma_f32_32x32x16_bf16 — the same 1024 matrix-pipe cycles per phase for
// twice the arithmetic — and the dequantisation acc * (sa[m] * sw[n]) in the epilogue (ifx_gemm_q8's contract).  The K order inside a
// fragment is whatever the hardware uses: both operands are read with the same lane -> byte map.
// Q8 = 2: int8 operands, four v_mfma_i32_32x32x32_i8 per accumulator and K-step (the bf16 loop with another instruction; exact int32 sums kept
// as bit patterns in the accumulator registers), the same dequantising epilogue.
template <int EPI, int TJ, int KS, int Q8 = 0>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const unsigned short* __restrict__ x, int ldx,
                                                         const unsigned short* __restrict__ w, unsigned short* __restrict__ y,
                                                         int ldy, int M, int N, int K, int tiles_m, int total, int per_xcd,
                                                         int wg_per_xcd, EpiArgsP ea, unsigned long long* __restrict__ trace, int dbg,
                                                         float* __restrict__ ws_part, unsigned* __restrict__ ws_flag,
                                                         unsigned* __restrict__ err_word, long long spin_ticks) {
  using namespace gpp;
  dbg = IFX_PP_LAB ? dbg : (dbg & 32);   // (bit 32 = the split-K fault injection of the tests, outside the K loop)
  constexpr int BM = 64 * TJ;               // tokens per tile: two groups x TJ blocks of 32
  constexpr int GM = 4;                     // row tiles per rasterisation group (see ifx_gemm_glds.hip)
  constexpr bool RES = EPI == IFX_EPI_RESIDUAL || EPI == IFX_EPI_GATE_RES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;
  // KS = 2: every tile is TWO work items, the K halves, on two workgroups (neighbouring ids: same XCD, same position of their lists).
  // The second half's workgroup leaves its fp32 accumulators in the workspace (write-through stores) and raises the tile's flag; the
  // first half's workgroup adds them to its own — always first + second — and runs the epilogue.  `total` counts ITEMS.
  // KS = 4, 8 (round 5; the shard-sized launches of a sequence-parallel rank): KS items per tile, parts 1 .. KS-1 each dump into their
  // own slot (tile * (KS-1) + part - 1) and raise their own flag, part 0 waits for all of them and adds them IN PART ORDER
  // (((p0 + p1) + p2) + ...: one fixed summation order, whoever finishes last).  Nobody waits for a workgroup that waits: owners wait
  // for producers with LARGER item ids only, and every workgroup walks its ids upwards (wg_per_xcd >= KS or one item per workgroup).
  constexpr int ES = Q8 ? 1 : 2;            // bytes per operand element
  const unsigned char* const xb = reinterpret_cast<const unsigned char*>(x);
  const unsigned char* const wb = reinterpret_cast<const unsigned char*>(w);
  // KS = 0: STREAM-K.  The launch's K-steps (tiles x S, tile-major) are cut into gridDim.x equal contiguous ranges; a workgroup's range
  // is the tail of one tile (K-steps k0 .. S: multiplied FIRST, accumulators dumped to the workgroup's slot of the workspace), whole
  // tiles, and the head of one more tile (K-steps 0 .. k1: multiplied LAST; this workgroup owns that tile, adds the slots of the
  // workgroups after it that hold the rest of the tile's K range — dumped long before, at the start of their runs — in ascending K
  // order and runs the epilogue).  Nobody waits for a workgroup that waits.  The K partition depends on the row count of the launch:
  // only callers that opted into that (ifx_set_option "gemm_small_split": the sequence-parallel shard sizes) get this path.
  constexpr bool SK = KS == 0;
  constexpr int KSD = KS > 0 ? KS : 1;
  const int S = K * ES / 128;               // K-steps (128 bytes of every operand row) of a whole tile
  const int KT = S / KSD;          // K-steps of one work item (stream-K: of a whole tile; items are shorter)
  const int tiles_n = total / KSD / tiles_m;

  // ---- this workgroup's tiles: XCD (bid & 7) owns ids [xcd * per_xcd, ...); its wg_per_xcd workgroups take them round-robin, so the
  //      workgroups resident on one XCD always work on consecutive ids = a GM x (32 / GM) block of tiles sharing operand panels in L2
  const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
  const int id_first = xcd * per_xcd + slot_i;
  const int id_end = min(xcd * per_xcd + per_xcd, total);
  if (!SK && id_first >= id_end) return;
  // stream-K: workgroup sk_p of sk_P (the workgroups of one XCD hold neighbouring ranges), K-steps [sk_u0, sk_u1) of total * S
  const int sk_P = (int)gridDim.x, sk_p = xcd * (sk_P >> 3) + slot_i;
  auto sk_start = [&](int q) __attribute__((always_inline)) { return (int)((long)q * total * S / sk_P); };
  const int sk_u0 = SK ? sk_start(sk_p) : 0, sk_u1 = SK ? sk_start(sk_p + 1) : 0;
  const int sk_t0 = SK ? sk_u0 / S : 0;
  const int n_my = SK ? (sk_u1 - 1) / S - sk_t0 + 1 : (id_end - id_first + wg_per_xcd - 1) / wg_per_xcd;
  const int G = SK ? sk_u1 - sk_u0 : n_my * KT;       // K-steps of the whole request stream
  auto item_split = [&](int i) __attribute__((always_inline)) { return KS > 1 ? (id_first + i * wg_per_xcd) % KSD : 0; };
  auto item_tile = [&](int i) __attribute__((always_inline)) { return SK ? sk_t0 + i : (id_first + i * wg_per_xcd) / KSD; };
  auto item_k0 = [&](int i) __attribute__((always_inline)) { return SK ? (i == 0 ? sk_u0 - sk_t0 * S : 0) : item_split(i) * KT; };
  auto item_len = [&](int i) __attribute__((always_inline)) {
    return SK ? min(S, sk_u1 - (sk_t0 + i) * S) - (i == 0 ? sk_u0 - sk_t0 * S : 0) : KT;
  };
  // an item whose accumulators go to the workspace instead of through the epilogue / the number of later partial sums its epilogue adds
  auto item_dumps = [&](int i) __attribute__((always_inline)) { return SK ? (i == 0 && sk_u0 > sk_t0 * S) : (KS > 1 && item_split(i) >= 1); };
  auto item_parts = [&](int i) __attribute__((always_inline)) {
    if (!SK) return KS > 1 ? KS - 1 : 0;
    const int t_end = (sk_t0 + i + 1) * S;            // first K-step of the next tile
    int n = 0;
    while (sk_p + 1 + n < sk_P && sk_start(sk_p + 1 + n) < t_end) ++n;
    return sk_u1 < t_end ? n : 0;
  };
  auto tile_base = [&](int i, int& m_base, int& n_base) __attribute__((always_inline)) {
    const int t_id = item_tile(i);
    const int grp_sz = GM * tiles_n;
    const int first_m = (t_id / grp_sz) * GM;
    const int gm = min(GM, tiles_m - first_m);
    const int rem = t_id % grp_sz;
    m_base = (first_m + rem % gm) * BM;
    n_base = (rem / gm) * BN;
  };

  // ---- loader side.  A piece = one wave instruction = 8 rows x 128 B, lane -> row lane >> 3, PHYSICAL 16-byte chunk lane & 7, which
  //      holds LOGICAL chunk (lane & 7) ^ ((row >> 1) & 7).  Wave w4 of the loading group moves pieces q * 4 + w4 of a slot: rows
  //      (q * 4 + w4) * 8 + (lane >> 3) — the swizzle phase does not depend on q, so ONE voffset per wave; the piece, the K-step and
  //      the tile half are scalar offsets.  Rows past M / N read zeros through the descriptor's bound.
  const int r8 = lane >> 3, pc = lane & 7;
  const int prow = w4 * 8 + r8;
  const int ld_op = grp == 0 ? K : ldx;              // row pitch of the operand this group requests (W / x)
  const int voff = prow * ld_op * ES + ((pc ^ ((prow >> 1) & 7)) << 4);
  const unsigned lds_piece0 = (unsigned)(unsigned long long)(lds_ptr_t)smem + w4 * 1024;
  auto cur_desc = [&](Cursor& c) __attribute__((always_inline)) {                   // descriptor (and lab rotation) of tile c.i
    int mb, nb;
    tile_base(c.i, mb, nb);
    if (dbg & 1) mb = nb = 0;                        // lab: every workgroup streams tile 0's operands (L2-hot), timing only
    c.rot = (!SK && (dbg & 4)) ? ((nb / BN) * 3) % KT : 0;    // lab: K rotation by column tile
    c.k0 = item_k0(c.i);
    c.len = item_len(c.i);
    c.kb = c.k0 * 128;
    if (grp == 0) c.rs = make_rsrc(wb + (size_t)nb * K * ES, (unsigned)min(((long)N - nb) * (long)K * ES, 0xffffffffL));
    else c.rs = make_rsrc(xb + (size_t)mb * ldx * ES, (unsigned)min(((long)M - mb - 1) * (long)ldx * ES + (long)K * ES, 0xffffffffL));
  };
  auto cur_init = [&](Cursor& c) __attribute__((always_inline)) {
    c.i = 0, c.kt = 0, c.g = 0, c.stg = 0;
    cur_desc(c);
  };
  auto cur_next = [&](Cursor& c) __attribute__((always_inline)) {
    ++c.g;
    c.stg ^= STAGE;
    c.kb += 128;
    if (++c.kt == c.len) {
      c.kt = 0;
      if (++c.i < n_my) cur_desc(c);
    }
  };
  // `pieces` 1 KiB pieces of the cursor's K-step: rows row0 + (q * 4 + w4) * 8 ... of the tile's operand -> slot `slot` of its stage
  auto issue = [&](const Cursor& c, int slot, int row0, int pieces) __attribute__((always_inline)) {
    // (every scalar instruction in front of a request block lengthens the loader phase: the K-step offset and the stage are running
    //  values of the cursor; the lab rotation keeps the recomputed form)
    const int kk = c.kt + c.rot;
    const int kb = IFX_PP_LAB ? ((!SK && kk >= KT ? kk - KT : kk) + c.k0) * 128 : c.kb;
    const unsigned p = lds_piece0 + c.stg + slot;
    if ((dbg & 8) && c.g >= 2) return;               // lab: no DMA after the first two K-steps (what would a loader phase of reads alone cost?)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < pieces) dma16(c.rs, p + q * 4096, voff, kb + (row0 + q * 32) * ld_op * ES);
  };
  Cursor ca, cb;                                     // group 0: ca = W.  group 1: ca = x-lower, cb = x-upper (one K-step ahead of ca)
  cur_init(ca);
  cur_init(cb);

  // ---- reader side: fragment row = block base + (lane & 31), logical chunk 2 ks + (lane >> 5)
  const int l31 = lane & 31, hi = lane >> 5;
  int lane_ks[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) lane_ks[ks] = l31 * 128 + (((2 * ks + hi) ^ ((l31 >> 1) & 7)) << 4);
  const int w_frag_off = (w4 >> 1) * SLOT + (w4 & 1) * 8192 + W_LO;     // this wave's 64 W rows: W-lo (waves 0, 1) / W-hi (waves 2, 3)
  const int x_frag_off = grp == 0 ? X_UP : X_DN;
  bf16x8 fx[4][TJ], fw[4][2];
  int rg = 0;                                        // K-step the next read_frags reads
  auto read_frags = [&]() __attribute__((always_inline)) {
    const unsigned char* st = smem + (rg & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fw[ks][i] = *reinterpret_cast<const bf16x8*>(st + w_frag_off + lane_ks[ks] + i * 4096);
#pragma unroll
      for (int j = 0; j < TJ; ++j) fx[ks][j] = *reinterpret_cast<const bf16x8*>(st + x_frag_off + lane_ks[ks] + j * 4096);
    }
    ++rg;
  };

  f32x16 acc[2][TJ];
  auto q8_mfma = [&](int kk, int i, int j, const f32x16 c) __attribute__((always_inline)) {
    const v8i a = __builtin_shufflevector(__builtin_bit_cast(v4i, fw[2 * kk][i]), __builtin_bit_cast(v4i, fw[2 * kk + 1][i]), 0, 1, 2, 3, 4, 5, 6, 7);
    const v8i b = __builtin_shufflevector(__builtin_bit_cast(v4i, fx[2 * kk][j]), __builtin_bit_cast(v4i, fx[2 * kk + 1][j]), 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  };
  auto i8_mfma = [&](int ks, int i, int j, const f32x16 c) __attribute__((always_inline)) {
    return __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(v4i, fw[ks][i]), __builtin_bit_cast(v4i, fx[ks][j]),
                                                                            __builtin_bit_cast(v16i, c), 0, 0, 0));
  };
  auto mfma_first = [&]() __attribute__((always_inline)) {                          // first K-step of a tile: the accumulators start from zero
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (Q8 == 2) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = i8_mfma(ks, i, j, ks == 0 ? z : acc[i][j]);
      return;
    }
    if constexpr (Q8 == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = q8_mfma(0, i, j, z);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = q8_mfma(1, i, j, acc[i][j]);
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[0][i], fx[0][j], z, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ks][i], fx[ks][j], acc[i][j], 0, 0, 0);
  };
  auto mfma_next = [&]() __attribute__((always_inline)) {
    if constexpr (Q8 == 2) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = i8_mfma(ks, i, j, acc[i][j]);
      return;
    }
    if constexpr (Q8 == 1) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = q8_mfma(kk, i, j, acc[i][j]);
      return;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ks][i], fx[ks][j], acc[i][j], 0, 0, 0);
  };

  // ---- epilogue of tile i_tile for THIS wave.  `between` = what the caller wants ahead of the epilogue's own memory operations in the
  //      in-order queue (this phase's DMA and the counted wait for the slot the NEXT phase reads).  Then bias (8 x 8 B per lane),
  //      residual (4 TJ x 16 B) and the gate rows (2 x 16 B) are fetched with ordinary loads — hipcc counts its waits over its own
  //      operations only, and everything older (the DMA just issued) returns first: one memory latency per epilogue, nothing drained
  //      that was requested later.  (Asm loads with hand-counted waits were tried: hipcc is free to spill or recycle the destination
  //      registers of an asm statement at once — the data of a load still in flight then lands in whatever lives there next: wild
  //      addresses in the following loads.)  32-token blocks go one at a time through the wave's 4 KiB of scratch: v = bf16(acc + bias)
  //      transposed to token-major rows of 128 B, then row-contiguous 16-byte accesses for the activation / residual / gate and the store.
  // Lane-derived addressing is re-derived from an opaque lane id: hoisted out of the K loop it would stay live across it (the loop runs
  // at ~232 VGPRs: 128 accumulators + 96 fragment registers).
  unsigned char* const tw = smem + SCRATCH + wave * 4096;
  // ---- split-K hand-off (KS = 2): a wave's accumulators as a register image, 16 B per lane and register quad, in the tile's 256 KiB of
  //      the workspace.  Write-through (sc1) stores on the producing side, sc1 loads on the consuming side, one flag per tile in between
  //      (raised once every wave of the producer has waited for its stores: vmcnt(0) + the workgroup barrier).
  const __amdgpu_buffer_rsrc_t rs_ws =
      __builtin_amdgcn_make_buffer_rsrc((void*)ws_part, 0,
                                        KS != 1 ? (int)min((long)(SK ? sk_P : total / KSD * (KSD - 1)) * BM * BN * 4L, 0x7fffffffL) : 0, 0x00020000);
  // slot of the workspace an item dumps to (split-K: part p >= 1 of tile t owns slot t (KS-1) + p - 1; stream-K: the workgroup's) / the
  // first slot its epilogue reads (the owner, part 0: slot t (KS-1))
  auto part_slot = [&](int i) __attribute__((always_inline)) {
    return SK ? sk_p : item_tile(i) * (KSD - 1) + (item_split(i) > 0 ? item_split(i) - 1 : 0);
  };
  auto part_soff = [&](int i) __attribute__((always_inline)) { return part_slot(i) * (BM * BN * 4) + wave * (TJ * 8192); };
  auto flag_of = [&](int i) __attribute__((always_inline)) { return part_slot(i); };
  auto dump_partial = [&](int i) __attribute__((always_inline)) {
    int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(ln));
    const int so = part_soff(i);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[ii][j][4 * q], acc[ii][j][4 * q + 1], acc[ii][j][4 * q + 2], acc[ii][j][4 * q + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_ws, ln * 16 + ((ii * TJ + j) * 4 + q) * 1024, so, 16);
        }
  };
  int flag_pending = 0;                              // 1 + tile whose partial this workgroup has dumped and not yet announced
  auto consumer_sync = [&](int i) __attribute__((always_inline)) {              // all eight waves: the partial sums the epilogue adds are complete and visible
    const int first = SK ? sk_p + 1 : item_tile(i) * (KSD - 1), n = item_parts(i);
    if (wave == 0) {
      // BOUNDED (round-4 verdict): a producer that is not resident (a grid larger than the chip admits, a preempted queue)
      // would otherwise hang the GPU.  After `spin_ticks` of the 100 MHz wall clock the wait gives up, raises the device error word
      // (ifx_last_error / ifx_device_error report it) and the launch runs to its end on whatever the workspace holds.
      const long long t0 = wall_clock64();
      bool gave_up = false;
      for (int q = 0; q < n && !gave_up; ++q)
        while (__hip_atomic_load(ws_flag + first + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          __builtin_amdgcn_s_sleep(4);
          if (wall_clock64() - t0 > spin_ticks) {
            gave_up = true;
            if (err_word != nullptr)
              __hip_atomic_store(err_word, (1u << 24) | ((unsigned)(first + q) & 0xffffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
          }
        }
    }
    __builtin_amdgcn_s_barrier();
    // s_barrier is IntrNoMem to hipcc: without this the partial-sum loads of the epilogue (raw_buffer_load, plain memory reads to the
    // compiler) could be scheduled above the spin loop that makes them valid
    asm volatile("" ::: "memory");
    if (wave == 0)
      for (int q = 0; q < n; ++q) __hip_atomic_store(ws_flag + first + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left as found: zero
  };
  auto epilogue = [&](int i_tile, auto between) __attribute__((always_inline)) {
    between();
    int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(ln));
    const int l31 = ln & 31, hi = ln >> 5, rr = ln >> 3, cc = ln & 7;
    int m_base, n_base;
    tile_base(i_tile, m_base, n_base);
    const int part_so = KS != 1 ? part_soff(i_tile) + (SK ? BM * BN * 4 : 0) : 0;      // stream-K: the slots of the workgroups after this one
    const int part_n = KS != 1 ? item_parts(i_tile) : 0;
    const int e_n0 = min(n_base + w4 * 64, N - 64);   // a wave whose 64 channels lie past N (N % 64 == 0) fetches valid addresses, stores nothing
    const bool n_ok = n_base + w4 * 64 < N;
    const int e_m0 = m_base + grp * (32 * TJ);
    const bool has_bias = ea.bias != nullptr;
    const unsigned short* const bias_p = has_bias ? ea.bias : y;      // any valid address: the values are not used without a bias
    const unsigned bias_mask = has_bias ? 0xffffffffu : 0u;
    // PRIV (8-bit operands, residual / gate epilogue, 256 tokens): the 48 registers of bias and weight scales do not fit next to 128
    // accumulators, the residual double buffer and the gate rows — hipcc then spills 36-164 bytes per lane and picks loop-carried
    // values for it (the K loop ran 4x slower: 326 us on the FFN down-projection).  There lane L keeps the bias and the scales of ONE
    // register quad, (L & 7), for its own channel half — 6 registers — and every lane fetches quad (i, g)'s from lane
    // (L & 32) | (i * 4 + g) with ds_bpermute when it gets there (48 crossbar reads per 32-token block; same values, same bits).
    // (Parking the vectors in private memory was tried: volatile reads are waited for one by one, + 40 us per tile; plain ones
    //  are hoisted back into registers by the scheduler and spill again.)
    constexpr bool PRIV = Q8 != 0 && RES && TJ == 4;
    u32x2 e_bias[2][4];
    int q_bias[2] = {0, 0}, q_sw[4] = {0, 0, 0, 0};   // PRIV: the quad (ln & 7) only, as separate scalars
    if constexpr (PRIV) {
      const u32x2 b = *reinterpret_cast<const u32x2*>(bias_p + e_n0 + (ln & 7) * 8 + hi * 4);
      q_bias[0] = (int)(b[0] & bias_mask), q_bias[1] = (int)(b[1] & bias_mask);
      const u32x4 sq = *reinterpret_cast<const u32x4*>(ea.sw + e_n0 + (ln & 7) * 8 + hi * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) q_sw[e] = (int)sq[e];
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // no bias: the loaded bits are masked to +0.0 and added all the same (a branch per register quad costs more than 16 v_and)
          const u32x2 b = *reinterpret_cast<const u32x2*>(bias_p + e_n0 + i * 32 + g * 8 + hi * 4);
          e_bias[i][g] = u32x2{b[0] & bias_mask, b[1] & bias_mask};
        }
    }
    f32x4 e_sw[Q8 ? 2 : 1][Q8 ? 4 : 1];               // Q8: per-channel weight scales of the lane's channels, per-token scales of its TJ tokens
    float e_sa[Q8 ? TJ : 1];
    f32x4 e_qd[2];                                    // quantised output: the divisors of the 8 channels this lane stores per row
    if constexpr (Q8) {
      if constexpr (!PRIV) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) e_sw[i][g] = *reinterpret_cast<const f32x4*>(ea.sw + e_n0 + i * 32 + g * 8 + hi * 4);
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) e_sa[j] = ea.sa[min(e_m0 + j * 32 + l31, M - 1)];
      if constexpr (EPI == IFX_EPI_GELU_TANH) {
        const float* const qd = ea.qdiv != nullptr ? ea.qdiv : ea.sw;       // any valid address when the output is bf16
        e_qd[0] = *reinterpret_cast<const f32x4*>(qd + e_n0 + cc * 8);
        e_qd[1] = *reinterpret_cast<const f32x4*>(qd + e_n0 + cc * 8 + 4);
      }
    }
    u32x4 e_res[2][4], e_gate[2];                     // residual rows of two 32-token blocks: the next block's are in flight under this one's work
    int e_split = 0;                                  // first token of the wave's second gate group
    auto fetch_res = [&](int j) __attribute__((always_inline)) {
      if constexpr (RES) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int m = min(e_m0 + j * 32 + p * 8 + rr, M - 1);      // rows past M: any valid address, the store is masked
          e_res[j & 1][p] = *reinterpret_cast<const u32x4*>(ea.residual + (size_t)m * ea.ld_res + e_n0 + cc * 8);
        }
      }
    };
    if constexpr (RES) {
      fetch_res(0);
      if constexpr (EPI == IFX_EPI_GATE_RES) {
        // the wave's 32 TJ tokens lie in at most two gate groups (rows_per_group >= 32 TJ, checked by the launcher)
        const int m_hi = min(e_m0 + 32 * TJ - 1, M - 1);
        const int g_lo = min(e_m0, M - 1) / ea.rows_per_group, g_hi = m_hi / ea.rows_per_group;
        e_split = (g_lo + 1) * ea.rows_per_group;
        e_gate[0] = *reinterpret_cast<const u32x4*>(ea.mod + ((size_t)g_lo * ea.mod_slots + ea.gate_slot) * N + e_n0 + cc * 8);
        e_gate[1] = *reinterpret_cast<const u32x4*>(ea.mod + ((size_t)g_hi * ea.mod_slots + ea.gate_slot) * N + e_n0 + cc * 8);
      }
    }
    unsigned char* const wr = tw + l31 * 128 + hi * 8;
    const unsigned char* const rd = tw + rr * 128;
    // (wave-uniform: a wave's 64 channels lie on one side of split_col, a multiple of 256)
    const bool to_y2 = ea.y2 != nullptr && e_n0 >= ea.split_col;
    const int ldo = to_y2 ? ea.ldy2 : ldy;
    unsigned short* const yrow = (to_y2 ? ea.y2 - ea.split_col : y) + (size_t)(e_m0 + rr) * ldo + e_n0 + cc * 8;
    // Every loaded register is TOUCHED (an empty asm that reads it) on every path: hipcc waits for a load where its value is used, and a
    // use that it sinks into a branch (the masked store of a ragged tile) leaves the load "maybe pending" at the loop back-edge — it then
    // protects the registers with vmcnt(3 .. 0) in the middle of the NEXT K-step's fragment reads, which drains the whole DMA queue
    // (measured: the loader phase 1100 -> 1900 cycles in the residual / gate instantiations).
    if constexpr (PRIV) {
      asm volatile("" : "+v"(q_bias[0]), "+v"(q_bias[1]), "+v"(q_sw[0]), "+v"(q_sw[1]), "+v"(q_sw[2]), "+v"(q_sw[3]));
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(e_bias[i][g]));
    }
    if constexpr (Q8) {
      if constexpr (!PRIV) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(e_sw[i][g]));
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) asm volatile("" : "+v"(e_sa[j]));
      if constexpr (EPI == IFX_EPI_GELU_TANH) asm volatile("" : "+v"(e_qd[0]), "+v"(e_qd[1]));
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      if (j + 1 < TJ) fetch_res(j + 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          if constexpr (Q8 == 2 && KS <= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (float)__builtin_bit_cast(int, v[e]);                  // the exact int32 sum
          }
          if constexpr (KS > 1) {
            // + the partners' partial sums of the later K ranges, read where the accumulators are consumed (writing them back into
            // the accumulator vectors first costs hipcc ~300 spilled registers): ((first + second) + third) ..., then the bias.  The
            // KS - 1 loads of a register quad are independent and requested together (compile-time trip count).
            f32x4 pq[KSD - 1];
#pragma unroll
            for (int q = 0; q < KSD - 1; ++q)
              pq[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ws, ln * 16 + ((i * TJ + j) * 4 + g) * 1024,
                                                                                      part_so + q * (BM * BN * 4), 16));
            if constexpr (Q8 == 2) {                   // int8: all parts are exact int32 sums (bit patterns): the split does not change a bit
              int vi[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) vi[e] = __builtin_bit_cast(int, v[e]);
#pragma unroll
              for (int q = 0; q < KSD - 1; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float pe = pq[q][e];           // (bit_cast of a vector ELEMENT expression reads element 0 whatever e is: hipcc 7.0)
                  vi[e] += __builtin_bit_cast(int, pe);
                }
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (float)vi[e];
            } else {
#pragma unroll
              for (int q = 0; q < KSD - 1; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += pq[q][e];
            }
          }
          if constexpr (SK) {                          // the later K ranges of the tile, in ascending K order
            for (int q = 0; q < part_n; ++q) {
              const f32x4 pq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                             rs_ws, ln * 16 + ((i * TJ + j) * 4 + g) * 1024, part_so + q * (BM * BN * 4), 16));
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += pq[e];
            }
          }
          if constexpr (Q8) {
            f32x4 swv;
            if constexpr (PRIV) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                swv[e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((ln & 32) | (i * 4 + g)) << 2, q_sw[e]));
            } else {
              swv = e_sw[i][g];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __fmul_rn(v[e], __fmul_rn(e_sa[j], swv[e]));   // not contracted with the bias add
          }
          u32x2 b;
          if constexpr (PRIV) {
            b[0] = (unsigned)__builtin_amdgcn_ds_bpermute(((ln & 32) | (i * 4 + g)) << 2, q_bias[0]);
            b[1] = (unsigned)__builtin_amdgcn_ds_bpermute(((ln & 32) | (i * 4 + g)) << 2, q_bias[1]);
          } else {
            b = e_bias[i][g];
          }
          v[0] += __builtin_bit_cast(float, b[0] << 16);
          v[1] += __builtin_bit_cast(float, b[0] & 0xffff0000u);
          v[2] += __builtin_bit_cast(float, b[1] << 16);
          v[3] += __builtin_bit_cast(float, b[1] & 0xffff0000u);
          u16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
          // channel nl = i * 32 + g * 8 + hi * 4 of token l31: 16-byte chunk (i * 4 + g) XOR (l31 & 7), half hi
          *reinterpret_cast<u16x4*>(wr + (((i * 4 + g) ^ (l31 & 7)) << 4)) = o;
        }
      }
      wait_lds();                                            // wave-private region: no barrier
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int mrow = p * 8 + rr;
        const int m = e_m0 + j * 32 + mrow;
        const u16x8 vv = *reinterpret_cast<const u16x8*>(rd + p * 1024 + ((cc ^ (mrow & 7)) << 4));
        if constexpr (RES) asm volatile("" : "+v"(e_res[j & 1][p]));
        if constexpr (EPI == IFX_EPI_GATE_RES) {
          if (j == 0 && p == 0) asm volatile("" : "+v"(e_gate[0]), "+v"(e_gate[1]));
        }
        u16x8 o;
        if constexpr (EPI == IFX_EPI_BIAS) {
          o = vv;
        } else if constexpr (EPI == IFX_EPI_GELU_TANH) {
          if (ea.gate_slot) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(gelu_erf_f(bf2f(vv[e])));
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(gelu_tanh_fast(bf2f(vv[e])));
          }
        } else {
          const u16x8 rv = __builtin_bit_cast(u16x8, e_res[j & 1][p]);
          if constexpr (EPI == IFX_EPI_RESIDUAL) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(rv[e]) + bf2f(vv[e]));
          } else {
            const u16x8 gv = __builtin_bit_cast(u16x8, m >= e_split ? e_gate[1] : e_gate[0]);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(rv[e]) + rbf(bf2f(vv[e]) * bf2f(gv[e])));
          }
        }
        if constexpr (Q8 && EPI == IFX_EPI_GELU_TANH) {
          if (ea.qdiv != nullptr) {                    // wave-uniform: the result leaves as the next linear's e4m3 input
            unsigned wq[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              float t[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float q = fminf(fmaxf(bf2f(o[4 * h + e]) / e_qd[h][e], -448.0f), 448.0f);
                t[e] = ea.q_via_bf16 ? rbf(q) : q;
              }
              unsigned pk = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], 0u, false);
              wq[h] = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], pk, true);
            }
            if (m < M && n_ok)
              *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(y) + (size_t)(e_m0 + rr + j * 32 + p * 8) * ldy + e_n0 + cc * 8) =
                  u32x2{wq[0], wq[1]};
            continue;
          }
        }
        if (m < M && n_ok) *reinterpret_cast<u16x8*>(yrow + (size_t)(j * 32 + p * 8) * ldo) = o;
      }
      // (no wait here: the LDS operations of one wave execute in order, the next block's writes cannot overtake these reads)
    }
  };

#if IFX_PP_TRACE
  // six segment sums per group (cycles), kept in scalar registers and written once: per-stamp stores would queue behind the DMA
  // stream and perturb exactly what is measured
  unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_readcyclecounter();
  int seg_i = 0;
#define PP_STAMP()                                                 \
  do {                                                             \
    const unsigned long long t_now = __builtin_readcyclecounter(); \
    seg[seg_i] += t_now - t_last;                                  \
    t_last = t_now;                                                \
    seg_i = seg_i == 5 ? 0 : seg_i + 1;                            \
  } while (0)
#else
#define PP_STAMP() \
  do {             \
  } while (0)
#endif
  // lab: loader waves 1 and 3 of a group read their fragments BEFORE they issue their DMA (the LDS serves two waves at a time instead of
  // four while the other two sit in their DMA issue stalls)
#ifndef IFX_PP_STAGGER
#define IFX_PP_STAGGER 0
#endif
  // (a run-time switch between the two placements INSIDE the loop costs 400 spilled registers — two definitions of the fragment set —
  //  so the whole role sequence is instantiated twice and the wave picks its copy once)
  auto roles = [&](auto stagger_tag) __attribute__((always_inline)) {
  constexpr bool reads_first = decltype(stagger_tag)::value;

  // ---- the two role sequences.  Both execute 2 G + 1 barriers.  Segments of the trace: 0 barrier behind the MFMA phase, 1 DMA issue
  //      (+ fragment reads of the staggered waves), 2 epilogue, 3 fragment reads + waits, 4 barrier behind the loader phase, 5 MFMAs
  int kt = 0, it = 0;                                // K-step within the item / item index of the K-step g being multiplied
  int klen = item_len(0);
  if (grp == 0) {
    issue(ca, W_LO, 0, 4), issue(ca, W_HI, 128, 4), cur_next(ca);          // W(0)
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();                    // -> phase -1: group 1 has x-upper(0)
    for (int g = 0; g < G; ++g) {
      // ---------------- phase 2g - 1: loader (W of K-step g + 1), epilogue of the tile that ended with K-step g - 1 ----------------
      PP_STAMP();
      const bool epi = g > 0 && kt == 0;
      const bool issued = ca.g < G;
      const bool early_reads = reads_first && !epi;
      if (epi && KS != 1 && item_dumps(it - 1)) {
        // second K half: the accumulators go to the workspace, the partner finishes the tile
        if (issued) issue(ca, W_LO, 0, 4), issue(ca, W_HI, 128, 4), cur_next(ca);
        PP_STAMP();
        dump_partial(it - 1);
        wait_vm<0>();
        flag_pending = flag_of(it - 1) + 1;
        PP_STAMP();
        read_frags();
      } else if (epi) {
        if (KS != 1) consumer_sync(it - 1);
        epilogue(it - 1, [&]() __attribute__((always_inline)) {
          if (issued) issue(ca, W_LO, 0, 4), issue(ca, W_HI, 128, 4), cur_next(ca);
          PP_STAMP();                                // 1: DMA issue
        });
        PP_STAMP();                                  // 2: the epilogue proper
        read_frags();                                // x-upper(g), W(g)
      } else if (early_reads) {
        read_frags();
        if (issued) issue(ca, W_LO, 0, 4), issue(ca, W_HI, 128, 4), cur_next(ca);
        PP_STAMP();
        PP_STAMP();
      } else {
        if (issued) issue(ca, W_LO, 0, 4), issue(ca, W_HI, 128, 4), cur_next(ca);
        PP_STAMP();
        PP_STAMP();
        read_frags();
      }
      wait_lds();
      __builtin_amdgcn_sched_barrier(0);
      PP_STAMP();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- phase 2g: MFMA ----------------
      PP_STAMP();
      if (KS != 1 && flag_pending) {                  // every wave waited for its dump before the barrier above
        if (wave == 0 && !(dbg & 32)) __hip_atomic_store(ws_flag + flag_pending - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag_pending = 0;
      }
      if (!(dbg & 2)) {
        if (kt == 0) mfma_first();
        else mfma_next();
      }
      if (++kt == klen) kt = 0, ++it, klen = it < n_my ? item_len(it) : 0;
      wait_vm<0>();                                  // W(g+1) (requested one phase ago) landed for the next phase's readers
      __builtin_amdgcn_sched_barrier(0);
      PP_STAMP();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    // phase 2G - 1: the last item, under group 1's last MFMAs
    if (KS != 1 && item_dumps(it - 1)) {
      dump_partial(it - 1);
      wait_vm<0>();
      flag_pending = flag_of(it - 1) + 1;
    } else {
      if (KS != 1) consumer_sync(it - 1);
      epilogue(it - 1, [&]() __attribute__((always_inline)) {});
    }
    if (KS != 1) {
      __builtin_amdgcn_s_barrier();                  // group 1's dump of the last item is complete as well
      if (flag_pending && wave == 0 && !(dbg & 32)) __hip_atomic_store(ws_flag + flag_pending - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    // x-upper(0); x-lower(0), x-upper(1): the x-upper cursor runs one K-step ahead of the x-lower cursor
    issue(cb, X_UP, 0, TJ), cur_next(cb);
    issue(ca, X_DN, 32 * TJ, TJ), cur_next(ca);
    if (cb.g < G) issue(cb, X_UP, 0, TJ), cur_next(cb);
    if (G > 1) wait_vm<2 * TJ>();                    // x-upper(0) landed
    else wait_vm<TJ>();
    __builtin_amdgcn_s_barrier();                    // -> phase -1 (group 0 reads K-step 0)
    if (G > 1) wait_vm<TJ>();                        // x-lower(0) landed
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                    // -> phase 0
    for (int g = 0; g < G; ++g) {
      // ---------------- phase 2g: loader (x-lower of K-step g + 1, x-upper of g + 2) ----------------
      PP_STAMP();
      const bool lower = ca.g < G, upper = cb.g < G;
      if (reads_first) {
        read_frags();
        if (lower) issue(ca, X_DN, 32 * TJ, TJ), cur_next(ca);
        if (upper) issue(cb, X_UP, 0, TJ), cur_next(cb);
        PP_STAMP();
        PP_STAMP();
      } else {
        if (lower) issue(ca, X_DN, 32 * TJ, TJ), cur_next(ca);
        if (upper) issue(cb, X_UP, 0, TJ), cur_next(cb);
        PP_STAMP();
        PP_STAMP();
        read_frags();                                // x-lower(g), W(g)
      }
      wait_lds();
      // x-upper(g+1) (requested two phases ago) has to have landed before group 0's phase: everything but this phase's requests
      if (upper) wait_vm<2 * TJ>();
      else if (lower) wait_vm<TJ>();
      else wait_vm<0>();
      __builtin_amdgcn_sched_barrier(0);
      PP_STAMP();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- phase 2g + 1: MFMA, then the epilogue if the tile is complete ----------------
      PP_STAMP();
      if (!(dbg & 2)) {
        if (kt == 0) mfma_first();
        else mfma_next();
      }
      if (++kt == klen) kt = 0, ++it, klen = it < n_my ? item_len(it) : 0;
      // x-lower(g+1) landed for this group's next phase: all but the x-upper(g+2) pieces behind it
      if (upper) wait_vm<TJ>();
      else wait_vm<0>();
      __builtin_amdgcn_sched_barrier(0);
      PP_STAMP();
      if (kt == 0) {                                 // K-step g closed item it - 1
        if (KS != 1 && item_dumps(it - 1)) {
          dump_partial(it - 1);
          wait_vm<0>();
        } else {
          if (KS != 1) consumer_sync(it - 1);
          epilogue(it - 1, [&]() __attribute__((always_inline)) {});
        }
      }
      if (g + 1 < G || KS != 1) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  };
  if (IFX_PP_STAGGER && (w4 & 1)) roles(std::true_type{});
  else roles(std::false_type{});
#if IFX_PP_TRACE
  if (trace != nullptr && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) trace[grp * 8 + k] = seg[k];
    trace[grp * 8 + 6] = (unsigned long long)G;
  }
#endif
}

// (Row invariance holds up to 1024 tiles of a launch — 43 690 rows at N = 1536, five times the largest block of the path — which is what
//  the flag page and a workspace of at most 256 MiB cover; a larger launch runs unsplit, i.e. with the other summation order.)
// Split K in two for long-K launches with few column tiles (the block's FFN down-projection, 1536 x 8960: 114 tiles of 256 x 256 for
// 256 CUs).  Depends on (N, K) ONLY: a row's summation order must not change with the number of rows in the launch.
bool gemm_pp_split(int N, int K) {
  static int min_k = -1;
  if (min_k < 0) {
    const char* e = getenv("IFX_PP_SPLIT_MIN_K");      // lab: where the two-workgroup split starts to pay
    min_k = e ? atoi(e) : 4096;
  }
  return K >= min_k && N <= 2048 && (K / 64) % 2 == 0;
}
// stream-K (128-token tile): one 128 KiB slot per workgroup behind the 4 KiB of flags
size_t gemm_pp_stream_k_workspace_bytes() { return 4096 + (size_t)256 * 128 * 256 * 4; }
// (the split runs on the 256- or the 192-token tile — whichever needs fewer rounds, pick_pp; the tile height does not change a bit of
//  the result — so the workspace covers the larger of the two tilings: 4096 bytes of flags + one fp32 tile image per tile)
size_t gemm_pp_workspace_bytes(int M, int N, int K) {
  if (!gemm_pp_split(N, K)) return 0;
  const size_t tn = (size_t)(N + 255) / 256;
  const size_t t4 = (size_t)((M + 255) / 256) * tn, t3 = (size_t)((M + 191) / 192) * tn;
  if (t4 > 1024) return 0;
  const size_t b4 = t4 * 256 * 256 * 4, b3 = t3 <= 1024 ? t3 * 192 * 256 * 4 : 0;
  return 4096 + (b4 > b3 ? b4 : b3);
}

// shard-sized launches (gemm_small_split): the 128-token tile with K split over `ks` workgroups per tile — 4096 bytes of flags + one
// fp32 tile image per (tile, part >= 1)
size_t gemm_pp_small_workspace_bytes(int M, int N, int ks) {
  if (ks <= 1) return 0;
  const size_t tiles = (size_t)((M + 127) / 128) * ((N + 255) / 256);
  return 4096 + tiles * (size_t)(ks - 1) * 128 * 256 * 4;
}

int launch_gemm_pp(const unsigned short* x, int ldx, const unsigned short* w, unsigned short* y, int ldy, int M, int N, int K,
                   int mode, const unsigned short* bias, const unsigned short* residual, int ld_res, const unsigned short* mod,
                   int mod_slots, int gate_slot, int rows_per_group, hipStream_t s, int tj, void* workspace, const float* q8_sa,
                   const float* q8_sw, const float* q8_qdiv, int q8_via_bf16, int stream_k, int q8_int8, int force_ks,
                   unsigned short* y2, int ldy2, int split_col) {
  using namespace gpp;
  const bool q8 = q8_sa != nullptr;                  // e4m3 operands: x / w point at bytes, ldx and K count elements = bytes
  EpiArgsP ea{bias, residual, ld_res, mod, mod_slots, gate_slot, rows_per_group, q8_sa, q8_sw, q8_qdiv, q8_via_bf16, y2, ldy2, split_col};
  if (y2 != nullptr && (q8 || split_col <= 0 || split_col % BN != 0 || split_col >= N || ldy2 % 8 != 0 || ((uintptr_t)y2 & 15))) {
    set_error("ifx_gemm_bf16: the second destination needs bf16 operands, 0 < split_col < N a multiple of %d, ldy2 %% 8 == 0 and a 16-byte aligned y2", BN);
    return IFX_EINVAL;
  }
  if (q8 && (K % 128 != 0 || ldx % 16 != 0 || ((uintptr_t)q8_sw & 15) || ((uintptr_t)q8_qdiv & 15))) {
    set_error("ifx_gemm_q8: the ping-pong tile needs K %% 128 == 0, ldx %% 16 == 0 and 16-byte aligned scale vectors (K = %d, ldx = %d)", K, ldx);
    return IFX_EINVAL;
  }
  if (N % 64 != 0 || K % 64 != 0) {
    set_error("ifx_gemm_bf16: the ping-pong tile needs N and K to be multiples of 64 (N = %d, K = %d)", N, K);
    return IFX_EINVAL;
  }
  if (mode == IFX_EPI_GATE_RES && rows_per_group < 32 * tj) {
    set_error("ifx_gemm_bf16: the ping-pong tile needs gate groups of at least %d rows (rows_per_group = %d)", 32 * tj, rows_per_group);
    return IFX_EINVAL;
  }
  const bool res = mode == IFX_EPI_RESIDUAL || mode == IFX_EPI_GATE_RES;
  if (((uintptr_t)bias & 7) || (res && (((uintptr_t)residual & 15) || ld_res % 8 != 0)) ||
      (mode == IFX_EPI_GATE_RES && ((uintptr_t)mod & 15))) {
    set_error("ifx_gemm_bf16: the ping-pong tile needs 8-byte aligned bias and 16-byte aligned residual / gate rows");
    return IFX_EINVAL;
  }
  // CUs of the CURRENT device (one process may drive several): the persistent grid must not exceed them, a split-K consumer spins on
  // a producer that has to be resident
  static int cu_of_dev[16] = {0};
  int dev = 0, n_cu = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) {
    if (cu_of_dev[dev] == 0) {
      int v = 0;
      cu_of_dev[dev] = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0 ? v : 256;
    }
    n_cu = cu_of_dev[dev];
  }
  if (n_cu <= 0) n_cu = 256;
  const int BM = 64 * tj;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  // split K between two workgroups per tile when a workspace is given and the shape asks for it (gemm_pp_split: a function of N and K
  // only, so that a row's bits do not depend on how many rows the launch has)
  // (the 256- and 192-token tiles are instantiated with the split: a caller that forces the 128-token tile on a split shape — gemm_variant
  //  24 through ifx_gemm_bf16_ws — gets the unsplit launch, not an error)
  // (8-bit operands: a K-step is 128 elements, K counts elements; the same (N, K)-only rule on the step count)
  // force_ks = 2 / 4 / 8 (bf16, 128-token tile): the caller's choice for shard-sized launches (gemm_small_split; pick_pp_small) —
  // the caller has checked the workspace against gemm_pp_small_workspace_bytes
  int ks = stream_k ? 0 : ((tj == 4 || (tj == 3 && !q8)) && workspace != nullptr && gemm_pp_split(N, q8 ? K / 2 : K) &&
                           tiles_m * tiles_n <= 1024) ? 2 : 1;
  if (force_ks > 1) {
    if (q8 || stream_k || tj != 2 || workspace == nullptr || (K / 64) % force_ks != 0 || tiles_m * tiles_n * (force_ks - 1) > 1024 ||
        (force_ks != 2 && force_ks != 4 && force_ks != 8)) {
      set_error("ifx_gemm_bf16: the %d-way K split needs the bf16 128-token tile, a workspace, K / 64 divisible by it and at most 1024 partial slots",
                force_ks);
      return IFX_EINVAL;
    }
    ks = force_ks;
  }
  const int total = tiles_m * tiles_n * (ks ? ks : 1), per_xcd = (total + 7) / 8;
  int wg_per_xcd = min(per_xcd, max(1, n_cu / 8));
  if (stream_k) {
    // equal K-step ranges over as many workgroups as leave each at least six K-steps (a dump + a partial read cost about two)
    const long units = (long)total * (K * (q8 ? 1 : 2) / 128);
    if (q8 || tj != 2 || workspace == nullptr || units < 8) {
      set_error("ifx_gemm_bf16: stream-K needs the bf16 128-token tile, a workspace and at least 8 K-steps");
      return IFX_EINVAL;
    }
    wg_per_xcd = (int)max(1L, min((long)min(n_cu, 256) / 8, units / 6 / 8));
  }
  const dim3 grid(wg_per_xcd * 8), block(512);
  unsigned* ws_flag = (unsigned*)workspace;          // the first 4096 bytes: zero on entry, zero on exit
  float* ws_part = workspace ? (float*)((char*)workspace + 4096) : nullptr;
  static unsigned long long* trace = nullptr;
  static int dbg = -1;
  if (dbg < 0) {
    const char* e = getenv("IFX_PP_DEBUG");       // lab only: 1 all workgroups stream tile 0, 2 no MFMAs, 4 K rotation by column tile, 16 staggered loaders
    dbg = e ? atoi(e) : 0;
  }
  // 32 = fault injection (ifx_set_option "spin_fault"): the producers of a split-K / stream-K launch keep their flags down, so that every
  // consumer's wait runs into its budget — the timeout path's test (tests/test_hip_kernels.py)
  const int dbg_launch = dbg | (spin_fault() ? 32 : 0);
  unsigned* const err_word = device_error_word();
  const long long spin_ticks = spin_timeout_ticks();
#if IFX_PP_TRACE
  if (!trace) {
    const char* e = getenv("IFX_PP_TRACE_PTR");
    if (e) trace = (unsigned long long*)strtoull(e, nullptr, 0);
  }
#endif
#define IFX_LAUNCH_PP(E, T, S, Q)                                                                                                    \
  do {                                                                                                                               \
    static bool attr_set = false;                                                                                                    \
    if (!attr_set) {                                                                                                                 \
      (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<E, T, S, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES); \
      attr_set = true;                                                                                                               \
    }                                                                                                                                \
    hipLaunchKernelGGL((gemm_pp_kernel<E, T, S, Q>), grid, block, LDS_BYTES, s, x, ldx, w, y, ldy, M, N, K, tiles_m, total, per_xcd, \
                       wg_per_xcd, ea, trace, dbg_launch, ws_part, ws_flag, err_word, spin_ticks);                                   \
  } while (0)
#define IFX_SWITCH_PP(T, S, Q)                                              \
  switch (mode) {                                                           \
    case IFX_EPI_BIAS: IFX_LAUNCH_PP(IFX_EPI_BIAS, T, S, Q); break;         \
    case IFX_EPI_GELU_TANH: IFX_LAUNCH_PP(IFX_EPI_GELU_TANH, T, S, Q); break; \
    case IFX_EPI_RESIDUAL: IFX_LAUNCH_PP(IFX_EPI_RESIDUAL, T, S, Q); break; \
    case IFX_EPI_GATE_RES: IFX_LAUNCH_PP(IFX_EPI_GATE_RES, T, S, Q); break; \
    default: set_error("ifx_gemm: the ping-pong tile has no epilogue %d", mode); return IFX_EINVAL; \
  }
  if (q8 && ks == 2) {
    if (q8_int8) { IFX_SWITCH_PP(4, 2, 2) }
    else { IFX_SWITCH_PP(4, 2, 1) }
  } else if (q8 && q8_int8) {
    if (tj == 4) { IFX_SWITCH_PP(4, 1, 2) }
    else if (tj == 3) { IFX_SWITCH_PP(3, 1, 2) }
    else if (tj == 2) { IFX_SWITCH_PP(2, 1, 2) }
    else { set_error("ifx_gemm_q8: no ping-pong tile of %d tokens", 64 * tj); return IFX_EINVAL; }
  } else if (q8) {
    if (tj == 4) { IFX_SWITCH_PP(4, 1, 1) }
    else if (tj == 3) { IFX_SWITCH_PP(3, 1, 1) }
    else if (tj == 2) { IFX_SWITCH_PP(2, 1, 1) }
    else { set_error("ifx_gemm_q8: no ping-pong tile of %d tokens", 64 * tj); return IFX_EINVAL; }
  } else if (ks == 0) {
    IFX_SWITCH_PP(2, 0, 0)
  } else if (tj == 2 && ks == 2) { IFX_SWITCH_PP(2, 2, 0)
  } else if (tj == 2 && ks == 4) { IFX_SWITCH_PP(2, 4, 0)
  } else if (tj == 2 && ks == 8) { IFX_SWITCH_PP(2, 8, 0)
  } else if (ks == 2) {                              // split K: long-K, narrow-N shapes on the 256- or the 192-token tile
    if (tj == 4) { IFX_SWITCH_PP(4, 2, 0) }
    else { IFX_SWITCH_PP(3, 2, 0) }
  } else if (tj == 4) { IFX_SWITCH_PP(4, 1, 0) }
  else if (tj == 3) { IFX_SWITCH_PP(3, 1, 0) }
  else if (tj == 2) { IFX_SWITCH_PP(2, 1, 0) }
  else { set_error("ifx_gemm_bf16: no ping-pong tile of %d tokens", 64 * tj); return IFX_EINVAL; }
#undef IFX_SWITCH_PP
#undef IFX_LAUNCH_PP
  return check_launch("ifx_gemm_bf16(pp)");
}

}  // namespace ifx
