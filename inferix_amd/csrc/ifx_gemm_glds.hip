// bf16 linear layer, large-shape kernel (gfx950): 256(tokens) x 128(channels) x 64(K) tiles, 8 waves,
// LDS-DMA staging (global_load_lds, 16 B/lane) into a 3-stage LDS ring with COUNTED vmcnt so that two
// K-tiles stay in flight across the (single) barrier per K-tile.
//
//   y[M,N] = epilogue( x[M,K] @ W[N,K]^T + bias[N] )            (same contract as ifx_gemm.hip)
//
// Why this shape: the v1 kernel (128^2, register staging, 2 buffers) spends 39 % of its wave cycles in
// s_waitcnt/s_barrier (profiles/r1_pmc_attn_gemm.md) because a register-staged double buffer must drain
// its loads before every barrier.  Here:
//   * LDS-DMA needs no staging VGPRs and no ds_write pass, so a third stage costs only LDS (3 x 48 KiB =
//     144 of the CU's 160 KiB -> one 8-wave workgroup per CU, 2 waves per SIMD);
//   * per K-tile: s_waitcnt vmcnt(6) [tile kt landed, tile kt+1 still in flight] -> s_barrier -> issue
//     tile kt+2 -> 16 x ds_read_b128 + 16 x v_mfma_f32_32x32x16_bf16 per wave;
//   * the DMA writes LDS lane-linearly (wave-uniform base + lane*16), so the bank-conflict swizzle is
//     applied on the per-lane SOURCE address and on the fragment reads (same involution both sides):
//     128-byte rows, 16-byte chunk index XOR ((row >> 1) & 7): conflict free for the 32-row fragments;
//   * transposed tile (A-operand = W rows, B-operand = x rows): each lane owns 4 consecutive output
//     channels of one token -> 8-byte epilogue accesses for bias / gate / residual / store;
//   * XCD-aware tile order: each XCD works a contiguous token-fastest range so concurrently resident
//     workgroups share W panels and walk x once per range in its L2.
#include <stdlib.h>

#include "ifx_common.h"

#ifndef IFX_GEMM_GM
#define IFX_GEMM_GM 4       // row tiles per rasterisation group (tile ids sweep GM rows x all column tiles before the next GM rows)
#endif
#ifndef IFX_T8_NST
#define IFX_T8_NST 2
#endif
#ifndef IFX_SMALL_NST64
#define IFX_SMALL_NST64 3
#endif
#ifndef IFX_SMALL_NST
#define IFX_SMALL_NST 2     // stages of the 128x128 tile: 2 = 64 KiB of LDS -> two workgroups per CU (4 stages, one per CU: 15-35 % slower)
#endif
#ifndef IFX_GEMM_LOADERS
#define IFX_GEMM_LOADERS 8   // 4 (older waves only) measured neutral here: FFN up 171.8 vs 170 us — this kernel is bound by operand delivery, not by DMA issue
#endif

namespace ifx {

namespace g2 {
constexpr int BM = 256, BN = 128, BK = 64;
constexpr int STAGE = (BM + BN) * BK * 2;   // 49152 B
constexpr int NSTAGE = 3;
constexpr int A_OFF = 0;                    // x tile  [256][128 B]
constexpr int B_OFF = BM * BK * 2;          // W tile  [128][128 B]
}  // namespace g2

struct EpiArgs2 {
  const unsigned short* bias;
  const unsigned short* residual;
  int ld_res;
  const unsigned short* mod;
  int mod_slots, gate_slot, rows_per_group;
};

// exact (erf) GELU as torch.nn.functional.gelu evaluates it on a bf16 tensor: fp32 math, one rounding (MAGI CustomMLP,
// inferix/models/magi/dit/dit_module.py:552).  Selected at run time inside the GELU epilogue instantiation: the epilogue's
// otherwise unused `gate_slot` field carries 1 for IFX_EPI_GELU_ERF.

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_glds_kernel(const unsigned short* __restrict__ x, int ldx,
                                                           const unsigned short* __restrict__ w,
                                                           unsigned short* __restrict__ y, int ldy, int M, int N,
                                                           int K, int tiles_m, int total, int per_xcd, EpiArgs2 ea, int ablate) {
  using namespace g2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware order: XCD (bid & 7) owns tiles [xcd*per, (xcd+1)*per), token(m)-fastest
  const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
  const int t_id = xcd * per_xcd + slot_i;
  if (slot_i >= per_xcd || t_id >= total) return;
  // grouped raster: consecutive ids walk GM token-tiles, then the next channel-tile -> the ~32 workgroups
  // resident on one XCD form a GM x (32/GM) block of tiles sharing GM x-panels and 32/GM W-panels in its L2
  // (v1's id = m-fastest order made every XCD touch every panel: 0.9-1.2 GB of memory-side fetches per GEMM).
  constexpr int GM = IFX_GEMM_GM;
  const int tiles_n = total / tiles_m;
  const int grp_sz = GM * tiles_n;
  const int first_m = (t_id / grp_sz) * GM;
  const int gm = min(GM, tiles_m - first_m);
  const int rem = t_id % grp_sz;
  const int tile_m = first_m + rem % gm, tile_n = rem / gm;
  const int m_base = tile_m * BM, n_base = tile_n * BN;
  const int wm = wave & 3, wn = wave >> 2;     // 4 x 2 waves, 64 x 64 each

  // ---- LDS-DMA source addressing: one wave-instruction = 8 rows x 128 B; lane l -> row l>>3, physical
  //      chunk l&7, which must hold LOGICAL chunk (l&7) ^ ((row>>1)&7)
  const int r8 = lane >> 3, pc = lane & 7;
  const unsigned short* src_a[4];
  const unsigned short* src_b[2];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = (r * 8 + wave) * 8 + r8;                 // 0..255
    const int lc = pc ^ ((row >> 1) & 7);
    src_a[r] = x + (size_t)min(m_base + row, M - 1) * ldx + lc * 8;
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = (r * 8 + wave) * 8 + r8;                 // 0..127
    const int lc = pc ^ ((row >> 1) & 7);
    src_b[r] = w + (size_t)min(n_base + row, N - 1) * K + lc * 8;
  }
  auto issue = [&](int kt) {
    unsigned char* st = smem + (kt % NSTAGE) * STAGE;
    const size_t ko = (size_t)kt * BK;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src_a[r] + ko),
                                       (lds_ptr_t)(st + A_OFF + (r * 8 + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int r = 0; r < 2; ++r)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src_b[r] + ko),
                                       (lds_ptr_t)(st + B_OFF + (r * 8 + wave) * 1024), 16, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = K / BK;
  issue(0);
  if (KT > 1) issue(1);

  // fragment read addressing: row = base + (lane & 31), logical chunk 2*ks + (lane >> 5)
  const int l31 = lane & 31, hi = lane >> 5;
  int a_row_off[2], b_row_off[2], a_swz[2], b_swz[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wm * 64 + j * 32 + l31;
    a_row_off[j] = A_OFF + row * 128;
    a_swz[j] = (row >> 1) & 7;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wn * 64 + i * 32 + l31;
    b_row_off[i] = B_OFF + row * 128;
    b_swz[i] = (row >> 1) & 7;
  }

  // Ping-pong schedule: the two waves that share a SIMD (wave w and w+4) alternate roles every phase —
  // one streams its 16 fragment reads of a K-tile from LDS into registers while the other runs its 16 MFMAs
  // back to back from registers — so the matrix pipe of every SIMD always has a wave in its math phase.
  //   barrier 2kt   : G0 issues DMA(kt+2), load(kt)   | G1 math(kt-1)
  //   barrier 2kt+1 : G0 math(kt)                     | G1 issues DMA(kt+2), load(kt)
  // Tile kt is read between barriers 2kt..2kt+2 and its ring slot is refilled (tile kt+3) after barrier
  // 2kt+2 at the earliest; every LDS read is retired (lgkmcnt(0)) before the reading wave reaches a barrier.
  const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);       // 0: waves 0-3, 1: waves 4-7
  bf16x8 fa[4][2], fb[4][2];
  auto load_frags = [&](int kt) {
    const unsigned char* st = smem + (kt % NSTAGE) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 2 * ks + hi;
#pragma unroll
      for (int i = 0; i < 2; ++i) fb[ks][i] = *reinterpret_cast<const bf16x8*>(st + b_row_off[i] + ((c ^ b_swz[i]) << 4));
#pragma unroll
      for (int j = 0; j < 2; ++j) fa[ks][j] = *reinterpret_cast<const bf16x8*>(st + a_row_off[j] + ((c ^ a_swz[j]) << 4));
    }
  };
  auto math = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][i], fa[ks][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  // Both groups run the SAME straight-line loop; group 1 is shifted by one barrier (one phase), which is what
  // makes the roles alternate.  Each wave waits for its own DMA pieces of the NEXT needed tile before every
  // barrier (counted vmcnt: one tile stays in flight), so after a barrier the tile is complete for everyone.
  // vmcnt discipline: tile kt is first read after barrier 2kt (by G0).  G0 retires its own pieces of tile kt
  // just before that barrier (top of its iteration kt), G1 just before the same barrier (middle of ITS
  // iteration kt-1) — one counted wait per wave per K-tile, issued 4 (G0) / 3 (G1) phases after the DMA.
  if (grp == 1) {
    if (KT > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // own pieces of tile 0 landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  for (int kt = 0; kt < KT; ++kt) {
    if (grp == 0) {
      if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // tile kt landed (kt+1 in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < KT && !(ablate & 1)) issue(kt + 2);
    if (!(ablate & 2)) load_frags(kt);
    if (grp == 1) {
      if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // tile kt+1 landed (kt+2 in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!(ablate & 4)) math();
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();

  // ---- epilogue.  The MFMA layout gives each lane 4 consecutive channels of ONE token (32 tokens per
  // instruction): stored directly that is 32 partial cache lines per store instruction (measured: ~40 us of
  // a 117 us QKV GEMM).  Instead every wave transposes its 64x64 tile of v = bf16(acc + bias) through LDS
  // (8 KiB per wave in the now idle ring; 16-byte chunk index XOR (token & 7): conflict-free b128 reads) and
  // then reads/writes 8 tokens x 128 contiguous bytes per instruction: residual / gate loads and the output
  // store are full 128-byte lines.  Rounding is unchanged (v is the bf16 Linear output in every epilogue).
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // every wave is done reading the ring
  unsigned char* tw = smem + wave * 8192;                // [64 tokens][128 B]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int mrow = j * 32 + l31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = i * 32 + g * 8 + hi * 4;           // channel within the wave tile (multiple of 4)
        const int n = n_base + wn * 64 + nl;
        float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (ea.bias && n < N) {
          const u16x4 bv = *reinterpret_cast<const u16x4*>(ea.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bf2f(bv[e]);
        }
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        const int chunk = (nl >> 3) ^ (mrow & 7);
        *reinterpret_cast<u16x4*>(tw + mrow * 128 + chunk * 16 + (nl & 4) * 2) = o;
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // wave-private region: no barrier needed
  {
    const int rr = lane >> 3, cc = lane & 7;               // 8 tokens x 8 chunks of 16 B per instruction
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int mrow = p * 8 + rr;
      const int m = m_base + wm * 64 + mrow;
      const int n = n_base + wn * 64 + cc * 8;
      const u16x8 vv = *reinterpret_cast<const u16x8*>(tw + mrow * 128 + ((cc ^ (mrow & 7)) << 4));
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
      *reinterpret_cast<u16x8*>(y + (size_t)m * ldy + n) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Small-M kernel family (sequence-parallel shards: M = 2340 / 1170 / 585 token rows per GPU).  With 256x128
// tiles such launches leave most CUs idle (M=585, N=1536: 36 workgroups for 256 CUs) and the register-staged
// 128^2 kernel is latency-bound at one workgroup per CU (28 us for 585x1536x1536).  Here: BM x BN in
// {128x128, 64x64}, 4 waves (2x2), same LDS-DMA staging / swizzle / transposed MFMA tile / LDS-transposed
// epilogue as above, a 4-stage ring with THREE K-tiles in flight and one barrier per K-tile; the 64x64 shape
// uses 64 KiB of LDS so two workgroups share a CU and hide each other's waits.
// KG > 1: split-K INSIDE the workgroup — KG groups of four waves each take a contiguous 1/KG of the K tiles of the SAME output tile
// through their own operand ring and are summed through LDS (fixed order: deterministic) before the epilogue.  For launches with at
// most one workgroup per CU (one rank's M = 4680 / P rows of a sequence-parallel shard) the K loop of a single four-wave workgroup
// serialises DMA issue -> fragment reads -> MFMA on each SIMD (~770 clk per 64-deep step, the matrix pipe busy for ~130 of them);
// KG = 4 puts four waves on every SIMD without needing more output tiles.
template <int BM, int BN, int NST, int EPI, int KG = 1>
__global__ __launch_bounds__(256 * KG) void gemm_small_kernel(const unsigned short* __restrict__ x, int ldx,
                                                         const unsigned short* __restrict__ w,
                                                         unsigned short* __restrict__ y, int ldy, int M, int N, int K,
                                                         int tiles_m, int total, int per_xcd, EpiArgs2 ea) {
  constexpr int BK = 64;
  constexpr int STAGE = (BM + BN) * BK * 2;
  constexpr int A_OFF = 0, B_OFF = BM * BK * 2;
  constexpr int PA = BM / 32, PB = BN / 32, P = PA + PB;     // DMA instructions per wave per K-tile
  constexpr int TJ = BM / 64, TI = BN / 64;                  // 32-blocks per wave: tokens, channels
  constexpr int WM = BM / 2, WN = BN / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = KG > 1 ? wave_all >> 2 : 0, wave = wave_all & 3;
  unsigned char* const ring = smem + kg * (NST * STAGE);
  const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
  const int t_id = xcd * per_xcd + slot_i;
  if (slot_i >= per_xcd || t_id >= total) return;
  constexpr int GM = IFX_GEMM_GM;
  const int tiles_n = total / tiles_m;
  const int grp_sz = GM * tiles_n;
  const int first_m = (t_id / grp_sz) * GM;
  const int gm = min(GM, tiles_m - first_m);
  const int rem = t_id % grp_sz;
  const int tile_m = first_m + rem % gm, tile_n = rem / gm;
  const int m_base = tile_m * BM, n_base = tile_n * BN;
  const int wm = wave & 1, wn = wave >> 1;

  const int r8 = lane >> 3, pc = lane & 7;
  const unsigned short* src_a[PA];
  const unsigned short* src_b[PB];
#pragma unroll
  for (int r = 0; r < PA; ++r) {
    const int row = (r * 4 + wave) * 8 + r8;
    src_a[r] = x + (size_t)min(m_base + row, M - 1) * ldx + (pc ^ ((row >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int r = 0; r < PB; ++r) {
    const int row = (r * 4 + wave) * 8 + r8;
    src_b[r] = w + (size_t)min(n_base + row, N - 1) * K + (pc ^ ((row >> 1) & 7)) * 8;
  }
  const int KT = (K / BK) / KG;                               // K tiles of this group (the host checks divisibility)
  const int kt0 = kg * KT;
  auto issue = [&](int kt) {
    unsigned char* st = ring + (kt % NST) * STAGE;
    const size_t ko = (size_t)(kt0 + kt) * BK;
#pragma unroll
    for (int r = 0; r < PA; ++r)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src_a[r] + ko), (lds_ptr_t)(st + A_OFF + (r * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int r = 0; r < PB; ++r)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src_b[r] + ko), (lds_ptr_t)(st + B_OFF + (r * 4 + wave) * 1024), 16, 0, 0);
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < KT) issue(i);

  const int l31 = lane & 31, hi = lane >> 5;
  int a_row_off[TJ], b_row_off[TI], a_swz[TJ], b_swz[TI];
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int row = wm * WM + j * 32 + l31;
    a_row_off[j] = A_OFF + row * 128;
    a_swz[j] = (row >> 1) & 7;
  }
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int row = wn * WN + i * 32 + l31;
    b_row_off[i] = B_OFF + row * 128;
    b_swz[i] = (row >> 1) & 7;
  }

  for (int kt = 0; kt < KT; ++kt) {
    // own pieces of tile kt landed; up to two later tiles stay in flight
    const int later = min(KT - 1 - kt, NST - 2);
    static_assert((NST - 2) * P <= 63 && NST <= 8, "vmcnt is a 6-bit counter");
#define IFX_WAIT_LATER(n)                                                                  \
  else if (NST - 2 >= (n) && later == (n)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(((n) * P) & 63) : "memory")
    if (later == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IFX_WAIT_LATER(1);
    IFX_WAIT_LATER(2);
    IFX_WAIT_LATER(3);
    IFX_WAIT_LATER(4);
    IFX_WAIT_LATER(5);
    IFX_WAIT_LATER(6);
#undef IFX_WAIT_LATER
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own fragment reads of tile kt-1 retired
    __builtin_amdgcn_s_barrier();          // tile kt complete for everyone; the slot of tile kt-1 drained by everyone
    if (kt + NST - 1 < KT) issue(kt + NST - 1);
    const unsigned char* st = ring + (kt % NST) * STAGE;
    bf16x8 fa[4][TJ], fb[4][TI];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 2 * ks + hi;
#pragma unroll
      for (int i = 0; i < TI; ++i) fb[ks][i] = *reinterpret_cast<const bf16x8*>(st + b_row_off[i] + ((c ^ b_swz[i]) << 4));
#pragma unroll
      for (int j = 0; j < TJ; ++j) fa[ks][j] = *reinterpret_cast<const bf16x8*>(st + a_row_off[j] + ((c ^ a_swz[j]) << 4));
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][i], fa[ks][j], acc[i][j], 0, 0, 0);
  }

  // ---- epilogue: per-wave LDS transpose of v = bf16(acc + bias), then row-contiguous 16-byte accesses
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (KG > 1) {
    // partial sums of groups 1..KG-1 -> LDS (behind the transposes of the epilogue), [group][wave][i][j][quad][lane] x 16 B
    float* part = reinterpret_cast<float*>(smem + BM * BN * 2);
    if (kg > 0) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int slot = ((((kg - 1) * 4 + wave) * TI + i) * TJ + j) * 4 + q;
            *reinterpret_cast<f32x4*>(part + (size_t)slot * 256 + lane * 4) =
                f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kg > 0) return;
#pragma unroll
    for (int g = 1; g < KG; ++g)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int slot = ((((g - 1) * 4 + wave) * TI + i) * TJ + j) * 4 + q;
            const f32x4 pv = *reinterpret_cast<const f32x4*>(part + (size_t)slot * 256 + lane * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += pv[e];
          }
  }
  constexpr int RB = WN * 2, CR = RB / 16, RP = 64 / CR;     // row bytes, 16-B chunks per row, rows per instruction
  unsigned char* tw = smem + wave * (WM * RB);
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int mrow = j * 32 + l31;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = i * 32 + g * 8 + hi * 4;
        const int n = n_base + wn * WN + nl;
        float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (ea.bias && n < N) {
          const u16x4 bv = *reinterpret_cast<const u16x4*>(ea.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bf2f(bv[e]);
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
      *reinterpret_cast<u16x8*>(y + (size_t)m * ldy + n) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// 256(tokens) x 256(channels) x 32(K) tiles, 8 waves (2 x 4, 128 x 64 outputs each), 4-stage LDS-DMA ring.
// Why: the 256x128x64 kernel is bound by operand DELIVERY, not by the matrix pipe — with MFMAs and fragment reads
// ablated it still takes 127 of 209 us on the 4680x8960x1536 GEMM (profiles/r1d_gemm_ablation.md): the ring keeps
// two 48 KiB tiles (96 KiB) in flight per CU against ~2 us of L2 latency = ~20 B/clk/CU.  A 256x256 tile needs a
// third fewer operand bytes per FLOP (64 KiB per 8.4 MFLOP against 48 KiB per 4.2), and K-steps of 32 in four
// 32 KiB stages keep 96 KiB in flight in a 128 KiB ring.  Used when the tile count still fills the chip
// (pick_tile in ifx_gemm.hip); same operand roles, swizzled DMA and LDS-transposed epilogue as above.
//   LDS rows are 64 B (32 bf16): physical 16-byte chunk = logical chunk XOR ((row >> 2) & 3).
// ---------------------------------------------------------------------------------------------------------------------
// Warp-specialised tile: waves 0-3 (one per SIMD, the older ones) only compute, waves 4-7 only issue the LDS-DMA of the operand
// stages.  An LDS-DMA instruction stalls its issuing wave ~65 cycles and serialises per SIMD, but does not slow the MFMAs of the
// other wave on that SIMD (tools/probe_overlap.hip) — in the other kernels every wave spends a third to a half of each K-step
// issuing loads with nothing queued on the matrix pipe.  BM x BN x 64 tile, consumer wave = (BM/2) x (BN/2).
template <int BM, int BN, int NST, int EPI>
__global__ __launch_bounds__(512) void gemm_ws_kernel(const unsigned short* __restrict__ x, int ldx,
                                                      const unsigned short* __restrict__ w, unsigned short* __restrict__ y, int ldy,
                                                      int M, int N, int K, int tiles_m, int total, int per_xcd, EpiArgs2 ea) {
  constexpr int BK = 64;
  constexpr int STAGE = (BM + BN) * BK * 2;
  constexpr int A_OFF = 0, B_OFF = BM * BK * 2;
  constexpr int PA = BM / 32, PB = BN / 32, P = PA + PB;     // 1 KiB pieces per PRODUCER wave per K-tile (8 rows each)
  constexpr int WM = BM / 2, WN = BN / 2, TJ = WM / 32, TI = WN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
  const int t_id = xcd * per_xcd + slot_i;
  if (slot_i >= per_xcd || t_id >= total) return;
  constexpr int GM = IFX_GEMM_GM;
  const int tiles_n = total / tiles_m;
  const int grp_sz = GM * tiles_n;
  const int first_m = (t_id / grp_sz) * GM;
  const int gm = min(GM, tiles_m - first_m);
  const int rem = t_id % grp_sz;
  const int tile_m = first_m + rem % gm, tile_n = rem / gm;
  const int m_base = tile_m * BM, n_base = tile_n * BN;
  const int KT = K / BK;
  const int l31 = lane & 31, hi = lane >> 5;

  if (producer) {
    const int pw = wave - 4;
    const int r8 = lane >> 3, pc = lane & 7;
    const unsigned short* src_a[PA];
    const unsigned short* src_b[PB];
#pragma unroll
    for (int r = 0; r < PA; ++r) {
      const int row = (r * 4 + pw) * 8 + r8;
      src_a[r] = x + (size_t)min(m_base + row, M - 1) * ldx + (pc ^ ((row >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int r = 0; r < PB; ++r) {
      const int row = (r * 4 + pw) * 8 + r8;
      src_b[r] = w + (size_t)min(n_base + row, N - 1) * K + (pc ^ ((row >> 1) & 7)) * 8;
    }
    auto issue = [&](int kt) {
      unsigned char* st = smem + (kt % NST) * STAGE;
      const size_t ko = (size_t)kt * BK;
#pragma unroll
      for (int r = 0; r < PA; ++r)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src_a[r] + ko), (lds_ptr_t)(st + A_OFF + (r * 4 + pw) * 1024), 16, 0, 0);
#pragma unroll
      for (int r = 0; r < PB; ++r)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src_b[r] + ko), (lds_ptr_t)(st + B_OFF + (r * 4 + pw) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
      if (i < KT) issue(i);
    for (int kt = 0; kt < KT; ++kt) {
      const int later = min(KT - 1 - kt, NST - 2);          // stages that may stay in flight behind stage kt
      if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                         // stage kt published; the slot of stage kt-1 is drained
      if (kt + NST - 1 < KT) issue(kt + NST - 1);
    }
    __builtin_amdgcn_s_barrier();                           // the consumers' epilogue barrier
    return;
  }

  const int wm = wave & 1, wn = wave >> 1;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int a_row_off[TJ], b_row_off[TI], a_swz[TJ], b_swz[TI];
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int row = wm * WM + j * 32 + l31;
    a_row_off[j] = A_OFF + row * 128;
    a_swz[j] = (row >> 1) & 7;
  }
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int row = wn * WN + i * 32 + l31;
    b_row_off[i] = B_OFF + row * 128;
    b_swz[i] = (row >> 1) & 7;
  }
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // own fragment reads of stage kt-1 retired
    __builtin_amdgcn_s_barrier();
    const unsigned char* st = smem + (kt % NST) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 2 * ks + hi;
      bf16x8 fa[TJ], fb[TI];
#pragma unroll
      for (int i = 0; i < TI; ++i) fb[i] = *reinterpret_cast<const bf16x8*>(st + b_row_off[i] + ((c ^ b_swz[i]) << 4));
#pragma unroll
      for (int j = 0; j < TJ; ++j) fa[j] = *reinterpret_cast<const bf16x8*>(st + a_row_off[j] + ((c ^ a_swz[j]) << 4));
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i], fa[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue (consumer waves): per-wave LDS transpose of v = bf16(acc + bias), then row-contiguous 16-byte accesses
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  constexpr int RB = WN * 2, CR = RB / 16, RP = 64 / CR;
  unsigned char* tw = smem + wave * (WM * RB);
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int mrow = j * 32 + l31;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = i * 32 + g * 8 + hi * 4;
        const int n = n_base + wn * WN + nl;
        float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (ea.bias && n < N) {
          const u16x4 bv = *reinterpret_cast<const u16x4*>(ea.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bf2f(bv[e]);
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
      *reinterpret_cast<u16x8*>(y + (size_t)m * ldy + n) = o;
    }
  }
}

// BK = 32: operand rows of 64 B in LDS; BK = 64: rows of 128 B.  LDS-DMA moves rows of >= 128 B from L2 at 52-58 B/clk/CU and
// 64-byte rows at HALF of that (tools/probe_dma.hip: a 64-byte request still costs a 128-byte line), so the 64-deep variant pays
// half the DMA time per byte; it needs (BM + BN) * 128 B per stage (64 KiB for 256x256: two stages).
template <int BM, int BN, int WAVES_M, int NST, int EPI, int BK = 32>
__global__ __launch_bounds__(512, (BM * BN <= 128 * 128) ? (NST <= 2 ? 3 : 2) : ((BM * BN <= 256 * 128 && NST <= 2) ? 2 : 1)) void gemm_big_kernel(const unsigned short* __restrict__ x, int ldx,
                                                       const unsigned short* __restrict__ w,
                                                       unsigned short* __restrict__ y, int ldy, int M, int N, int K,
                                                       int tiles_m, int total, int per_xcd, EpiArgs2 ea) {
  constexpr int CPR = BK / 8, RPP = 64 / CPR;           // 16-byte chunks per row (4 | 8), rows per 1 KiB DMA piece (16 | 8)
  constexpr int SW_SH = BK == 32 ? 2 : 1;                // swizzle phase: (row >> SW_SH) & (CPR - 1)
  constexpr int STAGE = (BM + BN) * BK * 2;              // 32 KiB (256x256) / 24 KiB (256x128)
  constexpr int A_OFF = 0, B_OFF = BM * BK * 2;          // x tile [BM][64 B], W tile [BN][64 B]
  constexpr int WAVES_N = 8 / WAVES_M;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TJ = WM / 32, TI = WN / 32;
  // LDS-DMA issue costs the issuing wave ~65 cycles per instruction and serialises per SIMD, but leaves the other wave of the
  // SIMD free (tools/probe_overlap.hip): with IFX_GEMM_LOADERS = 4 only waves 0-3 (the older wave of each SIMD) issue the
  // pieces of both waves, the younger ones go straight to the matrix pipe after the barrier.
  constexpr int LW = IFX_GEMM_LOADERS, LM = 8 / LW;      // loader waves, wave-slots per loader
  constexpr int PA = BM / (8 * RPP) * LM, PB = BN / (8 * RPP) * LM, P = PA + PB;   // DMA instructions per LOADER wave per K-tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
  const int t_id = xcd * per_xcd + slot_i;
  if (slot_i >= per_xcd || t_id >= total) return;
  constexpr int GM = IFX_GEMM_GM;
  const int tiles_n = total / tiles_m;
  const int grp_sz = GM * tiles_n;
  const int first_m = (t_id / grp_sz) * GM;
  const int gm = min(GM, tiles_m - first_m);
  const int rem = t_id % grp_sz;
  const int tile_m = first_m + rem % gm, tile_n = rem / gm;
  const int m_base = tile_m * BM, n_base = tile_n * BN;
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;

  // LDS-DMA: one wave-instruction = RPP rows x (BK * 2) B; lane -> row (lane / CPR), physical chunk (lane % CPR)
  const int r16 = lane / CPR, pc = lane % CPR;
  const unsigned short* src_a[PA];
  const unsigned short* src_b[PB];
  const bool loader = wave < LW;
  // piece index of (r, wave): r * LW + wave, covering rows [16 piece, 16 piece + 16)
#pragma unroll
  for (int r = 0; r < PA; ++r) {
    const int row = (r * LW + wave) * RPP + r16;
    src_a[r] = x + (size_t)min(m_base + row, M - 1) * ldx + (pc ^ ((row >> SW_SH) & (CPR - 1))) * 8;
  }
#pragma unroll
  for (int r = 0; r < PB; ++r) {
    const int row = (r * LW + wave) * RPP + r16;
    src_b[r] = w + (size_t)min(n_base + row, N - 1) * K + (pc ^ ((row >> SW_SH) & (CPR - 1))) * 8;
  }
  auto issue = [&](int kt) {
    unsigned char* st = smem + (kt % NST) * STAGE;
    const size_t ko = (size_t)kt * BK;
#pragma unroll
    for (int r = 0; r < PA; ++r)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src_a[r] + ko), (lds_ptr_t)(st + A_OFF + (r * LW + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int r = 0; r < PB; ++r)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src_b[r] + ko), (lds_ptr_t)(st + B_OFF + (r * LW + wave) * 1024), 16, 0, 0);
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = K / BK;
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < KT && loader) issue(i);

  const int l31 = lane & 31, hi = lane >> 5;
  int a_off[TJ], b_off[TI], a_swz[TJ], b_swz[TI];
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int row = wm * WM + j * 32 + l31;
    a_off[j] = A_OFF + row * (BK * 2);
    a_swz[j] = (row >> SW_SH) & (CPR - 1);
  }
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int row = wn * WN + i * 32 + l31;
    b_off[i] = B_OFF + row * (BK * 2);
    b_swz[i] = (row >> SW_SH) & (CPR - 1);
  }

  for (int kt = 0; kt < KT; ++kt) {
    const int later = min(KT - 1 - kt, NST - 2);           // own pieces of tile kt landed; later tiles stay in flight
    if (later >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * P) : "memory");
    else if (later == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * P) : "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + NST - 1 < KT && loader) issue(kt + NST - 1);
    const unsigned char* st = smem + (kt % NST) * STAGE;
#pragma unroll
    for (int half = 0; half < BK / 32; ++half) {           // 32 channels at a time: two k-steps of fragments in registers
      bf16x8 fa[2][TJ], fb[2][TI];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int c = 4 * half + 2 * ks + hi;
#pragma unroll
        for (int i = 0; i < TI; ++i) fb[ks][i] = *reinterpret_cast<const bf16x8*>(st + b_off[i] + ((c ^ b_swz[i]) << 4));
#pragma unroll
        for (int j = 0; j < TJ; ++j) fa[ks][j] = *reinterpret_cast<const bf16x8*>(st + a_off[j] + ((c ^ a_swz[j]) << 4));
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][i], fa[ks][j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: per-wave LDS transpose of v = bf16(acc + bias) (128 tokens x 128 B, 16 KiB per wave = the whole
  //      ring), then row-contiguous 16-byte accesses for residual / gate / store
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  constexpr int RB = WN * 2, CR = RB / 16, RP = 64 / CR;   // row bytes, 16-B chunks per row, rows per instruction
  // CR a power of two (WN = 32 / 64 / 128): XOR swizzle, all 64 lanes read.  Otherwise (WN = 96 of the 256 x 192 tile: 12 chunks) the
  // chunk index is rotated by the row and lanes RP * CR .. 63 sit the read-out phase out.
  constexpr bool P2 = (CR & (CR - 1)) == 0;
  auto phys = [&](int chunk, int mrow) { return P2 ? (chunk ^ (mrow & (CR - 1))) : (chunk + mrow) % CR; };
  unsigned char* tw = smem + wave * (WM * RB);
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int mrow = j * 32 + l31;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = i * 32 + g * 8 + hi * 4;
        const int n = n_base + wn * WN + nl;
        float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (ea.bias && n < N) {
          const u16x4 bv = *reinterpret_cast<const u16x4*>(ea.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bf2f(bv[e]);
        }
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        const int chunk = phys(nl >> 3, mrow);
        *reinterpret_cast<u16x4*>(tw + mrow * RB + chunk * 16 + (nl & 4) * 2) = o;
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {
    const int rr = lane / CR, cc = lane % CR;
#pragma unroll
    for (int p = 0; p < (WM + RP - 1) / RP; ++p) {
      const int mrow = p * RP + rr;
      if (!P2 && (rr >= RP || mrow >= WM)) continue;
      const int m = m_base + wm * WM + mrow;
      const int n = n_base + wn * WN + cc * 8;
      const u16x8 vv = *reinterpret_cast<const u16x8*>(tw + mrow * RB + (phys(cc, mrow) << 4));
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
      *reinterpret_cast<u16x8*>(y + (size_t)m * ldy + n) = o;
    }
  }
}

template <int BM, int BN, int WAVES_M, int NST, int BK = 32>
static int launch_big(const unsigned short* x, int ldx, const unsigned short* w, unsigned short* y, int ldy, int M, int N,
                      int K, int mode, const EpiArgs2& ea, hipStream_t s) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int total = tiles_m * tiles_n, per_xcd = (total + 7) / 8;
  const dim3 grid(per_xcd * 8), block(512);
  constexpr size_t lds_main = (size_t)NST * (BM + BN) * BK * 2, lds_epi = (size_t)BM * BN * 2;   // rings / per-wave transposes
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
#define IFX_LAUNCH_GB(E)                                                                                             \
  do {                                                                                                               \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      (void)hipFuncSetAttribute((const void*)gemm_big_kernel<BM, BN, WAVES_M, NST, E, BK>,                               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    hipLaunchKernelGGL((gemm_big_kernel<BM, BN, WAVES_M, NST, E, BK>), grid, block, lds, s, x, ldx, w, y, ldy, M, N, K,  \
                       tiles_m, total, per_xcd, ea);                                                                 \
  } while (0)
  switch (mode) {
    case IFX_EPI_BIAS: IFX_LAUNCH_GB(IFX_EPI_BIAS); break;
    case IFX_EPI_GELU_TANH: IFX_LAUNCH_GB(IFX_EPI_GELU_TANH); break;
    case IFX_EPI_RESIDUAL: IFX_LAUNCH_GB(IFX_EPI_RESIDUAL); break;
    case IFX_EPI_GATE_RES: IFX_LAUNCH_GB(IFX_EPI_GATE_RES); break;
    default: return IFX_EINVAL;
  }
#undef IFX_LAUNCH_GB
  return check_launch("ifx_gemm_bf16(k32)");
}

template <int BM, int BN, int NST>
static int launch_ws(const unsigned short* x, int ldx, const unsigned short* w, unsigned short* y, int ldy, int M, int N, int K,
                     int mode, const EpiArgs2& ea, hipStream_t s) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int total = tiles_m * tiles_n, per_xcd = (total + 7) / 8;
  const dim3 grid(per_xcd * 8), block(512);
  constexpr size_t lds_main = (size_t)NST * (BM + BN) * 128, lds_epi = (size_t)BM * BN * 2;
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
#define IFX_LAUNCH_WS(E)                                                                                             \
  do {                                                                                                               \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      (void)hipFuncSetAttribute((const void*)gemm_ws_kernel<BM, BN, NST, E>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                           \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, NST, E>), grid, block, lds, s, x, ldx, w, y, ldy, M, N, K, tiles_m, total, \
                       per_xcd, ea);                                                                                 \
  } while (0)
  switch (mode) {
    case IFX_EPI_BIAS: IFX_LAUNCH_WS(IFX_EPI_BIAS); break;
    case IFX_EPI_GELU_TANH: IFX_LAUNCH_WS(IFX_EPI_GELU_TANH); break;
    case IFX_EPI_RESIDUAL: IFX_LAUNCH_WS(IFX_EPI_RESIDUAL); break;
    case IFX_EPI_GATE_RES: IFX_LAUNCH_WS(IFX_EPI_GATE_RES); break;
    default: return IFX_EINVAL;
  }
#undef IFX_LAUNCH_WS
  return check_launch("ifx_gemm_bf16");
}

template <int BM, int BN, int NST, int KG = 1>
static int launch_small(const unsigned short* x, int ldx, const unsigned short* w, unsigned short* y, int ldy, int M,
                        int N, int K, int mode, const EpiArgs2& ea, hipStream_t s) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int total = tiles_m * tiles_n, per_xcd = (total + 7) / 8;
  const dim3 grid(per_xcd * 8), block(256 * KG);
  constexpr size_t lds_ring = (size_t)KG * NST * (BM + BN) * 128, lds_sum = (size_t)BM * BN * 2 + (size_t)(KG - 1) * BM * BN * 4;
  constexpr size_t lds = KG > 1 && lds_sum > lds_ring ? lds_sum : lds_ring;
  static_assert(lds <= 160 * 1024, "LDS");
  if (KG > 1 && (K / 64) % KG != 0) {
    set_error("ifx_gemm_bf16: K/64 = %d is not a multiple of the %d K-groups of this tile", K / 64, KG);
    return IFX_EINVAL;
  }
#define IFX_LAUNCH_GS(E)                                                                                             \
  do {                                                                                                               \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      (void)hipFuncSetAttribute((const void*)gemm_small_kernel<BM, BN, NST, E, KG>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                           \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    hipLaunchKernelGGL((gemm_small_kernel<BM, BN, NST, E, KG>), grid, block, lds, s, x, ldx, w, y, ldy, M, N, K, tiles_m, total, \
                       per_xcd, ea);                                                                                 \
  } while (0)
  switch (mode) {
    case IFX_EPI_BIAS: IFX_LAUNCH_GS(IFX_EPI_BIAS); break;
    case IFX_EPI_GELU_TANH: IFX_LAUNCH_GS(IFX_EPI_GELU_TANH); break;
    case IFX_EPI_RESIDUAL: IFX_LAUNCH_GS(IFX_EPI_RESIDUAL); break;
    case IFX_EPI_GATE_RES: IFX_LAUNCH_GS(IFX_EPI_GATE_RES); break;
    default: return IFX_EINVAL;
  }
#undef IFX_LAUNCH_GS
  return check_launch("ifx_gemm_bf16(small)");
}

// tile: 0 = 256x128x64 (8 waves, ping-pong), 1 = 128x128, 2 = 64x64, 3 = 256x256x32 (4 stages), 4 = 128x64 (3 stages).  (A 256x128x32 six-stage instantiation of the same template, 120 KiB in flight,
// measured equal to tile 0 — 90.7 / 36.1 / 201 / 150 us on the four block GEMMs — and is not built.)
int launch_gemm_lds_dma(int tile, const unsigned short* x, int ldx, const unsigned short* w, unsigned short* y, int ldy,
                        int M, int N, int K, int mode, const unsigned short* bias, const unsigned short* residual,
                        int ld_res, const unsigned short* mod, int mod_slots, int gate_slot, int rows_per_group,
                        hipStream_t s);

int launch_gemm_w4(const unsigned short* x, int ldx, const unsigned short* w, unsigned short* y, int ldy, int M, int N, int K,
                   int mode, const unsigned short* bias, const unsigned short* residual, int ld_res, const unsigned short* mod,
                   int mod_slots, int gate_slot, int rows_per_group, hipStream_t s, int splits, void* workspace);

// host launcher used by ifx_gemm_bf16 (ifx_gemm.hip) for large shapes
int launch_gemm_glds(const unsigned short* x, int ldx, const unsigned short* w, unsigned short* y, int ldy, int M,
                     int N, int K, int mode, const unsigned short* bias, const unsigned short* residual, int ld_res,
                     const unsigned short* mod, int mod_slots, int gate_slot, int rows_per_group, hipStream_t s) {
  using namespace g2;
  EpiArgs2 ea{bias, residual, ld_res, mod, mod_slots, gate_slot, rows_per_group};
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int total = tiles_m * tiles_n, per_xcd = (total + 7) / 8;
  const dim3 grid(per_xcd * 8), block(512);
  const size_t lds = (size_t)NSTAGE * STAGE;
  static int ablate = -1;
  if (ablate < 0) { const char* e = getenv("IFX_GEMM_ABLATE"); ablate = e ? atoi(e) : 0; }
#define IFX_LAUNCH_G2(E)                                                                                      \
  do {                                                                                                        \
    static bool attr_set = false;                                                                             \
    if (!attr_set) {                                                                                          \
      (void)hipFuncSetAttribute((const void*)gemm_glds_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                    \
      attr_set = true;                                                                                        \
    }                                                                                                         \
    hipLaunchKernelGGL((gemm_glds_kernel<E>), grid, block, lds, s, x, ldx, w, y, ldy, M, N, K, tiles_m, total, \
                       per_xcd, ea, ablate);                                                                  \
  } while (0)
  switch (mode) {
    case IFX_EPI_BIAS: IFX_LAUNCH_G2(IFX_EPI_BIAS); break;
    case IFX_EPI_GELU_TANH: IFX_LAUNCH_G2(IFX_EPI_GELU_TANH); break;
    case IFX_EPI_RESIDUAL: IFX_LAUNCH_G2(IFX_EPI_RESIDUAL); break;
    case IFX_EPI_GATE_RES: IFX_LAUNCH_G2(IFX_EPI_GATE_RES); break;
    default: return IFX_EINVAL;
  }
#undef IFX_LAUNCH_G2
  return check_launch("ifx_gemm_bf16(glds)");
}

int launch_gemm_lds_dma(int tile, const unsigned short* x, int ldx, const unsigned short* w, unsigned short* y, int ldy,
                        int M, int N, int K, int mode, const unsigned short* bias, const unsigned short* residual,
                        int ld_res, const unsigned short* mod, int mod_slots, int gate_slot, int rows_per_group,
                        hipStream_t s) {
  if (tile == 0)
    return launch_gemm_glds(x, ldx, w, y, ldy, M, N, K, mode, bias, residual, ld_res, mod, mod_slots, gate_slot,
                            rows_per_group, s);
  EpiArgs2 ea{bias, residual, ld_res, mod, mod_slots, gate_slot, rows_per_group};
  if (tile == 1) return launch_small<128, 128, IFX_SMALL_NST>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);
  if (tile == 4) return launch_small<128, 64, IFX_SMALL_NST64>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);
  if (tile == 3) return launch_big<256, 256, 2, 2, 64>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);   // 128-byte operand rows: full-rate LDS-DMA (FFN up 171 -> 164 us)
  if (tile == 5) return launch_big<256, 128, 2, 2>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);   // two workgroups per CU
  if (tile == 9) return launch_big<256, 256, 2, 4>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);       // the 32-deep four-stage predecessor of tile 3
  if (tile == 7) return launch_ws<256, 128, 3>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);       // warp-specialised (producer / consumer waves)
  if (tile == 8) return launch_ws<128, 128, 2>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);       // same, 64 KiB: two workgroups per CU
  if (tile == 6) return launch_big<128, 128, 2, IFX_T8_NST>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);   // eight waves, several per CU
  // split-K inside the workgroup for launches of at most one workgroup per CU (a sequence-parallel rank's M = 4680 / P rows);
  // deeper rings of the single-group tiles do NOT help there (64x64 x 8 stages: 12.0 vs 11.4 us on 585x1536x1536)
  if (tile == 10) return launch_small<64, 64, 2, 4>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);    // 16 waves, four K-groups, 128 KiB
  if (tile == 11) return launch_small<64, 64, 4, 2>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);    // 8 waves, two K-groups, 128 KiB
  if (tile == 12) return launch_small<128, 128, 2, 2>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);  // 8 waves, two K-groups, 128 KiB
  if (tile == 13) return launch_small<128, 64, 2>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);      // 48 KiB: three per CU
  if (tile == 14) return launch_small<64, 128, 2>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);      // 48 KiB: three per CU
  if (tile == 15) return launch_small<64, 128, 3>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);      // 72 KiB: two per CU
  // FOUR waves of 128 x 128 on a 256 x 256 x 64 tile (256 accumulator registers per lane, one wave per SIMD): the eight-wave
  // tiles read (128 + 64) fragment rows per 8192 outputs from LDS = ~96 B/clk/CU of a 128 B/clk LDS next to 32 B/clk of DMA
  // writes; 128 x 128 wave tiles read a third less per FLOP
  if (tile == 16) return launch_small<256, 256, 2>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);
  // 256 x 192: the QKV projection's 4608 columns are 24 x 192 -> 456 tiles = 1.8 rounds where 256 x 256 has 1.3 (two rounds, a third idle)
  if (tile == 19) return launch_big<256, 192, 4, 2, 64>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);
  if (tile == 17) return launch_gemm_w4(x, ldx, w, y, ldy, M, N, K, mode, bias, residual, ld_res, mod, mod_slots, gate_slot, rows_per_group, s, 1, nullptr);   // ifx_gemm_w4.hip
  return launch_small<64, 64, 4>(x, ldx, w, y, ldy, M, N, K, mode, ea, s);
}

}  // namespace ifx
