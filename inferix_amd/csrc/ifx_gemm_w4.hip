// bf16 linear layer, FOUR-wave 256 x 256 x 64 tile (gfx950): one wave per SIMD, 128 x 128 outputs per wave in 256 accumulator
// registers, operands global -> VGPR -> LDS (no LDS-DMA), instruction-level software pipeline pinned with sched_group_barrier.
//
//   y[M,N] = epilogue( x[M,K] @ W[N,K]^T + bias[N] )            (same contract and epilogues as ifx_gemm_glds.hip)
//
// Why (profiles/r2_gemm_lds_budget.md): the eight-wave tiles of ifx_gemm_glds.hip are bound by LDS bandwidth, not by the matrix
// pipe and not only by operand delivery.  A 128 x 64 wave tile reads (128 + 64) fragment rows x 128 B per 8192 outputs and K-step:
// 8 waves x 24 KiB = 192 KiB per 2048 matrix-pipe cycles = 96 B/clk/CU, next to 32 B/clk/CU of LDS-DMA writes, of a 128 B/clk/CU
// LDS; the 128 x 128 two-per-CU tile needs 128 + 32.  A 128 x 128 wave tile reads 32 KiB per wave and K-step: 4 waves = 64 B/clk/CU.
// One wave per SIMD cannot hide anything behind a sibling wave, so
//   * the operands are fetched with plain buffer loads into staging registers TWO K-steps ahead (two register sets, 128 KiB in
//     flight per CU) and written to LDS with ds_write_b128 one K-step ahead — an LDS-DMA instruction would stall the only wave of
//     the SIMD ~65 cycles with nothing queued on the matrix pipe (tools/probe_overlap.hip);
//   * fragment reads of sub-step ks + 1, the LDS writes of tile kt + 1 and the global loads of tile kt + 3 are interleaved with the
//     16 MFMAs of sub-step ks (sched_group_barrier); the single barrier per K-step sits in front of the LAST sub-step's MFMAs, so
//     the first fragment reads of the next tile run under them.
// Layout, swizzle (16-byte chunk XOR ((row >> 1) & 7)), transposed MFMA tile and the LDS-transposed epilogue are those of the
// eight-wave kernels.
#include "ifx_common.h"

namespace ifx {

struct EpiArgsW4 {
  const unsigned short* bias;
  const unsigned short* residual;
  int ld_res;
  const unsigned short* mod;
  int mod_slots, gate_slot, rows_per_group;
};

namespace w4 {
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int A_OFF = 0, B_OFF = BM * BK * 2, STAGE = (BM + BN) * BK * 2;     // 64 KiB per stage, two stages
constexpr int WM = 128, WN = 128, TJ = 4, TI = 4;

}  // namespace w4

#define W4_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
// one sub-step: 16 MFMAs, 8 fragment reads, 8 "other memory" instructions (LDS writes or buffer loads)
#define W4_PIN_SUBSTEP()                \
  do {                                  \
    _Pragma("unroll") for (int _n = 0; _n < 8; ++_n) { \
      W4_SG(0x008, 2);  /* MFMA */      \
      W4_SG(0x100, 1);  /* DS read */   \
      W4_SG(0x200, 1);  /* DS write */  \
      W4_SG(0x020, 1);  /* VMEM read */ \
    }                                   \
    __builtin_amdgcn_sched_barrier(0);  \
  } while (0)

template <int EPI, int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w4_kernel(
    const unsigned short* __restrict__ x, int ldx, const unsigned short* __restrict__ w, unsigned short* __restrict__ y, int ldy,
    int M, int N, int K, int tiles_m, int total, int per_xcd, EpiArgsW4 ea, float* __restrict__ ws_part,
    unsigned* __restrict__ ws_cnt) {
  using namespace w4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
  const int w_id = xcd * per_xcd + slot_i;                       // work item = (tile, K split); the two splits of a tile are neighbours
  if (slot_i >= per_xcd || w_id >= total * KS) return;
  const int t_id = w_id / KS, split = w_id % KS, tile_lin = t_id;
  constexpr int GM = 4;
  const int tiles_n = total / tiles_m;
  const int grp_sz = GM * tiles_n;
  const int first_m = (t_id / grp_sz) * GM;
  const int gm = min(GM, tiles_m - first_m);
  const int rem = t_id % grp_sz;
  const int tile_m = first_m + rem % gm, tile_n = rem / gm;
  const int m_base = tile_m * BM, n_base = tile_n * BN;
  const int wm = wave & 1, wn = wave >> 1;

  // ---- global -> register staging: a wave instruction = 8 rows x 128 B (lane -> row lane >> 3, PHYSICAL chunk lane & 7, which holds
  //      LOGICAL chunk (lane & 7) ^ ((row >> 1) & 7)); piece p of this wave = rows (p * 4 + wave) * 8 ... + 7.  Rows past M / N read
  //      zeros through the buffer bounds.
  const int r8 = lane >> 3, pc = lane & 7;
  const int row0 = wave * 8 + r8;                                   // row of piece 0; piece p adds 32 rows: same (row >> 1) & 7 phase
  const int lc = pc ^ ((row0 >> 1) & 7);
  const long a_rows = (long)M - m_base, b_rows = (long)N - n_base;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(x + (size_t)m_base * ldx), 0, (int)min(a_rows * ldx * 2L, 0x7fffffffL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(w + (size_t)n_base * K), 0, (int)min(b_rows * (long)K * 2L, 0x7fffffffL), 0x00020000);
  const int voff_a = row0 * ldx * 2 + lc * 16, voff_b = row0 * K * 2 + lc * 16;
  const int pstep_a = 32 * ldx * 2, pstep_b = 32 * K * 2;
  // a row past the last one must not alias the next row's bytes: the per-row bound is enforced by making such lanes read offset
  // beyond num_records (ragged edge tiles only)
  const int dead = 0x7ffffff0;
  int va[8], vb[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    va[p] = (row0 + 32 * p < a_rows) ? voff_a + p * pstep_a : dead;
    vb[p] = (row0 + 32 * p < b_rows) ? voff_b + p * pstep_b : dead;
  }
  u32x4 ga0[8], gb0[8];
  const unsigned lds_piece = wave * 1024 + lane * 16;             // + (p * 4) * 1024 per piece

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = (K / BK) / KS, kt0 = split * KT;                // K steps of this split (the host checks divisibility)
  const int l31 = lane & 31, hi = lane >> 5;
  int a_row_off[TJ], b_row_off[TI];
  const int swz = (l31 >> 1) & 7;                                  // rows of a 32-block: block base is a multiple of 32
#pragma unroll
  for (int j = 0; j < TJ; ++j) a_row_off[j] = A_OFF + (wm * WM + j * 32 + l31) * 128;
#pragma unroll
  for (int i = 0; i < TI; ++i) b_row_off[i] = B_OFF + (wn * WN + i * 32 + l31) * 128;

#define W4_GLOAD_HALF(GA, GB, kt, h)                                                                                  \
  do {                                                                                                                \
    _Pragma("unroll") for (int p = 4 * (h); p < 4 * (h) + 4; ++p) {                                                   \
      GA[p] = __builtin_amdgcn_raw_buffer_load_b128(ra, va[p], (kt0 + (kt)) * 128, 0);                                \
      GB[p] = __builtin_amdgcn_raw_buffer_load_b128(rb, vb[p], (kt0 + (kt)) * 128, 0);                                \
    }                                                                                                                 \
  } while (0)
#define W4_LSTORE_HALF(GA, GB, st, h)                                                                                 \
  do {                                                                                                                \
    _Pragma("unroll") for (int p = 4 * (h); p < 4 * (h) + 4; ++p) {                                                   \
      *reinterpret_cast<u32x4*>((st) + A_OFF + p * 4096 + lds_piece) = GA[p];                                         \
      *reinterpret_cast<u32x4*>((st) + B_OFF + p * 4096 + lds_piece) = GB[p];                                         \
    }                                                                                                                 \
  } while (0)
#define W4_READ_FRAGS(FA, FB, st, ks)                                                                                 \
  do {                                                                                                                \
    const int _c = ((2 * (ks) + hi) ^ swz) << 4;                                                                      \
    _Pragma("unroll") for (int i = 0; i < TI; ++i) FB[i] = *reinterpret_cast<const bf16x8*>((st) + b_row_off[i] + _c); \
    _Pragma("unroll") for (int j = 0; j < TJ; ++j) FA[j] = *reinterpret_cast<const bf16x8*>((st) + a_row_off[j] + _c); \
  } while (0)
#define W4_MFMAS(FA, FB)                                                                                              \
  do {                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < TI; ++i)                                                                    \
      _Pragma("unroll") for (int j = 0; j < TJ; ++j)                                                                  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB[i], FA[j], acc[i][j], 0, 0, 0);                        \
  } while (0)

  bf16x8 fa0[TJ], fb0[TI], fa1[TJ], fb1[TI];
  // ---- prologue: tile 0 through the staging registers into stage 0, tile 1 requested, first fragments read
  W4_GLOAD_HALF(ga0, gb0, 0, 0);
  W4_GLOAD_HALF(ga0, gb0, 0, 1);
  W4_LSTORE_HALF(ga0, gb0, smem, 0);
  W4_LSTORE_HALF(ga0, gb0, smem, 1);
  {
    const int k1 = min(1, KT - 1);
    W4_GLOAD_HALF(ga0, gb0, k1, 0);
    W4_GLOAD_HALF(ga0, gb0, k1, 1);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  W4_READ_FRAGS(fa0, fb0, smem, 0);
  __builtin_amdgcn_sched_barrier(0);

  // One K-step on LDS stage kt & 1.  The staging registers hold tile kt + 1 (requested one K-step ago): its halves go to the other
  // stage during sub-steps 1 and 2 and each half is re-requested (tile kt + 2) one sub-step after it was stored, so a request has
  // three sub-steps (~1500 matrix-pipe cycles) to land.  Past the end the last tile is re-read (cache hit) and stored into the idle
  // stage: harmless, and the loop body stays branch-free so that the instruction interleave below can be pinned.
  for (int kt = 0; kt < KT; ++kt) {
    const unsigned char* cur = smem + (kt & 1) * STAGE;
    unsigned char* nxt = smem + ((kt + 1) & 1) * STAGE;
    const int kt2 = min(kt + 2, KT - 1);
    W4_READ_FRAGS(fa1, fb1, cur, 1);
    W4_MFMAS(fa0, fb0);
    W4_PIN_SUBSTEP();
    W4_READ_FRAGS(fa0, fb0, cur, 2);
    W4_LSTORE_HALF(ga0, gb0, nxt, 0);
    W4_MFMAS(fa1, fb1);
    W4_PIN_SUBSTEP();
    W4_READ_FRAGS(fa1, fb1, cur, 3);
    W4_LSTORE_HALF(ga0, gb0, nxt, 1);
    W4_GLOAD_HALF(ga0, gb0, kt2, 0);
    W4_MFMAS(fa0, fb0);
    W4_PIN_SUBSTEP();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    W4_READ_FRAGS(fa0, fb0, nxt, 0);
    W4_GLOAD_HALF(ga0, gb0, kt2, 1);
    W4_MFMAS(fa1, fb1);
    W4_PIN_SUBSTEP();
  }

  // ---- split-K across TWO workgroups of one tile (KS == 2): whoever finishes second adds the other's fp32 partial, read back in the
  //      accumulator layout it was dumped in, and runs the epilogue.  a + b == b + a in fp32, so the result does not depend on
  //      which of the two that is.  The per-tile arrival counter is reset by the finisher (the workspace starts zeroed).
  if (KS > 1) {
    float* part = ws_part + (size_t)tile_lin * (2 * 4 * WM * WN) + (size_t)wave * (WM * WN);   // [tile][split][wave][128 x 128]
    unsigned* cnt = ws_cnt + tile_lin;
    __shared__ unsigned s_old;
    // optimistic: dump, fence, count
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(part + (size_t)(((i * TJ + j) * 4 + q) * 64 + lane) * 4 + (size_t)split * (4 * WM * WN)) =
              f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
    __threadfence();
    __syncthreads();
    if (tid == 0) s_old = atomicAdd(cnt, 1u);
    __syncthreads();
    if (s_old == 0) return;                                     // first to arrive: the partner finishes the tile
    __threadfence();
    if (tid == 0) *cnt = 0;
    const float* other = part + (size_t)(1 - split) * (4 * WM * WN);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 pv = *reinterpret_cast<const f32x4*>(other + (size_t)(((i * TJ + j) * 4 + q) * 64 + lane) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += pv[e];
        }
  }
  // ---- epilogue: per-wave LDS transpose of v = bf16(acc + bias), then row-contiguous 16-byte accesses (direct 16-byte stores
  //      from the accumulators after a v_permlane32_swap were measured slower: FFN up 166 -> 198 us, partial-line writes)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  constexpr int RB = WN * 2, CR = RB / 16, RP = 64 / CR;     // 256-byte rows, 16 chunks per row, 4 rows per instruction
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
#pragma unroll 4
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
        if (ea.gate_slot) {
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

size_t gemm_w4_workspace_bytes(int M, int N, int splits) {
  using namespace w4;
  if (splits <= 1) return 0;
  const size_t tiles = (size_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  return 4096 + tiles * (size_t)splits * BM * BN * 4;       // arrival counters (up to 1024 tiles) in front, then the partials
}

// splits == 2: K halves in two workgroups per tile, `workspace` = gemm_w4_workspace_bytes(M, N, 2) bytes whose FIRST 4096 bytes
// (the arrival counters; the same place for every shape that shares the workspace) are zero on entry and are left zero on exit
int launch_gemm_w4(const unsigned short* x, int ldx, const unsigned short* w, unsigned short* y, int ldy, int M, int N, int K,
                   int mode, const unsigned short* bias, const unsigned short* residual, int ld_res, const unsigned short* mod,
                   int mod_slots, int gate_slot, int rows_per_group, hipStream_t s, int splits, void* workspace) {
  using namespace w4;
  EpiArgsW4 ea{bias, residual, ld_res, mod, mod_slots, gate_slot, rows_per_group};
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int total = tiles_m * tiles_n;
  const int items = total * (splits > 1 ? 2 : 1), per_xcd = (items + 7) / 8;
  const dim3 grid(per_xcd * 8), block(256);
  constexpr size_t lds = 2 * STAGE;
  unsigned* ws_cnt = (unsigned*)workspace;
  float* ws_part = workspace ? (float*)((char*)workspace + 4096) : nullptr;
  if (splits > 1 && ((K / BK) % 2 != 0 || workspace == nullptr || total > 1024)) {
    set_error("ifx_gemm_bf16: the split-K tile needs K/64 even (K = %d) and a workspace", K);
    return IFX_EINVAL;
  }
#define IFX_LAUNCH_W4(E, KSV)                                                                                                \
  do {                                                                                                                       \
    static bool attr_set = false;                                                                                            \
    if (!attr_set) {                                                                                                         \
      (void)hipFuncSetAttribute((const void*)gemm_w4_kernel<E, KSV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      attr_set = true;                                                                                                       \
    }                                                                                                                        \
    hipLaunchKernelGGL((gemm_w4_kernel<E, KSV>), grid, block, lds, s, x, ldx, w, y, ldy, M, N, K, tiles_m, total, per_xcd, ea, \
                       ws_part, ws_cnt);                                                                                     \
  } while (0)
#define IFX_SWITCH_W4(KSV)                                                  \
  switch (mode) {                                                           \
    case IFX_EPI_BIAS: IFX_LAUNCH_W4(IFX_EPI_BIAS, KSV); break;             \
    case IFX_EPI_GELU_TANH: IFX_LAUNCH_W4(IFX_EPI_GELU_TANH, KSV); break;   \
    case IFX_EPI_RESIDUAL: IFX_LAUNCH_W4(IFX_EPI_RESIDUAL, KSV); break;     \
    case IFX_EPI_GATE_RES: IFX_LAUNCH_W4(IFX_EPI_GATE_RES, KSV); break;     \
    default: return IFX_EINVAL;                                             \
  }
  if (splits > 1) { IFX_SWITCH_W4(2) } else { IFX_SWITCH_W4(1) }
#undef IFX_SWITCH_W4
#undef IFX_LAUNCH_W4
  return check_launch("ifx_gemm_bf16(w4)");
}

}  // namespace ifx
