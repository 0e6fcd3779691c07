// bf16 linear layer on MFMA with fused epilogues (gfx950).
//
//   y[M,N] = epilogue( x[M,K] @ W[N,K]^T + bias[N] )
//
// Both operands are K-contiguous (activations token-major, nn.Linear weights
// [out,in]), which is the natural MFMA layout: every lane's 8-element fragment is one
// 16-byte load.  The kernel computes the TRANSPOSED tile D = W_tile · x_tile^T
// (A-operand = W rows, B-operand = x rows) so that each lane ends up with 4
// CONSECUTIVE output channels of one token: bias / gate / residual / store are
// 8-byte vector accesses along the contiguous dimension.
//
// Tiling (v1): 128(tokens) x 128(channels) x 64(K) per workgroup, 4 waves (2x2),
// each wave 64x64 = 4x4 fragments of v_mfma_f32_16x16x32_bf16; operands staged
// global -> registers -> LDS (double buffered, one barrier per K-tile, loads for
// tile t+1 issued before the MFMAs of tile t); LDS rows are 128 B with the 16-byte
// chunk index XOR-swizzled by (row & 7) so ds_read_b128 fragment reads are
// bank-conflict free.  64 KiB LDS -> 2 workgroups per CU.
#include <stdlib.h>

#include "ifx_common.h"

namespace ifx {

constexpr int BM = 128, BN = 128, BK = 64;

struct EpiArgs {
  const unsigned short* bias;
  const unsigned short* residual;
  int ld_res;
  const unsigned short* mod;
  int mod_slots, gate_slot, rows_per_group;
};

// exact (erf) GELU as torch.nn.functional.gelu evaluates it on a bf16 tensor: fp32 math, one rounding (MAGI CustomMLP,
// inferix/models/magi/dit/dit_module.py:552).  Selected at run time inside the GELU epilogue instantiation: the epilogue's
// otherwise unused `gate_slot` field carries 1 for IFX_EPI_GELU_ERF.

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const unsigned short* __restrict__ x, int ldx,
                                                           const unsigned short* __restrict__ w,
                                                           unsigned short* __restrict__ y, int ldy, int M, int N,
                                                           int K, int tiles_m, int tiles_n, EpiArgs ea) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [buf][X|W][128 rows][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  // XCD-aware grouped raster (see ifx_gemm_glds.hip): XCD (bid & 7) owns a contiguous id range; ids walk
  // GM token-tiles then the next channel-tile, so co-resident workgroups share operand panels in L2.
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

  // staging assignment: 4 chunks of 16 B per matrix per thread
  const unsigned short* gx[4];
  const unsigned short* gw[4];
  int lds_off[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int q = tid + 256 * p, row = q >> 3, c = q & 7;
    const int mr = min(m_base + row, M - 1), nr = min(n_base + row, N - 1);
    gx[p] = x + (size_t)mr * ldx + c * 8;
    gw[p] = w + (size_t)nr * K + c * 8;
    lds_off[p] = row * 128 + ((c ^ (row & 7)) << 4);
  }
  u32x4 rx[4], rw[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      rx[p] = *reinterpret_cast<const u32x4*>(gx[p] + (size_t)kt * BK);
      rw[p] = *reinterpret_cast<const u32x4*>(gw[p] + (size_t)kt * BK);
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

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KT = K / BK;
  gload(0);
  lstore(0);
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) gload(kt + 1);
    const unsigned char* bx = smem + buf * 32768;
    const unsigned char* bw = bx + 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], b[4];
      const int c = ks * 4 + fq;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wn * 64 + i * 16 + fr;
        a[i] = *reinterpret_cast<const bf16x8*>(bw + row * 128 + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wm * 64 + j * 16 + fr;
        b[j] = *reinterpret_cast<const bf16x8*>(bx + row * 128 + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < KT) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds D[n = n0+16i+4*fq+{0..3}][m = m0+16j+fr]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m_base + wm * 64 + j * 16 + fr;
    if (m >= M) continue;
    const unsigned short* gate_row = nullptr;
    if (EPI == IFX_EPI_GATE_RES)
      gate_row = ea.mod + ((size_t)(m / ea.rows_per_group) * ea.mod_slots + ea.gate_slot) * N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n_base + wn * 64 + i * 16 + fq * 4;
      if (n >= N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
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
      *reinterpret_cast<u16x4*>(y + (size_t)m * ldy + n) = o;
    }
  }
}

int launch_gemm_glds(const unsigned short* x, int ldx, const unsigned short* w, unsigned short* y, int ldy, int M,
                     int N, int K, int mode, const unsigned short* bias, const unsigned short* residual, int ld_res,
                     const unsigned short* mod, int mod_slots, int gate_slot, int rows_per_group, hipStream_t s);

int launch_gemm_w4(const unsigned short* x, int ldx, const unsigned short* w, unsigned short* y, int ldy, int M, int N, int K,
                   int mode, const unsigned short* bias, const unsigned short* residual, int ld_res, const unsigned short* mod,
                   int mod_slots, int gate_slot, int rows_per_group, hipStream_t s, int splits, void* workspace);
size_t gemm_w4_workspace_bytes(int M, int N, int splits);
// persistent ping-pong tiles (ifx_gemm_pp.hip): 64 tj tokens x 256 channels, K split in two when gemm_pp_split(N, K) and a workspace is given
int launch_gemm_pp(const unsigned short* x, int ldx, const unsigned short* w, unsigned short* y, int ldy, int M, int N, int K,
                   int mode, const unsigned short* bias, const unsigned short* residual, int ld_res, const unsigned short* mod,
                   int mod_slots, int gate_slot, int rows_per_group, hipStream_t s, int tj, void* workspace, const float* q8_sa = nullptr,
                   const float* q8_sw = nullptr, const float* q8_qdiv = nullptr, int q8_via_bf16 = 0, int stream_k = 0, int q8_int8 = 0,
                   int force_ks = 0, unsigned short* y2 = nullptr, int ldy2 = 0, int split_col = 0);
size_t gemm_pp_small_workspace_bytes(int M, int N, int ks);
size_t gemm_pp_stream_k_workspace_bytes();
bool gemm_pp_split(int N, int K);
size_t gemm_pp_workspace_bytes(int M, int N, int K);

// Which ping-pong tile (tokens = 64 tj) for a launch, 0 = none.  Model fitted to tools/gemm_lab.cpp on the block's shapes (1 x MI355X,
// profiles/r3_gemm_pp.md): a workgroup needs ~1.6 / 1.4 / 1.2 us per 64-deep K-step on the 256 / 192 / 128-token tile (the smaller
// tiles are bound by their loader phase, not by the matrix pipe) plus ~3 us per tile (8 with a GELU epilogue); the launch takes
// ceil(tiles / CUs) of those.  With K split in two (gemm_pp_split) there are twice the work items of half the length, 256-token tile only.
static int pick_pp(int M, int N, int K, int mode, bool have_ws) {
  if (N % 64 != 0 || K % 64 != 0) return 0;
  // the split shapes at ANY row count (a row's summation order must not depend on it) unless the caller opted into the row-count
  // dependent in-workgroup splits for small launches
  if (have_ws && gemm_pp_split(N, K) && ((M + 255) / 256) * ((N + 255) / 256) <= 1024 && !(gemm_small_split() && M < 2048)) {
    // 256 or 192 tokens per tile: twice the work items of half the K-steps; the height that needs less time by the rounds model (the
    // bits do not depend on it).  4680 rows: 228 items = one round of the 256-token tile; 10800 rows (720p): 516 items = three rounds
    // against three shorter ones of the 192-token tile's 684
    const int tn = (N + 255) / 256, steps = K / 64 / 2;
    const int i4 = ((M + 255) / 256) * tn * 2, i3 = ((M + 191) / 192) * tn * 2;
    const float t4 = ((i4 + 255) / 256) * (steps * 1.6f + 9.f), t3 = ((i3 + 255) / 256) * (steps * 1.4f + 9.f);
    static int allow3 = -1;
    if (allow3 < 0) {
      const char* e = getenv("IFX_PP_SPLIT_TJ3");     // lab: 0 = the split on the 256-token tile only (rounds 3-4 before this rule)
      allow3 = e ? atoi(e) : 1;
    }
    return (allow3 && i3 <= 2048 && t3 < t4 && mode != IFX_EPI_GELU_TANH) ? 3 : 4;
  }
  if (M < 1024) return 0;
  static const float step_us[5] = {0.f, 0.f, 1.2f, 1.4f, 1.6f};
  const float tile_us = (mode == IFX_EPI_GELU_TANH ? 8.f : 3.f);
  int best = 0, best_tiles = 0;
  float best_t = 1e30f;
  for (int tj = 4; tj >= 2; --tj) {
    const int tiles = ((M + 64 * tj - 1) / (64 * tj)) * ((N + 255) / 256);
    const int rounds = (tiles + 255) / 256;
    const float t = rounds * ((K / 64) * step_us[tj] + tile_us);
    if (t < best_t) best_t = t, best = tj, best_tiles = tiles;
  }
  // 1024 .. 2047 rows (one MAGI chunk of a cp rank, 1519 rows): only where the chosen tile still gives most CUs a tile — 1519 x 8192 x
  // 3072 99 -> 70 us, x 12288 153 -> 128; with fewer tiles the LDS-DMA tiles with several workgroups per CU are ahead (1170 x 1536^2: 15 vs 27 us)
  if (M < 2048 && best_tiles < 160) return 0;
  return best;
}

// Split-K over two workgroups on the four-wave 256 x 256 tile (ifx_gemm_w4.hip): long-K launches whose 256 x 256 tiles fill less
// than half of the chip — the FFN down-projection, 4680 x 1536 x 8960: 114 tiles, 228 workgroups with the split.  Needs a caller
// workspace, so only ifx_gemm_bf16_ws takes this path.
static bool want_w4_splitk(int M, int N, int K) {
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
  return K >= 4096 && (K / 64) % 2 == 0 && 2 * tiles <= 256 && 2 * tiles >= 160;
}

int launch_gemm_lds_dma(int tile, const unsigned short* x, int ldx, const unsigned short* w, unsigned short* y, int ldy,
                        int M, int N, int K, int mode, const unsigned short* bias, const unsigned short* residual,
                        int ld_res, const unsigned short* mod, int mod_slots, int gate_slot, int rows_per_group,
                        hipStream_t s);

// LDS-DMA tile choice: score = (measured relative throughput of the tile at full occupancy) x (how well the launch's
// workgroups fill whole rounds of the chip).  256 CUs; the 64-wide tiles run two workgroups per CU.
//   tile 3 = 256x256x32, 0 = 256x128x64, 1 = 128x128, 4 = 128x64, 2 = 64x64   (bench: tools/bench_kernels.py gemm M)
//   tile 10 = 64x64 with four K-groups, 11 = 64x64 with two K-groups, 12 = 128x128 with two K-groups (split-K inside the workgroup)
//   tile 13 = 128x64 with two stages (three workgroups per CU), 14 / 15 = 64x128 with two / three stages
static int pick_tile(int M, int N, int K) {
  struct Cand { int tile, bm, bn, slots; float base; };
  // 128x128 runs double-buffered with TWO workgroups per CU (64 KiB of LDS each): the co-resident workgroup hides the operand
  // latency better than the deep rings of the big tiles do — QKV 92 -> 83 us, O 38 -> 35, FFN down 150 -> 137 (tools/bench_gemm_tiles.py)
  // 256x192 (tile 19, round 2): 7 % more outputs per round-microsecond than the two-per-CU 128x128 tile, 8 % fewer than 256x256; it wins
  // where 192-wide columns quantise better — the QKV projection, N = 4608 = 24 x 192: 456 tiles = 1.8 rounds against 1.3 of 256x256
  // (interleaved A/B, tools/scratch/ab_gemm_tiles.py: 4680x4608x1536 82.4 -> 75.3 us, 2340x4608x1536 47.1 -> 42.1, 2340x8960x1536
  // 83.7 -> 79.3; not picked, and slower, at 4680x8960 147 vs 142, 9360x4608 154 vs 144, 10800x4608 193 vs 181).  Same K order as the
  // other single-pass tiles: bit-identical outputs.
  static const Cand cands[] = {{3, 256, 256, 256, 1.00f}, {19, 256, 192, 256, 0.99f}, {0, 256, 128, 256, 0.85f}, {1, 128, 128, 512, 0.95f},
                               {4, 128, 64, 512, 0.68f},  {2, 64, 64, 512, 0.50f}};
  // Few output tiles (one rank's M = 4680 / P rows of a sequence-parallel shard): the parallelism has to come from K.  Tiles 10-12
  // split K between wave groups of ONE workgroup (no workspace, fixed summation order).  Measured, tools/bench_gemm_tiles.py 585 / 2340:
  //   585x1536x1536 12.3 -> 9.8 us (tile 10), 585x1536x8960 43.2 -> 33.8 (tile 11)
  // They add the K halves / quarters in an order that is NOT the single-pass order, and whether a launch gets them depends on its ROW
  // count — so they are opt-in (ifx_set_option("gemm_small_split", 1); inferix_amd.sequence_parallel turns it on for its ranks, whose
  // row count is fixed by the rank count).  Without it the auto choice's summation order is a function of (N, K) alone: single pass
  // everywhere except the two-workgroup K split of ifx_gemm_pp.hip (gemm_pp_split(N, K), any row count, needs the workspace of
  // ifx_gemm_bf16_ws) — a row's bits do not change with the number of rows in the launch (tests/test_hip_kernels.py).
  const int kt = K / 64;
  const int wgs64 = ((M + 63) / 64) * ((N + 63) / 64);
  if (gemm_small_split()) {
    if (wgs64 <= 256 && kt >= 64 && kt % 2 == 0) return 11;
    if (wgs64 <= 256 && kt >= 16 && kt % 4 == 0) return 10;
  }
  // one round of 128x64 workgroups at THREE per CU (two stages, 48 KiB) instead of 1.4 rounds at two per CU:
  //   585x8960x1536 31.8 -> 29.2 us, 585x4608x1536 19.2 -> 17.7, 1170x4608x1536 28.6 -> 27.2
  const int wgs12864 = ((M + 127) / 128) * ((N + 63) / 64);
  if (wgs12864 > 256 && wgs12864 <= 768) return 13;
  // (tile 12 on 128 < wgs128 <= 256 launches is worth 10 % on 2340x1536x8960; it stays opt-in because it would flip the umT5
  //  encoder's M = 512 x batch launches between summation orders.)
  // Long K and at least ~1.5 rounds of 256 x 256 tiles: the four-wave register-staged tile (ifx_gemm_w4.hip, tile 17), whose K loop
  // needs a third less LDS bandwidth than the eight-wave tiles; its fixed cost per tile (one wave per SIMD: nothing overlaps the
  // prologue and the epilogue) is only amortised by K >= 2048.  MAGI-4.5B shapes, tools/bench_gemm_shapes.py: 6075 x 8192 x 3072
  // 326 -> 261 us, 6075 x 12288 x 3072 494 -> 409, 12150 x 3072 x 12288 1033 -> 900; the Wan block's K = 1536 GEMMs stay where they were.
  {
    const int t256 = ((M + 255) / 256) * ((N + 255) / 256), rounds = (t256 + 255) / 256;
    if (K >= 2048 && t256 >= 384 && (float)t256 >= 0.7f * 256.f * rounds) return 17;
  }
  int best = 2;
  float best_score = -1.f;
  for (const Cand& c : cands) {
    const int tm = (M + c.bm - 1) / c.bm, tn = (N + c.bn - 1) / c.bn, wgs = tm * tn;
    const int rounds = (wgs + c.slots - 1) / c.slots;
    const float edge = ((float)M * (float)N) / ((float)(tm * c.bm) * (float)(tn * c.bn));   // work in ragged edge tiles
    // less than one round of 256x256 tiles WITH a ragged edge: 128x128 wins (2340x4608: 46.6 vs 48.7 us, 1170x8960: 51.4 vs 53.9);
    // without one (the text encoder's 512x20480x4096) the big tile stays ahead (9.5 vs 10.0 ms per prompt)
    if (c.tile == 3 && wgs < c.slots && edge < 0.95f) continue;
    const float score = c.base * edge * (float)wgs / (float)(c.slots * rounds);
    if (score > best_score) best_score = score, best = c.tile;
  }
  return best;
}

// kernel selection: 0 = auto (LDS-DMA kernels, tile by shape), 1 = force the register-staged 128x128 kernel,
// 2 = force 256x128, 3 = force LDS-DMA 128x128, 4 = force LDS-DMA 64x64, 5 = force 256x256x32, 6 = force LDS-DMA 128x64,
// 7 / 8 = the two-per-CU 256x128x32 and eight-wave 128x128x32 experiments (slower than 3, kept selectable),
// 12 / 13 / 14 = the split-K-inside-the-workgroup tiles 10 / 11 / 12 (small launches), 15 = 128x64 three per CU, 16 / 17 = 64x128,
// 18 / 19 = the four-wave 256x256 tiles, 20 = split-K over two workgroups (needs a workspace), 21 = 256x192
}  // namespace ifx

using namespace ifx;

// ONE definition of "this launch takes the in-workgroup split 128 x 128 tile ahead of the ping-pong tiles", used by the launcher and by
// ifx_gemm_workspace_bytes (the two had drifted — the workspace query still reported up to 75 MB of split-K scratch for
// launches that never touch it)
static bool small_split_takes_tile12(int M, int N, int K) {
  if (!gemm_small_split() || N > 2048 || (K / 64) % 2 != 0 || K < 1024) return false;
  const int wgs128 = ((M + 127) / 128) * ((N + 127) / 128);
  return wgs128 > 128 && wgs128 <= 256;
}

// Shard-sized long-K launches (gemm_small_split): the 128-token ping-pong tile with K split over FOUR workgroups per tile where that
// makes one round of 129 .. 256 work items — a 4-way rank's FFN down-projection, 1170 x 1536 x 8960: 60 tiles x 4 = 240 items of 35
// K-steps, 52.0 us against 56.6 for the in-workgroup split 64 x 64 tiles (tools/bench_gemm_tiles.py 1170 0,27,28,29, round 5).  Measured
// and NOT taken everywhere else: 585 rows 46.7 (4-way) against 35.5, 2340 rows 80.2 (2-way) against 81.0, every K = 1536 launch behind
// the auto choice by 1.2 - 3x (the owner reads ks - 1 partial tiles: 128 KiB each through one CU).
static bool small_split_takes_pp_ks4(int M, int N, int K) {
  if (!gemm_small_split() || M >= 2048 || N > 2048 || N % 64 != 0 || K < 4096 || K % 64 != 0 || (K / 64) % 4 != 0) return false;
  const int items = ((M + 127) / 128) * ((N + 255) / 256) * 4;
  return items > 128 && items <= 256;
}

static int gemm_bf16_impl(const ifx_bf16* x, int32_t ldx, const ifx_bf16* w, const ifx_bf16* bias, ifx_bf16* y,
                          int32_t ldy, int32_t M, int32_t N, int32_t K, const ifx_epilogue* epi, void* stream, void* workspace,
                          int64_t workspace_bytes) {
  IFX_REQUIRE(x && w && y && M >= 0 && N > 0 && K > 0, "ifx_gemm_bf16: null/empty operand");
  IFX_REQUIRE(K % BK == 0, "ifx_gemm_bf16: K (%d) must be a multiple of %d", K, BK);   // 64; also covers the 32-deep tiles
  IFX_REQUIRE(N % 4 == 0 && ldx % 8 == 0 && ldy % 4 == 0, "ifx_gemm_bf16: N %% 4, ldx %% 8, ldy %% 4 required");
  int mode = epi ? epi->epilogue : IFX_EPI_BIAS;
  EpiArgs ea{bias, nullptr, 0, nullptr, 1, 0, 1};
  if (mode == IFX_EPI_GELU_ERF) {       // the GELU instantiation with the exact-erf activation selected at run time
    mode = IFX_EPI_GELU_TANH;
    ea.gate_slot = 1;
  }
  if (mode == IFX_EPI_RESIDUAL || mode == IFX_EPI_GATE_RES) {
    IFX_REQUIRE(epi->residual && epi->ld_res % 4 == 0, "ifx_gemm_bf16: residual epilogue needs residual/ld_res");
    ea.residual = epi->residual;
    ea.ld_res = epi->ld_res;
  }
  if (mode == IFX_EPI_GATE_RES) {
    IFX_REQUIRE(epi->mod && epi->rows_per_group > 0 && epi->gate_slot >= 0 && epi->gate_slot < epi->mod_slots,
                "ifx_gemm_bf16: gate epilogue needs mod/mod_slots/gate_slot/rows_per_group");
    ea.mod = epi->mod;
    ea.mod_slots = epi->mod_slots;
    ea.gate_slot = epi->gate_slot;
    ea.rows_per_group = epi->rows_per_group;
  }
  if (M == 0) return IFX_OK;
  const int variant = gemm_variant();
  const bool wide_ok = N % 8 == 0 && ldy % 8 == 0 && (ea.residual == nullptr || ea.ld_res % 8 == 0);
  // (measured on the FFN down-projection: 191 us against 164 us for the 128 x 128 two-per-CU tile — with 228 workgroups streaming
  //  1 GB of operands the launch is paced by memory-side latency x bytes in flight, not by the K loop — so it is opt-in: variant 20)
  if (wide_ok && variant == 20 && workspace != nullptr && want_w4_splitk(M, N, K) &&
      workspace_bytes >= (int64_t)gemm_w4_workspace_bytes(M, N, 2))
    return launch_gemm_w4(x, ldx, w, y, ldy, M, N, K, mode, ea.bias, ea.residual, ea.ld_res, ea.mod, ea.mod_slots, ea.gate_slot,
                          ea.rows_per_group, (hipStream_t)stream, 2, workspace);
  // persistent ping-pong tiles: 22 / 23 / 24 force the 256 / 192 / 128-token tile (25 = 256 without the K split), 0 = auto picks one for
  // launches of at least 2048 rows; the gate epilogue needs groups of at least a wave's token rows, the operands 16-byte rows
  // stream-K on the 128-token ping-pong tile, meant for the shard-sized launches of a sequence-parallel rank (the K partition, hence the
  // summation order of a row, depends on the row count of the launch).
  // MEASURED AND NOT IN THE AUTO CHOICE (gemm_variant 26 only, profiles/r3_gemm_pp.md): at 585 rows a tile's K range is spread over 3-9
  // workgroups and the owner's epilogue reads that many partial sums one memory latency after the other — 36 us against 19 us (QKV), 79
  // against 37 us (FFN down) for the in-workgroup split tiles below.
  if (wide_ok && variant == 26 && N % 64 == 0 && K % 64 == 0 && workspace != nullptr &&
      workspace_bytes >= (int64_t)gemm_pp_stream_k_workspace_bytes() && (long)((M + 127) / 128) * ((N + 255) / 256) * (K / 64) >= 48) {
    const bool res = mode == IFX_EPI_RESIDUAL || mode == IFX_EPI_GATE_RES;
    const bool fits = !((uintptr_t)bias & 7) && (!res || (!((uintptr_t)ea.residual & 15) && ea.ld_res % 8 == 0)) &&
                      (mode != IFX_EPI_GATE_RES || (!((uintptr_t)ea.mod & 15) && ea.rows_per_group >= 64));
    if (fits)
      return launch_gemm_pp(x, ldx, w, y, ldy, M, N, K, mode, ea.bias, ea.residual, ea.ld_res, ea.mod, ea.mod_slots, ea.gate_slot,
                            ea.rows_per_group, (hipStream_t)stream, 2, workspace, nullptr, nullptr, nullptr, 0, 1);
  }
  // second destination (ifx_epilogue.y2: the q|k|v projection storing its V columns straight into the KV cache): the ping-pong tiles
  // carry it; every other tile would drop the columns, so a launch that cannot take that path is an error, never a silent fallback
  if (epi != nullptr && epi->y2 != nullptr) {
    IFX_REQUIRE(mode == IFX_EPI_BIAS && wide_ok && N % 64 == 0 && !((uintptr_t)bias & 7) && (variant == 0 || (variant >= 22 && variant <= 25)),
                "ifx_gemm_bf16: the second destination (y2) needs the bias epilogue, N %% 64 == 0, 8-byte aligned bias and the ping-pong tiles");
    int tj = variant == 0 ? pick_pp(M, N, K, mode, false) : (variant == 22 || variant == 25 ? 4 : variant == 23 ? 3 : 2);
    if (tj == 0) tj = 2;
    return launch_gemm_pp(x, ldx, w, y, ldy, M, N, K, mode, ea.bias, nullptr, 0, nullptr, 1, 0, 1, (hipStream_t)stream, tj, nullptr, nullptr,
                          nullptr, nullptr, 0, 0, 0, 0, epi->y2, epi->ldy2, epi->split_col);
  }
  // lab (gemm_variant 27 / 28 / 29): the 128-token ping-pong tile with K split over 2 / 4 / 8 workgroups per tile, whatever the shape
  if (wide_ok && variant >= 27 && variant <= 29) {
    const int ks = 2 << (variant - 27);
    const bool res = mode == IFX_EPI_RESIDUAL || mode == IFX_EPI_GATE_RES;
    const bool fits = N % 64 == 0 && K % 64 == 0 && (K / 64) % ks == 0 && workspace != nullptr &&
                      workspace_bytes >= (int64_t)gemm_pp_small_workspace_bytes(M, N, ks) &&
                      (long)((M + 127) / 128) * ((N + 255) / 256) * (ks - 1) <= 1024 && !((uintptr_t)bias & 7) &&
                      (!res || (!((uintptr_t)ea.residual & 15) && ea.ld_res % 8 == 0)) &&
                      (mode != IFX_EPI_GATE_RES || (!((uintptr_t)ea.mod & 15) && ea.rows_per_group >= 64));
    if (fits)
      return launch_gemm_pp(x, ldx, w, y, ldy, M, N, K, mode, ea.bias, ea.residual, ea.ld_res, ea.mod, ea.mod_slots, ea.gate_slot,
                            ea.rows_per_group, (hipStream_t)stream, 2, workspace, nullptr, nullptr, nullptr, 0, 0, 0, ks);
  }
  // Under gemm_small_split (a sequence-parallel rank: row-count dependent summation orders are allowed) narrow-N launches whose 128 x 128
  // tiles make ONE round of at most 256 workgroups take tile 12 — 128 x 128 with K split between the two wave groups of the workgroup —
  // ahead of the ping-pong tiles: at 2340 rows (P = 2) 19 x 12 = 228 workgroups against 60-120 work items of the 256-token ping-pong
  // tile.  tools/bench_gemm_tiles.py 2340: 1536^2 28.1 / 25.8 -> 23.3 / 19.3 us (+ residual / bias only), 1536 x 8960 102.6 -> 82.1.
  if (wide_ok && variant == 0 && small_split_takes_pp_ks4(M, N, K) && workspace != nullptr &&
      workspace_bytes >= (int64_t)gemm_pp_small_workspace_bytes(M, N, 4)) {
    const bool res = mode == IFX_EPI_RESIDUAL || mode == IFX_EPI_GATE_RES;
    const bool fits = !((uintptr_t)bias & 7) && (!res || (!((uintptr_t)ea.residual & 15) && ea.ld_res % 8 == 0)) &&
                      (mode != IFX_EPI_GATE_RES || (!((uintptr_t)ea.mod & 15) && ea.rows_per_group >= 64));
    if (fits)
      return launch_gemm_pp(x, ldx, w, y, ldy, M, N, K, mode, ea.bias, ea.residual, ea.ld_res, ea.mod, ea.mod_slots, ea.gate_slot,
                            ea.rows_per_group, (hipStream_t)stream, 2, workspace, nullptr, nullptr, nullptr, 0, 0, 0, 4);
  }
  if (wide_ok && variant == 0 && small_split_takes_tile12(M, N, K))
    return launch_gemm_lds_dma(12, x, ldx, w, y, ldy, M, N, K, mode, ea.bias, ea.residual, ea.ld_res, ea.mod, ea.mod_slots, ea.gate_slot,
                               ea.rows_per_group, (hipStream_t)stream);
  if (wide_ok && (variant == 0 || (variant >= 22 && variant <= 25))) {
    const bool ws_ok = workspace != nullptr && workspace_bytes >= (int64_t)gemm_pp_workspace_bytes(M, N, K) && variant != 25;
    int tj = variant == 0 ? pick_pp(M, N, K, mode, ws_ok) : (variant == 22 || variant == 25 ? 4 : variant == 23 ? 3 : 2);
    const bool res = mode == IFX_EPI_RESIDUAL || mode == IFX_EPI_GATE_RES;
    const bool fits = N % 64 == 0 && !((uintptr_t)bias & 7) && (!res || (!((uintptr_t)ea.residual & 15) && ea.ld_res % 8 == 0)) &&
                      (mode != IFX_EPI_GATE_RES || (!((uintptr_t)ea.mod & 15) && ea.rows_per_group >= 32 * tj));
    if (tj && (fits || variant != 0))
      return launch_gemm_pp(x, ldx, w, y, ldy, M, N, K, mode, ea.bias, ea.residual, ea.ld_res, ea.mod, ea.mod_slots, ea.gate_slot,
                            ea.rows_per_group, (hipStream_t)stream, tj, ws_ok ? workspace : nullptr);
  }
  if (wide_ok && variant != 1) {
    const int tile = (variant >= 2 && variant != 20 && variant != 26) ? variant - 2 : pick_tile(M, N, K);
    return launch_gemm_lds_dma(tile, x, ldx, w, y, ldy, M, N, K, mode, ea.bias, ea.residual, ea.ld_res, ea.mod,
                               ea.mod_slots, ea.gate_slot, ea.rows_per_group, (hipStream_t)stream);
  }
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const dim3 grid(((tiles_m * tiles_n + 7) / 8) * 8), block(256);
  const size_t lds = 65536;
  hipStream_t s = (hipStream_t)stream;
#define IFX_LAUNCH_GEMM(E)                                                                                    \
  do {                                                                                                        \
    static bool attr_set = false;                                                                             \
    if (!attr_set) {                                                                                          \
      (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                    \
      attr_set = true;                                                                                        \
    }                                                                                                         \
    hipLaunchKernelGGL((gemm_bf16_kernel<E>), grid, block, lds, s, x, ldx, w, y, ldy, M, N, K, tiles_m, tiles_n, ea);  \
  } while (0)
  switch (mode) {
    case IFX_EPI_BIAS: IFX_LAUNCH_GEMM(IFX_EPI_BIAS); break;
    case IFX_EPI_GELU_TANH: IFX_LAUNCH_GEMM(IFX_EPI_GELU_TANH); break;
    case IFX_EPI_RESIDUAL: IFX_LAUNCH_GEMM(IFX_EPI_RESIDUAL); break;
    case IFX_EPI_GATE_RES: IFX_LAUNCH_GEMM(IFX_EPI_GATE_RES); break;
    default: set_error("ifx_gemm_bf16: unknown epilogue %d", mode); return IFX_EINVAL;
  }
#undef IFX_LAUNCH_GEMM
  return check_launch("ifx_gemm_bf16");
}

extern "C" int ifx_gemm_bf16(const ifx_bf16* x, int32_t ldx, const ifx_bf16* w, const ifx_bf16* bias, ifx_bf16* y,
                             int32_t ldy, int32_t M, int32_t N, int32_t K, const ifx_epilogue* epi, void* stream) {
  return gemm_bf16_impl(x, ldx, w, bias, y, ldy, M, N, K, epi, stream, nullptr, 0);
}

extern "C" int64_t ifx_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int v = gemm_variant();
  if (v == 20) return want_w4_splitk(M, N, K) ? (int64_t)gemm_w4_workspace_bytes(M, N, 2) : 0;
  if (v == 26 && N % 64 == 0 && K % 64 == 0) return (int64_t)gemm_pp_stream_k_workspace_bytes();
  if (v >= 27 && v <= 29) return (N % 64 == 0 && K % 64 == 0 && (K / 64) % (2 << (v - 27)) == 0) ? (int64_t)gemm_pp_small_workspace_bytes(M, N, 2 << (v - 27)) : 0;
  if (v == 0 && small_split_takes_pp_ks4(M, N, K)) return (int64_t)gemm_pp_small_workspace_bytes(M, N, 4);
  if (v == 0 && N % 8 == 0 && small_split_takes_tile12(M, N, K)) return 0;      // the launcher's shortcut (wide_ok needs N % 8 == 0)
  if ((v == 0 && N % 64 == 0 && K % 64 == 0 && !(gemm_small_split() && M < 2048)) || v == 22) return (int64_t)gemm_pp_workspace_bytes(M, N, K);
  return 0;
}

extern "C" int ifx_gemm_bf16_ws(const ifx_bf16* x, int32_t ldx, const ifx_bf16* w, const ifx_bf16* bias, ifx_bf16* y,
                                int32_t ldy, int32_t M, int32_t N, int32_t K, const ifx_epilogue* epi, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  return gemm_bf16_impl(x, ldx, w, bias, y, ldy, M, N, K, epi, stream, workspace, workspace_bytes);
}
