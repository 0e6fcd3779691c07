// Channels-last causal 3-D convolution for the Wan VAE decoder, gfx950 only.
//
// Replaces `CausalConv3d.forward` (inferix/models/wan_base/vae.py:26-34) + the feature-cache concatenation around it
// (vae.py:207-216), `Upsample` + `Conv2d` of `Resample` (vae.py:58-64, 83-90, 139-141) and the residual add of
// `ResidualBlock.forward` (vae.py:219).
//
// Design (LDS-tiled direct convolution on the matrix cores, not an im2col GEMM):
//   * activations are [frame][h][w][C] bf16; a workgroup (8 waves) owns an 8 x 64 pixel tile of ONE output frame and BN
//     output channels; wave w computes image row w of the tile (two 32-pixel MFMA blocks) for all BN channels.
//   * the K loop runs over stages (32-channel chunk cc outer, input frame dt inner: round 6, see conv_pp_kernel).  Per stage the (8+2) x (64+2) pixel halo patch of
//     that frame / chunk is DMA'd into LDS once (`global_load_lds`, 64 B per pixel, zero page for padding pixels and for
//     the all-zero frames in front of the stream) and ALL nine spatial taps read their A fragments from it at a per-tap
//     offset — 3.9 patch loads per output pixel-chunk instead of the 27 an implicit GEMM gathers.  The nearest-2x
//     upsample in front of the `Resample` conv2d is folded into the fragment address (source pixel = (o + d - 1) >> 1),
//     so the 4x larger upsampled tensor never exists.
//   * weights, packed [tap][Cin/32][Cout][32] so that a 1 KiB DMA piece (16 rows x 64 B) is CONTIGUOUS (LDS-DMA moves 64-byte rows
//     128 bytes apart at half rate, tools/probe_dma.hip), stream through a ring of groups (one kernel row = 3 taps x 32 channels x BN rows), one or
//     two groups ahead; ONE barrier per group, and inside a group the fragments of tap i+1 are read under the MFMAs of
//     tap i (a barrier per tap left the matrix pipe idle for the LDS latency of every tap: 845 -> see DESIGN.md).
//   * both DMA streams share the wave's vmcnt; the wait before group g allows exactly the instructions issued after
//     weights(g): the weight pieces of the groups issued since + the next stage's patch when it went out last group.
//   * epilogue: bias, bf16 rounding, optional residual add (second rounding, as `x + h` in bf16 upstream), stores into
//     caller-chosen frame slots (ring buffers of the next conv, or the even/odd frames of the temporal upsampler).
#include <stdlib.h>

#include "ifx_common.h"

namespace ifx {
namespace conv {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int TH = 8, TW = 64;          // output tile (pixels)
constexpr int MAXF = 16;                // frames per call (inputs incl. history / outputs)

struct ConvArgs {
  const unsigned short* x;
  const unsigned short* w;
  const unsigned short* bias;
  const unsigned short* res;
  const unsigned short* zero;
  unsigned short* y;
  long long in_frame_stride, out_frame_stride;
  int in_slot[MAXF], out_slot[MAXF];
  int Hs, Ws, Cin, Ho, Wo, Cout, KT, t_out;
  int tiles_w, tiles_h, tiles_n, per_xcd, total;
  unsigned long long* trace;      // lab builds (-DIFX_CONVPP_TRACE=1): segment cycle sums of workgroup 0's waves 0 and 4
  int in_planar;   // input frames [Cin/32][Hs][Ws][32] instead of [Hs][Ws][Cin]
  int ablate;      // IFX_CONV_ABLATE bit mask (timing experiments only): 1 no DMA, 2 no fragment reads, 4 no MFMA, 8 no stores
};

__device__ __forceinline__ void wait_vm(int n) {
#define IFX_WV(K) case K: asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory"); break;
  switch (n) {
    IFX_WV(0) IFX_WV(1) IFX_WV(2) IFX_WV(3) IFX_WV(4) IFX_WV(5) IFX_WV(6) IFX_WV(7) IFX_WV(8) IFX_WV(9) IFX_WV(10) IFX_WV(11)
    IFX_WV(12) IFX_WV(13) IFX_WV(14) IFX_WV(15) IFX_WV(16) IFX_WV(17) IFX_WV(18) IFX_WV(19) IFX_WV(20) IFX_WV(21) IFX_WV(22)
    IFX_WV(23) IFX_WV(24)
    default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
  }
#undef IFX_WV
}

#ifndef IFX_CONV_LOADER_WAVES
#define IFX_CONV_LOADER_WAVES 4
#endif

// Timing experiments (tools/ablate_conv.sh): build with -DIFX_CONV_ABLATE_RT=1 to honour IFX_CONV_ABLATE at run time;
// in normal builds the mask is a compile-time zero and the branches vanish (they cost registers in the main loop).
#ifdef IFX_CONV_ABLATE_RT
#define ABL(A) ((A).ablate)
#else
#define ABL(A) 0
#endif

template <int BN, int UPS, int KS>
struct Geo {
  static constexpr int TAPS = KS * KS;
  static constexpr int TG = TAPS == 9 ? 3 : 1;          // taps per barrier interval ("group" = one kernel row)
  static constexpr int GPS = TAPS / TG;                 // groups per stage
  static constexpr int PH = UPS ? TH / 2 + 2 : TH + KS - 1;
  static constexpr int PW = UPS ? TW / 2 + 2 : TW + KS - 1;
  static constexpr int NP = PH * PW;                    // patch pixels
  static constexpr int NPI = (NP + 15) / 16;            // 1 KiB DMA instructions per patch
  // LDS-DMA costs its issuing wave ~65 cycles per instruction and is serialised per SIMD, but does not hold back the other
  // wave of the SIMD (tools/probe_overlap.hip): only waves 0 .. LW-1 (the older wave of each SIMD) issue DMA, the younger
  // ones go straight to the matrix pipe after the barrier.
  static constexpr int LW = IFX_CONV_LOADER_WAVES;
  static constexpr int PPW = (NPI + LW - 1) / LW;       // patch pieces per loader wave
  static constexpr int P_SLOT = NPI * 1024;
  static constexpr int WP = BN / 16;                    // 1 KiB weight pieces per tap (16 rows x 32 channels)
  static constexpr int W_TAP = BN * 64;
  static constexpr int WPW = (TG * WP + LW - 1) / LW;    // weight pieces per loader wave per group
  static constexpr int WRG = TAPS == 9 ? (BN == 128 ? 2 : 3) : 2;   // weight ring depth in groups
  static constexpr int L = WRG - 1;                     // groups of weight lookahead
  static constexpr int P_OFF = 0, W_OFF = 2 * P_SLOT, S_OFF = W_OFF + WRG * TG * W_TAP, LDS_MAIN = S_OFF + LW * 1024;
  static constexpr int LDS_EPI = 8 * 64 * (BN * 2 + 16);      // per-wave transpose regions of the epilogue (reuse the rings)
  static constexpr int LDS = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
  static constexpr bool PF = BN <= 96;                  // fragments of the next tap prefetched under the MFMAs of this one
  static_assert(L == 1 || GPS >= 3, "patch of the next stage is issued in group 0 and must precede weights two groups on");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int BN, int UPS, int KS>
__global__ __launch_bounds__(512) void conv_cl_kernel(ConvArgs A) {
  using G = Geo<BN, UPS, KS>;
  constexpr int TAPS = G::TAPS, TG = G::TG, GPS = G::GPS, PW = G::PW, PPW = G::PPW, WPW = G::WPW, WRG = G::WRG, L = G::L, LW = G::LW;
  constexpr int TI = BN / 32, TJ = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
  const int t_id = xcd * A.per_xcd + slot_i;
  if (slot_i >= A.per_xcd || t_id >= A.total) return;
  // work order: channel tile fastest (the workgroups that read the same halo patch sit on one XCD's L2)
  int rem = t_id;
  const int nt = rem % A.tiles_n;
  rem /= A.tiles_n;
  const int tw = rem % A.tiles_w;
  rem /= A.tiles_w;
  const int th = rem % A.tiles_h;
  const int to = rem / A.tiles_h;
  const int h0 = th * TH, w0 = tw * TW, n_base = nt * BN;

  const bool loader = wave < LW;
  // ---- patch DMA sources: piece p = wave + LW r covers patch pixels [16 p, 16 p + 16) x 4 chunks of 16 B
  const int ph0 = UPS ? h0 / 2 - 1 : h0 - KS / 2, pw0 = UPS ? w0 / 2 - 1 : w0 - KS / 2;
  int poff[PPW];
#pragma unroll
  for (int r = 0; r < PPW; ++r) {
    const int p = wave + LW * r;
    const int px = p * 16 + (lane >> 2);
    const int pr = px / PW, pc = px - pr * PW;
    const int sh = ph0 + pr, sw = pw0 + pc;
    const bool ok = p < G::NPI && px < G::NP && sh >= 0 && sh < A.Hs && sw >= 0 && sw < A.Ws;
    poff[r] = ok ? (sh * A.Ws + sw) * (A.in_planar ? 32 : A.Cin) + (((lane & 3) ^ ((px >> 2) & 3)) << 3) : -1;
  }
  const int CC = A.Cin >> 5;
  const int S = A.KT * CC, totalG = S * GPS;
  auto issue_patch = [&](int s, int r) {            // piece r of this wave for stage s
    const int cc = s / A.KT, dt = s - cc * A.KT;      // stage order: channel chunk outer, input frame inner (round 6, see conv_pp_kernel)
    const int f = A.in_slot[to + dt];
    const unsigned short* base = A.x + (long long)f * A.in_frame_stride + (A.in_planar ? (long long)cc * A.Hs * A.Ws * 32 : cc * 32);
    const int p = wave + LW * r;
    const unsigned short* src = (poff[r] < 0 || f < 0) ? A.zero : base + poff[r];
    unsigned char* dst = p < G::NPI ? smem + G::P_OFF + (s & 1) * G::P_SLOT + p * 1024 : smem + G::S_OFF + wave * 1024;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
  };
  // ---- weight DMA: piece idx = wave + LW q of a group = (tap idx / WP of the group, rows [16 (idx % WP), +16))
  int woff[WPW];
#pragma unroll
  for (int q = 0; q < WPW; ++q) {
    const int idx = wave + LW * q;
    const int wrow = (idx % G::WP) * 16 + (lane >> 2);
    woff[q] = min(n_base + wrow, A.Cout - 1) * 32 + (((lane & 3) ^ ((wrow >> 2) & 3)) << 3);     // [tap][cc][cout][32]: 64-byte rows, contiguous
  }
  auto issue_w = [&](int g) {                       // all weight pieces of this wave for group g
    const int s = g / GPS, gi = g - s * GPS;
    const int cc = s / A.KT, dt = s - cc * A.KT;      // stage order: channel chunk outer, input frame inner (round 6, see conv_pp_kernel)
    const unsigned short* wbase = A.w + ((size_t)(dt * TAPS + gi * TG) * CC + cc) * A.Cout * 32;
    unsigned char* ring = smem + G::W_OFF + (g % WRG) * (TG * G::W_TAP);
#pragma unroll
    for (int q = 0; q < WPW; ++q) {
      const int idx = wave + LW * q;
      const int tg = idx / G::WP;                   // tap within the group
      const bool real = idx < TG * G::WP;
      const unsigned short* src = wbase + (size_t)(real ? min(tg, TG - 1) : 0) * CC * A.Cout * 32 + woff[q];
      unsigned char* dst = real ? ring + tg * G::W_TAP + (idx % G::WP) * 1024 : smem + G::S_OFF + wave * 1024;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
    }
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: patch of stage 0, weights of groups 0 .. L-1
  if (loader) {
#pragma unroll
    for (int r = 0; r < PPW; ++r) issue_patch(0, r);
#pragma unroll
    for (int g = 0; g < L; ++g)
      if (g < totalG) issue_w(g);
  }

  int b_off[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int n = i * 32 + l31;
    b_off[i] = n * 64 + (((n >> 2) & 3) << 4);        // row base with the swizzle phase folded in as an XOR operand below
  }

  constexpr int NB = G::PF ? 2 : 1;
  bf16x8 fa[NB][2][TJ], fb[NB][2][TI];
  for (int s = 0; s < S; ++s) {
    const bool has_next = s + 1 < S;
    const unsigned char* pb = smem + G::P_OFF + (s & 1) * G::P_SLOT;
#pragma unroll
    for (int gi = 0; gi < GPS; ++gi) {
      const int g = s * GPS + gi;
      // in flight after weights(g): the weight groups issued since + the next stage's patch when it was issued last group
      wait_vm(WPW * min(L - 1, totalG - 1 - g) + ((L == 2 && gi == 1 && has_next) ? PPW : 0));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own fragment reads of group g-1 retired
      __builtin_amdgcn_s_barrier();
      if (loader && gi == 0 && has_next && !(ABL(A) & 1)) {
#pragma unroll
        for (int r = 0; r < PPW; ++r) issue_patch(s + 1, r);
      }
      if (loader && g + L < totalG && !(ABL(A) & 1)) issue_w(g + L);

      const unsigned char* wg = smem + G::W_OFF + (g % WRG) * (TG * G::W_TAP);
      auto load = [&](int buf, int tg) {
        if (ABL(A) & 2) return;
        const int tap = gi * TG + tg;
        const int dh = tap / KS, dw = tap - dh * KS;
        const unsigned char* wb = wg + tg * G::W_TAP;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          int prow;
          if (UPS) prow = ((wave + dh + 1) >> 1) * PW + ((j * 32 + l31 + dw + 1) >> 1);
          else prow = (wave + dh) * PW + (j * 32 + l31 + dw);
          const int sw = (prow >> 2) & 3;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            fa[buf][ks][j] = *reinterpret_cast<const bf16x8*>(pb + prow * 64 + (((2 * ks + hi) ^ sw) << 4));
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            fb[buf][ks][i] = *reinterpret_cast<const bf16x8*>(wb + (b_off[i] ^ ((2 * ks + hi) << 4)));
      };
      auto mma = [&](int buf) {
        if (ABL(A) & 4) return;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[buf][ks][i], fa[buf][ks][j], acc[i][j], 0, 0, 0);
      };
      if (G::PF) {
        load(0, 0);
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) {
          if (tg + 1 < TG) load((tg + 1) & 1, tg + 1);
          mma(tg & 1);
        }
      } else {
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) {
          load(0, tg);
          mma(0);
        }
      }
    }
  }

  // ---- epilogue.  Lane (l31, hi) holds, for pixel l31 of block j, channels i*32 + g*8 + hi*4 + e: stored directly that is
  //      8 bytes per lane scattered over 32 pixel rows (measured: a quarter of the 480p launch).  Instead every wave
  //      transposes its 64 pixels x BN channels of bf16(acc + bias) through its own LDS region (row pitch BN*2 + 16 B:
  //      16-byte aligned, bank-spread) and moves whole pixel rows: consecutive lanes = consecutive 16-byte chunks, 1 KiB
  //      contiguous per instruction; the residual comes in the same way.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                        // everyone is done with the patch / weight rings
  if (ABL(A) & 8) return;
  constexpr int PITCH = BN * 2 + 16, CR = BN / 8;
  unsigned char* tr = smem + wave * (64 * PITCH);
  const bool vec_bias = A.bias != nullptr && (A.Cout & 3) == 0;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nl = i * 32 + g * 8 + hi * 4;
      const int n = n_base + nl;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (vec_bias) {
        const u16x4 b4 = *reinterpret_cast<const u16x4*>(A.bias + min(n, A.Cout - 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = bf2f(b4[e]);
      } else if (A.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = bf2f(A.bias[min(n + e, A.Cout - 1)]);
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[i][j][4 * g + e] + bv[e]);
        *reinterpret_cast<u16x4*>(tr + (j * 32 + l31) * PITCH + nl * 2) = o;
      }
      __builtin_amdgcn_sched_barrier(0);               // keep the bias loads from piling up in registers
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own region only: no barrier needed
  const int oh = h0 + wave;
  if (oh >= A.Ho) return;
  unsigned short* yf = A.y + (long long)A.out_slot[to] * A.out_frame_stride;
  const unsigned short* rf = A.res ? A.res + (long long)to * A.Ho * A.Wo * A.Cout : nullptr;
#pragma unroll
  for (int it = 0; it < CR; ++it) {
    const int idx = it * 64 + lane;
    const int p = idx / CR, c = idx - p * CR;
    const int ow = w0 + p, n = n_base + c * 8;
    if (ow >= A.Wo || n >= A.Cout) continue;
    u16x8 v = *reinterpret_cast<const u16x8*>(tr + p * PITCH + c * 16);
    const size_t off = ((size_t)oh * A.Wo + ow) * A.Cout + n;
    if (n + 7 < A.Cout && (A.Cout & 7) == 0) {
      if (rf) {
        const u16x8 rv = *reinterpret_cast<const u16x8*>(rf + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) + bf2f(rv[e]));
      }
      *reinterpret_cast<u16x8*>(yf + off) = v;
    } else {
      for (int e = 0; e < 8 && n + e < A.Cout; ++e) {
        unsigned short o = v[e];
        if (rf) o = f2bf(bf2f(o) + bf2f(rf[off + e]));
        yf[off + e] = o;
      }
    }
  }
}

template <int BN, int UPS, int KS>
static void launch(const ConvArgs& a, hipStream_t s) {
  using G = Geo<BN, UPS, KS>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_cl_kernel<BN, UPS, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    attr = true;
  }
  hipLaunchKernelGGL((conv_cl_kernel<BN, UPS, KS>), dim3(a.per_xcd * 8), dim3(512), G::LDS, s, a);
}

// =====================================================================================================================
// Round 6: the same convolution as a PERSISTENT PING-PONG kernel (`conv_pp_kernel`), the structure of ifx_gemm_pp.hip.
//
// Why (tools/ablate_conv.sh, round 6, 96 -> 96 @ 480 x 832 x 12 frames): with the kernel's parts switched off one at a time the
// lock-step loop above loses 25 % without MFMAs, 25 % without fragment reads, 27 % without DMA and 13 % without the epilogue — the
// four serialise instead of overlapping (both waves of a SIMD issue DMA, then both read fragments, then both want the matrix pipe;
// the epilogue of a tile runs with the matrix pipe idle because the rings double as its transpose scratch: one workgroup per CU).
// Here, per SIMD, the two waves alternate roles every phase (one barrier per phase):
//
//   phase 2g   : group 0 (waves 0-3, tile rows 0-3)  36 MFMAs of step g from registers | group 1 (rows 4-7)  patch DMA, fragments of step g
//   phase 2g+1 : group 0  weight DMA, fragment reads of step g+1                     | group 1  36 MFMAs of step g
//
//   * a step = one kernel row (3 taps) of one stage (input frame dt, 32-channel chunk cc): 12 patch + 18 weight ds_read_b128 = 120
//     fragment registers, ALL read in the wave's loader phase; the MFMA phase holds nothing but MFMAs.  (A first version issued the
//     weight DMA at the head of group 1's MFMA phase: an LDS-DMA instruction blocks its wave until the CU's vector-memory path takes
//     it — 200-380 cycles per 1 KiB piece in this loop, s_memtime trace in DESIGN.md — and that wave's MFMAs wait behind it.)
//   * LDS: two patch slots (the (8+2) x (64+2) halo of a stage, as above), a weight ring of TWO steps, and four 32-pixel transpose
//     regions for the epilogue that are NOT part of the rings — so the request stream runs across output tiles: the first stages of
//     tile i+1 are in flight while tile i's last steps are multiplied, and a group's epilogue (bias, bf16, + residual, whole-pixel
//     stores) runs in its loader phase under the other group's MFMAs.
//   * group 0 requests the weights of step g+2 in its loader phase 2g+1 (the slot's last reader was group 1 in phase 2g) and waits
//     vmcnt(0) at the end of its MFMA phase 2g+2: visible from phase 2g+3, where group 0 itself reads them.  Group 1 requests the
//     patch of stage t+1 in its loader phases 6t and 6t+2 (half of its pieces each; the slot's last reader was group 1 itself in phase
//     6t-2) and waits at the head of its loader phase 6t+4: visible from phase 6t+5, where group 0 reads the first fragments of
//     stage t+1.  Group 0 moves 12 of a stage's 42 patch pieces as well (one per wave in each of its loader phases 6t-1, 6t+1, 6t+3,
//     covered by the same end-of-MFMA-phase wait): its weight pieces cost ~90 cycles of issue each, a patch piece ~200-300, and the
//     two loader phases are what the step time is made of.  Every request has at least one whole MFMA phase (>= 1152 matrix-pipe
//     cycles) to land, the patch (HBM) three.
//   * epilogue: bias from an LDS copy, bf16(acc + bias) transposed through the wave's LDS region 32 pixels at a time, residual rows
//     (their cache lines touched three steps ahead by two 4-byte-per-lane DMA requests into a sink) and output rows moved by buffer
//     instructions whose offsets are TWO lane terms + wave-uniform bases (no per-access address arithmetic: 998 -> 604 VALU
//     instructions, the largest single gain after the schedule).
//   * DMA by inline asm through buffer descriptors (hipcc fences every later ds_read behind a `global_load_lds` builtin with
//     vmcnt(0)); padding pixels and the zero frames in front of the stream are out-of-range offsets / empty descriptors, which
//     return zeros: no zero page.  A loader wave interleaves its DMA instructions with its fragment reads (the reads do not wait
//     for the vector-memory path).
//   * patch swizzle by COLUMN ((col >> 2) & 3, not by linear patch pixel): a fragment address is a wave-uniform row base + a
//     lane term that does not depend on the row — half the address arithmetic of the lock-step kernel's loader.
//   * tile order: channel tile fastest, then OUTPUT FRAME, then the spatial tile: the workgroups resident on an XCD work on
//     consecutive ids = the same spatial tile of consecutive frames, whose three input frames overlap two by two — and the stage order
//     is channel chunk OUTER, input frame INNER (both kernels, round 6): output frame `to` reads chunk cc of frame f = to + dt at stage
//     3 cc + dt, its neighbours to - 1 / to - 2 read the same chunk one / two stages later, while it is still in that XCD's L2
//     (FETCH_SIZE 1.212e6 -> 0.944e6 KiB per launch: fabric reads 2.3x -> 1.77x algorithmic, profiles/r6_pmc_conv.md).
//   * arithmetic, K order (cc, dt, dh, dw, 16-channel k-step), epilogue rounding: those of conv_cl_kernel — bit-identical outputs
//     (tools/bench_conv.py compares the two kernels bit for bit on every shape it times).
namespace pp {
typedef int v4i __attribute__((ext_vector_type(4)));

#ifndef IFX_CONVPP_TRACE
#define IFX_CONVPP_TRACE 0      // 1: s_memtime segment sums of workgroup 0 (printed by the launcher after a synchronisation; lab only)
#endif
// Further lab switches for the trace build (`make convtrace CXXFLAGS+=-DIFX_CONVPP_NOFRAG ...`; results are garbage, only the segment times
// matter): IFX_CONVPP_NOFRAG no fragment reads, IFX_CONVPP_NOMFMA no MFMAs behind a tile's first step, IFX_CONVPP_NORES no residual rows,
// IFX_CONVPP_NOSTORE no output stores — how DESIGN 13's "what bounds the kernel" numbers were taken.
#if IFX_CONVPP_TRACE
#define CP_STAMP(i)                                                \
  do {                                                             \
    const unsigned long long t_now = __builtin_readcyclecounter(); \
    seg[i] += t_now - t_last;                                      \
    t_last = t_now;                                                \
  } while (0)
#else
#define CP_STAMP(i) \
  do {              \
  } while (0)
#endif

__device__ __forceinline__ void dma16(v4i rsrc, unsigned lds, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}
__device__ __forceinline__ void dma4(v4i rsrc, unsigned lds, int voff) {      // 4 bytes per lane (a cache-line touch: the data is not used)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" : : "s"(lds), "v"(voff), "s"(rsrc) : "memory");
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
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_lds() { __builtin_amdgcn_s_waitcnt(0xC07F); }       // lgkmcnt(0), known to hipcc's scoreboard

template <int BN, int UPS>
struct GeoP {
  static constexpr int PH = UPS ? TH / 2 + 2 : TH + 2, PW = UPS ? TW / 2 + 2 : TW + 2;
  static constexpr int NP = PH * PW, NPI = (NP + 15) / 16;
  // a stage's NPI patch pieces: group 0 moves the first 4 * PG0 (one per wave in each of its loader phases 6t-1, 6t+1, 6t+3: its loader
  // phases carry the 4-5 weight pieces per wave, ~90 cycles each, and a patch piece costs ~300 — 64-byte rows 192+ bytes apart), group 1
  // the rest in two batches (loader phases 6t, 6t+2)
  static constexpr int PG0 = UPS ? 1 : 3;               // (1 or 2 for the plain kernel: group 1's lane terms no longer fit, 1-3 registers spill)
  static constexpr int NP1 = NPI - 4 * PG0;
  static constexpr int PPW = (NP1 + 3) / 4;             // patch piece slots per wave of group 1 and stage
  static constexpr int QA = (PPW + 1) / 2;              // of which in the first of the stage's two batches
  static constexpr int P_SLOT = NPI * 1024;
  static constexpr int WP = BN / 16;                    // 1 KiB weight pieces per tap
  static constexpr int W_TAP = BN * 64, W_STEP = 3 * W_TAP;
  static constexpr int WPS = 3 * WP, WPW = (WPS + 3) / 4;      // weight pieces per step / piece slots per wave of group 0 and step
  static constexpr int PITCH = BN * 2 + 16, SCR = 32 * PITCH;  // transpose region of one wave: 32 pixels x BN channels
  static constexpr int P_OFF = 0, W_OFF = 2 * P_SLOT, E_OFF = W_OFF + 2 * W_STEP;
  static constexpr int B_OFF = E_OFF + 4 * SCR;         // the launch's bias vector (Cout <= 512 channels) as bf16
  static constexpr int S_OFF = B_OFF + 1024;            // 256-byte sinks of the residual warm-up requests, one per wave
  static constexpr int LDS = S_OFF + 8 * 256;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(SCR % 16 == 0 && P_SLOT % 16 == 0 && W_STEP % 16 == 0, "alignment");
};

struct Tile {
  int to, h0, w0, n_base;
};

template <int BN, int UPS>
__global__ __launch_bounds__(512) void conv_pp_kernel(ConvArgs A) {
  using G = GeoP<BN, UPS>;
  constexpr int TI = BN / 32, TJ = 2, PW = G::PW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;

  // ---- this workgroup's tiles: XCD (bid & 7) owns ids [xcd * per_xcd, ...), its workgroups take them round-robin
  const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3, wgx = (int)gridDim.x >> 3;
  const int id_first = xcd * A.per_xcd + slot_i;
  const int id_end = min(xcd * A.per_xcd + A.per_xcd, A.total);
  if (id_first >= id_end) return;
  const int n_my = (id_end - id_first + wgx - 1) / wgx;
  const int CC = A.Cin >> 5, S = A.KT * CC, U = 3 * S, GT = n_my * U;
  const unsigned lds0 = (unsigned)(unsigned long long)(lds_ptr_t)smem;
  auto tile_of = [&](int i) __attribute__((always_inline)) {
    int rem = id_first + i * wgx;
    Tile t;
    const int nt = rem % A.tiles_n;
    rem /= A.tiles_n;
    t.to = rem % A.t_out;
    rem /= A.t_out;
    const int tw = rem % A.tiles_w;
    t.h0 = (rem / A.tiles_w) * TH;
    t.w0 = tw * TW;
    t.n_base = nt * BN;
    return t;
  };

  // ---- reader side.  (Lane-derived addressing is re-derived from an opaque lane id at every call: hipcc otherwise hoists the
  //      per-tap pixel columns and the weight-row offsets out of the step loop — ~20 registers held across a loop that runs at 96
  //      accumulators + 120 fragment registers — and spills fragments into the MFMA phase.)
  bf16x8 fa[3][2][TJ], fb[3][2][TI];
  // The column swizzle makes a fragment address = wave-uniform base (slot, patch row) + a lane term that depends on (tap, block, k-step)
  // only: the 12 + 6 lane terms live in registers across the loop (the loop then runs at ~240 of 256) and a loader phase spends 18
  // v_add instead of ~100 VALU operations on addresses (s_memtime trace: a loader phase was ~800 cycles of reads + address arithmetic
  // against the 1152 of the MFMA phase it must hide under).
  // (k-step 1 of a fragment is k-step 0's address with bit 5 flipped — the chunk index is (2 ks + hi) ^ swizzle — so only the
  //  k-step-0 terms are kept: 9 registers, and one v_xor per k-step-1 read)
  int a_term[3][TJ], b_term[TI];
  {
    const int f31 = (int)(threadIdx.x & 31), fhi = (int)((threadIdx.x >> 5) & 1);
#pragma unroll
    for (int tg = 0; tg < 3; ++tg)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = UPS ? (j * 32 + f31 + tg + 1) >> 1 : j * 32 + f31 + tg;
        a_term[tg][j] = col * 64 + ((fhi ^ ((col >> 2) & 3)) << 4);
      }
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int n = i * 32 + f31;
      b_term[i] = (n * 64 + (((n >> 2) & 3) << 4)) ^ (fhi << 4);
    }
  }
  // fragments of kernel row r of the stage in patch slot ps, weights in ring slot ws; `between(tg)` runs BEHIND tap tg's reads (the
  // loader's DMA instructions go there: the reads just issued are served by the LDS while the wave sits in a DMA issue stall, and the
  // other group's requests of the phase before have drained from the CU's vector-memory path by then — the first request of a phase
  // waited ~480 cycles for them when it was issued straight behind the barrier)
  auto read_frags = [&](int r, int ps, int ws, auto between) __attribute__((always_inline)) {
    const unsigned char* pb = smem + G::P_OFF + ps * G::P_SLOT + (UPS ? (wave + r + 1) >> 1 : wave + r) * (PW * 64);
    const unsigned char* wg = smem + G::W_OFF + ws * G::W_STEP;
#pragma unroll
    for (int tg = 0; tg < 3; ++tg) {
#ifndef IFX_CONVPP_NOFRAG
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[tg][ks][j] = *reinterpret_cast<const bf16x8*>(pb + (a_term[tg][j] ^ (ks << 5)));
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fb[tg][ks][i] = *reinterpret_cast<const bf16x8*>(wg + tg * G::W_TAP + (b_term[i] ^ (ks << 5)));
#endif
      between(tg);
    }
  };
  f32x16 acc[TI][TJ];
  auto mfma_first = [&]() __attribute__((always_inline)) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][0][i], fa[0][0][j], z, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][1][i], fa[0][1][j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int tg = 1; tg < 3; ++tg)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[tg][ks][i], fa[tg][ks][j], acc[i][j], 0, 0, 0);
  };
  auto mfma_next = [&]() __attribute__((always_inline)) {
#ifdef IFX_CONVPP_NOMFMA
    return;                                           // lab
#endif
#pragma unroll
    for (int tg = 0; tg < 3; ++tg)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[tg][ks][i], fa[tg][ks][j], acc[i][j], 0, 0, 0);
  };

  // ---- epilogue of tile `t` for this wave (image row `wave` of the tile): 32-pixel blocks one at a time through the wave's transpose
  //      region (shared by wave w4 of both groups: their epilogues run in different phases), whole pixel rows out, as above.  The
  //      residual rows of BOTH blocks are requested first (one memory round trip per epilogue; the fragment registers are free here).
  auto epilogue = [&](const Tile& t) __attribute__((always_inline)) {
    int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(ln));                      // lane-derived addressing re-derived here: it would stay live across the K loop
    const int e31 = ln & 31, ehi = ln >> 5;
    constexpr int PITCH = G::PITCH, CG = BN / 32, NIT = 2 * CG;      // an iteration moves 16 pixels x one 32-channel group (4 chunks of 16 B)
    unsigned char* tr = smem + G::E_OFF + w4 * G::SCR;
    const int oh = t.h0 + wave;
    if (oh >= A.Ho) return;                            // (wave-uniform: an image row below the frame)
    // Whole-pixel-row accesses through buffer descriptors: a lane's byte offset = a term that depends on the lane alone (pixel of the
    // 32-pixel block and 16-byte chunk it moves: computed once per epilogue) + a wave-uniform base per block — no per-access address
    // arithmetic (the first version spent ~300 of the epilogue's ~1000 VALU instructions on 64-bit addresses, divisions by the chunk
    // count and bounds).  Pixels right of the frame (last tile column only) get an out-of-range offset: loads return 0, stores are dropped.
    const int frame_out_bytes = A.Ho * A.Wo * A.Cout * 2;
    const __amdgpu_buffer_rsrc_t rs_y =
        __builtin_amdgcn_make_buffer_rsrc((void*)(A.y + (long long)A.out_slot[t.to] * A.out_frame_stride), 0, frame_out_bytes, 0x00020000);
    const bool has_res = A.res != nullptr;
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(has_res ? A.res + (long long)t.to * A.Ho * A.Wo * A.Cout : A.y), 0, has_res ? frame_out_bytes : 0, 0x00020000);
    // lane -> (pixel ln >> 2 of a 16-pixel half block, chunk ln & 3 of a 32-channel group): iteration it = (half hb, group cg) adds
    // compile-time constants to two lane terms — 64-byte runs per pixel and instruction, the three groups of a pixel back to back
    const int pl = ln >> 2;
    const int l_lane = pl * PITCH + (ln & 3) * 16, g_lane = (pl * A.Cout + (ln & 3) * 8) * 2;
    const int cols_left = A.Wo - t.w0;                 // pixels of the tile's 64 columns that lie inside the frame
    u32x4 rv[TJ][NIT];
    if (has_res) {
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int sbase = ((oh * A.Wo + t.w0 + 32 * j) * A.Cout + t.n_base) * 2;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int hb = it / CG, cg = it - hb * CG;
          rv[j][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, pl < cols_left - 32 * j - 16 * hb ? g_lane : (int)0x80000000,
                                                            sbase + (16 * hb * A.Cout + cg * 32) * 2, 0);
        }
      }
    }
    u32x2 e_bias[TI][4];                               // from the copy the workgroup made in LDS (zeros without a bias): no memory round trip
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        e_bias[i][g] = *reinterpret_cast<const u32x2*>(smem + G::B_OFF + (t.n_base + i * 32 + g * 8 + ehi * 4) * 2);
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          const u32x2 b = e_bias[i][g];
          v[0] += __builtin_bit_cast(float, b[0] << 16);
          v[1] += __builtin_bit_cast(float, b[0] & 0xffff0000u);
          v[2] += __builtin_bit_cast(float, b[1] << 16);
          v[3] += __builtin_bit_cast(float, b[1] & 0xffff0000u);
          u16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
          *reinterpret_cast<u16x4*>(tr + e31 * PITCH + (i * 32 + g * 8 + ehi * 4) * 2) = o;
        }
      wait_lds();                                     // wave-private region: no barrier
      const int sbase = ((oh * A.Wo + t.w0 + 32 * j) * A.Cout + t.n_base) * 2;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int hb = it / CG, cg = it - hb * CG;
        u16x8 v = *reinterpret_cast<const u16x8*>(tr + l_lane + 16 * hb * PITCH + cg * 64);
        if (has_res) {
          const u16x8 r8 = __builtin_bit_cast(u16x8, rv[j][it]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) + bf2f(r8[e]));
        }
#ifndef IFX_CONVPP_NOSTORE
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_y, pl < cols_left - 32 * j - 16 * hb ? g_lane : (int)0x80000000,
                                               sbase + (16 * hb * A.Cout + cg * 32) * 2, 0);
#endif
      }
      // (the LDS operations of one wave execute in order: the next block's writes cannot overtake these reads)
    }
  };

  const unsigned frame_bytes = (unsigned)((long long)A.Hs * A.Ws * A.Cin * 2);
  // compute-side cursor: step c_g of the stream = kernel row c_r of stage c_s of tile c_it; c_sg = global stage index
  int c_it = 0, c_s = 0, c_r = 0, c_sg = 0;
  Tile c_t = tile_of(0);
  auto c_first = [&]() __attribute__((always_inline)) { return c_s == 0 && c_r == 0; };
  auto c_last = [&]() __attribute__((always_inline)) { return c_s == S - 1 && c_r == 2; };
  auto c_next = [&]() __attribute__((always_inline)) {
    if (++c_r == 3) {
      c_r = 0;
      ++c_sg;
      if (++c_s == S) c_s = 0, ++c_it;
    }
  };
  // ---- patch requests: every wave runs its own cursor over the stages of the stream (group 0 and group 1 advance it at different
  //      phases); piece p = 16 patch pixels x 64 B
  int p_it = 0, p_dt = 0, p_cc = 0, p_sg = 0;
  Tile p_t = c_t;
  v4i p_rs;
  auto p_desc = [&]() __attribute__((always_inline)) {             // planar frames: one descriptor per 32-channel plane
    const int f = A.in_slot[p_t.to + p_dt];
    const long long plane = (long long)A.Hs * A.Ws * 32;
    p_rs = make_rsrc(A.x + (long long)max(f, 0) * A.in_frame_stride + (A.in_planar ? p_cc * plane : 0),
                     f < 0 ? 0u : (A.in_planar ? (unsigned)(plane * 2) : frame_bytes));
  };
  // A piece's source offsets split into a TILE-INVARIANT lane term — (patch row * Ws + patch column) pixels + the swizzled 16-byte chunk,
  // computed once per kernel and kept in a register per piece slot — and a wave-uniform term (tile origin, channel chunk) added per
  // request.  Rows above / below the image fall outside the descriptor (negative or past-the-end offsets: zeros); the halo COLUMNS of
  // the first / last tile column do not, so the lane term carries two flags in its free low bits — bit 0: patch column 0, bit 1: a
  // column at or beyond the right image edge as seen from the LAST tile column — that the request masks with the tile's position.
  // (The first version derived row, column, bounds and address per request: ~25 dependent VALU operations, 200-500 cycles per piece in
  //  the s_memtime trace against ~90 for a weight piece whose address is scalar.)
  const int pix_bytes = A.in_planar ? 64 : A.Cin * 2;
  const int w0_last = (A.tiles_w - 1) * TW;
  const int col_limit_last = A.Ws + 1 - (UPS ? w0_last / 2 : w0_last);         // first invalid patch column of the last tile column
  auto p_lane_term = [&](int p) __attribute__((always_inline)) {
    const int ln = (int)(threadIdx.x & 63);
    const int px = p * 16 + (ln >> 2);
    const int pr = px / PW, pc = px - pr * PW;
    if (p >= G::NPI || px >= G::NP) return (int)0x80000000;
    return (pr * A.Ws + pc) * pix_bytes + (((ln & 3) ^ ((pc >> 2) & 3)) << 4) + (pc == 0 ? 1 : 0) + (pc >= col_limit_last ? 2 : 0);
  };
  int p_tile = 0, p_edge = 0;                          // wave-uniform term of the cursor's tile / its edge mask (bit 0: first, bit 1: last tile column)
  auto p_tile_terms = [&]() __attribute__((always_inline)) {
    const int ph0 = UPS ? p_t.h0 / 2 - 1 : p_t.h0 - 1, pw0 = UPS ? p_t.w0 / 2 - 1 : p_t.w0 - 1;
    p_tile = (ph0 * A.Ws + pw0) * pix_bytes;
    p_edge = (p_t.w0 == 0 ? 1 : 0) | (p_t.w0 == w0_last ? 2 : 0);
  };
  auto p_piece = [&](int p, int lane_term) __attribute__((always_inline)) {
    if (p_it >= n_my || p >= G::NPI) return;
    const int voff = (lane_term & p_edge) ? (int)0x80000000 : (lane_term & ~3) + p_tile + (A.in_planar ? 0 : p_cc * 64);
    dma16(p_rs, lds0 + G::P_OFF + (p_sg & 1) * G::P_SLOT + p * 1024, voff, 0);
  };
  auto p_next = [&]() __attribute__((always_inline)) {
    ++p_sg;
    if (++p_dt == A.KT) {                              // stage order: channel chunk outer, input frame inner
      p_dt = 0;
      if (++p_cc == CC) {
        p_cc = 0;
        if (++p_it < n_my) p_t = tile_of(p_it), p_tile_terms();
      }
    }
    if (p_it < n_my) p_desc();
  };
  p_desc();
  p_tile_terms();
  // ---- residual warm-up: three steps before a tile's epilogue every wave touches the cache lines of its residual row (64 pixels x BN
  //      channels: two lines per pixel) with two 4-byte-per-lane DMA requests into a sink — the epilogue's own loads then hit in L2
  //      instead of paying an HBM round trip inside the phase the other group sits out (trace: 4.7k of the epilogue's 12k cycles)
  auto res_warm = [&](const Tile& t) __attribute__((always_inline)) {
    if (A.res == nullptr) return;
    int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(ln));
    const v4i rs = make_rsrc(A.res + (long long)t.to * A.Ho * A.Wo * A.Cout, (unsigned)((long long)A.Ho * A.Wo * A.Cout * 2));
    const int oh = t.h0 + wave, ow = t.w0 + ln;
    const int voff = (oh < A.Ho && ow < A.Wo) ? ((oh * A.Wo + ow) * A.Cout + t.n_base) * 2 : (int)0x80000000;
    dma4(rs, lds0 + G::S_OFF + wave * 256, voff);
    dma4(rs, lds0 + G::S_OFF + wave * 256, voff + 128);
  };
  // ---- the launch's bias vector into LDS (zeros without one); visible behind barrier B0
  if (tid < 64) {
    u16x8 bv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (A.bias != nullptr && tid * 8 < A.Cout) bv = *reinterpret_cast<const u16x8*>(A.bias + tid * 8);
    *reinterpret_cast<u16x8*>(smem + G::B_OFF + tid * 16) = bv;
  }
#if IFX_CONVPP_TRACE
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_readcyclecounter();
#endif

  if (grp == 0) {
    // ---- weight requests: a cursor over the steps of the stream; wave w4 moves pieces w4, w4 + 4, ... of a step's 3 x WP
    int ln0 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int lane_w = (ln0 >> 2) * 64 + (((ln0 & 3) ^ ((ln0 >> 4) & 3)) << 4);
    const v4i w_rs = make_rsrc(A.w, (unsigned)((long long)A.KT * 9 * CC * A.Cout * 64));
    int w_it = 0, w_dt = 0, w_cc = 0, w_r = 0, w_g = 0;
    int w_nb = c_t.n_base;
    // byte offset of the cursor's step (tap 0 of its kernel row, channel row 0 of the tile) in the packed weights, refreshed ONCE per step;
    // a piece adds its tap and row-block terms.  (The five multiplications per piece this replaces sat in front of every request of the
    // loader phase, whose length bounds the kernel: -1.5 ... -3 % per launch at 192 / 384 channels.  Hoisting the patch requests' slot and
    // chunk terms the same way bought nothing: what it saves in scalar instructions it costs in scalar registers spilled to lanes.)
    const int w_tap_stride = CC * A.Cout * 64;
    int w_step_base = w_nb * 64;
    auto w_piece = [&](int q) __attribute__((always_inline)) {       // piece slot q of the cursor's step
      const int idx = w4 + 4 * q;
      if (w_it >= n_my || idx >= G::WPS) return;
      const int tg = idx / G::WP, pr = idx - tg * G::WP;
      const int base = w_step_base + tg * w_tap_stride + pr * 1024;
      dma16(w_rs, lds0 + G::W_OFF + (w_g & 1) * G::W_STEP + tg * G::W_TAP + pr * 1024, lane_w, base);
    };
    auto w_next = [&]() __attribute__((always_inline)) {
      ++w_g;
      if (++w_r == 3) {
        w_r = 0;
        if (++w_dt == A.KT) {
          w_dt = 0;
          if (++w_cc == CC) {
            w_cc = 0;
            if (++w_it < n_my) w_nb = tile_of(w_it).n_base;
          }
        }
      }
      w_step_base = ((w_dt * 9 + w_r * 3) * CC + w_cc) * (A.Cout * 64) + w_nb * 64;
    };
    auto w_all = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < G::WPW; ++q) w_piece(q);
      w_next();
    };
    // pieces of the cursor's step spread over the three taps' reads of a loader phase
    auto w_between = [&](int tg) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < G::WPW; ++q)
        if (q * 3 / G::WPW == tg) w_piece(q);
      if (tg == 2) w_next();
    };
    // this group's patch pieces: w4 + 4 k, k < PG0, of every stage; piece k in the k-th of the three loader phases in front of the stage
    int p_k = 0;
    int p_lt[G::PG0];
#pragma unroll
    for (int k = 0; k < G::PG0; ++k) p_lt[k] = p_lane_term(w4 + 4 * k);
    auto p_mine = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < G::PG0; ++k)
        if (p_k == k) p_piece(w4 + 4 * k, p_lt[k]);
      if (++p_k == 3) p_k = 0, p_next();
    };
    w_all(), w_all();                                 // steps 0 and 1
    p_mine(), p_mine(), p_mine();                     // stage 0
    wait_lds();                                       // (the bias copy)
    wait_vm0();
    __builtin_amdgcn_s_barrier();                     // B0: patch 0, weights 0 and 1, the bias are in LDS
    p_mine();                                         // phase -1: stage 1
    read_frags(0, 0, 0, [&](int) __attribute__((always_inline)) {});
    wait_lds();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    CP_STAMP(7);
    for (int g = 0; g < GT; ++g) {
      // ---------------- phase 2g: MFMA ----------------
      if (c_first()) mfma_first();
      else mfma_next();
      const bool tile_done = c_last();
      c_next();
      __builtin_amdgcn_sched_barrier(0);
      CP_STAMP(0);
      wait_vm0();                                     // the weights of step g+1 (requested one phase ago) have landed; so have an epilogue's stores
      __builtin_amdgcn_sched_barrier(0);
      CP_STAMP(1);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      CP_STAMP(2);
      // ---------------- phase 2g+1: loader ----------------
      if (tile_done) {
        epilogue(c_t);
        if (c_it < n_my) c_t = tile_of(c_it);
      }
      CP_STAMP(3);
      if (c_s == S - 1 && c_r == 1) res_warm(c_t);
      CP_STAMP(7);
      // (the reads are unconditional: behind the last step they fetch stale LDS that nobody multiplies — a conditional read would keep the
      //  OLD fragments alive through the epilogue on the not-taken path, 120 registers next to the accumulators)
      read_frags(c_r, c_sg & 1, (g + 1) & 1, [&](int tg) __attribute__((always_inline)) {      // + the weights of step g+2 into the slot step g was read from
        w_between(tg);
        if (tg == 2) p_mine();
      });
      __builtin_amdgcn_sched_barrier(0);
      CP_STAMP(4);
      wait_lds();
      __builtin_amdgcn_sched_barrier(0);
      CP_STAMP(5);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      CP_STAMP(6);
    }
#if IFX_CONVPP_TRACE
    if (blockIdx.x == 0 && wave == 0 && ln0 == 0 && A.trace != nullptr)
      for (int i = 0; i < 8; ++i) A.trace[i] = seg[i];
#endif
  } else {
    // ---- this group's patch pieces: 4 PG0 + w4 + 4 q, q < PPW, of every stage, in two batches (loader phases 6t and 6t+2)
    int p_lt[G::PPW];
#pragma unroll
    for (int q = 0; q < G::PPW; ++q) p_lt[q] = p_lane_term(4 * G::PG0 + w4 + 4 * q);
#pragma unroll
    for (int q = 0; q < G::PPW; ++q) p_piece(4 * G::PG0 + w4 + 4 * q, p_lt[q]);      // stage 0
    p_next();
    wait_lds();                                       // (the bias copy)
    wait_vm0();
    __builtin_amdgcn_s_barrier();                     // B0
    __builtin_amdgcn_s_barrier();                     // phase -1: nothing to read yet
    CP_STAMP(7);
    for (int g = 0; g < GT; ++g) {
      // ---------------- phase 2g: loader ----------------
      // head of phase 6t+4: this wave's pieces of stage t+1 (requested in phases 6t and 6t+2) have landed
      if (c_r == 2) wait_vm0();
      CP_STAMP(0);
      if (g > 0 && c_first()) {                        // (c_t still names the tile that ended one phase ago)
        epilogue(c_t);
        c_t = tile_of(c_it);
      }
      CP_STAMP(1);
      const int batch = c_r;                          // 0 / 1: that half of the next stage's pieces; 2: none
      read_frags(c_r, c_sg & 1, g & 1, [&](int tg) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < G::PPW; ++q) {
          const int b = q < G::QA ? 0 : 1, qq = b == 0 ? q : q - G::QA, nb = b == 0 ? G::QA : G::PPW - G::QA;
          if (b == batch && qq * 3 / nb == tg) p_piece(4 * G::PG0 + w4 + 4 * q, p_lt[q]);
        }
        if (tg == 2 && batch == 1) p_next();
      });
      if (c_s == S - 1 && c_r == 1) res_warm(c_t);
      __builtin_amdgcn_sched_barrier(0);
      CP_STAMP(2);
      wait_lds();
      __builtin_amdgcn_sched_barrier(0);
      CP_STAMP(3);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      CP_STAMP(4);
      // ---------------- phase 2g+1: MFMA ----------------
      if (c_first()) mfma_first();
      else mfma_next();
      c_next();
      __builtin_amdgcn_sched_barrier(0);
      CP_STAMP(5);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      CP_STAMP(6);
    }
#if IFX_CONVPP_TRACE
    if (blockIdx.x == 0 && wave == 4 && (threadIdx.x & 63) == 0 && A.trace != nullptr)
      for (int i = 0; i < 8; ++i) A.trace[8 + i] = seg[i];
#endif
    epilogue(c_t);
  }
}

template <int BN, int UPS>
static void launch(const ConvArgs& a, hipStream_t s) {
  using G = GeoP<BN, UPS>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_pp_kernel<BN, UPS>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    attr = true;
  }
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const int wgx = max(1, min(cus / 8, a.per_xcd));     // one workgroup per CU (LDS), the same number on every XCD
#if IFX_CONVPP_TRACE
  static unsigned long long* tr = nullptr;
  if (tr == nullptr) (void)hipMalloc((void**)&tr, 16 * sizeof(unsigned long long));
  (void)hipMemsetAsync(tr, 0, 16 * sizeof(unsigned long long), s);
  ConvArgs b = a;
  b.trace = tr;
  hipLaunchKernelGGL((conv_pp_kernel<BN, UPS>), dim3(wgx * 8), dim3(512), G::LDS, s, b);
  unsigned long long h[16];
  (void)hipMemcpyAsync(h, tr, sizeof(h), hipMemcpyDeviceToHost, s);
  (void)hipStreamSynchronize(s);
  const int n_my = (a.per_xcd + wgx - 1) / wgx, steps = n_my * 3 * a.KT * (a.Cin >> 5);
  fprintf(stderr, "conv_pp<%d,%d> %dx%d cin %d cout %d t %d: %d tiles/wg, %d steps; cycles per step\n  G0: mfma %.0f  wait_vm %.0f  barrier %.0f  epilogue %.0f  "
          "dma+frags %.0f  wait_lds %.0f  barrier %.0f  (patch piece %.0f)\n  G1: wait_vm %.0f  epilogue %.0f  dma+frags %.0f  wait_lds %.0f  barrier %.0f  mfma %.0f  barrier %.0f  (prologue %.0f)\n",
          BN, UPS, a.Ho, a.Wo, a.Cin, a.Cout, a.t_out, n_my, steps, (double)h[0] / steps, (double)h[1] / steps, (double)h[2] / steps, (double)h[3] / steps,
          (double)h[4] / steps, (double)h[5] / steps, (double)h[6] / steps, (double)h[7] / steps, (double)h[8] / steps, (double)h[9] / steps, (double)h[10] / steps,
          (double)h[11] / steps, (double)h[12] / steps, (double)h[13] / steps, (double)h[14] / steps, (double)h[15]);
#else
  hipLaunchKernelGGL((conv_pp_kernel<BN, UPS>), dim3(wgx * 8), dim3(512), G::LDS, s, a);
#endif
}
}  // namespace pp

// ---------------------------------------------------------------------------------------------------------------------
// Per-pixel channel RMS norm (+ SiLU) on channels-last frames, written into caller-chosen frame slots.
// Follows the bf16 op chain of `RMS_norm.forward` (vae.py:52-55) + `nn.SiLU`: n = bf16(||x||), y = bf16(x / max(n, eps)),
// y = bf16(y * sqrt(C)), y = bf16(y * gamma), y = bf16(silu(y)).  G lanes (a power of two >= C / 8) share one pixel.
struct NormArgs {
  const unsigned short* x;
  const unsigned short* gamma;
  unsigned short* y;
  long long out_frame_stride;
  int out_slot[MAXF];
  int frame_pixels, C, silu, planar;
  long long pixels;
  float scale;
};

// G lanes share one pixel; every lane holds CPL 16-byte chunks of it (chunk k G + lane: each load instruction of the group is
// contiguous).  CPL = 3 for the decoder's 96 / 192 / 384 channels (G = 4 / 8 / 16: every lane busy; with one chunk per lane a
// quarter of the lanes idled in a VALU-bound kernel), CPL = 1 for powers of two.  NPX pixels per group are loaded up front.
template <int G, int CPL>
__global__ __launch_bounds__(256) void rmsnorm_cl_kernel(NormArgs A) {
  constexpr int NPX = CPL == 1 ? 4 : 2;     // 16-byte loads in flight per lane: 4
  constexpr int GPB = 256 / G;              // lane groups per block
  const int lane_in = threadIdx.x % G, grp = threadIdx.x / G;
  const int pix0 = blockIdx.x * (GPB * NPX) + grp;          // pixel indices fit 32 bits (checked by the launcher)
  u16x8 xv[NPX][CPL], gv[CPL];
  bool on[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    on[k] = (k * G + lane_in) * 8 < A.C;
    gv[k] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (on[k]) gv[k] = *reinterpret_cast<const u16x8*>(A.gamma + (k * G + lane_in) * 8);
  }
#pragma unroll
  for (int i = 0; i < NPX; ++i) {
    const int pix = pix0 + i * GPB;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      xv[i][k] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (on[k] && pix < A.pixels) xv[i][k] = *reinterpret_cast<const u16x8*>(A.x + (long long)pix * A.C + (k * G + lane_in) * 8);
    }
  }
#pragma unroll
  for (int i = 0; i < NPX; ++i) {
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = bf2f(xv[i][k][e]);
        ss += f * f;
      }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const int pix = pix0 + i * GPB;
    if (pix >= A.pixels) continue;
    const float n = fmaxf(rbf(sqrtf(ss)), 1e-12f);
    // x / n without the IEEE division sequence (the kernel was VALU-bound on it): reciprocal + one residual step gives
    // the correctly rounded quotient except for ties no bf16 rounding can see
    const float rn = __builtin_amdgcn_rcpf(n);
    const int f = pix / A.frame_pixels, pp = pix - f * A.frame_pixels;
    // planar output: 16-byte chunk ch of the pixel goes to plane ch >> 2, [plane][pixel][32]
    unsigned short* yf = A.y + (long long)A.out_slot[f] * A.out_frame_stride;
    unsigned short* yp = yf + (long long)pp * A.C;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      if (!on[k]) continue;
      u16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = bf2f(xv[i][k][e]);
        float q = x * rn;
        q = __builtin_fmaf(__builtin_fmaf(-q, n, x), rn, q);
        float v = rbf(q);
        v = rbf(v * A.scale);
        v = rbf(v * bf2f(gv[k][e]));
        if (A.silu) v = v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
        o[e] = f2bf(v);
      }
      const int ch = k * G + lane_in;
      if (A.planar) *reinterpret_cast<u16x8*>(yf + ((long long)(ch >> 2) * A.frame_pixels + pp) * 32 + (ch & 3) * 8) = o;
      else *reinterpret_cast<u16x8*>(yp + ch * 8) = o;
    }
  }
}

// Row softmax of bf16 scores (single-head spatial attention of the VAE middle block, vae.py:250-254): one wave per row.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const unsigned short* __restrict__ s, unsigned short* __restrict__ p,
                                                           int rows, int cols, int ld, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const unsigned short* sr = s + (size_t)row * ld;
  unsigned short* pr = p + (size_t)row * ld;
  float m = -INFINITY;
  for (int c = lane * 8; c < cols; c += 512) {
    const u16x8 v = *reinterpret_cast<const u16x8*>(sr + c);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c + e < cols) m = fmaxf(m, bf2f(v[e]));
  }
  m = wave_max(m);
  float sum = 0.f;
  for (int c = lane * 8; c < cols; c += 512) {
    const u16x8 v = *reinterpret_cast<const u16x8*>(sr + c);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c + e < cols) sum += __expf((bf2f(v[e]) - m) * scale);
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int c = lane * 8; c < cols; c += 512) {
    const u16x8 v = *reinterpret_cast<const u16x8*>(sr + c);
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = c + e < cols ? f2bf(__expf((bf2f(v[e]) - m) * scale) * inv) : (unsigned short)0;
    *reinterpret_cast<u16x8*>(pr + c) = o;
  }
}

}  // namespace conv
}  // namespace ifx

using namespace ifx;
using namespace ifx::conv;

extern "C" int ifx_conv3d_cl(const ifx_conv3d_desc* d, void* stream) {
  IFX_REQUIRE(d && d->x && d->w && d->y && d->zero_page, "ifx_conv3d_cl: null argument");
  IFX_REQUIRE(d->kt == 1 || d->kt == 3, "ifx_conv3d_cl: temporal kernel %d not built (1 or 3)", d->kt);
  IFX_REQUIRE(d->ks == 1 || d->ks == 3, "ifx_conv3d_cl: spatial kernel %d not built (1 or 3)", d->ks);
  IFX_REQUIRE(d->upsample == 0 || (d->upsample == 1 && d->ks == 3), "ifx_conv3d_cl: upsample needs the 3x3 kernel");
  IFX_REQUIRE(d->cin > 0 && d->cin % 32 == 0, "ifx_conv3d_cl: cin %d must be a multiple of 32", d->cin);
  IFX_REQUIRE(d->in_planar == 0 || d->in_planar == 1, "ifx_conv3d_cl: in_planar %d (0 or 1)", d->in_planar);
  IFX_REQUIRE(d->cout > 0 && (d->cout % 4 == 0 || d->cout < 4), "ifx_conv3d_cl: cout %d must be a multiple of 4 (or < 4)", d->cout);
  IFX_REQUIRE(d->t_out >= 1 && d->t_out + d->kt - 1 <= MAXF, "ifx_conv3d_cl: %d output frames per call (max %d inputs)",
              d->t_out, MAXF);
  IFX_REQUIRE(d->hs > 0 && d->ws > 0, "ifx_conv3d_cl: empty frame");
  IFX_REQUIRE((long long)d->hs * d->ws * d->cin < (1ll << 31) && (long long)d->hs * d->ws * d->cout * (d->upsample ? 4 : 1) < (1ll << 31),
              "ifx_conv3d_cl: frame too large for 32-bit pixel offsets");
  ConvArgs a;
  a.x = d->x;
  a.w = d->w;
  a.bias = d->bias;
  a.res = d->residual;
  a.zero = (const unsigned short*)d->zero_page;
  a.y = d->y;
  a.in_frame_stride = d->in_frame_stride;
  a.out_frame_stride = d->out_frame_stride;
  for (int i = 0; i < MAXF; ++i) {
    a.in_slot[i] = i < d->t_out + d->kt - 1 ? d->in_slots[i] : -1;
    a.out_slot[i] = i < d->t_out ? d->out_slots[i] : 0;
  }
  a.Hs = d->hs;
  a.Ws = d->ws;
  a.Cin = d->cin;
  a.Ho = d->hs << d->upsample;
  a.Wo = d->ws << d->upsample;
  a.Cout = d->cout;
  a.KT = d->kt;
  a.t_out = d->t_out;
  // channel tile: 96 wherever it divides (fragment prefetch + two weight groups ahead fit the register / LDS budget,
  // and 384 = 4 x 96 gives the 60 x 104 level more workgroups), else 128 / 64 / 32
  const int bn = d->cout <= 32 ? 32 : (d->cout % 96 == 0 ? 96 : (d->cout % 128 == 0 ? 128 : (d->cout <= 64 ? 64 : 128)));
  a.tiles_w = (a.Wo + TW - 1) / TW;
  a.tiles_h = (a.Ho + TH - 1) / TH;
  a.tiles_n = (d->cout + bn - 1) / bn;
  a.total = a.tiles_w * a.tiles_h * a.tiles_n * d->t_out;
  a.per_xcd = (a.total + 7) / 8;
  static int ablate = -1;
  if (ablate < 0) {
    const char* e = getenv("IFX_CONV_ABLATE");
    ablate = e ? atoi(e) : 0;
  }
  a.ablate = ablate;
  a.trace = nullptr;
  a.in_planar = d->in_planar ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  // the persistent ping-pong kernel serves the 3 x 3 spatial kernels on 96-channel tiles (every expensive layer of the decoder);
  // option conv_variant = 1 keeps the lock-step kernel (A/B, tools/bench_vae.py --variant)
  if (d->ks == 3 && bn == 96 && conv_variant() != 1 && d->cout <= 512 && (long long)d->hs * d->ws * d->cin * 2 < (1ll << 31) &&
      (long long)a.Ho * a.Wo * d->cout * 2 < (1ll << 31) && (long long)d->kt * 9 * d->cin * d->cout * 2 < (1ll << 31)) {
    if (d->upsample) pp::launch<96, 1>(a, s);
    else pp::launch<96, 0>(a, s);
    return check_launch("ifx_conv3d_cl");
  }
#define IFX_CONV_BN(UPS, KS)                              \
  switch (bn) {                                           \
    case 32: launch<32, UPS, KS>(a, s); break;            \
    case 64: launch<64, UPS, KS>(a, s); break;            \
    case 96: launch<96, UPS, KS>(a, s); break;            \
    default: launch<128, UPS, KS>(a, s); break;           \
  }
  if (d->ks == 1) {
    IFX_CONV_BN(0, 1)
  } else if (d->upsample) {
    IFX_CONV_BN(1, 3)
  } else {
    IFX_CONV_BN(0, 3)
  }
#undef IFX_CONV_BN
  return check_launch("ifx_conv3d_cl");
}

extern "C" int ifx_rmsnorm_cl(const ifx_bf16* x, const ifx_bf16* gamma, ifx_bf16* y, int64_t out_frame_stride,
                              const int32_t* out_slots, int32_t frames, int32_t frame_pixels, int32_t channels,
                              int32_t flags, void* stream) {
  IFX_REQUIRE(x && gamma && y && out_slots, "ifx_rmsnorm_cl: null argument");
  IFX_REQUIRE(frames >= 1 && frames <= MAXF && frame_pixels > 0, "ifx_rmsnorm_cl: %d frames per call (max %d)", frames, MAXF);
  IFX_REQUIRE((long long)frames * frame_pixels < (1ll << 31) - (1 << 16), "ifx_rmsnorm_cl: too many pixels per call");
  IFX_REQUIRE(channels % 8 == 0 && channels >= 8 && channels <= 512, "ifx_rmsnorm_cl: channels %d not in [8, 512] step 8", channels);
  NormArgs a;
  a.x = x;
  a.gamma = gamma;
  a.y = y;
  a.out_frame_stride = out_frame_stride;
  for (int i = 0; i < MAXF; ++i) a.out_slot[i] = i < frames ? out_slots[i] : 0;
  a.frame_pixels = frame_pixels;
  a.C = channels;
  IFX_REQUIRE((flags & ~3) == 0 && (!(flags & IFX_NORM_OUT_PLANAR) || channels % 32 == 0),
              "ifx_rmsnorm_cl: flags %d (IFX_NORM_SILU | IFX_NORM_OUT_PLANAR; planar output needs channels %% 32 == 0)", flags);
  a.silu = flags & IFX_NORM_SILU;
  a.planar = (flags & IFX_NORM_OUT_PLANAR) ? 1 : 0;
  a.pixels = (long long)frames * frame_pixels;
  a.scale = sqrtf((float)channels);
  const int chunks = channels / 8;
  hipStream_t s = (hipStream_t)stream;
#define IFX_NORM_G(GG, CC)                                                                                     \
  {                                                                                                            \
    const long long per_block = (256 / GG) * (CC == 1 ? 4 : 2);                                                \
    const long long blocks = (a.pixels + per_block - 1) / per_block;                                           \
    hipLaunchKernelGGL((rmsnorm_cl_kernel<GG, CC>), dim3((unsigned)blocks), dim3(256), 0, s, a);               \
  }
  if (chunks % 3 == 0 && (chunks / 3 & (chunks / 3 - 1)) == 0 && chunks / 3 <= 64) {      // 96 / 192 / 384 ... channels
    const int g3 = chunks / 3;
    if (g3 == 1) IFX_NORM_G(1, 3)
    else if (g3 == 2) IFX_NORM_G(2, 3)
    else if (g3 == 4) IFX_NORM_G(4, 3)
    else if (g3 == 8) IFX_NORM_G(8, 3)
    else if (g3 == 16) IFX_NORM_G(16, 3)
    else if (g3 == 32) IFX_NORM_G(32, 3)
    else IFX_NORM_G(64, 3)
  } else if (chunks <= 1) IFX_NORM_G(1, 1)
  else if (chunks <= 2) IFX_NORM_G(2, 1)
  else if (chunks <= 4) IFX_NORM_G(4, 1)
  else if (chunks <= 8) IFX_NORM_G(8, 1)
  else if (chunks <= 16) IFX_NORM_G(16, 1)
  else if (chunks <= 32) IFX_NORM_G(32, 1)
  else IFX_NORM_G(64, 1)
#undef IFX_NORM_G
  return check_launch("ifx_rmsnorm_cl");
}

extern "C" int ifx_softmax_rows(const ifx_bf16* scores, ifx_bf16* probs, int32_t rows, int32_t cols, int32_t ld, float scale,
                                void* stream) {
  IFX_REQUIRE(scores && probs && rows > 0 && cols > 0 && ld >= cols && ld % 8 == 0, "ifx_softmax_rows: bad arguments");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, scores, probs, rows, cols, ld,
                     scale);
  return check_launch("ifx_softmax_rows");
}
