// Channels-last causal 3-D convolution for the Wan VAE decoder, gfx950 only.
//
// Replaces `CausalConv3d.forward` (inferix/models/wan_base/vae.py:26-34) + the feature-cache concatenation around it
// (vae.py:207-216), `Upsample` + `Conv2d` of `Resample` (vae.py:58-64, 83-90, 139-141) and the residual add of
// `ResidualBlock.forward` (vae.py:219).
//
// Design (LDS-tiled direct convolution on the matrix cores, not an im2col GEMM):
//   * activations are [frame][h][w][C] bf16; a workgroup (8 waves) owns an 8 x 64 pixel tile of ONE output frame and BN
//     output channels; wave w computes image row w of the tile (two 32-pixel MFMA blocks) for all BN channels.
//   * the K loop runs over stages (input frame dt, 32-channel chunk cc).  Per stage the (8+2) x (64+2) pixel halo patch of
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
    poff[r] = ok ? (sh * A.Ws + sw) * A.Cin + (((lane & 3) ^ ((px >> 2) & 3)) << 3) : -1;
  }
  const int CC = A.Cin >> 5;
  const int S = A.KT * CC, totalG = S * GPS;
  auto issue_patch = [&](int s, int r) {            // piece r of this wave for stage s
    const int dt = s / CC, cc = s - dt * CC;
    const int f = A.in_slot[to + dt];
    const unsigned short* base = A.x + (long long)f * A.in_frame_stride + cc * 32;
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
    const int dt = s / CC, cc = s - dt * CC;
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
  int frame_pixels, C, silu;
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
    unsigned short* yp = A.y + (long long)A.out_slot[f] * A.out_frame_stride + (long long)pp * A.C;
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
      *reinterpret_cast<u16x8*>(yp + (k * G + lane_in) * 8) = o;
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
  hipStream_t s = (hipStream_t)stream;
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
                              int32_t silu, void* stream) {
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
  a.silu = silu;
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
