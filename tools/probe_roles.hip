// Throughput shape of a ROLE-SPECIALISED attention step on gfx950 (no real data, only the instruction / LDS traffic mix):
// waves 0-3 (older) run the softmax of tile t, waves 4-7 (younger, same SIMDs) run QK(t+1) and PV(t-1) on the matrix pipe;
// S goes MFMA wave -> LDS -> softmax wave as fp32, P comes back as bf16.  One workgroup barrier per step.
// Question: how many cycles per 64-key step for 128 queries, against 2118 (= 4237 / 2) of the shipped ping-pong kernel
// where both waves of a SIMD alternate roles?  (tools/probe_overlap.hip: an OLDER VALU / transcendental wave overlaps
// with a YOUNGER MFMA wave; the other way round the VALU wave starves.)
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_roles.hip -o tools/bin/probe_roles
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// MODE 0: specialised roles (softmax waves older).  MODE 1: same but softmax waves are the YOUNGER ones (4-7).
// MODE 2: every wave does everything for its own 32 queries, two waves per SIMD, no hand-off (the shipped structure's mix).
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
// FEAT bits (MODE 2 only): 1 = V fragments through pairs of ds_read_tr16_b64, 2 = four LDS-DMA instructions per wave per step,
// 4 = row-sum ballot of the lazy maximum, 8 = packed exponent arguments (v_pk_fma_f32)
template <int MODE, int FEAT = 0>
__global__ __launch_bounds__(512) void steps(float* out, int iters, const unsigned char* gsrc = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* KV = smem;                       // 32 KiB: one K tile + one V tile
  unsigned char* SB = smem + 32768;               // [2][4][8 KiB] fp32 S
  unsigned char* PB = SB + 65536;                 // [2][4][4 KiB] bf16 P
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool softmax_role = MODE == 0 ? wave < 4 : (MODE == 1 ? wave >= 4 : false);
  const int pair = wave & 3;
  for (int i = threadIdx.x; i < 32768 / 4; i += 512) reinterpret_cast<float*>(KV)[i] = 0.001f * i;
  for (int i = threadIdx.x; i < 98304 / 4; i += 512) reinterpret_cast<float*>(SB)[i] = 0.25f;
  __syncthreads();
  f32x16 s0 = {}, s1 = {}, o0 = {}, o1 = {}, o2 = {}, o3 = {};
  bf16x8 q[8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) q[i][j] = (__bf16)(0.01f * (i + j));
  float acc = 0.f, m = 0.5f;
  const long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < iters; ++t) {
    const int buf = t & 1;
    if ((FEAT & 2) && gsrc) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gsrc + ((size_t)((blockIdx.x & 7) * 64 + (t & 63)) * 32768) + (wave * 4 + i) * 1024 + lane * 16),
                                         (lds_ptr_t)(smem + 98304 + 32768 + (wave * 4 + i) * 1024 - 32768), 16, 0, 0);
    }
    if (MODE == 2 || !softmax_role) {
      // ---- QK: 16 K fragments -> 16 MFMAs
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(KV + ((ks * 64 + lane) << 4));
        const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(KV + 8192 + ((ks * 64 + lane) << 4));
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q[ks], s1, 0, 0, 0);
      }
    }
    if (MODE != 2 && !softmax_role) {
      // ---- hand S to the softmax wave: 8 x ds_write_b128
      unsigned char* sp = SB + (buf * 4 + pair) * 8192 + lane * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4*>(sp + i * 1024) = f32x4{s0[4 * i], s0[4 * i + 1], s0[4 * i + 2], s0[4 * i + 3]};
        *reinterpret_cast<f32x4*>(sp + 4096 + i * 1024) = f32x4{s1[4 * i], s1[4 * i + 1], s1[4 * i + 2], s1[4 * i + 3]};
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.f;
    }
    bf16x8 p[4];
    if (MODE == 2 || softmax_role) {
      // ---- softmax of one 32 x 64 tile: 32 scores per lane
      float s[32];
      if (MODE == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = s0[r]; s[16 + r] = s1[r]; s0[r] = s1[r] = 0.f; }
      } else {
        const unsigned char* sp = SB + ((buf ^ 1) * 4 + pair) * 8192 + lane * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(sp + i * 1024);
          s[4 * i] = v[0]; s[4 * i + 1] = v[1]; s[4 * i + 2] = v[2]; s[4 * i + 3] = v[3];
        }
      }
      float sum = 0.f;
      if (FEAT & 8) {
        const f32x2 cv = {0.127f, 0.127f}, mv = {m, m};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const f32x2 e = f32x2{s[2 * r], s[2 * r + 1]} * cv - mv;
          s[2 * r] = __builtin_amdgcn_exp2f(e[0]);
          s[2 * r + 1] = __builtin_amdgcn_exp2f(e[1]);
          sum += s[2 * r] + s[2 * r + 1];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
          s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], 0.127f, -m));
          sum += s[r];
        }
      }
      if (FEAT & 4) {
        if (__any(!(sum < 1048576.f))) m += 1.0f;
      }
      acc += sum;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) p[i][j] = (__bf16)s[i * 8 + j];
      if (MODE != 2) {
        unsigned char* pp = PB + ((buf ^ 1) * 4 + pair) * 4096 + lane * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<bf16x8*>(pp + i * 1024) = p[i];
      }
    }
    if (MODE == 2 || !softmax_role) {
      if (MODE != 2) {
        const unsigned char* pp = PB + (buf * 4 + pair) * 4096 + lane * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = *reinterpret_cast<const bf16x8*>(pp + i * 1024);
      }
      // ---- PV: 16 V fragments (two transposed 8-byte reads each) -> 16 MFMAs
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          bf16x8 v;
          if (FEAT & 16) {
            const int vi = lane & 15, vg1 = (lane >> 4) & 1, hi = lane >> 5, v_rowq = vi >> 2;
            const int v_in = (vg1 << 5) | ((vi & 3) << 3);
            const int bb = ks >> 1, s2 = ks & 1;
            const unsigned char* vr0 = KV + 16384 + (32 * bb + 16 * s2 + 4 * hi + v_rowq) * 256 + v_in;
            const unsigned char* vr1 = vr0 + 8 * 256;
            const int ch = (d ^ v_rowq) << 6;
            const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vr0 + ch));
            const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vr1 + ch));
            v = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
          } else if (FEAT & 1) {
            const unsigned char* vp = KV + 16384 + (((ks * 4 + d) * 64 + lane) << 4);
            const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vp));
            const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vp + 8));
            v = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
          } else {
            v = *reinterpret_cast<const bf16x8*>(KV + 16384 + (((ks * 4 + d) * 64 + lane) << 4));
          }
          f32x16& o = d == 0 ? o0 : (d == 1 ? o1 : (d == 2 ? o2 : o3));
          o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v, p[ks], o, 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const long long t1 = __builtin_readcyclecounter();
  float r = acc + s0[0] + s1[1] + o0[0] + o1[1] + o2[2] + o3[3];
  if (r == 12345.678f) out[threadIdx.x] = r;
  if (blockIdx.x == 0 && lane == 0) out[1024 + wave] = (float)(t1 - t0);
}

template <int MODE, int FEAT = 0>
static void run(float* d, int iters, const char* what, int queries, const unsigned char* g = nullptr) {
  hipFuncSetAttribute((const void*)steps<MODE, FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  steps<MODE, FEAT><<<256, 512, 140000>>>(d, iters, g);
  hipDeviceSynchronize();
  hipEventRecord(s);
  steps<MODE, FEAT><<<256, 512, 140000>>>(d, iters, g);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const double flops = 256.0 * iters * queries * 64.0 * 128.0 * 4.0;
  printf("%-58s: %7.1f us, %6.0f ns/step, %5.0f TFLOP/s-equivalent (%d queries / step / CU)\n", what, ms * 1e3, ms * 1e6 / iters,
         flops / ms / 1e9, queries);
}

int main() {
  float* d;
  hipMalloc(&d, 16384);
  const int it = 20000;
  run<2>(d, it, "all 8 waves do QK + softmax + PV (no hand-off)", 256);
  run<0>(d, it, "specialised: softmax waves 0-3 (older), MFMA waves 4-7", 128);
  run<1>(d, it, "specialised: MFMA waves 0-3 (older), softmax waves 4-7", 128);
  unsigned char* g;
  hipMalloc(&g, (size_t)256 * 64 * 32768);
  hipMemset(g, 0, (size_t)256 * 64 * 32768);
  run<2, 1>(d, it, "no hand-off + V via ds_read_tr16_b64", 256);
  run<2, 2>(d, it, "no hand-off + 4 LDS-DMA per wave per step", 256, g);
  run<2, 4>(d, it, "no hand-off + lazy-max ballot", 256);
  run<2, 8>(d, it, "no hand-off + packed exponent arguments", 256);
  run<2, 15>(d, it, "no hand-off + all four", 256, g);
  run<2, 16>(d, it, "no hand-off + V tr reads with the shipped swizzle", 256);
  return 0;
}
