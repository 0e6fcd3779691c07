// Does a SIMD co-issue MFMA from one wave with VALU / transcendental ops from another?  (gfx950)
// 8-wave workgroups (waves w and w+4 share a SIMD); waves 0-3 run role A, waves 4-7 role B.
// roles: 0 idle, 1 MFMA 32x32x16 bf16 (4 independent accumulators), 2 v_fma_f32 (16 independent chains),
//        3 v_exp_f32 (16 independent chains), 4 v_pk_fma_f32 (8 independent chains of 2), 5 MFMA with ONE accumulator
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_overlap.hip -o tools/bin/probe_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ const unsigned char* g_dma_src;

template <int ROLE>
__device__ __forceinline__ float run_role(int iters, float seed) {
  if (ROLE == 1 || ROLE == 5) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; ++it) {
      if (ROLE == 1) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      } else {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      }
    }
    return c0[0] + c1[1] + c2[2] + c3[3];
  } else if (ROLE == 6) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4, 0, 0, 0);
      c5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c5, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c6, 0, 0, 0);
      c7 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c7, 0, 0, 0);
    }
    return c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
  } else if (ROLE == 7) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    s16x4 a = {1, 2, 3, 4}, b = {4, 3, 2, 1};
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c3, 0, 0, 0);
    }
    return c0[0] + c1[1] + c2[2] + c3[3];
  } else if (ROLE == 8) {
    float x[8];
    f32x2 y[4], k = {seed, seed};
    for (int i = 0; i < 8; ++i) x[i] = seed * 0.001f + i * 0.01f;
    for (int i = 0; i < 4; ++i) y[i] = f32x2{seed + i, seed - i};
    unsigned pk[4] = {0, 0, 0, 0};
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(k));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(k));
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(x[2 * i]), "v"(x[2 * i + 1]));
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x[i]));
    }
    float s = acc + (float)(pk[0] + pk[1] + pk[2] + pk[3]);
    for (int i = 0; i < 4; ++i) s += y[i][0] + y[i][1];
    return s;
  } else if (ROLE == 9 || ROLE == 10) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    const unsigned char* base = lds + (threadIdx.x & 63) * 16;
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    f32x4 c[8] = {};
    f32x16 d0 = {}, d1 = {}, d2 = {}, d3 = {};
    for (int it = 0; it < iters; ++it) {
      bf16x8 f0 = *reinterpret_cast<const bf16x8*>(base + ((it & 3) << 10));
      bf16x8 f1 = *reinterpret_cast<const bf16x8*>(base + 4096 + ((it & 3) << 10));
      asm volatile("" : "+v"(f0), "+v"(f1));
      if (ROLE == 9) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(i & 1 ? f1 : f0, b, c[i], 0, 0, 0);
      } else {
        d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, b, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, b, d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, b, d3, 0, 0, 0);
      }
    }
    float s = d0[0] + d1[1] + d2[2] + d3[3];
    for (int i = 0; i < 8; ++i) s += c[i][0];
    return s;
  } else if (ROLE == 11 || ROLE == 12 || ROLE == 13) {
    // ONE wave interleaving MFMAs with its own VALU work: per iteration 4 x { MFMA 32x32x16 ; NV x (v_fma + v_exp) }
    // 11: 2 fma + 2 exp per MFMA (the softmax's ratio), 12: 4 plain fma per MFMA, 13: 2 exp only per MFMA
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed * 0.001f + i * 0.01f;
    for (int it = 0; it < iters; ++it) {
#define STEP(C, I)                                                                        \
      C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, C, 0, 0, 0);                      \
      if (ROLE == 11) {                                                                   \
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[2 * I]) : "v"(seed));            \
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[2 * I]));                                \
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[2 * I + 1]) : "v"(seed));        \
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[2 * I + 1]));                            \
      } else if (ROLE == 12) {                                                            \
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[2 * I]) : "v"(seed));            \
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[2 * I + 1]) : "v"(seed));        \
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[2 * I]) : "v"(seed));            \
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[2 * I + 1]) : "v"(seed));        \
      } else {                                                                            \
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[2 * I]));                                \
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[2 * I + 1]));                            \
      }
      STEP(c0, 0) STEP(c1, 1) STEP(c2, 2) STEP(c3, 3)
#undef STEP
    }
    float s = c0[0] + c1[1] + c2[2] + c3[3];
    for (int i = 0; i < 8; ++i) s += x[i];
    return s;
  } else if (ROLE == 14 || ROLE == 15) {
    // as 12 (4 plain fma behind every MFMA) but the fma READ registers an MFMA wrote half an iteration earlier
    // (ROLE 14: two accumulator sets ping-pong, like S(t) / S(t+1) of the software-pipelined attention loop);
    // ROLE 15: the fma read the accumulators of the set the in-flight MFMAs are NOT touching, values never consumed
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    f32x16 c[4] = {}, d[4] = {};
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = d[i][4 * e];
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(v), "v"(seed));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        d[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d[i], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = c[i][4 * e];
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(v), "v"(seed));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float s = acc;
    for (int i = 0; i < 4; ++i) s += c[i][0] + d[i][1];
    return s;
  } else if (ROLE == 16) {
    // four LDS-DMA instructions (64 lanes x 16 B, L2-resident source) per iteration, nothing else
    __shared__ __attribute__((aligned(16))) unsigned char dlds[8 * 4096];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    const unsigned char* src = g_dma_src + ((blockIdx.x & 7) * 64) * 4096 + (threadIdx.x & 63) * 16;
    const int w = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + ((it & 63) * 4 + i) * 1024), (lds_ptr_t)(dlds + w * 4096 + i * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return dlds[0];
  } else if (ROLE == 2) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(seed));
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    return s;
  } else if (ROLE == 3) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed * 0.001f + i * 0.01f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    return s;
  } else if (ROLE == 4) {
    f32x2 x[8], k = {seed, seed};
    for (int i = 0; i < 8; ++i) x[i] = f32x2{seed + i, seed - i};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(k));
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    return s;
  }
  return 0.f;
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  const long long t0 = __builtin_readcyclecounter();
  float r = wave < 4 ? run_role<RA>(iters, seed) : run_role<RB>(iters, seed);
  const long long t1 = __builtin_readcyclecounter();
  if (r == 12345.678f) out[threadIdx.x] = r;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[1024 + wave] = (float)(t1 - t0);   // s_memtime ticks (100 MHz)
}

static float g_ta, g_tb;
// three waves per SIMD: waves 0-3 role RA, 4-7 role RB, 8-11 role RC
template <int RA, int RB, int RC>
__global__ __launch_bounds__(768) void probe3(float* out, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  const long long t0 = __builtin_readcyclecounter();
  float r = wave < 4 ? run_role<RA>(iters, seed) : (wave < 8 ? run_role<RB>(iters, seed) : run_role<RC>(iters, seed));
  const long long t1 = __builtin_readcyclecounter();
  if (r == 12345.678f) out[threadIdx.x] = r;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[1024 + wave] = (float)(t1 - t0);
}
template <int RA, int RB, int RC>
static void time3(float* d, int iters, const char** names) {
  probe3<RA, RB, RC><<<256, 768>>>(d, iters, 1.0f);
  hipDeviceSynchronize();
  probe3<RA, RB, RC><<<256, 768>>>(d, iters, 1.0f);
  hipDeviceSynchronize();
  float h[12];
  hipMemcpy(h, d + 1024, sizeof(h), hipMemcpyDeviceToHost);
  printf("A=%-12s B=%-12s C=%-12s : cycles/iter A %6.1f  B %6.1f  C %6.1f\n", names[RA], names[RB], names[RC],
         h[0] / iters, h[4] / iters, h[8] / iters);
}

// as probe, but role B raises its wave priority (s_setprio 3) and role A lowers it (0)
template <int RA, int RB>
__global__ __launch_bounds__(512) void probe_prio(float* out, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  if (wave >= 4) __builtin_amdgcn_s_setprio(3);
  else __builtin_amdgcn_s_setprio(0);
  const long long t0 = __builtin_readcyclecounter();
  float r = wave < 4 ? run_role<RA>(iters, seed) : run_role<RB>(iters, seed);
  const long long t1 = __builtin_readcyclecounter();
  if (r == 12345.678f) out[threadIdx.x] = r;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[1024 + wave] = (float)(t1 - t0);
}
template <int RA, int RB>
static void time_prio(float* d, int iters, const char** names) {
  probe_prio<RA, RB><<<256, 512>>>(d, iters, 1.0f);
  hipDeviceSynchronize();
  probe_prio<RA, RB><<<256, 512>>>(d, iters, 1.0f);
  hipDeviceSynchronize();
  float h[8];
  hipMemcpy(h, d + 1024, sizeof(h), hipMemcpyDeviceToHost);
  printf("PRIO(B=3,A=0) A=%-12s B=%-12s : cycles/iter A %6.1f  B %6.1f\n", names[RA], names[RB], h[0] / iters, h[4] / iters);
}

template <int RA, int RB>
static float time_it(float* d, int iters) {
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  probe<RA, RB><<<256, 512>>>(d, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(s);
  probe<RA, RB><<<256, 512>>>(d, iters, 1.0f);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  float h[8];
  hipMemcpy(h, d + 1024, sizeof(h), hipMemcpyDeviceToHost);
  g_ta = h[0]; g_tb = h[4];
  return ms * 1e3f;
}

int main() {
  float* d; hipMalloc(&d, 8192);
  const int it = 20000;
  const char* names[] = {"idle", "mfma(4 acc)", "v_fma x16", "v_exp x16", "v_pk_fma x8", "mfma(1 acc)", "mfma16x16x32 x8", "mfma32x32x8 x4", "softmax mix 24", "M16+lds", "M32+lds", "M+2fma2exp", "M+4fma", "M+2exp", "M+4fma(acc)", "x", "4 LDS-DMA"};
#define T(A, B) { float us = time_it<A, B>(d, it); printf("A=%-12s B=%-12s : %8.1f us   role A %8.1f us  role B %8.1f us\n", names[A], names[B], us, g_ta * 0.01f, g_tb * 0.01f); }
  T(1, 0) T(5, 0) T(0, 2) T(0, 3) T(0, 4) T(1, 1) T(2, 2) T(3, 3) T(1, 2) T(1, 3) T(1, 4) T(5, 2) T(5, 3) T(2, 3)
  T(2, 1) T(6, 0) T(6, 2) T(7, 0) T(7, 2) T(2, 7)
  T(3, 1) T(4, 1) T(3, 7) T(2, 6)
  T(8, 0) T(9, 0) T(10, 0) T(8, 10) T(10, 8) T(8, 9) T(9, 8) T(8, 8)
  {
    unsigned char* g;
    hipMalloc(&g, 8 * 64 * 4096 + 65536);
    hipMemset(g, 0, 8 * 64 * 4096 + 65536);
    hipMemcpyToSymbol(HIP_SYMBOL(g_dma_src), &g, sizeof(g));
  }
  T(16, 0) T(0, 16) T(16, 16) T(16, 1) T(1, 16) T(16, 2) T(2, 16)
  T(14, 0) T(14, 14)
  T(11, 0) T(12, 0) T(13, 0) T(11, 11) T(12, 12) T(13, 13)
  T(6, 3) T(3, 6) T(6, 4) T(4, 6) T(6, 8) T(8, 6) T(9, 9) T(6, 6)
  time_prio<1, 2>(d, it, names);
  time_prio<1, 3>(d, it, names);
  time_prio<1, 4>(d, it, names);
  time_prio<2, 1>(d, it, names);
  time3<1, 2, 0>(d, it, names);
  time3<1, 2, 2>(d, it, names);
  time3<1, 3, 3>(d, it, names);
  time3<1, 2, 3>(d, it, names);
  time3<1, 4, 4>(d, it, names);
  time3<1, 1, 2>(d, it, names);
  time3<0, 2, 2>(d, it, names);
  time3<2, 2, 2>(d, it, names);
  // per iteration: mfma role = 4 MFMA (4*32 = 128 pipe cycles); v_fma role = 16 VALU (64 issue cycles); v_exp = 16 trans
  printf("iters %d; at 2.4 GHz 128 cycles * %d = %.1f us\n", it, it, 128.0 * it / 2400.0);
  return 0;
}
