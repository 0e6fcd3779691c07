// Does a software exponential (range reduction + polynomial on the FMA pipe) relieve v_exp_f32 on gfx950?  (round-2 verdict, item 6)
// One wave per SIMD (or two: blocks of 512) runs the softmax segment of the attention kernel in miniature: 16 MFMA 32x32x16 bf16
// with 32 exponentials of independent values interleaved 2 per MFMA, in three forms:
//   mode 0: no exponentials (the MFMA floor)          mode 1: 32 x v_exp_f32
//   mode 2: 16 x v_exp_f32 + 16 x software exp2        mode 3: 32 x software exp2
//   mode 4..6: the same VALU without the MFMAs (pure issue cost)
// software exp2(x), x <= 0: t = x + 1.5 * 2^23 (integer part lands in the low mantissa bits), f = x - (t - 1.5 * 2^23) in [-0.5, 0.5],
// p = c0 + f (c1 + f (c2 + f c3)) (degree 3, rel. error 1e-4 < bf16's 2^-9), result = bits(p) + (bits(t) << 23): 2 add, 3 fma,
// 1 v_lshl_add_u32 = 6 full-rate VALU per value.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_exp.hip -o tools/bin/probe_exp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float soft_exp2(float x) {
  const float magic = 12582912.0f;
  const float t = x + magic;
  const float f = x - (t - magic);
  float p = 0.0555041086f;
  p = __builtin_fmaf(p, f, 0.2402265069f);
  p = __builtin_fmaf(p, f, 0.6931471806f);
  p = __builtin_fmaf(p, f, 1.0f);
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p) + (__builtin_bit_cast(unsigned, t) << 23));
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
  float v[32];
  for (int e = 0; e < 32; ++e) v[e] = -0.001f * (lane + e);
  constexpr bool MF = MODE < 4;
  constexpr int EX = MODE & 3;       // 0 none, 1 hw, 2 half, 3 soft
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (MF) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = 2 * m + k;
        if (EX == 1 || (EX == 2 && k == 0)) v[e] = __builtin_amdgcn_exp2f(v[e] - 0.25f);
        else if (EX == 3 || (EX == 2 && k == 1)) v[e] = soft_exp2(v[e] - 0.25f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) v[e] = -v[e] * 3.0f;       // keep the arguments negative and alive
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int e = 0; e < 32; ++e) s += v[e];
  for (int i = 0; i < 4; ++i) s += acc[i][lane & 15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int threads, float* out, long long* cyc, int iters) {
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < 256; ++i) mean += h[i];
  mean /= 256;
  // s_memtime counts at 100 MHz on gfx950 builds of this counter; report wall time per iteration instead (ns) alongside
  printf("| %-44s | %d | %8.1f | %9.1f |\n", name, threads / 256, ms * 1e6 / iters, mean / iters);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 8);
  const int iters = 20000;
  printf("| segment (16 MFMA 32x32x16 + 32 exponentials) | waves/SIMD | ns / iteration | counter ticks / iteration |\n|---|---:|---:|---:|\n");
  for (int threads : {256, 512}) {
    run<0>("MFMAs only", threads, out, cyc, iters);
    run<1>("MFMAs + 32 v_exp_f32", threads, out, cyc, iters);
    run<2>("MFMAs + 16 v_exp_f32 + 16 software exp2", threads, out, cyc, iters);
    run<3>("MFMAs + 32 software exp2", threads, out, cyc, iters);
    run<5>("32 v_exp_f32 alone", threads, out, cyc, iters);
    run<6>("16 v_exp_f32 + 16 software exp2 alone", threads, out, cyc, iters);
    run<7>("32 software exp2 alone", threads, out, cyc, iters);
  }
  return 0;
}
