// What does a grid-wide phase boundary INSIDE a persistent launch cost on this part, against the same boundary as a kernel boundary?
// (Round-4 verdict item 1 prices an in-kernel boundary at 1.1-1.4 us; MI355X_MICROARCH.md's price list says 4.1-7.2 us for a grid
//  barrier and 1.1-1.9 us for a dependent kernel boundary.  This probe measures both on the box it runs on.)
//
// P dependent phases, 256 (or 512) workgroups of 256 threads.  A phase: every workgroup reads the 4 KiB slab ANOTHER workgroup
// wrote in the previous phase (so the boundary has to publish data across CUs / XCDs, not only order execution), adds 1 and writes its
// own slab.  After P phases every element must be P: a boundary that does not make the data visible fails the check.
//   mode L : P launches of the phase kernel on one stream
//   mode F : ONE launch, flat barrier — lane 0: release fence, atomicAdd on one counter, relaxed poll (sc1) + s_sleep, acquire fence
//   (both barriers also in a form without fences: the slabs travel through write-through sc1 stores + sc1 loads, "sc1 data")
//   mode X : ONE launch, XCD-hierarchical barrier — arrive on the XCD's counter; the XCD's last arriver arrives on the top counter and
//            polls it, then publishes a generation word per XCD that the other workgroups of the XCD poll
// `body_us` > 0 adds that much busy work per phase (a stand-in for a kernel body: skewed arrivals).
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_grid_barrier.hip -o tools/bin/probe_grid_barrier
// run:   tools/bin/probe_grid_barrier [phases=64] [wgs=256] [body_us=0]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                                                   \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); }              \
  } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// WT: the slab travels through write-through (sc1) stores and sc1 loads (raw buffer operations, aux = 16) — the hand-off form of the
// split-K GEMM (ifx_gemm_pp.hip) and of MI355X_MICROARCH.md's recipe: no L2 write-back / invalidate fence is needed around the barrier
template <bool WT = false>
__device__ __forceinline__ void phase_body(const unsigned* src, unsigned* dst, int p, int nwg, long long body_ticks) {
  const int from = (int)(((long)blockIdx.x * 37 + 11 + p) % nwg);            // another workgroup's slab (changes every phase)
  u32x4 v;
  if (WT) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nwg * 4096, 0x00020000);
    v = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16, from * 4096, 16);
  } else {
    v = *reinterpret_cast<const u32x4*>(src + (size_t)from * 1024 + threadIdx.x * 4);
  }
  if (body_ticks > 0) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < body_ticks) __builtin_amdgcn_s_sleep(1);
  }
  v += 1u;
  if (WT) {
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, nwg * 4096, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, rd, threadIdx.x * 16, blockIdx.x * 4096, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // the store has left before this wave arrives at the barrier
  } else {
    *reinterpret_cast<u32x4*>(dst + (size_t)blockIdx.x * 1024 + threadIdx.x * 4) = v;
  }
}

__global__ __launch_bounds__(256) void phase_kernel(const unsigned* src, unsigned* dst, int p, int nwg, long long body_ticks) {
  phase_body<false>(src, dst, p, nwg, body_ticks);
}

template <bool WT>
__device__ __forceinline__ void barrier_flat(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!WT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    if (!WT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// words: [0..7] per-XCD arrival counters (64 B apart), [8] top counter, [9..16] per-XCD generation words — all 64 B apart
template <bool WT>
__device__ __forceinline__ void barrier_xcd(unsigned* w, unsigned gen, int per_xcd) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int xcd = blockIdx.x & 7;
    unsigned* arrive = w + xcd * 16;
    unsigned* top = w + 8 * 16;
    unsigned* generation = w + (9 + xcd) * 16;
    if (!WT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned n = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (n == gen * (unsigned)per_xcd) {                                        // the XCD's last arriver of this generation
      __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * 8u) __builtin_amdgcn_s_sleep(1);
      __hip_atomic_store(generation, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(2);
    }
    if (!WT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void persistent_kernel(unsigned* a, unsigned* b, int phases, int nwg, long long body_ticks, unsigned* words) {
  for (int p = 0; p < phases; ++p) {
    constexpr bool WT = MODE >= 2;
    phase_body<WT>((p & 1) ? b : a, (p & 1) ? a : b, p, nwg, body_ticks);
    if (p + 1 < phases) {
      if (MODE == 0 || MODE == 2) barrier_flat<WT>(words, (unsigned)(p + 1) * (unsigned)nwg);
      else barrier_xcd<WT>(words, (unsigned)(p + 1), nwg / 8);
    }
  }
}

static bool check(const unsigned* dev, int nwg, unsigned want) {
  std::vector<unsigned> h((size_t)nwg * 1024);
  CHECK(hipMemcpy(h.data(), dev, h.size() * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < h.size(); ++i)
    if (h[i] != want) { printf("  WRONG: element %zu = %u, want %u (a boundary did not publish its phase)\n", i, h[i], want); return false; }
  return true;
}

int main(int argc, char** argv) {
  const int phases = argc > 1 ? atoi(argv[1]) : 64, nwg = argc > 2 ? atoi(argv[2]) : 256;
  const double body_us = argc > 3 ? atof(argv[3]) : 0.0;
  const long long body_ticks = (long long)(body_us * 100.0);                  // wall_clock64 ticks at 100 MHz
  unsigned *a, *b, *words;
  CHECK(hipMalloc(&a, (size_t)nwg * 4096));
  CHECK(hipMalloc(&b, (size_t)nwg * 4096));
  CHECK(hipMalloc(&words, 4096));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("phases %d, workgroups %d x 256 threads, body %.1f us per phase\n", phases, nwg, body_us);
  for (int rep = 0; rep < 3; ++rep) {
    // ---- launches
    CHECK(hipMemsetAsync(a, 0, (size_t)nwg * 4096, s));
    CHECK(hipMemsetAsync(b, 0, (size_t)nwg * 4096, s));
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(e0, s));
    for (int p = 0; p < phases; ++p)
      hipLaunchKernelGGL(phase_kernel, dim3(nwg), dim3(256), 0, s, (p & 1) ? b : a, (p & 1) ? a : b, p, nwg, body_ticks);
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const bool ok_l = check((phases & 1) ? b : a, nwg, (unsigned)phases);
    printf("  launches              : %7.2f us per phase%s\n", ms * 1e3 / phases, ok_l ? "" : "  (WRONG)");
    // ---- one launch, flat / hierarchical barrier
    for (int mode = 0; mode < 4; ++mode) {
      CHECK(hipMemsetAsync(a, 0, (size_t)nwg * 4096, s));
      CHECK(hipMemsetAsync(b, 0, (size_t)nwg * 4096, s));
      CHECK(hipMemsetAsync(words, 0, 4096, s));
      CHECK(hipStreamSynchronize(s));
      CHECK(hipEventRecord(e0, s));
      if (mode == 0) hipLaunchKernelGGL(persistent_kernel<0>, dim3(nwg), dim3(256), 0, s, a, b, phases, nwg, body_ticks, words);
      else if (mode == 1) hipLaunchKernelGGL(persistent_kernel<1>, dim3(nwg), dim3(256), 0, s, a, b, phases, nwg, body_ticks, words);
      else if (mode == 2) hipLaunchKernelGGL(persistent_kernel<2>, dim3(nwg), dim3(256), 0, s, a, b, phases, nwg, body_ticks, words);
      else hipLaunchKernelGGL(persistent_kernel<3>, dim3(nwg), dim3(256), 0, s, a, b, phases, nwg, body_ticks, words);
      CHECK(hipEventRecord(e1, s));
      CHECK(hipStreamSynchronize(s));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const bool ok = check((phases & 1) ? b : a, nwg, (unsigned)phases);
      static const char* names[4] = {"flat barrier, fences         ", "XCD  barrier, fences         ", "flat barrier, sc1 data       ", "XCD  barrier, sc1 data       "};
      printf("  one launch, %s: %7.2f us per phase%s\n", names[mode], ms * 1e3 / phases, ok ? "" : "  (WRONG)");
    }
  }
  return 0;
}
