// How fast can LDS-DMA deliver?  Every workgroup (8 waves) streams `bytes_per_wg` from global memory into an LDS ring
// with NOUT wave-instructions (1 KiB each) in flight per wave and does nothing else.
//   footprint: each workgroup reads its own region of `region` bytes round-robin (small region -> L2 hits, large -> HBM / MALL)
//   ROWB     : bytes per row an instruction touches: 64 B x 16 rows, 128 B x 8 rows, 256 B x 4 rows, 1024 B x 1 row
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_dma.hip -o tools/bin/probe_dma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int ROWB, int NOUT>
__global__ __launch_bounds__(512) void stream(const unsigned char* __restrict__ src, size_t region, size_t row_pitch,
                                              int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int ROWS = 1024 / ROWB;                      // rows per instruction
  const int lr = lane / (ROWB / 16), lc = lane % (ROWB / 16);
  const unsigned char* base = src + (size_t)blockIdx.x * region;
  // wave w walks rows w*ROWS .. ; one "row" is ROWB contiguous bytes, consecutive rows row_pitch apart
  size_t pos = 0;
  const size_t wave_off = (size_t)(wave * ROWS + lr) * row_pitch + lc * 16;
  const size_t step = (size_t)8 * ROWS * row_pitch;      // all 8 waves together advance this much per instruction round
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + (pos + wave_off) % region),
                                       (lds_ptr_t)(smem + ((it * NOUT + k) % 16) * 8192 + wave * 1024), 16, 0, 0);
      pos += step;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NOUT / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0 && out) out[blockIdx.x] = smem[0];
}

// Same traffic through registers: global_load_dwordx4 -> VGPR -> ds_write_b128.
template <int ROWB, int NOUT>
__global__ __launch_bounds__(512) void stream_vgpr(const unsigned char* __restrict__ src, size_t region, size_t row_pitch,
                                                   int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int ROWS = 1024 / ROWB;
  const int lr = lane / (ROWB / 16), lc = lane % (ROWB / 16);
  const unsigned char* base = src + (size_t)blockIdx.x * region;
  size_t pos = 0;
  const size_t wave_off = (size_t)(wave * ROWS + lr) * row_pitch + lc * 16;
  const size_t step = (size_t)8 * ROWS * row_pitch;
  for (int it = 0; it < iters; ++it) {
    uint4 r[NOUT];
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
      r[k] = *(const uint4*)(base + (pos + wave_off) % region);
      pos += step;
    }
#pragma unroll
    for (int k = 0; k < NOUT; ++k) *(uint4*)(smem + (k % 16) * 8192 + wave * 1024 + lane * 16) = r[k];
  }
  __syncthreads();
  if (threadIdx.x == 0 && out) out[blockIdx.x] = smem[0];
}

template <int ROWB, int NOUT>
static void run_vgpr(const unsigned char* d, size_t region, size_t row_pitch, const char* what) {
  const int iters = 2000 / NOUT * 4;
  hipFuncSetAttribute((const void*)stream_vgpr<ROWB, NOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  stream_vgpr<ROWB, NOUT><<<256, 512, 131072>>>(d, region, row_pitch, iters, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(s);
  stream_vgpr<ROWB, NOUT><<<256, 512, 131072>>>(d, region, row_pitch, iters, nullptr);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const double bytes = 256.0 * 8 * 1024.0 * NOUT * iters;
  printf("VGPR %-23s row %4d B x %2d rows, %2d in flight/wave (%3d KiB/CU): %7.2f TB/s = %5.1f B/clk/CU @2.4GHz\n", what, ROWB,
         1024 / ROWB, NOUT, NOUT * 8, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

template <int ROWB, int NOUT>
static void run(const unsigned char* d, size_t region, size_t row_pitch, const char* what) {
  const int iters = 2000 / NOUT * 4;
  hipFuncSetAttribute((const void*)stream<ROWB, NOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  stream<ROWB, NOUT><<<256, 512, 131072>>>(d, region, row_pitch, iters, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(s);
  stream<ROWB, NOUT><<<256, 512, 131072>>>(d, region, row_pitch, iters, nullptr);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const double bytes = 256.0 * 8 * 1024.0 * NOUT * iters;
  printf("%-28s row %4d B x %2d rows, %2d in flight/wave (%3d KiB/CU): %7.2f TB/s = %5.1f B/clk/CU @2.4GHz\n", what, ROWB,
         1024 / ROWB, NOUT, NOUT * 8, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

// Shared-panel pattern of a GEMM: groups of `share` consecutive workgroups (same XCD via blockIdx & 7 kept equal) stream the
// SAME 64 KiB panel; rot = 1 rotates each workgroup's starting offset inside the panel (stagger-K).
template <int NOUT>
__global__ __launch_bounds__(512) void stream_shared(const unsigned char* __restrict__ src, int share, int rot, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;            // idx-th workgroup of this XCD
  const int panel = xcd * 64 + idx / share;                         // which 64 KiB panel
  const unsigned char* base = src + (size_t)panel * 65536;
  size_t pos = rot ? (size_t)(idx % share) * (65536 / share) : 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + ((pos + wave * 1024 + lane * 16) & 65535)),
                                       (lds_ptr_t)(smem + ((it * NOUT + k) % 16) * 8192 + wave * 1024), 16, 0, 0);
      pos += 8192;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NOUT / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0 && out) out[blockIdx.x] = smem[0];
}

// GEMM operand pattern with cheap addressing: one instruction = ROWS rows x (1024 / ROWS) bytes, rows `pitch` bytes apart; the
// workgroup walks K (columns) of a [512 rows][pitch] panel that stays L2-resident (32 workgroups of an XCD share it).
template <int ROWS, int NOUT>
__global__ __launch_bounds__(512) void stream_rows(const unsigned char* __restrict__ src, int pitch, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int RB = 1024 / ROWS, CPR = RB / 16;                    // row bytes per instruction, 16-byte chunks per row
  const int row = wave * ROWS + lane / CPR, chunk = lane % CPR;
  const unsigned char* p = src + (size_t)(blockIdx.x & 7) * (512 * 3072) + (size_t)row * pitch + chunk * 16;
  int col = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
      // k-th instruction of the step: next block of 8 * ROWS rows, same K columns
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p + (size_t)((k & 7) * 8 * ROWS) * pitch + col),
                                       (lds_ptr_t)(smem + ((it * NOUT + k) % 16) * 8192 + wave * 1024), 16, 0, 0);
    }
    col = (col + RB) & (pitch - 1 >= 2047 ? 2047 : 1023);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NOUT / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0 && out) out[blockIdx.x] = smem[0];
}

template <int ROWS>
static void run_rows(const unsigned char* d, int pitch) {
  constexpr int NOUT = 8;
  const int iters = 4000;
  hipFuncSetAttribute((const void*)stream_rows<ROWS, NOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  stream_rows<ROWS, NOUT><<<256, 512, 131072>>>(d, pitch, iters, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(s);
  stream_rows<ROWS, NOUT><<<256, 512, 131072>>>(d, pitch, iters, nullptr);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const double bytes = 256.0 * 8 * 1024.0 * NOUT * iters;
  printf("L2 hits, GEMM rows: %2d rows x %4d B per instruction, pitch %4d B: %7.2f TB/s = %5.1f B/clk/CU @2.4GHz\n", ROWS,
         1024 / ROWS, pitch, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

static void run_shared(const unsigned char* d, int share, int rot) {
  constexpr int NOUT = 12;
  const int iters = 2000 / NOUT * 4;
  hipFuncSetAttribute((const void*)stream_shared<NOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  stream_shared<NOUT><<<256, 512, 131072>>>(d, share, rot, iters, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(s);
  stream_shared<NOUT><<<256, 512, 131072>>>(d, share, rot, iters, nullptr);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const double bytes = 256.0 * 8 * 1024.0 * NOUT * iters;
  printf("L2 hits, %2d workgroups of an XCD share one 64 KiB panel%s: %7.2f TB/s = %5.1f B/clk/CU @2.4GHz\n", share,
         rot ? ", rotated start" : ", same order   ", bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  unsigned char* d;
  const size_t total = (size_t)256 * (64 << 20);           // 16 GiB: 64 MiB per workgroup
  hipMalloc(&d, total);
  hipMemset(d, 1, total);
  // (a) L2-resident: each workgroup re-reads 64 KiB ; (b) streaming from HBM: 64 MiB per workgroup
  for (int pass = 0; pass < 2; ++pass) {
    const size_t region = pass == 0 ? (64 << 10) : (64 << 20);
    const char* what = pass == 0 ? "L2 hits (64 KiB / WG)" : "HBM stream (64 MiB / WG)";
    run<64, 12>(d, region, 3072, what);
    run<128, 12>(d, region, 3072, what);
    run<256, 12>(d, region, 3072, what);
    run<1024, 12>(d, region, 1024, what);
    run<128, 4>(d, region, 3072, what);
    run<128, 8>(d, region, 3072, what);
    run<128, 16>(d, region, 3072, what);
    run_vgpr<64, 8>(d, region, 3072, what);
    run_vgpr<128, 8>(d, region, 3072, what);
    run_vgpr<1024, 8>(d, region, 1024, what);
    run_vgpr<128, 4>(d, region, 3072, what);
    run_vgpr<128, 16>(d, region, 3072, what);
  }
  run_rows<1>(d, 3072);
  run_rows<4>(d, 3072);
  run_rows<8>(d, 3072);
  run_rows<16>(d, 3072);
  run_rows<8>(d, 2048);
  run_rows<16>(d, 2048);
  for (int share : {1, 8, 32}) {
    run_shared(d, share, 0);
    if (share > 1) run_shared(d, share, 1);
  }
  return 0;
}
