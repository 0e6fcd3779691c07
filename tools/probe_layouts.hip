// GPU probe: verifies the gfx950 hardware facts the attention / GEMM kernels rely on
//   (1) v_mfma_f32_32x32x16_bf16 C/D layout and A-row / B-col lane mapping
//   (2) v_mfma_f32_16x16x32_bf16 C/D layout
//   (3) ds_read_b64_tr_b16 semantics, with the swizzled V-tile addressing of ifx_attn.hip
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_layouts.hip -o tools/bin/probe_layouts
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe(float* o32a, float* o32b, float* o16a, float* o16b, short* otr_lin, short* otr_v, unsigned* osw) {
  const int lane = threadIdx.x;
  {
    // (4) v_permlane32_swap: vdst = lane, src = 100 + lane (distinct), and the x,x form used for a half-wave max
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)lane, (unsigned)(100 + lane), false, false);
    osw[lane * 4 + 0] = r[0]; osw[lane * 4 + 1] = r[1];
    unsigned u = 1000u + lane;
    auto q = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    osw[lane * 4 + 2] = q[0]; osw[lane * 4 + 3] = q[1];
  }
  __shared__ __attribute__((aligned(16))) short lds[64 * 128];
  // ---- (1) 32x32x16: A[i][k] = i for all k ; B[k][n] = (k == 0 slot: lane>>5==0 && j==0) ? 1 : 0
  {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(lane & 31); b[j] = (__bf16)((lane >> 5) == 0 && j == 0 ? 1.f : 0.f); }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) o32a[lane * 16 + r] = c[r];           // expect row index i
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)((lane >> 5) == 1 && j == 3 ? 1.f : 0.f); b[j] = (__bf16)(float)(lane & 31); }
    f32x16 d = {0};
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 16; ++r) o32b[lane * 16 + r] = d[r];           // expect col index n = lane&31
  }
  // ---- (2) 16x16x32
  {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(lane & 15); b[j] = (__bf16)((lane >> 4) == 2 && j == 5 ? 1.f : 0.f); }
    f32x4 c = {0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) o16a[lane * 4 + r] = c[r];             // expect row i = 4*(lane>>4)+r
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)((lane >> 4) == 1 && j == 0 ? 1.f : 0.f); b[j] = (__bf16)(float)(lane & 15); }
    f32x4 d = {0};
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) o16b[lane * 4 + r] = d[r];             // expect col n = lane&15
  }
  // ---- (3a) tr16_b64 with linear per-lane addresses (8*lane bytes), lds[e] = e
  for (int e = lane; e < 64 * 128; e += 64) lds[e] = (short)e;
  __syncthreads();
  {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + 4 * lane));
    for (int j = 0; j < 4; ++j) otr_lin[lane * 4 + j] = t[j];           // guide: (l&15) + 16*j + 64*(l>>4)
  }
  __syncthreads();
  // ---- (3b) V tile [64 keys][128 d], value = key*128 + d, stored with the attention kernel's swizzle
  for (int e = lane; e < 64 * 128; e += 64) {
    const int key = e >> 7, d = e & 127;
    const int colb = d * 2;
    const int phys = key * 256 + ((((colb >> 6) ^ (key & 3)) << 6) | (colb & 63));
    lds[phys >> 1] = (short)e;
  }
  __syncthreads();
  {
    const int hi = lane >> 5, vi = lane & 15, vg1 = (lane >> 4) & 1;
    const int v_rowq = vi >> 2, v_in = (vg1 << 5) | ((vi & 3) << 3);
    int n = 0;
    for (int b = 0; b < 2; ++b) for (int s2 = 0; s2 < 2; ++s2) {
      const int key0 = 32 * b + 16 * s2 + 4 * hi;
      for (int d = 0; d < 4; ++d) for (int h = 0; h < 2; ++h) {
        const unsigned char* p = (const unsigned char*)lds + (key0 + 8 * h + v_rowq) * 256 + v_in + ((d ^ v_rowq) << 6);
        s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
        for (int j = 0; j < 4; ++j) otr_v[(n * 64 + lane) * 4 + j] = t[j];   // expect (key0+8h+j)*128 + 32d + (lane&31)
        ++n;
      }
    }
  }
}

int main() {
  float *a, *b, *c, *d; short *e, *f; unsigned* gsw; hipMalloc(&gsw, 64 * 4 * 4);
  hipMalloc(&a, 64 * 16 * 4); hipMalloc(&b, 64 * 16 * 4); hipMalloc(&c, 64 * 4 * 4); hipMalloc(&d, 64 * 4 * 4);
  hipMalloc(&e, 64 * 4 * 2); hipMalloc(&f, 32 * 64 * 4 * 2);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, c, d, e, f, gsw);
  if (hipDeviceSynchronize() != hipSuccess) { printf("probe: kernel failed\n"); return 2; }
  std::vector<float> ha(1024), hb(1024), hc(256), hd(256); std::vector<short> he(256), hf(32 * 256);
  hipMemcpy(ha.data(), a, 4096, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, 4096, hipMemcpyDeviceToHost);
  hipMemcpy(hc.data(), c, 1024, hipMemcpyDeviceToHost); hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
  hipMemcpy(he.data(), e, 512, hipMemcpyDeviceToHost); hipMemcpy(hf.data(), f, 32 * 512, hipMemcpyDeviceToHost);
  int bad = 0, fails = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    if (ha[l * 16 + r] != (float)((r & 3) + 8 * (r >> 2) + 4 * (l >> 5))) bad++;
  }
  printf("mfma32 C rows  (r&3)+8*(r>>2)+4*(lane>>5): %s (%d bad)\n", bad ? "FAIL" : "ok", bad); fails += bad != 0;
  if (bad) { printf("  lane0:"); for (int r = 0; r < 16; ++r) printf(" %g", ha[r]); printf("\n  lane32:"); for (int r = 0; r < 16; ++r) printf(" %g", ha[32 * 16 + r]); printf("\n"); }
  bad = 0; for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) if (hb[l * 16 + r] != (float)(l & 31)) bad++;
  printf("mfma32 C col = lane&31 / B n = lane&31: %s (%d bad)\n", bad ? "FAIL" : "ok", bad); fails += bad != 0;
  bad = 0; for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hc[l * 4 + r] != (float)(4 * (l >> 4) + r)) bad++;
  printf("mfma16 C rows 4*(lane>>4)+r: %s (%d bad)\n", bad ? "FAIL" : "ok", bad); fails += bad != 0;
  if (bad) { for (int l = 0; l < 64; l += 16) { printf("  lane%d:", l); for (int r = 0; r < 4; ++r) printf(" %g", hc[l * 4 + r]); printf("\n"); } }
  bad = 0; for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hd[l * 4 + r] != (float)(l & 15)) bad++;
  printf("mfma16 C col = lane&15: %s (%d bad)\n", bad ? "FAIL" : "ok", bad); fails += bad != 0;
  bad = 0; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (he[l * 4 + j] != (short)((l & 15) + 16 * j + 64 * (l >> 4))) bad++;
  printf("ds_read_tr16_b64 linear: %s (%d bad)\n", bad ? "FAIL" : "ok", bad); fails += bad != 0;
  if (bad) { for (int l = 0; l < 20; ++l) { printf("  lane%d:", l); for (int j = 0; j < 4; ++j) printf(" %d", he[l * 4 + j]); printf("\n"); } }
  bad = 0; int n = 0;
  for (int bb = 0; bb < 2; ++bb) for (int s2 = 0; s2 < 2; ++s2) for (int dd = 0; dd < 4; ++dd) for (int h = 0; h < 2; ++h) {
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      const int key0 = 32 * bb + 16 * s2 + 4 * (l >> 5);
      const int want = (key0 + 8 * h + j) * 128 + 32 * dd + (l & 31);
      if (hf[(n * 64 + l) * 4 + j] != (short)want) { if (bad < 8) printf("  n%d lane%d j%d got %d want %d\n", n, l, j, hf[(n * 64 + l) * 4 + j], want); bad++; }
    }
    ++n;
  }
  printf("ds_read_tr16_b64 V-tile addressing: %s (%d bad)\n", bad ? "FAIL" : "ok", bad); fails += bad != 0;
  { std::vector<unsigned> hs(256); hipMemcpy(hs.data(), gsw, 1024, hipMemcpyDeviceToHost);
    printf("permlane32_swap(vdst=lane, src=100+lane): lane0 -> (%u,%u) lane1 -> (%u,%u) lane32 -> (%u,%u) lane33 -> (%u,%u)\n", hs[0], hs[1], hs[4], hs[5], hs[128], hs[129], hs[132], hs[133]);
    printf("permlane32_swap(x,x) x=1000+lane: lane0 -> (%u,%u) lane32 -> (%u,%u)\n", hs[2], hs[3], hs[130], hs[131]); }
  printf("PROBE %s\n", fails ? "FAILED" : "PASSED");
  return fails ? 1 : 0;
}
