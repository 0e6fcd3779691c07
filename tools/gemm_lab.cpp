// GEMM laboratory: the block's linear-layer shapes through ifx_gemm_bf16 (C ABI of libinferix_hip.so) under several
// "gemm_variant" settings, in ONE process without torch: bit comparison against the first variant, then interleaved timing rounds
// (GEMM noise is +-3 %: only within-process interleaved numbers are comparable).
//
//   build: hipcc --offload-arch=gfx950 -O2 tools/gemm_lab.cpp -o tools/bin/gemm_lab -ldl
//   run  : tools/bin/gemm_lab [-l lib.so] [-r rounds] [-i inner] [-t] VARIANTS SHAPE [SHAPE ...]
//          VARIANTS = comma list of gemm_variant values (first = reference for the bit comparison)
//          SHAPE    = M,N,K[,epi]   epi: 0 bias, 1 gelu-tanh, 2 residual, 3 gate+residual     or a name: qkv o cq co up down block
//          -t       : read back the s_memtime stamps of ifx_gemm_pp.hip (library built with -DIFX_PP_TRACE=1)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../include/inferix_hip.h"

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                        \
    }                                                                                 \
  } while (0)

typedef int (*gemm_fn)(const ifx_bf16*, int32_t, const ifx_bf16*, const ifx_bf16*, ifx_bf16*, int32_t, int32_t, int32_t, int32_t,
                       const ifx_epilogue*, void*);
typedef int (*gemm_ws_fn)(const ifx_bf16*, int32_t, const ifx_bf16*, const ifx_bf16*, ifx_bf16*, int32_t, int32_t, int32_t, int32_t,
                          const ifx_epilogue*, void*, int64_t, void*);
typedef int64_t (*ws_bytes_fn)(int32_t, int32_t, int32_t);
typedef int (*opt_fn)(const char*, int32_t);
typedef const char* (*err_fn)(void);

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rnd32() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 32);
}
static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// sum of 12 uniforms - 6: close enough to a normal for operand statistics (and power draw) comparable to torch.randn
static void fill(std::vector<uint16_t>& v, float scale) {
  for (auto& e : v) {
    float s = 0.f;
    for (int k = 0; k < 4; ++k) s += (float)(rnd32() & 0xffffff) * (1.0f / 16777216.0f);
    e = f2bf((s - 2.0f) * 1.7320508f * scale);
  }
}
static uint16_t* to_dev(const std::vector<uint16_t>& h) {
  uint16_t* d;
  CK(hipMalloc(&d, h.size() * 2));
  CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  return d;
}

struct Shape { int M, N, K, epi; std::string name; };

int main(int argc, char** argv) {
  const char* libpath = "inferix_amd/libinferix_hip.so";
  int rounds = 7, inner = 5;
  bool want_trace = false;
  int a = 1;
  for (; a < argc && argv[a][0] == '-'; ++a) {
    if (!strcmp(argv[a], "-l")) libpath = argv[++a];
    else if (!strcmp(argv[a], "-r")) rounds = atoi(argv[++a]);
    else if (!strcmp(argv[a], "-i")) inner = atoi(argv[++a]);
    else if (!strcmp(argv[a], "-t")) want_trace = true;
  }
  if (argc - a < 2) {
    fprintf(stderr, "usage: gemm_lab [-l lib] [-r rounds] [-i inner] [-t] VARIANTS SHAPE...\n");
    return 1;
  }
  std::vector<int> variants;
  for (char* t = strtok(argv[a], ","); t; t = strtok(nullptr, ",")) variants.push_back(atoi(t));
  ++a;
  std::vector<Shape> shapes;
  auto add = [&](const char* nm) -> bool {
    const int M = 4680, d = 1536, f = 8960;
    if (!strcmp(nm, "qkv")) shapes.push_back({M, 3 * d, d, 0, "qkv"});
    else if (!strcmp(nm, "o")) shapes.push_back({M, d, d, 3, "o+gate"});
    else if (!strcmp(nm, "cq")) shapes.push_back({M, d, d, 0, "cross-q"});
    else if (!strcmp(nm, "co")) shapes.push_back({M, d, d, 2, "cross-o+res"});
    else if (!strcmp(nm, "up")) shapes.push_back({M, f, d, 1, "ffn-up+gelu"});
    else if (!strcmp(nm, "down")) shapes.push_back({M, d, f, 3, "ffn-down+gate"});
    else return false;
    return true;
  };
  for (; a < argc; ++a) {
    if (!strcmp(argv[a], "block")) {
      for (const char* nm : {"qkv", "o", "cq", "co", "up", "down"}) add(nm);
    } else if (!add(argv[a])) {
      Shape s{0, 0, 0, 0, argv[a]};
      if (sscanf(argv[a], "%d,%d,%d,%d", &s.M, &s.N, &s.K, &s.epi) < 3) {
        fprintf(stderr, "bad shape %s\n", argv[a]);
        return 1;
      }
      shapes.push_back(s);
    }
  }
  void* lib = dlopen(libpath, RTLD_NOW | RTLD_GLOBAL);
  if (!lib) {
    fprintf(stderr, "dlopen %s: %s\n", libpath, dlerror());
    return 2;
  }
  gemm_fn gemm = (gemm_fn)dlsym(lib, "ifx_gemm_bf16");
  opt_fn set_opt = (opt_fn)dlsym(lib, "ifx_set_option");
  err_fn last_err = (err_fn)dlsym(lib, "ifx_last_error");
  gemm_ws_fn gemm_ws = (gemm_ws_fn)dlsym(lib, "ifx_gemm_bf16_ws");
  ws_bytes_fn ws_bytes = (ws_bytes_fn)dlsym(lib, "ifx_gemm_workspace_bytes");
  if (!gemm || !set_opt || !last_err || !gemm_ws || !ws_bytes) return 2;
  // one zero-initialised workspace for every launch (as inferix_amd.hip_ops.linear keeps one per stream): 64 MiB covers the block
  const int64_t ws_cap = 64ll << 20;
  void* ws = nullptr;
  CK(hipMalloc(&ws, ws_cap));
  CK(hipMemset(ws, 0, ws_cap));

  unsigned long long* trace = nullptr;
  if (want_trace) {
    CK(hipMalloc(&trace, 16 * 8));
    CK(hipMemset(trace, 0, 16 * 8));
    char buf[64];
    snprintf(buf, sizeof buf, "%llu", (unsigned long long)(uintptr_t)trace);
    setenv("IFX_PP_TRACE_PTR", buf, 1);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K, groups = (M + 1559) / 1560;
    std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K), hb(N), hr((size_t)M * N), hm((size_t)groups * 6 * N);
    fill(hx, 1.0f);
    fill(hw, 0.02f);
    fill(hb, 0.1f);
    fill(hr, 1.0f);
    fill(hm, 0.5f);
    uint16_t *dx = to_dev(hx), *dw = to_dev(hw), *db = to_dev(hb), *dr = to_dev(hr), *dm = to_dev(hm);
    std::vector<uint16_t*> dy(variants.size());
    for (auto& p : dy) {
      CK(hipMalloc(&p, (size_t)M * N * 2));
      CK(hipMemset(p, 0xff, (size_t)M * N * 2));
    }
    ifx_epilogue epi;
    memset(&epi, 0, sizeof epi);
    epi.epilogue = sh.epi;
    epi.residual = dr;
    epi.ld_res = N;
    epi.mod = dm;
    epi.mod_slots = 6;
    epi.gate_slot = 2;
    epi.rows_per_group = 1560;
    auto run = [&](int vi) {
      set_opt("gemm_variant", variants[vi]);
      const int64_t need = ws_bytes(M, N, K);        // what the library asks for under this variant (0: the plain entry point)
      const int rc = (need > 0 && need <= ws_cap) ? gemm_ws(dx, K, dw, db, dy[vi], N, M, N, K, &epi, ws, ws_cap, nullptr)
                                                  : gemm(dx, K, dw, db, dy[vi], N, M, N, K, &epi, nullptr);
      if (rc != 0) {
        fprintf(stderr, "variant %d on %s: rc %d: %s\n", variants[vi], sh.name.c_str(), rc, last_err());
        exit(3);
      }
    };
    // ---- correctness: every variant against the first
    std::vector<uint16_t> ref((size_t)M * N), got((size_t)M * N);
    std::vector<std::string> verdicts(variants.size());
    for (size_t vi = 0; vi < variants.size(); ++vi) {
      run((int)vi);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(vi == 0 ? ref.data() : got.data(), dy[vi], (size_t)M * N * 2, hipMemcpyDeviceToHost));
      if (vi == 0) {
        verdicts[0] = "ref";
        continue;
      }
      size_t bad = 0, first_bad = 0;
      double maxd = 0;
      for (size_t i = 0; i < ref.size(); ++i)
        if (ref[i] != got[i]) {
          if (!bad) first_bad = i;
          ++bad;
          maxd = std::max(maxd, (double)fabsf(bf2f(ref[i]) - bf2f(got[i])));
        }
      char buf[160];
      if (!bad) snprintf(buf, sizeof buf, "bit-identical");
      else
        snprintf(buf, sizeof buf, "MISMATCH %zu of %zu (first at row %zu col %zu: %g vs %g, max |d| %g)", bad, ref.size(),
                 first_bad / N, first_bad % N, bf2f(ref[first_bad]), bf2f(got[first_bad]), maxd);
      verdicts[vi] = buf;
    }
    // ---- timing: interleaved rounds
    std::vector<std::vector<float>> us(variants.size());
    for (int r = 0; r < rounds + 1; ++r)
      for (size_t vi = 0; vi < variants.size(); ++vi) {
        CK(hipEventRecord(e0, nullptr));
        for (int k = 0; k < inner; ++k) run((int)vi);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) us[vi].push_back(ms * 1000.f / inner);
      }
    printf("%-14s %5d x %5d x %5d epi %d\n", sh.name.c_str(), M, N, K, sh.epi);
    for (size_t vi = 0; vi < variants.size(); ++vi) {
      std::sort(us[vi].begin(), us[vi].end());
      const float med = us[vi][us[vi].size() / 2], mn = us[vi][0];
      printf("   v%-3d  median %8.1f us  min %8.1f us  %6.0f TF/s  (%.3f of 2.5 PF)   %s\n", variants[vi], med, mn,
             2.0 * M * N * K / med / 1e6, 2.0 * M * N * K / med / 1e6 / 2500.0, verdicts[vi].c_str());
    }
    fflush(stdout);
    for (auto p : dy) CK(hipFree(p));
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dr)); CK(hipFree(dm));
  }
  if (want_trace) {
    std::vector<unsigned long long> t(16);
    CK(hipMemcpy(t.data(), trace, 16 * 8, hipMemcpyDeviceToHost));
    const char* names[6] = {"barrier behind mfma", "dma issue", "epilogue", "frag reads + waits", "barrier behind loader", "mfma + wait"};
    for (int g = 0; g < 2; ++g) {
      const double G = (double)t[g * 8 + 6];
      printf("trace group %d (workgroup 0, last launch), %g K-steps, mean cycles per K-step:\n", g, G);
      double tot = 0;
      for (int k = 0; k < 6; ++k) tot += (double)t[g * 8 + k];
      for (int k = 1; k <= 6; ++k) printf("   %-22s %8.0f\n", names[k % 6], (double)t[g * 8 + (k % 6)] / G);
      printf("   %-22s %8.0f\n", "total", tot / G);
    }
  }
  return 0;
}
