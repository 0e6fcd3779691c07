"""Per-K-step and fixed cost of a GEMM tile variant: time vs K at fixed (M, N).  usage: bench_gemm_k.py VARIANTS [M] [N...]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from inferix_amd import hip_ops as ops
variants = [int(v) for v in sys.argv[1].split(",")]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4680
Ns = [int(n) for n in sys.argv[3:]] or [1536, 4608, 8960]
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device="cuda").to(torch.bfloat16)
def timeit(fn, iters=7, inner=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record()
        for _ in range(inner): fn()
        e.record(); e.synchronize(); ts.append(s.elapsed_time(e) / inner)
    ts.sort(); return ts[len(ts) // 2] * 1e3
for N in Ns:
    for v in variants:
        ops.set_option("gemm_variant", v)
        row = []
        for K in (512, 1536, 3072, 6144, 8960):
            a, w = rnd(M, K), rnd(N, K) * 0.02
            out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            t = timeit(lambda: ops.linear(a, w, None, out=out))
            row.append(f"K={K}: {t:7.1f} us ({2.0 * M * N * K / t / 1e6:6.0f} TF/s)")
        print(f"M={M} N={N} v{v}: " + "  ".join(row))
ops.set_option("gemm_variant", 0)
