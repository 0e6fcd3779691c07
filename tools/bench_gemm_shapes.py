"""GEMM tile variants on explicit (M, N, K) shapes.  usage: bench_gemm_shapes.py VARIANTS M,N,K [M,N,K ...]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from inferix_amd import hip_ops as ops
variants = [int(v) for v in sys.argv[1].split(",")]
shapes = [tuple(int(t) for t in a.split(",")) for a in sys.argv[2:]]
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device="cuda").to(torch.bfloat16)
def timeit(fn, iters=7, inner=5):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record()
        for _ in range(inner): fn()
        e.record(); e.synchronize(); ts.append(s.elapsed_time(e) / inner)
    ts.sort(); return ts[len(ts) // 2] * 1e3
for M, N, K in shapes:
    a, w = rnd(M, K), rnd(N, K) * 0.02
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    row = []
    for v in variants:
        ops.set_option("gemm_variant", v)
        t = timeit(lambda: ops.linear(a, w, None, out=out))
        row.append(f"v{v}: {t:7.1f} us ({2.0 * M * N * K / t / 1e6:5.0f} TF/s)")
    print(f"{M}x{N}x{K}: " + "  ".join(row))
ops.set_option("gemm_variant", 0)
