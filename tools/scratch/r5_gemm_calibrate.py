"""Calibration of the ping-pong GEMM tile against the guide's 256^2 8-phase figure (1320-1340 TFLOP/s at 4096^3, random operands):
the same kernel at 4096^3 / 8192^3 and at the block's shapes, uniform random [-1, 1) operands."""
import sys, torch
sys.path.insert(0, "/root/repo")
from inferix_amd import hip_ops as ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def rnd(*s): return (torch.rand(*s, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
def timeit(fn, iters=7, inner=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record()
        for _ in range(inner): fn()
        e.record(); e.synchronize(); ts.append(s.elapsed_time(e) / inner)
    ts.sort(); return ts[len(ts) // 2]
for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 8192), (4608, 4608, 1536), (4608, 8960, 1536), (4608, 1536, 8960), (4680, 4608, 1536), (4680, 8960, 1536), (4680, 1536, 8960), (4680, 1536, 1536), (4608, 1536, 1536), (4608, 1536, 4096)):
    x, w, b = rnd(M, K), rnd(N, K), rnd(N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    row = []
    for v in (0, 22, 23, 24):
        ops.set_option("gemm_variant", v)
        try:
            t = timeit(lambda: ops.linear(x, w, b, out=out))
            row.append(f"v{v}: {t*1e3:7.1f} us {2.0*M*N*K/t/1e9:7.1f} TF")
        except Exception as e:
            row.append(f"v{v}: n/a")
    ops.set_option("gemm_variant", 0)
    print(f"M={M} N={N} K={K}  " + "  ".join(row), flush=True)
