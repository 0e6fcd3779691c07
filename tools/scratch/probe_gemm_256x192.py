import sys, torch
sys.path.insert(0, "/root/repo")
from inferix_amd import _hip, hip_ops as ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (M, N, K) in [(4680, 4608, 1536), (4680, 8960, 1536), (4680, 1536, 1536), (10800, 4608, 1536), (4680, 576, 1536), (1000, 200, 128)]:
    x, w, b = rnd(M, K), rnd(N, K) * 0.03, rnd(N)
    ref = (x.float() @ w.float().T + b.float())
    out = {}
    for v in (0, 21, 5):
        ops.set_option("gemm_variant", v)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.linear(x, w, b, out=y)
        err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
        t = timeit(lambda: ops.linear(x, w, b, out=y))
        out[v] = y.clone()
        print(f"M={M} N={N} K={K} variant {v:2d}: {t:8.1f} us  {2.0*M*N*K/t/1e6:7.1f} TFLOP/s  rel err {err:.2e}  eq_auto {torch.equal(out[0], y)}")
    # residual/gate epilogue on the new tile
    if N == 1536 or N == 4608:
        res = rnd(M, N); mod = rnd(3, 6, N)
        ops.set_option("gemm_variant", 21)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.linear(x, w, b, out=y, epilogue=_hip.IFX_EPI_GATE_RES, residual=res, mod=mod, gate_slot=2, rows_per_group=(M + 2) // 3)
        ops.set_option("gemm_variant", 5)
        y2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.linear(x, w, b, out=y2, epilogue=_hip.IFX_EPI_GATE_RES, residual=res, mod=mod, gate_slot=2, rows_per_group=(M + 2) // 3)
        print("  gate epilogue equal to 256x256 tile:", torch.equal(y, y2))
ops.set_option("gemm_variant", 0)
