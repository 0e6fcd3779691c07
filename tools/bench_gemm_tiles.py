import sys, torch
sys.path.insert(0, "/root/repo")
from inferix_amd import hip_ops as ops, _hip
dev = "cuda"
import os
if os.environ.get("IFX_SMALL_SPLIT") == "1":
    ops.set_option("gemm_small_split", 1)      # what a sequence-parallel model's forward scopes on
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
M, d, f = (int(sys.argv[1]) if len(sys.argv) > 1 else 4680), 1536, 8960
VARIANTS = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (0, 3, 5, 11)
def timeit(fn, iters=10, inner=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record()
        for _ in range(inner): fn()
        e.record(); e.synchronize(); ts.append(s.elapsed_time(e) / inner)
    ts.sort(); return ts[len(ts) // 2]
x = rnd(M, d); res = rnd(M, d); mod = rnd(3, 6, d)
shapes = [("o/q-cross N=1536 K=1536", x, rnd(d, d) * 0.03, rnd(d), dict(epilogue=_hip.IFX_EPI_RESIDUAL, residual=res)),
          ("q-cross N=1536 K=1536 bias", x, rnd(d, d) * 0.03, rnd(d), dict()),
          ("qkv N=4608 K=1536", x, rnd(3 * d, d) * 0.03, rnd(3 * d), dict()),
          ("ffn0 N=8960 K=1536 gelu", x, rnd(f, d) * 0.03, rnd(f), dict(epilogue=_hip.IFX_EPI_GELU_TANH)),
          ("ffn2 N=1536 K=8960 res", rnd(M, f), rnd(d, f) * 0.01, rnd(d), dict(epilogue=_hip.IFX_EPI_RESIDUAL, residual=res))]
for name, a, w, b, kw in shapes:
    out = torch.empty(M, w.shape[0], dtype=torch.bfloat16, device=dev)
    row = []
    ops.set_option("gemm_variant", 0)
    ref = ops.linear(a, w, b, **kw).float()
    for v in VARIANTS:
        ops.set_option("gemm_variant", v)
        t = timeit(lambda: ops.linear(a, w, b, out=out, **kw)) * 1e3
        bad = "" if torch.equal(out.float(), ref) else f"(!= v0, max diff {float((out.float() - ref).abs().max()):.3g})"
        row.append(f"v{v}: {t:6.1f}{bad}")
    ops.set_option("gemm_variant", 0)
    print(f"M={M}", name, " ".join(row), " (v2 256x128x64, v3 128x128, v4 64x64, v5 256x256x32, v6 128x64, v7 256x128x32 two per CU, v8 128x128x32 8 waves, v9 256x128x64 warp-specialised, v10 128x128x64 warp-specialised two per CU, v11 256x256x64 two stages)")
