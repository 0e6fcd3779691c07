import sys, torch
sys.path.insert(0, "/root/repo")
from inferix_amd import hip_ops as ops, _hip
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device="cuda").to(torch.bfloat16)
M, N, K = 4680, 1536, 8960
a, w, b, res = rnd(M, K), rnd(N, K) * 0.01, rnd(N), rnd(M, N)
mod = rnd(3, 6, N)
def timeit(fn, iters=7, inner=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record()
        for _ in range(inner): fn()
        e.record(); e.synchronize(); ts.append(s.elapsed_time(e) / inner)
    ts.sort(); return ts[len(ts) // 2] * 1e3
kw = dict(epilogue=_hip.IFX_EPI_GATE_RES, residual=res, mod=mod, gate_slot=5, rows_per_group=1560)
ops.set_option("gemm_variant", 3)
ref = ops.linear(a, w, b, **kw).float()
t3 = timeit(lambda: ops.linear(a, w, b, **kw))
ops.set_option("gemm_variant", 20)
out = ops.linear(a, w, b, **kw).float()
outs = [ops.linear(a, w, b, **kw) for _ in range(20)]
t0 = timeit(lambda: ops.linear(a, w, b, **kw))
exact = (torch.nn.functional.linear(a.float(), w.float(), b.float()))
print("v3 %.1f us, auto(split-K w4) %.1f us; max |auto - v3| %.4g; rel-L2 vs v3 %.3e; deterministic %s" % (
    t3, t0, float((out - ref).abs().max()), float((out - ref).norm() / ref.norm()), all(torch.equal(o, outs[0]) for o in outs)))
