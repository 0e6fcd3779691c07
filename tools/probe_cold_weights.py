"""GPU: what cold weights cost a block GEMM.  Each shape is launched back to back (a) with ONE weight matrix (what the lab timings do: it
stays in the memory-side cache), (b) cycling through enough different matrices to exceed the 256 MB cache (what a clip does: 2.5 GB of
weights per forward), (c) as (b) with a cheap read pass over the NEXT matrix on a second stream while the current launch runs."""
import sys, torch
sys.path.insert(0, ".")
from inferix_amd import hip_ops as ops, _hip
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
M, d, f = 4680, 1536, 8960
def med(fn, n, outer=7):
    for _ in range(2): fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(outer):
        e0.record()
        for i in range(n): fn(i)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1000)
    return sorted(ts)[len(ts) // 2]
side = torch.cuda.Stream()
for name, N, K, kw in (("o+res 1536x1536", d, d, "res"), ("qkv 4608x1536", 3 * d, d, ""), ("ffn up 8960x1536 gelu", f, d, "gelu"), ("ffn down 1536x8960 res", d, f, "res")):
    nw = max(2, int(600e6 / (N * K * 2)))
    ws = [rnd(N, K) * 0.03 for _ in range(nw)]
    b = rnd(N)
    xs = [rnd(M, K) for _ in range(4)]
    res = rnd(M, N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    k = dict(epilogue=_hip.IFX_EPI_RESIDUAL, residual=res) if kw == "res" else dict(epilogue=_hip.IFX_EPI_GELU_TANH) if kw == "gelu" else {}
    warm = med(lambda i: ops.linear(xs[0], ws[0], b, out=out, **k), 32)
    cold = med(lambda i: ops.linear(xs[i % 4], ws[i % nw], b, out=out, **k), nw)
    def pre(i):
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            ws[(i + 1) % nw].view(torch.int32).sum()          # a read pass over the next matrix
        ops.linear(xs[i % 4], ws[i % nw], b, out=out, **k)
    pref = med(pre, nw)
    print(f"{name}: one matrix {warm:6.1f} us, {nw} matrices in turn {cold:6.1f} us, with the next one read on a side stream {pref:6.1f} us")
