#!/usr/bin/env python3
"""Reference point only (NOT used by the product): what the vendor GEMM (hipBLASLt via torch) reaches on the
block's GEMM shapes, to size the headroom of the hand-written kernels."""
import torch
import torch.nn.functional as F
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
M, d, f = 4680, 1536, 8960
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); return ts[len(ts)//2]
for name, K, N in (("qkv", d, 3*d), ("o", d, d), ("ffn0", d, f), ("ffn2", f, d)):
    x, w, b = rnd(M, K), rnd(N, K) * 0.03, rnd(N)
    t = timeit(lambda: F.linear(x, w, b))
    print(f"hipBLASLt {name:5s} M={M} N={N} K={K}: {t*1e3:7.1f} us  {2.0*M*N*K/t/1e9:7.1f} TFLOP/s")
