import sys, os
sys.path.insert(0, "/root/repo")
import torch
from inferix_amd import hip_ops as ops
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device="cuda").to(torch.bfloat16)
def timeit(fn, iters=10, inner=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record()
        for _ in range(inner): fn()
        e.record(); e.synchronize(); ts.append(s.elapsed_time(e) / inner)
    ts.sort(); return ts[len(ts)//2] * 1e3
for M in (585, 1170, 2340, 4680):
    q = rnd(M, 12, 128); k = rnd(512, 12, 128); v = rnd(512, 12, 128); out = torch.empty_like(q)
    res = []
    for av, sp in ((1, 1), (5, 1), (6, 1), (6, 2), (6, 4), (5, 4)):
        ops.set_option("attn_variant", av)
        try:
            res.append(f"v{av}/s{sp}:{timeit(lambda: ops.attention(q, ops.KvCacheView(k, v), 512, out=out, splits=sp)):.1f}")
        except Exception as ex:
            res.append(f"v{av}/s{sp}:ERR")
    ops.set_option("attn_variant", 0)
    print(f"cross-attn M={M}: " + "  ".join(res))
