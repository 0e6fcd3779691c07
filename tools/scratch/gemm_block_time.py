"""Lab: us per launch of the five block GEMM shapes at 4680 rows (auto tile choice), ~60 ms samples, five interleaved rounds, median."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inferix_amd import hip_ops as ops, _hip
dev = "cuda"; g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
M, d, f = 4680, 1536, 8960
x, res, mod = rnd(M, d), rnd(M, d), rnd(1, 6, d)
shapes = [("o + gate + res 1536x1536", x, rnd(d, d) * 0.03, rnd(d), dict(epilogue=_hip.IFX_EPI_GATE_RES, residual=res, mod=mod, gate_slot=2, rows_per_group=M)),
          ("cross-q 1536x1536 bias", x, rnd(d, d) * 0.03, rnd(d), dict()),
          ("qkv 4608x1536", x, rnd(3 * d, d) * 0.03, rnd(3 * d), dict()),
          ("ffn up 8960x1536 gelu", x, rnd(f, d) * 0.03, rnd(f), dict(epilogue=_hip.IFX_EPI_GELU_TANH)),
          ("ffn down 1536x8960 res", rnd(M, f), rnd(d, f) * 0.01, rnd(d), dict(epilogue=_hip.IFX_EPI_RESIDUAL, residual=res))]
outs = [torch.empty(M, w.shape[0], dtype=torch.bfloat16, device=dev) for _, _, w, _, _ in shapes]
def timed(i, n):
    name, a, w, b, kw = shapes[i]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.linear(a, w, b, out=outs[i], **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
ns = [1500, 1500, 800, 400, 450]
try:
    for i in range(len(shapes)): timed(i, 20)
except Exception as e:
    print("skip gate shape:", e); shapes = shapes[1:]; outs = outs[1:]; ns = ns[1:]
ts = [[] for _ in shapes]
for rnd_i in range(5):
    for i in range(len(shapes)): ts[i].append(timed(i, ns[i]))
tot = 0.0
for i, (name, *_r) in enumerate(shapes):
    us = sorted(ts[i])[2]; tot += us
    print(f"{name:28s} {us:8.2f} us")
print(f"{'sum':28s} {tot:8.2f} us")
