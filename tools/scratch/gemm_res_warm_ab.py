"""Lab: the block's residual-epilogue GEMM launches with / without the residual warm-up (IFX_PP_DEBUG=64 in a fresh process each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inferix_amd import _hip, hip_ops as ops
dev = "cuda"; g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
N, d, f = 4680, 1536, 8960
x, u = rnd(N, d), rnd(N, f)
wo, w2, b1 = rnd(d, d) * 0.02, rnd(d, f) * 0.01, rnd(d)
mod = rnd(3, 6, d)
res = rnd(N, d)
cases = {"o+gate": lambda: ops.linear(x, wo, b1, epilogue=_hip.IFX_EPI_GATE_RES, residual=res, mod=mod, gate_slot=2, rows_per_group=1560),
         "cross-o+res": lambda: ops.linear(x, wo, b1, epilogue=_hip.IFX_EPI_RESIDUAL, residual=res),
         "ffn-down+gate": lambda: ops.linear(u, w2, b1, epilogue=_hip.IFX_EPI_GATE_RES, residual=res, mod=mod, gate_slot=5, rows_per_group=1560)}
junk = rnd(64 * 1024 * 1024)            # 128 MB: evict the residual from the L2s between launches, as a layer's other launches do
outs = {}
for name, fn in cases.items():
    outs[name] = fn().clone()
    ts = []
    for _ in range(15):
        junk.add_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    print(f"{name:14s} {ts[len(ts)//2]:7.1f} us (min {ts[0]:.1f})  checksum {int(outs[name].view(torch.int16).to(torch.int64).sum())}")
