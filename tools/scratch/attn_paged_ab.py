"""Lab: self-attention over a saturated 32760-key cache, contiguous vs paged (one-frame pages, rotated table) — us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inferix_amd import hip_ops as ops
dev = "cuda"; g = torch.Generator(device=dev).manual_seed(0)
N, H, D, L = 4680, 12, 128, 32760
qs, asc = ops.attn_q_prescale(D)
q = (torch.randn(N, H, D, generator=g, device=dev) * qs).to(torch.bfloat16)
k = torch.randn(L, H, D, generator=g, device=dev).to(torch.bfloat16); v = torch.randn(L, H, D, generator=g, device=dev).to(torch.bfloat16)
def paged_view(ps):
    pages = -(-k.shape[0] // ps)
    tbl = torch.roll(torch.arange(pages, dtype=torch.int32), 3).to(dev)
    # physical cache holding the SAME logical rows: physical page tbl[p] = logical page p
    kp = torch.zeros(pages * ps, H, D, dtype=torch.bfloat16, device=dev); vp = torch.zeros_like(kp)
    kl = torch.zeros(pages * ps, H, D, dtype=torch.bfloat16, device=dev); kl[:k.shape[0]] = k
    vl = torch.zeros_like(kl); vl[:v.shape[0]] = v
    kp.view(pages, ps, H, D)[tbl.long()] = kl.view(pages, ps, H, D)
    vp.view(pages, ps, H, D)[tbl.long()] = vl.view(pages, ps, H, D)
    return ops.KvCacheView(kp, vp, tbl, ps)
PS = (0, 1560, 64, 4680, 130, 1)
views = {ps: (paged_view(ps) if ps else ops.KvCacheView(k, v)) for ps in PS}
def timed(view, L, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.attention(q, view, L, scale=asc)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for L in (32760, 18720, 4680 + 1560 * 3 + 17):
    ref = ops.attention(q, views[0], L, scale=asc).clone()
    same = {ps: torch.equal(ops.attention(q, views[ps], L, scale=asc), ref) for ps in PS}
    n = max(20, int(60000 / (L / 32760 * 930)))          # ~60 ms per sample, interleaved, at the sustained clock
    for ps in PS: timed(views[ps], L, n)
    ts = {ps: [] for ps in PS}
    for rnd in range(5):
        for ps in PS: ts[ps].append(timed(views[ps], L, n))
    for ps in PS:
        us = sorted(ts[ps])[2]
        print(f"L {L:6d} page_size {ps or 'contiguous':>10}: {us:8.1f} us  {4.0*N*L*H*D/us*1e-6:7.1f} TFLOP/s  {'== contiguous' if same[ps] else '!= contiguous'}")
