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
for ps in (0, 1560, 64, 4680):
    if ps:
        pages = L // ps
        tbl = torch.roll(torch.arange(pages, dtype=torch.int32), 3).to(dev)
        view = ops.KvCacheView(k, v, tbl, ps)
    else:
        view = ops.KvCacheView(k, v)
    o = ops.attention(q, view, L, scale=asc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.attention(q, view, L, scale=asc)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"page_size {ps or 'contiguous':>10}: {us:8.1f} us  {4.0*N*L*H*D/us*1e-6:7.1f} TFLOP/s  splits plan {ops.attention_split_plan(N,H,L)}")
