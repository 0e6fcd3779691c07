"""Per-CU rate of the self-attention kernel against the number of busy CUs: M x H chosen so that q_tiles * H = 64 .. 256
workgroups of one 256-row tile each.  usage: tools/probe_attn_occupancy.py [M H]  (one configuration: for rocprofv3 --pmc GRBM_GUI_ACTIVE)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inferix_amd import hip_ops as ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2] * 1e6
L, D = 32760, 128
CFG = ((int(sys.argv[1]), int(sys.argv[2])),) if len(sys.argv) > 2 else ((4680, 12), (4680, 12), (4680, 13), (4096, 16), (4096, 12), (4096, 8), (4096, 4))
for (M, H) in CFG:
    q = torch.randn(M, H, D, device=dev).to(torch.bfloat16)
    k = torch.randn(L, H, D, device=dev).to(torch.bfloat16)
    v = torch.randn(L, H, D, device=dev).to(torch.bfloat16)
    o = torch.empty_like(q)
    view = ops.KvCacheView(k, v)
    t = timeit(lambda: ops.attention(q, view, L, out=o, splits=1))
    tiles = ((M + 255) // 256) * H
    print(f"M={M} H={H} tiles={tiles} {t:8.1f} us  {4.0*M*L*H*D/t/1e6:7.1f} TFLOP/s  per-tile-rate {4.0*256*L*D*tiles/t/1e6/tiles:6.2f} TF/s/CU", flush=True)
