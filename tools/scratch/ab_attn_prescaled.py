import math, sys, torch
sys.path.insert(0, "/root/repo")
from inferix_amd import hip_ops as ops
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
H, D, M, L = 12, 128, 4680, int(sys.argv[1]) if len(sys.argv) > 1 else 18720
qs, sc = ops.attn_q_prescale(D)
q = (torch.randn(M, H, D, generator=g, device=dev) * qs).to(torch.bfloat16)
k = torch.randn(L, H, D, generator=g, device=dev).to(torch.bfloat16); v = torch.randn(L, H, D, generator=g, device=dev).to(torch.bfloat16)
kv = ops.KvCacheView(k, v); out = torch.empty_like(q)
ts = []
for r in range(6):
    for _ in range(3): ops.attention(q, kv, L, scale=sc, out=out)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.attention(q, kv, L, scale=sc, out=out)
    e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 20 * 1e3)
ts.sort(); t = ts[len(ts) // 2]
print(f"L={L}: median {t:.1f} us  {4.0*M*L*H*D/t/1e6:.0f} TFLOP/s  (min {ts[0]:.1f})")
