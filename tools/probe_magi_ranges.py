"""MAGI rank-shape core attention (4 ranges x 12150 queries x 3 q-heads on 1 kv-head, keys 2..5 chunks): sequential split-KV launches
vs one multi-range launch vs four concurrent unsplit launches on side streams."""
import sys, torch
sys.path.insert(0, "/root/repo")
from inferix_amd import hip_ops as ops
clip, hq = 12150, 3
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(4 * clip, hq * 128, generator=g, device="cuda").to(torch.bfloat16)
k = torch.randn(5 * clip, 1, 128, generator=g, device="cuda").to(torch.bfloat16)
v = torch.randn(5 * clip, 1, 128, generator=g, device="cuda").to(torch.bfloat16)
view = ops.KvCacheView(k, v)
out = torch.empty_like(q)
qr = [(i * clip, (i + 1) * clip) for i in range(4)]
kr = [(0, (2 + i) * clip) for i in range(4)]
flops = sum(4.0 * clip * (ke - ks) * hq * 128 for ks, ke in kr)
def timeit(fn, iters=5, inner=3):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record()
        for _ in range(inner): fn()
        e.record(); e.synchronize(); ts.append(s.elapsed_time(e) / inner)
    ts.sort(); return ts[len(ts) // 2]
def seq(splits=None):
    for (qs, qe), (ks, ke) in zip(qr, kr):
        ops.attention(q[qs:qe].view(-1, hq, 128), view, ke, out=out[qs:qe].view(-1, hq, 128), kv_start=ks, splits=splits)
streams = [torch.cuda.Stream() for _ in range(4)]
def conc(splits=1):
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(cur)
    for st, (qs, qe), (ks, ke) in zip(streams, qr[::-1], kr[::-1]):       # longest first
        st.wait_event(ev)
        with torch.cuda.stream(st):
            ops.attention(q[qs:qe].view(-1, hq, 128), view, ke, out=out[qs:qe].view(-1, hq, 128), kv_start=ks, splits=splits)
        e2 = torch.cuda.Event(); e2.record(st); cur.wait_event(e2)
for name, fn in (("sequential, split plan", lambda: seq(None)), ("sequential, unsplit", lambda: seq(1)),
                 ("one multi-range launch", lambda: ops.attention_ranges(q, view, qr, kr, out, hq)),
                 ("4 streams, unsplit", lambda: conc(1)), ("4 streams, split plan", lambda: conc(None))):
    t = timeit(fn)
    print(f"{name:28s} {t:7.3f} ms  {flops / t / 1e9:7.1f} TFLOP/s")
