#!/usr/bin/env python3
"""Micro-benchmark of the C-ABI kernels at the 480p shapes (events on the launch stream, interleaved rounds).
usage: tools/bench_kernels.py [gemm|attn|norm|all] [M]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from inferix_amd import _hip, hip_ops as ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4680
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)


def timeit(fn, iters=10, warm=3, inner=10):
    """median / min over `iters` samples of the mean of `inner` back-to-back launches (hides launch gaps)"""
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record()
        for _ in range(inner):
            fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e) / inner)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


d, f, H, D = 1536, 8960, 12, 128
if what in ("gemm", "all"):
    x, u = rnd(M, d), rnd(M, f)
    res = rnd(M, d)
    mod = rnd(3, 6, d)
    shapes = [("qkv   N=4608 K=1536 bias", x, rnd(3 * d, d) * 0.03, rnd(3 * d), dict()),
              ("o     N=1536 K=1536 gate", x, rnd(d, d) * 0.03, rnd(d), dict(epilogue=_hip.IFX_EPI_GATE_RES, residual=res, mod=mod, gate_slot=2, rows_per_group=(M + 2) // 3)),
              ("ffn0  N=8960 K=1536 gelu", x, rnd(f, d) * 0.03, rnd(f), dict(epilogue=_hip.IFX_EPI_GELU_TANH)),
              ("ffn2  N=1536 K=8960 gate", u, rnd(d, f) * 0.01, rnd(d), dict(epilogue=_hip.IFX_EPI_GATE_RES, residual=res, mod=mod, gate_slot=5, rows_per_group=(M + 2) // 3))]
    for variant in [0]:
        ops.set_option("gemm_variant", variant)
        tot = 0.0
        for name, a, w, b, kw in shapes:
            out = torch.empty(M, w.shape[0], dtype=torch.bfloat16, device=dev)
            med, mn = timeit(lambda: ops.linear(a, w, b, out=out, **kw))
            fl = 2.0 * M * w.shape[0] * w.shape[1]
            tot += med
            print(f"gemm v{variant} {name}: {med*1e3:8.1f} us (min {mn*1e3:7.1f})  {fl/med/1e9:7.1f} TFLOP/s")
        print(f"gemm v{variant} sum of 4 = {tot*1e3:.1f} us")
    ops.set_option("gemm_variant", 0)
if what in ("attn", "all"):
    av = int(os.environ.get("ATTN_VARIANT", "0"))
    ops.set_option("attn_variant", av)
    q = rnd(M, H, D)
    for blk in (1, 4, 7):
        L = blk * 4680
        k, v = rnd(L, H, D), rnd(L, H, D)
        out = torch.empty_like(q)
        med, mn = timeit(lambda: ops.attention(q, ops.KvCacheView(k, v), L, out=out), iters=10)
        fl = 4.0 * M * L * H * D
        print(f"attn L={L:6d}: {med*1e3:8.1f} us (min {mn*1e3:7.1f})  {fl/med/1e9:7.1f} TFLOP/s")
    k, v = rnd(512, H, D), rnd(512, H, D)
    med, mn = timeit(lambda: ops.attention(q, ops.KvCacheView(k, v), 512))
    print(f"attn cross L=512: {med*1e3:8.1f} us  {4.0*M*512*H*D/med/1e9:7.1f} TFLOP/s")
if what in ("norm", "all"):
    x = rnd(M, d)
    mod = rnd(3, 6, d)
    out = torch.empty_like(x)
    med, _ = timeit(lambda: ops.layernorm(x, 1e-6, mod=mod, rows_per_group=(M + 2) // 3, out=out))
    print(f"adaln layernorm: {med*1e3:7.1f} us  {4.0*M*d/med/1e6:7.1f} GB/s")
    qkv = rnd(M, 3 * d)
    kc, vc = torch.zeros(M, H, D, dtype=torch.bfloat16, device=dev), torch.zeros(M, H, D, dtype=torch.bfloat16, device=dev)
    from inferix_amd.wan import components as C
    rope = ops.RopeGridSpec(C.rope_table(128).to(dev), 0, 30, 52, 0, M // 3)
    w1 = rnd(d)
    med, _ = timeit(lambda: ops.rmsnorm_rope_kv_append(qkv, w1, w1, 1e-6, rope, ops.KvCacheView(kc, vc), 0, d))
    print(f"rmsnorm+rope+append: {med*1e3:7.1f} us  {12.0*M*d/med/1e6:7.1f} GB/s")
if what in ("norm", "all"):
    # config 4's row passes: the per-token quantiser on the FFN activation and on a dim-wide row, LayerNorm + quantiser in one pass
    for K in (f, d):
        xk = rnd(M, K)
        qb, sb = torch.empty(M, K, dtype=torch.uint8, device=dev), torch.empty(M, dtype=torch.float32, device=dev)
        for fmt, nm in ((_hip.IFX_Q_FP8_E4M3, "fp8"), (_hip.IFX_Q_INT8, "int8")):
            med, _ = timeit(lambda: ops.quant_per_token(xk, fmt, q=qb, scale=sb))
            print(f"quant_per_token {nm} K={K}: {med*1e3:7.1f} us  {3.0*M*K/med/1e6:7.1f} GB/s")
    x = rnd(M, d)
    mod = rnd(3, 6, d)
    qb, sb = torch.empty(M, d, dtype=torch.uint8, device=dev), torch.empty(M, dtype=torch.float32, device=dev)
    med, _ = timeit(lambda: ops.layernorm_quant(x, 1e-6, _hip.IFX_Q_FP8_E4M3, q=qb, scale=sb, mod=mod, rows_per_group=(M + 2) // 3))
    print(f"adaln layernorm + fp8 quant: {med*1e3:7.1f} us  {3.0*M*d/med/1e6:7.1f} GB/s")
    w1 = rnd(d)
    med, _ = timeit(lambda: ops.rmsnorm(x, w1, 1e-6, out=x))
    print(f"rmsnorm (cross q): {med*1e3:7.1f} us  {4.0*M*d/med/1e6:7.1f} GB/s")
if what == "split":
    ops.set_option("attn_variant", int(os.environ.get("ATTN_VARIANT", "0")))
    q = rnd(M, H, D)
    for blk in (1, 4, 7):
        L = blk * 4680
        k, v = rnd(L, H, D), rnd(L, H, D)
        out = torch.empty_like(q)
        fl = 4.0 * M * L * H * D
        res = []
        for sp in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, None):
            med, mn = timeit(lambda: ops.attention(q, ops.KvCacheView(k, v), L, out=out, splits=sp), iters=10)
            res.append(f"{'auto' if sp is None else sp}:{med*1e3:.0f}")
        print(f"attn M={M} L={L:6d} us by splits  " + "  ".join(res) + f"   ideal@972TF {fl/972e6:.0f}")
