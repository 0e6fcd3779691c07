"""Is a hipGraph worth it for one rank of an sp-8 shard?  One layer's token-local launches (585 rows: AdaLN, qkv GEMM, o-proj,
cross-attention q GEMM, FFN) issued eagerly through the ctypes wrappers vs replayed as one captured graph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inferix_amd import _hip, hip_ops as ops  # noqa: E402

BF = torch.bfloat16


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 585
    dev = torch.device("cuda:0")
    d, ffn = 1536, 8960
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.05).to(BF)
    x, h, qkv, a, u = r(rows, d), r(rows, d), r(rows, 3 * d), r(rows, d), r(rows, ffn)
    mod = r(3, 6, d)
    W = dict(qkv=r(3 * d, d), o=r(d, d), cq=r(d, d), co=r(d, d), f0=r(ffn, d), f2=r(d, ffn))
    Bv = dict(qkv=r(3 * d), o=r(d), cq=r(d), co=r(d), f0=r(ffn), f2=r(d))
    g3, b3 = r(d), r(d)
    rpg = rows // 3

    def layer():
        ops.layernorm(x, 1e-6, mod=mod, shift_slot=0, scale_slot=1, rows_per_group=rpg, out=h)
        ops.linear(h, W["qkv"], Bv["qkv"], out=qkv)
        ops.linear(a, W["o"], Bv["o"], epilogue=_hip.IFX_EPI_GATE_RES, residual=x, mod=mod, gate_slot=2, rows_per_group=rpg, out=x)
        ops.layernorm(x, 1e-6, gamma=g3, beta=b3, out=h)
        ops.linear(h, W["cq"], Bv["cq"], out=a)
        ops.rmsnorm(a, g3, 1e-6, out=a)
        ops.linear(a, W["co"], Bv["co"], epilogue=_hip.IFX_EPI_RESIDUAL, residual=x, out=x)
        ops.layernorm(x, 1e-6, mod=mod, shift_slot=3, scale_slot=4, rows_per_group=rpg, out=h)
        ops.linear(h, W["f0"], Bv["f0"], epilogue=_hip.IFX_EPI_GELU_TANH, out=u)
        ops.linear(u, W["f2"], Bv["f2"], epilogue=_hip.IFX_EPI_GATE_RES, residual=x, mod=mod, gate_slot=5, rows_per_group=rpg, out=x)

    for _ in range(3):
        layer()
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        layer()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rows {rows} eager : enqueue {(t1 - t0) / n * 1e6:7.1f} us/layer, wall {(t2 - t0) / n * 1e6:7.1f} us/layer (10 launches)")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph.capture_begin()
        layer()
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rows {rows} graph : enqueue {(t1 - t0) / n * 1e6:7.1f} us/layer, wall {(t2 - t0) / n * 1e6:7.1f} us/layer (1 replay)")
    # three graphs of ~3 launches each (what a layer with a collective in the middle would replay)
    parts = []
    fns = [lambda: (ops.layernorm(x, 1e-6, mod=mod, shift_slot=0, scale_slot=1, rows_per_group=rpg, out=h),
                    ops.linear(h, W["qkv"], Bv["qkv"], out=qkv)),
           lambda: (ops.linear(a, W["o"], Bv["o"], epilogue=_hip.IFX_EPI_GATE_RES, residual=x, mod=mod, gate_slot=2, rows_per_group=rpg, out=x),
                    ops.layernorm(x, 1e-6, gamma=g3, beta=b3, out=h), ops.linear(h, W["cq"], Bv["cq"], out=a),
                    ops.rmsnorm(a, g3, 1e-6, out=a)),
           lambda: (ops.linear(a, W["co"], Bv["co"], epilogue=_hip.IFX_EPI_RESIDUAL, residual=x, out=x),
                    ops.layernorm(x, 1e-6, mod=mod, shift_slot=3, scale_slot=4, rows_per_group=rpg, out=h),
                    ops.linear(h, W["f0"], Bv["f0"], epilogue=_hip.IFX_EPI_GELU_TANH, out=u),
                    ops.linear(u, W["f2"], Bv["f2"], epilogue=_hip.IFX_EPI_GATE_RES, residual=x, mod=mod, gate_slot=5, rows_per_group=rpg, out=x))]
    for fn in fns:
        gg = torch.cuda.CUDAGraph()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            gg.capture_begin()
            fn()
            gg.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        parts.append(gg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for gg in parts:
            gg.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rows {rows} 3 graphs: enqueue {(t1 - t0) / n * 1e6:7.1f} us/layer, wall {(t2 - t0) / n * 1e6:7.1f} us/layer (3 replays)")


if __name__ == "__main__":
    main()
