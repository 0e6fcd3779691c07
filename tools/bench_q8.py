#!/usr/bin/env python3
"""FP8 GEMM A/B on one MI355X: the ping-pong tile (auto / forced 256-, 192-, 128-token forms) against the LDS-DMA tiles it replaces
(gemm_variant 3), interleaved launches, medians, outputs compared bit for bit.  usage: python tools/bench_q8.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from inferix_amd import _hip, hip_ops as ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
FP8 = _hip.IFX_Q_FP8_E4M3
SHAPES = [  # name, M, N, K, epilogue
    ("wan qkv", 4680, 4608, 1536, "bias"), ("wan o+gate", 4680, 1536, 1536, "gate"), ("wan cross q", 4680, 1536, 1536, "bias"),
    ("wan ffn up", 4680, 8960, 1536, "gelu"), ("wan ffn down", 4680, 1536, 8960, "gate"),
    ("magi q", 6075, 3072, 3072, "none"), ("magi k", 6075, 1024, 3072, "none"), ("magi fc1", 6075, 12288, 3072, "gelu_erf"),
    ("magi fc1 -> e4m3", 6075, 12288, 3072, "gelu_erf_q"), ("magi proj", 6075, 3072, 6144, "none"), ("magi fc2", 6075, 3072, 12288, "none"),
    ("magi fc1 3 chunks", 4557, 12288, 3072, "gelu_erf"), ("magi fc1 2 chunks", 3038, 12288, 3072, "gelu_erf"),
    ("magi q 1 chunk", 1519, 3072, 3072, "none"), ("magi k 1 chunk", 1519, 1024, 3072, "none"), ("magi fc1 1 chunk", 1519, 12288, 3072, "gelu_erf_q"),
    ("magi proj 1 chunk", 1519, 3072, 6144, "none"), ("magi fc2 1 chunk", 1519, 3072, 12288, "none"),
]
if len(sys.argv) > 2:
    SHAPES = [s_ for s_ in SHAPES if sys.argv[2] in s_[0]]
print("| launch | M x N x K | epilogue | LDS-DMA tiles us | ping-pong auto us | TFLOP/s | of 5 PF | forced 256 / 192 / 128 us | bits |")
print("|---|---|---|---:|---:|---:|---:|---|---|")
for name, M, N, K, epi in SHAPES:
    xq = torch.randn(M, K, generator=g, device=dev).to(torch.float8_e4m3fn).view(torch.uint8)
    wq = (torch.randn(N, K, generator=g, device=dev) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
    sx = (0.02 * (1 + torch.rand(M, generator=g, device=dev))).contiguous()
    sw = (0.002 * (1 + torch.rand(N, generator=g, device=dev))).contiguous()
    bias = torch.randn(N, generator=g, device=dev).to(torch.bfloat16) if epi in ("bias", "gate", "gelu") else None
    res = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16)
    mod = torch.randn(3, 6, N, generator=g, device=dev).to(torch.bfloat16)
    div = (0.02 * (1 + torch.rand(N, generator=g, device=dev))).contiguous()

    def run():
        if epi == "gate":
            return ops.linear_q8(xq, sx, wq, sw, bias, FP8, epilogue=_hip.IFX_EPI_GATE_RES, residual=res, mod=mod, gate_slot=2, rows_per_group=1560)
        if epi == "gelu":
            return ops.linear_q8(xq, sx, wq, sw, bias, FP8, epilogue=_hip.IFX_EPI_GELU_TANH)
        if epi == "gelu_erf":
            return ops.linear_q8(xq, sx, wq, sw, None, FP8, epilogue=_hip.IFX_EPI_GELU_ERF)
        if epi == "gelu_erf_q":
            return ops.linear_q8_quant_out(xq, sx, wq, sw, FP8, div, epilogue=_hip.IFX_EPI_GELU_ERF)
        return ops.linear_q8(xq, sx, wq, sw, bias, FP8)
    variants = [3, 0, 22, 23, 24]
    outs, times = {}, {v: [] for v in variants}
    for v in variants:
        ops.set_option("gemm_variant", v)
        outs[v] = run().clone()
    for _ in range(reps):
        for v in variants:
            ops.set_option("gemm_variant", v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            e1.synchronize()
            times[v].append(e0.elapsed_time(e1) * 1e3)
    ops.set_option("gemm_variant", 0)
    med = {v: sorted(t)[len(t) // 2] for v, t in times.items()}
    same = {v: bool(torch.equal(outs[v], outs[3])) for v in variants}
    nd = int((outs[0] != outs[3]).sum())
    tf = 2.0 * M * N * K / (med[0] * 1e-6) / 1e12
    print(f"| {name} | {M} x {N} x {K} | {epi} | {med[3]:.1f} | {med[0]:.1f} | {tf:.0f} | {tf / 5000:.3f} | {med[22]:.1f} / {med[23]:.1f} / {med[24]:.1f} | "
          f"{'identical' if all(same.values()) else f'{nd} of {outs[0].numel()} differ (auto vs LDS-DMA)'} |", flush=True)
