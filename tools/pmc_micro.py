#!/usr/bin/env python3
"""Micro workload for PMC passes: the self-attention kernel and the block's GEMMs at the real 480p shapes
(N=4680, d=1536, ffn=8960, 12 heads; prefix L = 4 blocks = 18720 keys), a few launches each."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from inferix_amd import _hip, hip_ops as ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "attn,gemm"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 18720
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
N, H, D, d, f = 4680, 12, 128, 1536, 8960
rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
if "attn" in which:
    # as the model calls it: q carries scale * log2(e), the call says scale = ln 2 (the exponent fast path, `attn_fwd_pp_kernel<.., 2, 7>`)
    q_scale, a_scale = ops.attn_q_prescale(D)
    q, k, v = (torch.randn(N, H, D, generator=g, device=dev) * q_scale).to(torch.bfloat16), rnd(L, H, D), rnd(L, H, D)
    for _ in range(reps):
        ops.attention(q, ops.KvCacheView(k, v), L, scale=a_scale)
if "gemm" in which:
    x = rnd(N, d)
    wqkv, wo, w1, w2 = rnd(3 * d, d), rnd(d, d), rnd(f, d), rnd(d, f)
    b3, b1, bf_ = rnd(3 * d), rnd(d), rnd(f)
    mod = rnd(3, 6, d)
    for _ in range(reps):
        ops.linear(x, wqkv, b3)                                                                      # qkv            (pp <0, 3, 1>)
        ops.linear(x, wo, b1, epilogue=_hip.IFX_EPI_GATE_RES, residual=x, mod=mod, gate_slot=2, rows_per_group=1560)   # o + gate  (pp <3, ., 1>)
        ops.linear(x, wo, b1)                                                                        # cross q        (pp <0, ., 1>)
        ops.linear(x, wo, b1, epilogue=_hip.IFX_EPI_RESIDUAL, residual=x)                            # cross o + res  (pp <2, ., 1>)
        u = ops.linear(x, w1, bf_, epilogue=_hip.IFX_EPI_GELU_TANH)                                  # ffn up + GELU  (pp <1, 4, 1>)
        ops.linear(u, w2, b1, epilogue=_hip.IFX_EPI_GATE_RES, residual=x, mod=mod, gate_slot=5, rows_per_group=1560)   # ffn down + gate, split-K (pp <3, 4, 2>)
if "w4" in which:
    # long-K shapes of the MAGI layer (one rank of cp = 8): the four-wave register-staged tile (auto) against the eight-wave 256x256 tile
    M, hdn, ffn = 6075, 3072, 12288
    a, w1 = rnd(M, hdn), rnd(ffn, hdn)
    for _ in range(reps):
        ops.set_option("gemm_variant", 0)
        ops.linear(a, w1, None)
        ops.set_option("gemm_variant", 5)
        ops.linear(a, w1, None)
    ops.set_option("gemm_variant", 0)
if "q8" in which:
    # FP8 launches on the 8-bit instantiations of the ping-pong tile: the block's shapes and MAGI's long-K ones
    FP8 = _hip.IFX_Q_FP8_E4M3
    for M, Nn, K in ((4680, 4608, 1536), (4680, 8960, 1536), (4680, 1536, 8960), (6075, 3072, 12288), (6075, 12288, 3072)):
        xq = torch.randn(M, K, generator=g, device=dev).to(torch.float8_e4m3fn).view(torch.uint8)
        wq = (torch.randn(Nn, K, generator=g, device=dev) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
        sx = torch.full((M,), 0.02, device=dev)
        sw = torch.full((Nn,), 0.002, device=dev)
        for _ in range(reps):
            ops.linear_q8(xq, sx, wq, sw, None, FP8)
if "norm" in which:
    x = rnd(N, d)
    mod = rnd(3, 6, d)
    for _ in range(reps):
        ops.layernorm(x, 1e-6, mod=mod, rows_per_group=1560)
if "conv" in which:
    # the three dominant decoder shapes: 96->96 @480x832 (12 frames), 192->192 @240x416 (12), 384->384 @120x208 (6)
    for c, h, w_, t in ((96, 480, 832, 12), (192, 240, 416, 12), (384, 120, 208, 6)):
        ring = ops.to_planar(rnd(t + 2, h, w_, c))           # as the decoder's frame rings hand the frames over (32-channel planes)
        wt = (rnd(27, c // 32, c, 32) * (27 * c) ** -0.5).contiguous()
        b = rnd(c)
        res = rnd(t, h, w_, c)
        y = torch.empty(t, h, w_, c, dtype=torch.bfloat16, device=dev)
        for _ in range(reps):
            ops.conv3d_cl(ring, list(range(t + 2)), wt, b, kt=3, ks=3, y=y, out_slots=list(range(t)), residual=res)
        del ring, res, y
torch.cuda.synchronize()
print("pmc_micro done", which, L)
