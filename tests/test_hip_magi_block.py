"""GPU parity tests of the MAGI transformer layer (BASELINE config 5) on the HIP path against the reference-generated
goldens tests/golden/magi_block_{tiny,real}.npz (the reference's own TransformerLayer run on CPU by
oracle/gen_golden_magi_block.py) and against oracle/magi_block_oracle.py, plus the kernels underneath it."""
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import magi_block_oracle as MB
from fixture_io import golden
from util import assert_bf16_parity, rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _configs(cfg: MB.MagiLayerConfig, n_layers: int, cp_strategy="none"):
    mc = SimpleNamespace(num_layers=max(n_layers, 3), hidden_size=cfg.hidden_size, ffn_hidden_size=cfg.ffn_hidden_size,
                         num_attention_heads=cfg.num_attention_heads, num_query_groups=cfg.num_query_groups,
                         kv_channels=cfg.kv_channels, layernorm_epsilon=cfg.layernorm_epsilon,
                         apply_layernorm_1p=cfg.apply_layernorm_1p, gated_linear_unit=cfg.gated_linear_unit,
                         params_dtype=BF)
    ec = SimpleNamespace(cp_size=1, cp_strategy=cp_strategy, fp8_quant=False, kv_offload=False, ulysses_overlap_degree=1)
    return mc, ec


def _meta(m: MB.LayerMeta):
    from inferix_amd.magi.types import ModelMetaArgs, PackedCoreAttnParams, PackedCrossAttnParams
    qr, kr = torch.tensor(m.q_ranges, dtype=torch.int32), torch.tensor(m.k_ranges, dtype=torch.int32)
    core = PackedCoreAttnParams(q_range=qr, k_range=kr, np_q_range=qr.numpy(), np_k_range=kr.numpy(),
                                max_seqlen_q=m.clip_token_nums, max_seqlen_k=int(kr[:, 1].max()))
    cross = PackedCrossAttnParams(cu_seqlens_q=torch.tensor(m.cu_seqlens_q, dtype=torch.int32),
                                  cu_seqlens_kv=torch.tensor(m.cu_seqlens_kv, dtype=torch.int32),
                                  max_seqlen_q=m.clip_token_nums, max_seqlen_kv=int(np.diff(m.cu_seqlens_kv).max()))
    return ModelMetaArgs(H=1, W=1, cp_pad_size=0, cp_split_sizes=None, slice_point=m.slice_point,
                         denoising_range_num=len(m.q_ranges), range_num=len(m.q_ranges) + m.slice_point,
                         extract_prefix_video_feature=False, fwd_extra_1st_chunk=m.use_cache and m.slice_point == 0,
                         distill_nearly_clean_chunk=m.distill_nearly_clean_chunk, clip_token_nums=m.clip_token_nums,
                         enable_cuda_graph=False, core_attn_params=core, cross_attn_params=cross)


@pytest.mark.parametrize("name", ["magi_block_tiny", "magi_block_real"])
def test_layer_stack_vs_reference_golden(name):
    """Every forward of the fixture through HipMagiTransformerLayer: outputs against the reference's, with the distance of
    BOTH from the float64 evaluation of the same layer printed (the bf16 floor: the reference's own rounding noise), and the
    stored cache rows against the reference's cache."""
    from inferix_amd.magi.dit import HipMagiTransformerLayer
    from inferix_amd.magi.types import InferenceParams
    fx = golden(name + ".npz")
    cfg, n_layers, clip, n_calls, wseed, max_tokens = MB.fixture_geometry(fx)
    mc, ec = _configs(cfg, n_layers)
    Ws = [MB.init_layer_weights(cfg, wseed + li) for li in range(n_layers)]
    layers = []
    for li in range(n_layers):
        layer = HipMagiTransformerLayer(mc, ec, li, "cuda")
        layer.load_state_dict(Ws[li])
        layers.append(layer)
    ip = InferenceParams(1, max_tokens)
    orc_caches = [MB.MagiLayerCache(max_tokens, cfg.num_query_groups, cfg.kv_channels) for _ in range(n_layers)]
    for ci in range(n_calls):
        inp, m = MB.fixture_call(fx, ci)
        meta = _meta(m)
        ip.update_kv_cache = m.update_kv_cache
        x = inp["x"].cuda()
        x_ref = inp["x"]
        for li, layer in enumerate(layers):
            # yardstick: the float64 layer on the REFERENCE's input of this layer, with the reference's cache prefix
            exact = MB.exact_layer_forward(Ws[li], cfg, x_ref, inp["condition"], inp["condition_map"], inp["y"], inp["rope"], m,
                                           orc_caches[li])
            ref = fx[f"c{ci}_out_l{li}"]
            x_in_hip = x
            x = layer(x, inp["condition"].cuda(), inp["condition_map"].cuda(), inp["y"].cuda(), inp["rope"].cuda(), ip, meta)
            if li == 0:                       # identical inputs on both sides: the per-layer bar
                floor = rel_l2(ref, exact)
                d_hip_exact, d_hip_ref = rel_l2(x.cpu(), exact), rel_l2(x.cpu(), ref)
                print(f"{name} call {ci} layer 0: ref-vs-exact {floor:.3e}  hip-vs-exact {d_hip_exact:.3e}  hip-vs-ref {d_hip_ref:.3e}")
                assert d_hip_exact <= 1.25 * floor + 5e-4, (ci, d_hip_exact, floor)
                # element bound: the post-norms re-scale rows by 1/std of (x * gate), so one flipped bf16 rounding upstream shows
                # up at a few ULPs of a typical element; the tensor-level bound above (1.25 x the reference's own distance) is the bar
                assert_bf16_parity(x, ref, max_ulp=8, max_mismatch_frac=0.6, rel=2.0 * floor + 5e-4, floor=1.0,
                                   what=f"{name} call {ci} layer 0")
            else:                             # chained layers: inputs already differ by the floor
                assert rel_l2(x.cpu(), ref) < 1e-2, (ci, li)
            # advance the oracle's cache with the reference's own stream so the yardstick prefix stays the reference's
            MB.layer_forward(Ws[li], cfg, x_ref, inp["condition"], inp["condition_map"], inp["y"], inp["rope"], m, orc_caches[li])
            x_ref = ref
            del x_in_hip
    written = int(fx["cache_written"])
    for li in range(n_layers):
        raw = ip.kv_cache_manager.get_raw(ip.kv_cache_request, f"layer_{li}")
        if li == 0:
            assert_bf16_parity(raw[0, :written, 0], fx[f"cache_l{li}"][0, :written, 0], max_ulp=1, floor=1.0, max_mismatch_frac=0.05,
                               what="cache K (LayerNorm + rotary)")
            # V is the raw K = hidden GEMM output: one rounding; 2 ULP covers a flip across a binade boundary
            assert_bf16_parity(raw[1, :written, 0], fx[f"cache_l{li}"][1, :written, 0], max_ulp=2, floor=0.05, what="cache V")
        else:
            assert rel_l2(raw[:, :written].cpu(), fx[f"cache_l{li}"][:, :written]) < 1e-2


def test_head_prep_vs_oracle():
    """ifx_magi_head_prep on identical inputs (the fused projection output): q / k against fp32 LayerNorm + rotary, qx / kx
    against the bf16 LayerNorm, v bit-exact, with the split destination (stored rows | scratch tail)."""
    from inferix_amd import hip_ops as ops
    g = torch.Generator().manual_seed(5)
    cfg = MB.MagiLayerConfig()
    hq, hk, hd, rows = cfg.num_attention_heads, cfg.num_query_groups, 128, 333
    mixed = (torch.randn(rows, (2 * hq + 2 * hk) * hd, generator=g) * 1.5).to(BF)
    r = torch.rand(rows, 64, generator=g) * 6.0
    rope = torch.cat([torch.sin(r), torch.cos(r)], dim=-1)
    qn = (0.1 * torch.randn(hd, generator=g), 0.1 * torch.randn(hd, generator=g))
    kn = (0.1 * torch.randn(hd, generator=g), 0.1 * torch.randn(hd, generator=g))
    xn = ((0.1 * torch.randn(hd, generator=g)).to(BF), (0.1 * torch.randn(hd, generator=g)).to(BF))
    cap, split = 400, 200
    kc = torch.zeros(cap + rows, hk, hd, dtype=BF, device="cuda")
    vc = torch.zeros_like(kc)
    q_out = torch.empty(rows, hq * hd, dtype=BF, device="cuda")
    qx_out = torch.empty_like(q_out)
    ops.magi_head_prep(mixed.cuda(), layout=0, q_heads=hq, kv_heads=hk, eps=1e-6, layernorm_1p=True, k_out=kc, v_out=vc,
                       kv_head_stride=hd, ld_kv=hk * hd, row0=100, split=split, row1=cap, rope=rope.cuda(),
                       qn=[t.cuda() for t in qn], kn=[t.cuda() for t in kn], xn=[t.cuda() for t in xn], q_out=q_out, qx_out=qx_out)
    Q, KV = hq * hd, hk * hd
    sin, cos = rope[:, :64], rope[:, 64:]

    def ref_qk(t, n):
        t = MB.fused_layer_norm(t.reshape(rows, 1, -1, hd).float(), n[0], n[1], cfg)
        return MB.apply_rotary(t.transpose(0, 1).contiguous(), cos, sin).to(BF)[0]
    q_ref = ref_qk(mixed[:, :Q], qn)
    k_ref = ref_qk(mixed[:, 2 * Q:2 * Q + KV], kn)
    qx_ref = MB.fused_layer_norm(mixed[:, Q:2 * Q].reshape(rows, hq, hd), xn[0], xn[1], cfg)
    v_ref = mixed[:, 2 * Q + KV:].reshape(rows, hk, hd)
    assert_bf16_parity(q_out.view(rows, hq, hd), q_ref, max_ulp=1, floor=1.0, what="q (LayerNorm + rotary)")
    assert_bf16_parity(qx_out.view(rows, hq, hd), qx_ref, max_ulp=1, floor=0.05, what="qx (bf16 LayerNorm)")
    k_got = torch.cat([kc[100:100 + split], kc[cap:cap + rows - split]])
    v_got = torch.cat([vc[100:100 + split], vc[cap:cap + rows - split]])
    assert_bf16_parity(k_got, k_ref, max_ulp=1, floor=1.0, what="k (LayerNorm + rotary)")
    assert torch.equal(v_got.cpu(), v_ref), "v is a copy"
    assert int(kc[:100].abs().sum()) == 0 and int(kc[100 + split:cap].abs().sum()) == 0, "rows outside the destination touched"
    # round 5: q (and k | v) written straight in the head -> rank all-to-all's send order — the bits of the plain launch, permuted
    for cp in (2, 4, 8):
        if hq % cp or hk != cp:
            continue
        hpr = hq // cp
        q_send = torch.zeros(cp, rows, hpr * hd, dtype=BF, device="cuda")
        kv_send = torch.zeros(cp, rows, 1, 2 * hd, dtype=BF, device="cuda")
        qx2 = torch.empty_like(qx_out)
        ops.magi_head_prep(mixed.cuda(), layout=0, q_heads=hq, kv_heads=hk, eps=1e-6, layernorm_1p=True, k_out=kv_send,
                           v_out=kv_send.view(-1)[hd:], kv_head_stride=rows * 2 * hd, ld_kv=2 * hd, rope=rope.cuda(),
                           qn=[t.cuda() for t in qn], kn=[t.cuda() for t in kn], xn=[t.cuda() for t in xn], q_out=q_send, qx_out=qx2,
                           q_group=hpr)
        want_q = q_out.view(rows, cp, hpr * hd).transpose(0, 1)
        assert torch.equal(q_send, want_q), f"cp = {cp}: q in send order differs from the plain launch"
        assert torch.equal(qx2, qx_out)
        assert torch.equal(kv_send[:, :, 0, :hd].transpose(0, 1), k_got.cuda() if False else torch.cat([kc[100:100 + split], kc[cap:cap + rows - split]]))
        assert torch.equal(kv_send[:, :, 0, hd:].transpose(0, 1).cpu(), v_ref)
    # layout 1: the caption keys / values
    yt = 77
    kvx = torch.randn(yt, 2 * KV, generator=g).to(BF)
    kx = torch.empty(yt, hk, hd, dtype=BF, device="cuda")
    vx = torch.empty_like(kx)
    ops.magi_head_prep(kvx.cuda(), layout=1, q_heads=0, kv_heads=hk, eps=1e-6, layernorm_1p=True, k_out=kx, v_out=vx,
                       kv_head_stride=hd, ld_kv=hk * hd, xn=[t.cuda() for t in xn])
    kv3 = kvx.view(yt, hk, 2 * hd)
    assert_bf16_parity(kx, MB.fused_layer_norm(kv3[..., :hd], xn[0], xn[1], cfg), max_ulp=1, floor=0.05, what="kx")
    assert torch.equal(vx.cpu(), kv3[..., hd:].contiguous())


@pytest.mark.parametrize("rows,dim", [(96, 3072), (333, 256), (4050, 3072)])
def test_gate_norm_residual_vs_oracle(rows, dim):
    from inferix_amd import hip_ops as ops
    g = torch.Generator().manual_seed(rows + dim)
    x = torch.randn(rows, dim, generator=g).to(BF)
    res = torch.randn(rows, dim, generator=g).to(BF)
    gate = torch.tanh(torch.randn(3, 2 * dim, generator=g)).to(BF)
    cmap = (torch.arange(rows) * 3 // rows).to(torch.int32)
    w, b = 0.1 * torch.randn(dim, generator=g), 0.1 * torch.randn(dim, generator=g)
    for half in (0, 1):
        gv = gate[:, half * dim:(half + 1) * dim]
        t = x.float() * gv.float()[cmap.long()]
        ref = (torch.nn.functional.layer_norm(t, (dim,), w + 1, b, 1e-6) + res.float()).to(BF)
        got = ops.magi_gate_norm_residual(x.cuda(), res.cuda(), cmap.cuda(), gate.cuda()[:, half * dim:(half + 1) * dim],
                                          w.cuda(), b.cuda(), 1e-6, True)
        assert_bf16_parity(got, ref, max_ulp=1, floor=1.0, what=f"bias_modulate_add half {half}")


def test_gate_path_and_gelu_erf_epilogue():
    """SiLU -> linear -> softcap (AdaModulateLayer + softcap) and the exact-GELU GEMM epilogue against torch on the same inputs."""
    from inferix_amd import _hip
    from inferix_amd import hip_ops as ops
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(5, 768, generator=g) * 2).to(BF)
    assert_bf16_parity(ops.act_rows(x.cuda(), _hip.IFX_ACT_SILU), torch.nn.functional.silu(x), max_ulp=1, what="silu")
    assert_bf16_parity(ops.act_rows(x.cuda(), _hip.IFX_ACT_TANH), torch.tanh(x.float()).to(BF), max_ulp=1, what="softcap")
    for M, N, K in ((96, 12288, 3072), (300, 512, 256), (4050, 12288, 3072)):
        a = torch.randn(M, K, generator=g).to(BF).cuda()
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).cuda()
        got = ops.linear(a, w, None, epilogue=_hip.IFX_EPI_GELU_ERF)
        pre = (a.double() @ w.double().t()).to(BF)                     # exact product, one bf16 rounding (the fc1 output)
        ref = torch.nn.functional.gelu(pre.float()).to(BF)
        assert_bf16_parity(got, ref, max_ulp=2, floor=1.0, max_mismatch_frac=0.05, what=f"fc1 + exact GELU {M}x{N}x{K}")
        plain = ops.linear(a, w, None)
        assert_bf16_parity(got, torch.nn.functional.gelu(plain.float()).to(BF), max_ulp=1, max_mismatch_frac=0.01,
                           what="epilogue == gelu(bf16 GEMM output)")


def test_gelu_erf_epilogue_every_bf16_input():
    """The exact-GELU epilogue (erf from the Chebyshev erfc, ifx_common.h) on EVERY finite bf16 value, passed through a GEMM whose
    weight picks one input column: bit-identical to torch's fp32 `0.5 x (1 + erf(x / sqrt 2))` rounded to bf16 for x >= -3; below
    that 1 + erf cancels in fp32 and torch's own vectorised erf and libm's differ in the last bits (a handful of inputs, results
    ~1e-3 and smaller) — there the bound is one ULP, absolute 2e-6 in the deep tail."""
    from inferix_amd import _hip
    from inferix_amd import hip_ops as ops
    bits = torch.arange(65536, dtype=torch.int32)
    v = (bits << 16).view(torch.float32)
    v = v[torch.isfinite(v) & (v.abs() < 1e30) & (v.abs() > 1e-30)].to(BF)       # (-0 cannot pass through a GEMM: 0 * w + (-0) = +0)
    v = torch.cat([torch.zeros(1, dtype=BF), v])
    M = v.numel()
    a = torch.zeros(M, 64, dtype=BF)
    a[:, 0] = v
    w = torch.zeros(64, 64, dtype=BF)
    w[0, 0] = 1.0
    got = ops.linear(a.cuda(), w.cuda(), None, epilogue=_hip.IFX_EPI_GELU_ERF)[:, 0].cpu()
    ref = torch.nn.functional.gelu(v.float()).to(BF)
    head = v.float() >= -3.0
    assert torch.equal(got[head].view(torch.int16), ref[head].view(torch.int16)), \
        f"{int((got[head].view(torch.int16) != ref[head].view(torch.int16)).sum())} of {int(head.sum())} inputs differ for x >= -3"
    tail = ~head
    assert int((got[tail].view(torch.int16) != ref[tail].view(torch.int16)).sum()) <= 64
    d = (got[tail].float() - ref[tail].float()).abs()
    assert bool((d <= torch.clamp(ref[tail].float().abs() * 2.0 ** -7, min=2e-6)).all())      # one bf16 ULP, or 2e-6 in the deep tail


def test_strided_attention_matches_dense():
    """ifx_attn_fwd_paged_ld: query rows taken from, and output rows written into, column blocks of wider matrices — bit-identical
    to the dense launch; neighbouring columns untouched.  Shapes of one MAGI rank and of a single-GPU layer."""
    from inferix_amd import hip_ops as ops
    g = torch.Generator().manual_seed(3)
    for rows, hq, hk, nk, start in ((300, 24, 8, 700, 0), (2025, 3, 1, 4000, 128), (96, 24, 8, 65, 40)):
        q = torch.randn(rows, hq, 128, generator=g).to(BF).cuda()
        k = torch.randn(nk, hk, 128, generator=g).to(BF).cuda()
        v = torch.randn(nk, hk, 128, generator=g).to(BF).cuda()
        view = ops.KvCacheView(k, v)
        dense = ops.attention(q, view, nk, kv_start=start)
        wide_q = torch.zeros(rows, 2 * hq * 128 + 64, dtype=BF, device="cuda")
        wide_q[:, 64:64 + hq * 128] = q.view(rows, -1)
        wide_o = torch.full((rows, 2 * hq * 128), 7.0, dtype=BF, device="cuda")
        ops.attention_ld(wide_q[:, 64:], view, nk, wide_o[:, hq * 128:], hq, kv_start=start)
        assert torch.equal(wide_o[:, hq * 128:].reshape(rows, hq, 128), dense)
        assert bool((wide_o[:, :hq * 128] == 7.0).all())


def test_multi_range_attention_equals_per_range_launches():
    """ifx_attn_fwd_ranges: the four denoising ranges of a MAGI forward (one rank: 3 q-heads on 1 kv-head; and all 24 / 8 heads)
    in one launch, ranges of unequal length with ragged tiles, equal the per-range launches bit for bit when those do not split
    keys, and to flash-attention noise when they do; rows outside every range stay untouched."""
    from inferix_amd import hip_ops as ops
    g = torch.Generator().manual_seed(8)
    for hq, hk, clip in ((3, 1, 2025), (24, 8, 1300)):
        nk = 5 * clip
        rows = 4 * clip + 77
        q = torch.randn(rows, hq * 128, generator=g).to(BF).cuda()
        k = torch.randn(nk, hk, 128, generator=g).to(BF).cuda()
        v = torch.randn(nk, hk, 128, generator=g).to(BF).cuda()
        view = ops.KvCacheView(k, v)
        qr = [(i * clip, (i + 1) * clip) for i in range(4)]
        kr = [(0, 2 * clip), (clip, 3 * clip + 11), (0, 4 * clip), (0, 5 * clip)]
        out = torch.full((rows, hq * 128), 3.0, dtype=BF, device="cuda")
        ops.attention_ranges(q, view, qr, kr, out, hq)
        ref = torch.full_like(out, 3.0)
        for (qs, qe), (ks, ke) in zip(qr, kr):
            ref[qs:qe] = ops.attention(q[qs:qe].view(-1, hq, 128).contiguous(), view, ke, kv_start=ks, splits=1).view(qe - qs, -1)
        assert torch.equal(out[4 * clip:], ref[4 * clip:]), "rows outside the ranges were written"
        assert rel_l2(out, ref) < 2e-3
        assert_bf16_parity(out, ref, max_ulp=4, max_mismatch_frac=0.3, rel=2e-3, what="multi-range vs per-range")


def test_static_and_per_tensor_quantisers_vs_reference_golden():
    """ifx_quant_static with the bf16 intermediate == the reference's div_clamp_to bytes (tests/golden/quant_fp8.npz, generated by
    importing inferix/models/magi/dit/dit_module.py); the static-scale FP8 linears on ifx_gemm_q8 against the reference modules'
    outputs; the dynamic per-tensor quantiser against its definition."""
    from inferix_amd import _hip
    from inferix_amd import hip_ops as ops
    from inferix_amd.quant import StaticFp8Linear
    fx = golden("quant_fp8.npz")
    x = fx["x"].cuda()
    K = x.shape[1]
    for name in ("vec", "one"):
        q = ops.quant_static(x, fx[f"div_{name}"].cuda(), _hip.IFX_Q_FP8_E4M3, via_bf16=True)
        assert torch.equal(q.cpu(), fx[f"q_{name}"]), f"div_clamp_to bytes ({name}) differ from the reference's"
    direct = ops.quant_static(x, fx["div_vec"].cuda(), _hip.IFX_Q_FP8_E4M3, via_bf16=False)
    assert int((direct.cpu() != fx["q_vec"]).sum()) == int(fx["double_rounding_diffs"]), "single-rounding variant is the direct cast"
    wq, ws, ins = fx["wq"].cuda(), fx["w_scale"].cuda(), fx["in_scale"].cuda()
    xin = x.view(4, 24, K)
    pt = StaticFp8Linear(wq, ws, ins, divisor=ins.expand(K).contiguous())
    pc = StaticFp8Linear(wq, ws, ins, divisor=fx["div_vec"].cuda())
    # the fp8 products are exact in fp32 up to summation order; one bf16 rounding at the end
    assert_bf16_parity(pt(xin), fx["y_per_tensor"], max_ulp=1, max_mismatch_frac=0.03, what="PerTensorQuantizedFp8Linear")
    assert_bf16_parity(pc(xin), fx["y_per_channel"], max_ulp=1, max_mismatch_frac=0.03, what="PerChannelQuantizedFp8Linear")
    # dynamic per-tensor (the qconfig family next to per-token): s = amax / QMAX over the whole tensor
    for fmt, qmax in ((_hip.IFX_Q_FP8_E4M3, 448.0), (_hip.IFX_Q_INT8, 127.0)):
        q, s = ops.quant_per_tensor(x, fmt)
        s_ref = x.float().abs().max() / qmax
        assert torch.equal(s.cpu(), s_ref.cpu().expand(x.shape[0])), "per-tensor scale"
        vals = (x.float() / s_ref).clamp(-qmax, qmax).cpu()
        ref = vals.to(torch.float8_e4m3fn).view(torch.uint8) if fmt == _hip.IFX_Q_FP8_E4M3 else torch.round(vals).to(torch.int8).view(torch.uint8)
        assert torch.equal(q.cpu(), ref)
    z = torch.zeros(8, 256, dtype=BF, device="cuda")
    q, s = ops.quant_per_tensor(z, _hip.IFX_Q_INT8)
    assert float(s[0]) == 1.0 and int(q.sum()) == 0


def test_fused_static_quantisers_are_the_separate_passes_bit_for_bit():
    """The two producers that now carry MAGI's static quantisers: LayerNorm + n quantisations of its row (ifx_layernorm_quant_static)
    == ifx_layernorm then ifx_quant_static per divisor; a GELU GEMM whose epilogue quantises for the next linear
    (ifx_gemm_q8_quant_out) == ifx_gemm_q8 then ifx_quant_static — every byte, on every kernel the dispatcher picks (128-row
    register-staged tile, 256 x 128 and 256 x 256 LDS-DMA tiles), ragged row counts included."""
    from inferix_amd import _hip
    from inferix_amd import hip_ops as ops
    g = torch.Generator(device="cuda").manual_seed(3)
    FP8 = _hip.IFX_Q_FP8_E4M3
    for rows, dim, n_out in ((1519, 3072, 4), (6075, 3072, 1), (37, 256, 3), (300, 1536, 2)):
        x = (torch.randn(rows, dim, generator=g, device="cuda") * 1.7 + 0.3).to(BF)
        gamma = (1 + 0.1 * torch.randn(dim, generator=g, device="cuda")).to(BF)
        beta = (0.1 * torch.randn(dim, generator=g, device="cuda")).to(BF)
        divs = (0.03 * (1 + 0.5 * torch.rand(n_out, dim, generator=g, device="cuda"))).contiguous()
        for affine in (True, False):
            kw = dict(gamma=gamma, beta=beta) if affine else {}
            fused = ops.layernorm_quant_static(x, 1e-6, divs, **kw)
            ln = ops.layernorm(x, 1e-6, **kw)
            for j in range(n_out):
                assert torch.equal(fused[:, j], ops.quant_static(ln, divs[j], FP8, via_bf16=True)), (rows, dim, j, affine)
        direct = ops.layernorm_quant_static(x, 1e-6, divs[:1].contiguous(), via_bf16=False)
        assert torch.equal(direct[:, 0], ops.quant_static(ops.layernorm(x, 1e-6), divs[0], FP8, via_bf16=False))
    for M, N, K in ((1519, 12288, 3072), (6075, 12288, 3072), (24300, 12288, 3072), (100, 512, 256), (300, 1024, 384)):
        xq = torch.randn(M, K, generator=g, device="cuda").to(torch.float8_e4m3fn).view(torch.uint8)
        wq = (torch.randn(N, K, generator=g, device="cuda") * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
        sx = torch.full((M,), 0.03, device="cuda")
        sw = (0.002 * (1 + torch.rand(N, generator=g, device="cuda"))).contiguous()
        div = (0.02 * (1 + torch.rand(N, generator=g, device="cuda"))).contiguous()
        for epi in (_hip.IFX_EPI_GELU_ERF, _hip.IFX_EPI_GELU_TANH):
            y = ops.linear_q8(xq, sx, wq, sw, None, FP8, epilogue=epi)
            want = ops.quant_static(y, div, FP8, via_bf16=True)
            got = ops.linear_q8_quant_out(xq, sx, wq, sw, FP8, div, epilogue=epi)
            assert torch.equal(got, want), (M, N, K, epi, int((got != want).sum()))
    with pytest.raises(RuntimeError):
        ops.linear_q8_quant_out(xq, sx, wq, sw, FP8, div, epilogue=_hip.IFX_EPI_BIAS)


def test_full_size_chunk_properties():
    """MAGI-4.5B at its workload size — one 720x720 chunk = 12150 tokens, 24 q-heads on 8 kv-groups, 2 denoising ranges, a
    prefix of one stored chunk — where the CPU oracle is out of reach: size-independent properties of the layer.
      * determinism: two runs, bit-identical;
      * range independence: the first range's output rows do not change when the second range's inputs do;
      * cache idempotence: re-running the storing forward leaves the cache bit-identical;
      * one rank's view (3 q-heads on 1 kv-head over all tokens, the cp = 8 split) of the core attention equals the
        corresponding head slice of the full launch."""
    from inferix_amd import hip_ops as ops
    from inferix_amd.magi.dit import HipMagiTransformerLayer
    from inferix_amd.magi.types import InferenceParams
    cfg = MB.MagiLayerConfig()
    mc, ec = _configs(cfg, 1)
    W = MB.init_layer_weights(cfg, 77)
    layer = HipMagiTransformerLayer(mc, ec, 0, "cuda")
    layer.load_state_dict(W)
    clip = 12150
    g = torch.Generator().manual_seed(1)
    s = 2 * clip
    x = torch.randn(s, 1, cfg.hidden_size, generator=g).to(BF).cuda()
    cond = torch.randn(1, 2, cfg.cond_size, generator=g).to(BF).cuda()
    cmap = (torch.arange(s, dtype=torch.int32) // clip).reshape(s, 1).cuda()
    y = torch.randn(2 * 200, cfg.xattn_size, generator=g).to(BF).cuda()
    r = torch.rand(s, 64, generator=g) * 6.0
    rope = torch.cat([torch.sin(r), torch.cos(r)], -1).cuda()
    m = MB.LayerMeta(q_ranges=[(0, clip), (clip, 2 * clip)], k_ranges=[(0, clip), (0, 2 * clip)], cu_seqlens_q=[0, clip, 2 * clip],
                     cu_seqlens_kv=[0, 200, 400], clip_token_nums=clip, slice_point=0, update_kv_cache=True, use_cache=True)
    meta = _meta(m)
    ip = InferenceParams(1, 4 * clip)
    ip.update_kv_cache = True
    out1 = layer(x, cond, cmap, y, rope, ip, meta)
    cache1 = ip.kv_cache_manager.get_raw(ip.kv_cache_request, "layer_0").clone()
    out2 = layer(x, cond, cmap, y, rope, ip, meta)
    assert torch.equal(out1, out2), "run-to-run determinism"
    assert torch.equal(cache1[:, :s], ip.kv_cache_manager.get_raw(ip.kv_cache_request, "layer_0")[:, :s]), "cache idempotence"
    assert torch.isfinite(out1.float()).all()
    x2 = x.clone()
    x2[clip:] = torch.randn(clip, 1, cfg.hidden_size, generator=g).to(BF).cuda()
    out3 = layer(x2, cond, cmap, y, rope, ip, meta)
    assert torch.equal(out3[:clip], out1[:clip]), "range 0 must not see range 1"
    assert not torch.equal(out3[clip:], out1[clip:])
    # one rank of cp = 8: query heads 3r..3r+2 read kv head r
    raw = ip.kv_cache_manager.get_raw(ip.kv_cache_request, "layer_0")
    k_all, v_all = raw[0, :s, 0], raw[1, :s, 0]
    q = torch.randn(clip, 24, 128, generator=g).to(BF).cuda()
    full = ops.attention(q, ops.KvCacheView(k_all.contiguous(), v_all.contiguous()), s)
    for rank in (0, 5):
        part = ops.attention(q[:, 3 * rank:3 * rank + 3].contiguous(),
                             ops.KvCacheView(k_all[:, rank:rank + 1].contiguous(), v_all[:, rank:rank + 1].contiguous()), s)
        # the 3-head launch fills the chip by splitting the key range (fp32 partials + merge): a different summation order of
        # the same bf16-P products, i.e. the noise of two flash-attention evaluations (~1.5e-3 rel-L2 on random data)
        assert rel_l2(part, full[:, 3 * rank:3 * rank + 3]) < 2e-3
        assert_bf16_parity(part, full[:, 3 * rank:3 * rank + 3], max_ulp=4, max_mismatch_frac=0.3, rel=2e-3, what="rank view")


# ---- the layer under cp_ulysses: 2 ranks on one GPU (gloo rendezvous; device tensors staged through the host by the exchange) --
def _cp_worker(rank, world, port, ret, strategy="cp_ulysses"):
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import dataclasses
        torch.cuda.set_device(0)
        from inferix_amd.magi import context_parallel as cpl
        from inferix_amd.magi.dit import HipMagiTransformerLayer
        from inferix_amd.magi.types import InferenceParams
        cpl.set_cp_group(dist.group.WORLD)
        fx = golden("magi_block_tiny.npz")
        cfg, n_layers, clip, n_calls, wseed, max_tokens = MB.fixture_geometry(fx)
        mc, ec = _configs(cfg, n_layers, cp_strategy=strategy)
        ec.cp_size = world
        layers = []
        for li in range(n_layers):
            layer = HipMagiTransformerLayer(mc, ec, li, "cuda")
            layer.load_state_dict(MB.init_layer_weights(cfg, wseed + li))
            layers.append(layer)
        ip = InferenceParams(1, max_tokens)
        outs = []
        for ci in range(3):                          # store two chunks / prefix + nearly-clean rule / read-only window
            inp, m = MB.fixture_call(fx, ci)
            meta = _meta(m)
            ardf = dict(denoising_range_num=meta.denoising_range_num, q_range=meta.core_attn_params.q_range,
                        k_range=meta.core_attn_params.k_range, max_seqlen_q=meta.core_attn_params.max_seqlen_q,
                        max_seqlen_k=meta.core_attn_params.max_seqlen_k)
            x, cmap, rope, pad, sizes, core_p, cross_p = cpl.cp_pre_process(
                world, strategy, inp["x"].cuda(), inp["condition_map"].cuda(), inp["rope"].cuda(), None, ardf,
                meta.core_attn_params, meta.cross_attn_params)
            meta = dataclasses.replace(meta, cp_pad_size=pad, cp_split_sizes=sizes, core_attn_params=core_p, cross_attn_params=cross_p)
            ip.update_kv_cache = m.update_kv_cache
            for layer in layers:
                x = layer(x, inp["condition"].cuda(), cmap, inp["y"].cuda(), rope, ip, meta)
            if strategy == "cp_shuffle_overlap":      # the shuffled shards are re-assembled by the strategy's own gather
                x = cpl.cp_post_process(world, strategy, x.cpu(), meta)
            outs.append(x.cpu())
        torch.cuda.synchronize()
        ret[rank] = outs
    finally:
        dist.destroy_process_group()


def test_layer_under_cp_ulysses_two_ranks_matches_single_device_golden():
    """The algebraic identity that pins the context-parallel path: the rank-order concatenation of the 2 ranks'
    outputs equals the single-device reference output.  4 q-heads / 2 kv-heads over 2 ranks = 2 q-heads on 1 kv-head each,
    head-sharded cache, prefix read in place on the second and third forward."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    with mp.get_context("spawn").Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_cp_worker, args=(world, port, ret), nprocs=world, join=True)
        outs = [ret[r] for r in range(world)]
    fx = golden("magi_block_tiny.npz")
    for ci in range(3):
        got = torch.cat([outs[r][ci] for r in range(world)], dim=0)
        ref = fx[f"c{ci}_out_l1"]
        assert got.shape == ref.shape
        assert rel_l2(got, ref) < 1e-2, (ci, rel_l2(got, ref))


def test_layer_under_cp_shuffle_overlap_two_ranks_matches_single_device_golden():
    """The reference's second context-parallel strategy on the HIP layer: every rank holds a slice of EVERY denoising chunk
    (padded to a multiple of the ranks), K/V in one message, queries / outputs chunk by chunk under the attention of the
    neighbouring chunk; gathered with the strategy's own `cp_post_process`, the result equals the single-device reference."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    with mp.get_context("spawn").Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_cp_worker, args=(world, port, ret, "cp_shuffle_overlap"), nprocs=world, join=True)
        outs = [ret[r] for r in range(world)]
    fx = golden("magi_block_tiny.npz")
    for ci in range(3):
        ref = fx[f"c{ci}_out_l1"]
        for r in range(world):
            got = outs[r][ci]
            assert got.shape == ref.shape
            assert rel_l2(got, ref) < 1e-2, (ci, r, rel_l2(got, ref))


def test_fp8_layer_shares_one_quantised_copy_between_linears_with_the_same_input_scale():
    """q / qx / k / v quantise the same LayerNorm row; a checkpoint that carries the same `input_scale` vector for several of them gets
    ONE e4m3 copy of the row for those (fewer bytes written by the fused LayerNorm + quantiser), with bit-identical layer output."""
    from inferix_amd.magi.dit import HipMagiTransformerLayer, synthetic_layer_state_dict
    from inferix_amd.magi.types import InferenceParams
    mc = SimpleNamespace(num_layers=3, hidden_size=512, ffn_hidden_size=1024, num_attention_heads=4, num_query_groups=2, kv_channels=128,
                         layernorm_epsilon=1e-6, apply_layernorm_1p=True, gated_linear_unit=False, cond_hidden_ratio=0.25,
                         xattn_cond_hidden_ratio=1.0, cond_gating_ratio=1.0)
    ec = SimpleNamespace(cp_size=1, cp_strategy="none", fp8_quant=True, kv_offload=False, ulysses_overlap_degree=1)
    sd = synthetic_layer_state_dict(mc, seed=3, device="cuda", fp8=True)
    for nm in ("qx", "k"):
        sd[f"self_attention.linear_qkv.{nm}.input_scale"] = sd["self_attention.linear_qkv.q.input_scale"].clone()
    layer = HipMagiTransformerLayer(mc, ec, 1, "cuda")
    layer.load_state_dict(sd)
    sa = layer.self_attention
    assert sa.qkv_divisors.shape[0] == 2 and sa.qkv_slot == {"q": 0, "qx": 0, "k": 0, "v": 1}
    ref = HipMagiTransformerLayer(mc, ec, 1, "cuda")
    ref.load_state_dict(sd)
    h = ref.self_attention.fp8["q"].in_features           # the same weights with the sharing switched off: four copies
    ref.self_attention.qkv_divisors = torch.stack([ref.self_attention.fp8[nm].divisor.expand(h) for nm in ("q", "qx", "k", "v")]).contiguous()
    ref.self_attention.qkv_slot = {"q": 0, "qx": 1, "k": 2, "v": 3}
    g = torch.Generator(device="cuda").manual_seed(1)
    s_len, clip = 96, 96
    x = torch.randn(s_len, 1, 512, generator=g, device="cuda").to(BF)
    cond = torch.randn(1, 1, 128, generator=g, device="cuda").to(BF)
    cmap = torch.zeros(s_len, 1, dtype=torch.int32, device="cuda")
    y = torch.randn(7, 512, generator=g, device="cuda").to(BF)
    ang = torch.rand(s_len, 48, generator=g, device="cuda") * 6.0
    rope = torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)
    from inferix_amd.magi.types import ModelMetaArgs, PackedCoreAttnParams, PackedCrossAttnParams
    qr = torch.tensor([[0, clip]], dtype=torch.int32)
    core = PackedCoreAttnParams(q_range=qr, k_range=qr, np_q_range=qr.numpy(), np_k_range=qr.numpy(), max_seqlen_q=clip, max_seqlen_k=clip)
    cross = PackedCrossAttnParams(q_ranges=qr, kv_ranges=torch.tensor([[0, 7]], dtype=torch.int32), cu_seqlens_q=torch.tensor([0, clip], dtype=torch.int32),
                                  cu_seqlens_kv=torch.tensor([0, 7], dtype=torch.int32), max_seqlen_q=clip, max_seqlen_kv=7)
    meta = ModelMetaArgs(H=8, W=12, cp_pad_size=0, cp_split_sizes=[s_len], slice_point=0, denoising_range_num=1, range_num=1,
                         extract_prefix_video_feature=False, fwd_extra_1st_chunk=False, distill_nearly_clean_chunk=False, clip_token_nums=clip,
                         enable_cuda_graph=False, core_attn_params=core, cross_attn_params=cross)
    out = layer(x, cond, cmap, y, rope, None, meta)
    want = ref(x, cond, cmap, y, rope, None, meta)
    assert torch.isfinite(out.float()).all() and torch.equal(out, want)


def test_fp8_quant_layer_stack_vs_reference_golden():
    """`engine_config.fp8_quant` (the 4.5B distill-quant config of BASELINE config 5): three layers, the middle one on the
    static-scale FP8 linears (q / qx / k / v / fc1 per-tensor form, linear_proj / fc2 per-channel form, dit_module.py:408-413,
    :434-490, :526-539, :867).  Every layer gets the REFERENCE's input of that layer (teacher forcing).  Yardstick for the FP8
    layer: an e4m3 code flips when a bf16 rounding in front of `div_clamp_to` differs, and the layer re-quantises five times, so
    two correct evaluations are far apart by bf16 standards — the reference layer ITSELF (the oracle, bit-identical to it) moves
    by 1.7e-2 rel-L2 when 5 % of its input elements move by one bf16 ULP; the HIP layer must be closer to the reference than that."""
    from inferix_amd.magi.dit import HipMagiTransformerLayer
    from inferix_amd.magi.types import InferenceParams
    fx = golden("magi_block_fp8_tiny.npz")
    assert int(fx["fp8_quant"]) == 1
    cfg, n_layers, clip, n_calls, wseed, max_tokens = MB.fixture_geometry(fx)
    mc, ec = _configs(cfg, n_layers)
    ec.fp8_quant = True
    layers, Ws = [], []
    for li in range(n_layers):
        layer = HipMagiTransformerLayer(mc, ec, li, "cuda")
        Ws.append(MB.init_layer_weights(cfg, wseed + li, fp8=MB.layer_is_fp8(li, mc.num_layers)))
        layer.load_state_dict(Ws[li])
        assert bool(layer.fp8) == (li == 1) and bool(layer.self_attention.fp8) == (li == 1)
        layers.append(layer)
    with pytest.raises(ValueError, match="FP8"):           # a bf16 checkpoint in a layer the config says is quantised
        HipMagiTransformerLayer(mc, ec, 1, "cuda").load_state_dict(MB.init_layer_weights(cfg, wseed + 1))
    ip = InferenceParams(1, max_tokens)
    g = torch.Generator().manual_seed(0)
    for ci in range(n_calls):
        inp, m = MB.fixture_call(fx, ci)
        meta = _meta(m)
        ip.update_kv_cache = m.update_kv_cache
        x_in = inp["x"]
        for li, layer in enumerate(layers):
            ref = fx[f"c{ci}_out_l{li}"]
            got = layer(x_in.cuda(), inp["condition"].cuda(), inp["condition_map"].cuda(), inp["y"].cuda(), inp["rope"].cuda(), ip, meta)
            r = rel_l2(got.cpu(), ref)
            bar = 6e-3
            if li == 1 and ci == 0:                          # the reference layer's own response to a 1-ULP nudge of 5 % of its input
                xi = x_in.view(torch.int16).clone()
                mask = torch.rand(x_in.shape, generator=g) < 0.05
                xi[mask] += (torch.randint(0, 2, x_in.shape, generator=g) * 2 - 1).to(torch.int16)[mask]
                cache = MB.MagiLayerCache(max_tokens, cfg.num_query_groups, cfg.kv_channels)
                nudged = MB.layer_forward(Ws[1], cfg, xi.view(BF), inp["condition"], inp["condition_map"], inp["y"], inp["rope"], m, cache)
                bar = rel_l2(nudged, ref)
                assert bar > 8e-3, bar                       # the yardstick really is of that size
            elif li == 1:
                bar = 2e-2
            print(f"fp8 stack call {ci} layer {li} ({'fp8' if li == 1 else 'bf16'}): hip-vs-ref {r:.3e}  (bar {bar:.3e})")
            assert r < bar, (ci, li, r, bar)
            x_in = ref
    written = int(fx["cache_written"])
    raw = ip.kv_cache_manager.get_raw(ip.kv_cache_request, "layer_1")           # K / V of the FP8 layer: fp8 GEMM outputs
    assert rel_l2(raw[:, :written].cpu(), fx["cache_l1"][:, :written]) < 1.5e-2


def test_kv_split_rows_is_the_four_copies():
    """ifx_kv_split_rows (round 5): the K | V rows of the all-to-all message into the cache planes under the store rule's two destination
    runs, in one launch — bit-identical to the four strided copies it replaces, nothing else of the cache touched, ragged splits included."""
    from inferix_amd import hip_ops as ops
    g = torch.Generator().manual_seed(11)
    for n, heads, row0, split, row1, slots in ((333, 1, 100, 200, 700, 900), (64, 2, 0, 64, 64, 128), (50, 3, 10, 0, 20, 80), (7, 1, 5, 7, 0, 16)):
        kv = torch.randn(n, heads, 256, generator=g).to(BF).cuda()
        kc = torch.full((slots, heads, 128), 3.0, dtype=BF, device="cuda")
        vc = torch.full((slots, heads, 128), 5.0, dtype=BF, device="cuda")
        kr, vr = kc.clone(), vc.clone()
        kr[row0:row0 + split] = kv[:split, :, :128]
        vr[row0:row0 + split] = kv[:split, :, 128:]
        kr[row1:row1 + n - split] = kv[split:, :, :128]
        vr[row1:row1 + n - split] = kv[split:, :, 128:]
        ops.kv_split_rows(kv, kc, vc, row0, split, row1)
        assert torch.equal(kc, kr) and torch.equal(vc, vr), (n, heads, row0, split, row1)
