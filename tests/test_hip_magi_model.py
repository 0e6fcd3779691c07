"""GPU: MAGI-1 as a MODEL step (BASELINE config 5; round-2 verdict missing #1) — `HipVideoDiTModel.forward` (embedders, rope table,
the HIP layer stack with its KV cache, final LayerNorm, final linear, unpatchify) against tests/golden/magi_model_tiny.npz, which the
reference's own `VideoDiTModel.forward` produced (oracle/gen_golden_magi_model.py)."""
from types import SimpleNamespace

import pytest
import torch

import magi_block_oracle as MB
import magi_model_oracle as MM
from fixture_io import golden
from util import rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _config(cfg: MM.MagiModelConfig):
    L = cfg.layer
    mc = SimpleNamespace(num_layers=cfg.num_layers, hidden_size=L.hidden_size, ffn_hidden_size=L.ffn_hidden_size,
                         num_attention_heads=L.num_attention_heads, num_query_groups=L.num_query_groups, kv_channels=L.kv_channels,
                         layernorm_epsilon=L.layernorm_epsilon, apply_layernorm_1p=L.apply_layernorm_1p,
                         gated_linear_unit=L.gated_linear_unit, params_dtype=BF, patch_size=cfg.patch_size, t_patch_size=cfg.t_patch_size,
                         in_channels=cfg.in_channels, out_channels=cfg.out_channels, caption_channels=cfg.caption_channels,
                         caption_max_length=cfg.caption_max_length, cond_hidden_ratio=L.cond_hidden_ratio,
                         xattn_cond_hidden_ratio=L.xattn_cond_hidden_ratio, cond_gating_ratio=L.cond_gating_ratio,
                         x_rescale_factor=cfg.x_rescale_factor, half_channel_vae=cfg.half_channel_vae)
    ec = SimpleNamespace(cp_size=1, cp_strategy="none", fp8_quant=False, kv_offload=False, ulysses_overlap_degree=1, distill=False)
    return SimpleNamespace(model_config=mc, engine_config=ec, runtime_config=None)


def _exact_model(cfg, EW, Ws, fx, n_calls):
    """The same forwards with every layer in float64 (magi_block_oracle.exact_layer_forward) between the fp32 embedders: the yardstick."""
    L = cfg.layer
    caches = [None] * cfg.num_layers
    outs = []
    exact_caches = [MB.MagiLayerCache(int(fx["geom"][5]), L.num_query_groups, L.kv_channels) for _ in range(cfg.num_layers)]
    for ci in range(n_calls):
        kw = dict(zip(("range_num", "dn", "slice_point", "fe", "di", "update"), [int(v) for v in fx[f"c{ci}_flags"]]))
        x, cond, cmap, yf, rope, meta = MM.pre_process(EW, cfg, fx[f"c{ci}_in_x"], fx[f"c{ci}_in_t"], fx[f"c{ci}_in_y"], fx[f"c{ci}_in_mask"],
                                                       fx[f"c{ci}_in_kv_range"], fx[f"c{ci}_in_drop"], range_num=kw["range_num"],
                                                       denoising_range_num=kw["dn"], slice_point=kw["slice_point"])
        lm = MB.LayerMeta(q_ranges=[tuple(r) for r in meta["q_range"].tolist()], k_ranges=[tuple(r) for r in fx[f"c{ci}_in_kv_range"].tolist()],
                          cu_seqlens_q=meta["cu_seqlens_q"].tolist(), cu_seqlens_kv=meta["cu_seqlens_kv"].tolist(),
                          clip_token_nums=meta["clip_token_nums"], slice_point=kw["slice_point"], update_kv_cache=bool(kw["update"]),
                          use_cache=bool(kw["fe"]) or kw["slice_point"] > 0, distill_nearly_clean_chunk=bool(kw["di"]))
        # exact_layer_forward READS the cache (never writes): the bf16 restatement runs beside it and keeps the caches as the reference
        # has them — exact layer first (the prefix as stored before this forward), then the bf16 layer (which stores this forward's rows)
        h, h_bf = x, x
        for W, cache in zip(Ws, exact_caches):
            h = MB.exact_layer_forward(W, L, h, cond, cmap, yf, rope, lm, cache)
            h_bf = MB.layer_forward(W, L, h_bf, cond, cmap, yf, rope, lm, cache)
        outs.append(MM.post_process(EW, cfg, h.float(), meta["H"], meta["W"]))
    return outs


def test_model_forward_vs_reference_golden():
    """Three forwards that walk the cache rule (first two chunks / prefix + nearly-clean / read-only window), each with its own
    timesteps, captions and caption-dropout flag.  The pre-processing tensors must equal the reference's (bit-exact for the integer
    maps, bf16-rounded fp32 math within one ULP for the embeddings); the model output is held to the bf16 floor: the reference's own
    result sits `floor` from the float64-layer evaluation, the HIP result has to be within floor x 1.25 (+ eps) of both."""
    from inferix_amd.magi.model import HipVideoDiTModel
    from inferix_amd.magi.types import InferenceParams
    fx = golden("magi_model_tiny.npz")
    n_layers, clip, n_calls, wseed, eseed, max_tokens = [int(v) for v in fx["geom"]]
    cfg = MM.tiny_model_config()
    EW = MM.init_embedder_weights(cfg, eseed)
    Ws = [MB.init_layer_weights(cfg.layer, wseed + li) for li in range(n_layers)]
    sd = dict(EW)
    for li, W in enumerate(Ws):
        sd.update({f"videodit_blocks.layers.{li}.{k}": v for k, v in W.items()})
    model = HipVideoDiTModel(_config(cfg), "cuda")
    model.load_state_dict(sd)
    ip = InferenceParams(1, max_tokens)
    exact = _exact_model(cfg, EW, Ws, fx, n_calls)
    for ci in range(n_calls):
        range_num, dn, sp, fe, di, upd = [int(v) for v in fx[f"c{ci}_flags"]]
        kw = dict(range_num=range_num, denoising_range_num=dn, slice_point=sp, fwd_extra_1st_chunk=bool(fe), distill_nearly_clean_chunk=bool(di))
        args = [fx[f"c{ci}_in_{k}"].cuda() for k in ("x", "t", "y", "drop", "mask", "kv_range")]
        pre = model.forward_pre_process(*args, **kw)
        for nm, got in zip(("x", "condition", "condition_map", "y_xattn_flat", "rope"), pre[:5]):
            want = fx[f"c{ci}_pre_{nm}"]
            if nm == "condition_map":
                assert torch.equal(got.cpu().long(), want.long()), (ci, nm)
            else:
                assert got.shape == want.shape and got.dtype == want.dtype, (ci, nm, got.dtype, want.dtype)
                assert rel_l2(got.cpu(), want) < (2e-3 if got.dtype == BF else 1e-5), (ci, nm, rel_l2(got.cpu(), want))
        meta = pre[5]
        assert meta.core_attn_params.np_q_range.tolist() == [[i * clip, (i + 1) * clip] for i in range(dn)]
        assert meta.cross_attn_params.cu_seqlens_kv.tolist() == [0] + fx[f"c{ci}_in_mask"].reshape(dn, -1).sum(-1).cumsum(0).int().tolist()
        ip.update_kv_cache = bool(upd)
        out = model(*args, inference_params=ip, **kw).cpu()
        ref = fx[f"c{ci}_out"]
        assert out.shape == ref.shape and out.dtype == torch.float32
        floor, mine, r = rel_l2(ref, exact[ci]), rel_l2(out, exact[ci]), rel_l2(out, ref)
        print(f"magi model call {ci}: floor (reference vs float64 layers) {floor:.3e}; HIP vs float64 {mine:.3e}; HIP vs reference {r:.3e}")
        assert mine <= 1.25 * floor + 5e-4 and r <= 1.25 * floor + 5e-4, (ci, floor, mine, r)
    with pytest.raises(ValueError):
        model.forward_pre_process(args[0], args[1], args[2], None, args[4], args[5], **kw)
