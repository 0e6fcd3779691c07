"""CPU: the MAGI model-level restatement (oracle/magi_model_oracle.py + magi_block_oracle.py) against the golden the reference's own
`VideoDiTModel.forward` produced (oracle/gen_golden_magi_model.py): every pre-processing tensor and every model output, bit for bit."""
import torch

import magi_block_oracle as MB
import magi_model_oracle as MM
from fixture_io import golden


def _call(fx, ci):
    kw = dict(zip(("range_num", "denoising_range_num", "slice_point", "fwd_extra_1st_chunk", "distill_nearly_clean_chunk", "update"),
                  [int(v) for v in fx[f"c{ci}_flags"]]))
    return {k: fx[f"c{ci}_in_{k}"] for k in ("x", "t", "y", "mask", "kv_range", "drop")}, kw


def test_model_oracle_matches_reference_golden():
    fx = golden("magi_model_tiny.npz")
    n_layers, clip, n_calls, wseed, eseed, max_tokens = [int(v) for v in fx["geom"]]
    cfg = MM.tiny_model_config()
    assert cfg.num_layers == n_layers
    L = cfg.layer
    EW = MM.init_embedder_weights(cfg, eseed)
    Ws = [MB.init_layer_weights(L, wseed + li) for li in range(n_layers)]
    caches = [MB.MagiLayerCache(max_tokens, L.num_query_groups, L.kv_channels) for _ in range(n_layers)]
    for ci in range(n_calls):
        c, kw = _call(fx, ci)
        x, cond, cmap, yf, rope, meta = MM.pre_process(EW, cfg, c["x"], c["t"], c["y"], c["mask"], c["kv_range"], c["drop"],
                                                       range_num=kw["range_num"], denoising_range_num=kw["denoising_range_num"],
                                                       slice_point=kw["slice_point"])
        for nm, got in (("x", x), ("condition", cond), ("condition_map", cmap), ("y_xattn_flat", yf), ("rope", rope)):
            assert torch.equal(got.to(fx[f"c{ci}_pre_{nm}"].dtype), fx[f"c{ci}_pre_{nm}"]), (ci, nm)
        lm = MB.LayerMeta(q_ranges=[tuple(r) for r in meta["q_range"].tolist()], k_ranges=[tuple(r) for r in c["kv_range"].tolist()],
                          cu_seqlens_q=meta["cu_seqlens_q"].tolist(), cu_seqlens_kv=meta["cu_seqlens_kv"].tolist(),
                          clip_token_nums=meta["clip_token_nums"], slice_point=kw["slice_point"], update_kv_cache=bool(kw["update"]),
                          use_cache=bool(kw["fwd_extra_1st_chunk"]) or kw["slice_point"] > 0,
                          distill_nearly_clean_chunk=bool(kw["distill_nearly_clean_chunk"]))
        h = x
        for W, cache in zip(Ws, caches):
            h = MB.layer_forward(W, L, h, cond, cmap, yf, rope, lm, cache)
        out = MM.post_process(EW, cfg, h, meta["H"], meta["W"])
        assert out.shape == fx[f"c{ci}_out"].shape and torch.equal(out, fx[f"c{ci}_out"]), ci


def test_product_model_host_side_on_cpu_vs_reference_golden():
    """The PRODUCT model's own pre- and post-processing (inferix_amd/magi/model.py: patch embedding as a linear, rope table, timestep /
    caption embedders, condition map, packed ranges; final LayerNorm + linear + unpatchify) run on the CPU — they are torch glue, only the
    layers need the HIP library — against the reference's tensors: integer maps bit for bit, fp32 embedders to 1e-5 before their one
    rounding to bf16, and the post-processing of the oracle's last hidden states against the reference's model output."""
    from types import SimpleNamespace
    from inferix_amd.magi.model import HipVideoDiTModel
    fx = golden("magi_model_tiny.npz")
    n_layers, clip, n_calls, wseed, eseed, max_tokens = [int(v) for v in fx["geom"]]
    cfg = MM.tiny_model_config()
    L = cfg.layer
    mc = SimpleNamespace(num_layers=0, hidden_size=L.hidden_size, ffn_hidden_size=L.ffn_hidden_size, num_attention_heads=L.num_attention_heads,
                         num_query_groups=L.num_query_groups, kv_channels=L.kv_channels, layernorm_epsilon=L.layernorm_epsilon,
                         apply_layernorm_1p=L.apply_layernorm_1p, patch_size=cfg.patch_size, t_patch_size=cfg.t_patch_size,
                         in_channels=cfg.in_channels, out_channels=cfg.out_channels, caption_channels=cfg.caption_channels,
                         caption_max_length=cfg.caption_max_length, cond_hidden_ratio=L.cond_hidden_ratio,
                         xattn_cond_hidden_ratio=L.xattn_cond_hidden_ratio, x_rescale_factor=cfg.x_rescale_factor, half_channel_vae=cfg.half_channel_vae)
    ec = SimpleNamespace(cp_size=1, cp_strategy="none", fp8_quant=False, distill=False)
    model = HipVideoDiTModel(SimpleNamespace(model_config=mc, engine_config=ec, runtime_config=None), "cpu")     # no layers: host side only
    EW = MM.init_embedder_weights(cfg, eseed)
    model.load_state_dict(EW)
    Ws = [MB.init_layer_weights(L, wseed + li) for li in range(n_layers)]
    caches = [MB.MagiLayerCache(max_tokens, L.num_query_groups, L.kv_channels) for _ in range(n_layers)]
    for ci in range(n_calls):
        c, kw = _call(fx, ci)
        kwargs = dict(range_num=kw["range_num"], denoising_range_num=kw["denoising_range_num"], slice_point=kw["slice_point"],
                      fwd_extra_1st_chunk=bool(kw["fwd_extra_1st_chunk"]), distill_nearly_clean_chunk=bool(kw["distill_nearly_clean_chunk"]))
        x, cond, cmap, yf, rope, meta = model.forward_pre_process(c["x"], c["t"], c["y"], c["drop"], c["mask"], c["kv_range"], **kwargs)
        assert torch.equal(cmap.long(), fx[f"c{ci}_pre_condition_map"].long())
        for nm, got in (("x", x), ("condition", cond), ("y_xattn_flat", yf), ("rope", rope)):
            want = fx[f"c{ci}_pre_{nm}"]
            assert got.shape == want.shape and got.dtype == want.dtype, (ci, nm)
            # bf16 tensors: the fp32 values agree to ~1e-6, so at most a rounding tie in a few elements flips
            frac = float((got != want).float().mean())
            assert frac < (2e-3 if got.dtype == torch.bfloat16 else 1.0) and torch.allclose(got.float(), want.float(), rtol=1e-2 if got.dtype == torch.bfloat16 else 1e-5, atol=1e-5), (ci, nm, frac)
        dn = kw["denoising_range_num"]
        assert meta.core_attn_params.np_q_range.tolist() == [[i * clip, (i + 1) * clip] for i in range(dn)]
        assert meta.core_attn_params.np_k_range.tolist() == c["kv_range"].tolist()
        assert meta.cross_attn_params.cu_seqlens_kv.tolist() == [0] + c["mask"].reshape(dn, -1).sum(-1).cumsum(0).int().tolist()
        assert (meta.slice_point, meta.range_num, meta.denoising_range_num, meta.clip_token_nums) == (kw["slice_point"], kw["range_num"], dn, clip)
        # the oracle's layers between the product's two halves
        lm = MB.LayerMeta(q_ranges=[tuple(r) for r in meta.core_attn_params.np_q_range.tolist()], k_ranges=[tuple(r) for r in c["kv_range"].tolist()],
                          cu_seqlens_q=meta.cross_attn_params.cu_seqlens_q.tolist(), cu_seqlens_kv=meta.cross_attn_params.cu_seqlens_kv.tolist(),
                          clip_token_nums=clip, slice_point=kw["slice_point"], update_kv_cache=bool(kw["update"]),
                          use_cache=bool(kw["fwd_extra_1st_chunk"]) or kw["slice_point"] > 0,
                          distill_nearly_clean_chunk=bool(kw["distill_nearly_clean_chunk"]))
        h = fx[f"c{ci}_pre_x"]
        for W, cache in zip(Ws, caches):
            h = MB.layer_forward(W, L, h, fx[f"c{ci}_pre_condition"], fx[f"c{ci}_pre_condition_map"], fx[f"c{ci}_pre_y_xattn_flat"], fx[f"c{ci}_pre_rope"], lm, cache)
        out = model.forward_post_process(h, meta)
        want = fx[f"c{ci}_out"]
        assert out.shape == want.shape and torch.allclose(out, want, rtol=1e-4, atol=1e-5), (ci, float((out - want).abs().max()))
