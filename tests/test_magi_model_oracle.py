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
