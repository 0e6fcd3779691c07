#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/magi_model_tiny.npz by running the REFERENCE's own `VideoDiTModel.forward`
(inferix/models/magi/dit/dit_model.py:337-362: forward_pre_process -> TransformerBlock -> forward_post_process) on CPU, one rank,
cp_strategy "none", with the reference's `MagiKVCacheManager` behind the layers.  Stand-ins as in gen_golden_magi_block.py
(oracle/_refstub.install_magi); additionally the `torch.autocast("cuda", dtype=torch.float32)` regions of the pre / post processing —
on a GPU they cast the inputs of the fp32 embedders' linear / conv calls up to fp32, on a CPU they are no-ops and the embedders
cannot run — are emulated by doing exactly that cast around F.linear / F.conv3d inside forward_pre_process / forward_post_process,
and `torch.cuda.current_device()` answers "cpu" for the rope table.  Only data is stored; the weights are
regenerated from seeds (magi_block_oracle.init_layer_weights per layer, magi_model_oracle.init_embedder_weights for the rest).

Three forwards at tiny dimensions (hidden 256, 4 q-heads on 2 kv-groups, 3 layers, 2 x 2 x 2 patches of a 16-channel 8 x 12 latent):
  0: first two chunks (fwd_extra_1st_chunk, cache update), two denoising ranges with their own timesteps and captions
  1: prefix of one clean chunk + two denoising ranges, nearly-clean rule
  2: one range over a key window, read-only
After writing, the restatement (magi_model_oracle + magi_block_oracle) is checked against every stored tensor, bit for bit.
usage (build container only; /root/reference must exist):  python oracle/gen_golden_magi_model.py
"""
from __future__ import annotations

import contextlib
import importlib
import os
import sys
import warnings

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refstub  # noqa: E402
import magi_block_oracle as MB  # noqa: E402
import magi_model_oracle as MM  # noqa: E402
from fixture_io import GOLDEN_DIR, save_npz  # noqa: E402

BF = torch.bfloat16
WSEED, ESEED = 900, 901


@contextlib.contextmanager
def fp32_autocast(*a, **k):
    """What torch.autocast("cuda", dtype=torch.float32) does to the ops of the embedders on a GPU: linear / conv inputs -> fp32."""
    lin, conv = F.linear, F.conv3d
    up = lambda t: t.float() if isinstance(t, torch.Tensor) and t.is_floating_point() else t
    F.linear = lambda x, w, b=None: lin(up(x), up(w), up(b))
    F.conv3d = lambda x, w, b=None, *aa, **kk: conv(up(x), up(w), up(b), *aa, **kk)
    try:
        yield
    finally:
        F.linear, F.conv3d = lin, conv


def calls(cfg: MM.MagiModelConfig, seed: int):
    g = torch.Generator().manual_seed(seed)
    Hl, Wl, Lc, Cc = 8, 12, cfg.caption_max_length, cfg.caption_channels
    clip = (Hl // cfg.patch_size) * (Wl // cfg.patch_size)           # one latent frame per denoising range
    out = []

    def mk(ranges, kv, caps, drop, **kw):
        x = torch.randn(1, cfg.in_channels, ranges, Hl, Wl, generator=g)
        t = torch.rand(1, ranges, generator=g)
        y = torch.randn(ranges, 1, Lc, Cc, generator=g)
        mask = torch.zeros(ranges, 1, 1, Lc)
        for r, c in enumerate(caps):
            mask[r, ..., :c] = 1
        out.append(dict(x=x, t=t, y=y, mask=mask, kv_range=torch.tensor(kv, dtype=torch.int32), drop=torch.tensor([drop]), kw=kw))
    mk(2, [(0, clip), (0, 2 * clip)], (7, 5), False, range_num=2, denoising_range_num=2, slice_point=0, fwd_extra_1st_chunk=True, update=True)
    mk(2, [(0, 2 * clip), (0, 3 * clip)], (12, 3), True, range_num=3, denoising_range_num=2, slice_point=1, fwd_extra_1st_chunk=False,
       distill_nearly_clean_chunk=True, update=True)
    mk(1, [(clip, 3 * clip)], (9,), False, range_num=3, denoising_range_num=1, slice_point=2, fwd_extra_1st_chunk=False, update=False)
    return out, clip


def build_reference(cfg: MM.MagiModelConfig, max_tokens: int):
    """-> (the reference's VideoDiTModel with the seeded weights, a factory of fresh InferenceParams with an empty CPU KV cache)."""
    dm = _refstub.import_magi_dit()
    torch.cuda.current_device = lambda: "cpu"
    M = importlib.import_module("inferix.models.magi.dit.dit_model")
    cfgm = importlib.import_module("inferix.core.config")
    mcfg = importlib.import_module("inferix.models.magi.config")
    types = importlib.import_module("inferix.core.types.inference")
    kvm = importlib.import_module("inferix.kvcache_manager.kvcache_manager")
    L = cfg.layer
    mc = cfgm.ModelConfig(model_name="golden", num_layers=cfg.num_layers, hidden_size=L.hidden_size, ffn_hidden_size=L.ffn_hidden_size,
                          num_attention_heads=L.num_attention_heads, num_query_groups=L.num_query_groups, kv_channels=L.kv_channels,
                          layernorm_epsilon=L.layernorm_epsilon, apply_layernorm_1p=L.apply_layernorm_1p, params_dtype=BF,
                          patch_size=cfg.patch_size, t_patch_size=cfg.t_patch_size, in_channels=cfg.in_channels,
                          out_channels=cfg.out_channels, cond_hidden_ratio=L.cond_hidden_ratio, caption_channels=cfg.caption_channels,
                          caption_max_length=cfg.caption_max_length, xattn_cond_hidden_ratio=L.xattn_cond_hidden_ratio,
                          cond_gating_ratio=L.cond_gating_ratio, gated_linear_unit=L.gated_linear_unit,
                          x_rescale_factor=cfg.x_rescale_factor, half_channel_vae=cfg.half_channel_vae)
    ec = cfgm.EngineConfig(cp_size=1, cp_strategy="none", fp8_quant=False, kv_offload=False)
    model = M.VideoDiTModel(mcfg.MagiConfig(model_config=mc, runtime_config=cfgm.RuntimeConfig(), engine_config=ec))
    M._high_precision_promoter(model)
    sd = dict(MM.init_embedder_weights(cfg, ESEED))
    for li in range(cfg.num_layers):
        for k, v in MB.init_layer_weights(L, WSEED + li).items():
            sd[f"videodit_blocks.layers.{li}.{k}"] = v
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for n, p in model.named_parameters():
        assert p.dtype == sd[n].dtype, (n, p.dtype, sd[n].dtype)
    model.eval()
    # The cast-up is applied around forward_pre_process / forward_post_process ONLY: those cannot run on a CPU at all without it (bf16
    # sinusoid into an fp32 Linear).  Inside the layers the reference's one autocast region (linear_proj of a non-quantised layer,
    # dit_module.py:1290-1293) stays what it is on the CPU path — disabled — as in tests/golden/magi_block_*.npz: the parity target is
    # the reference's CPU path (on a GPU that projection would additionally run in fp32).
    pre0, post0 = model.forward_pre_process, model.forward_post_process

    def pre1(*a, **k):
        with fp32_autocast():
            return pre0(*a, **k)

    def post1(*a, **k):
        with fp32_autocast():
            return post0(*a, **k)
    model.forward_pre_process, model.forward_post_process = pre1, post1
    count = [0]

    def make_ip():
        ip = object.__new__(types.InferenceParams)
        ip.max_sequence_length, ip.max_batch_size, ip.sequence_len_offset = max_tokens, 1, 0
        count[0] += 1
        ip.kv_cache_request = kvm.KVCacheRequest(request_id=f"magi{count[0]}")
        ip.kv_cache_manager = kvm.KVCacheManager(device="cpu")
        ip.key_value_memory_dict, ip.update_kv_cache = {}, False
        return ip
    return model, make_ip


def run_reference(cfg: MM.MagiModelConfig, cs, max_tokens: int):
    model, make_ip = build_reference(cfg, max_tokens)
    ip = make_ip()
    outs, pres = [], []
    for c in cs:
        kw = dict(c["kw"])
        ip.update_kv_cache = kw.pop("update")
        with torch.no_grad():
            pre = model.forward_pre_process(c["x"], c["t"], c["y"], c["drop"], c["mask"], c["kv_range"], **kw)
            pres.append(pre)
            outs.append(model(c["x"], c["t"], c["y"], c["drop"], c["mask"], c["kv_range"], inference_params=ip, **kw))
    return outs, pres


def run_oracle(cfg: MM.MagiModelConfig, cs, max_tokens: int):
    L = cfg.layer
    EW = MM.init_embedder_weights(cfg, ESEED)
    Ws = [MB.init_layer_weights(L, WSEED + li) for li in range(cfg.num_layers)]
    caches = [MB.MagiLayerCache(max_tokens, L.num_query_groups, L.kv_channels) for _ in range(cfg.num_layers)]
    outs, pres = [], []
    for c in cs:
        kw = c["kw"]
        x, cond, cmap, yf, rope, meta = MM.pre_process(EW, cfg, c["x"], c["t"], c["y"], c["mask"], c["kv_range"], c["drop"], range_num=kw["range_num"],
                                                       denoising_range_num=kw["denoising_range_num"], slice_point=kw["slice_point"])
        pres.append((x, cond, cmap, yf, rope))
        lm = MB.LayerMeta(q_ranges=[tuple(r) for r in meta["q_range"].tolist()], k_ranges=[tuple(r) for r in c["kv_range"].tolist()],
                          cu_seqlens_q=meta["cu_seqlens_q"].tolist(), cu_seqlens_kv=meta["cu_seqlens_kv"].tolist(),
                          clip_token_nums=meta["clip_token_nums"], slice_point=kw["slice_point"], update_kv_cache=kw["update"],
                          use_cache=kw["fwd_extra_1st_chunk"] or kw["slice_point"] > 0,
                          distill_nearly_clean_chunk=kw.get("distill_nearly_clean_chunk", False))
        h = x
        for W, cache in zip(Ws, caches):
            h = MB.layer_forward(W, L, h, cond, cmap, yf, rope, lm, cache)
        outs.append(MM.post_process(EW, cfg, h, meta["H"], meta["W"]))
    return outs, pres


def main():
    if not _refstub.available():
        raise SystemExit("reference tree not available: fixtures can only be generated in the build container")
    warnings.filterwarnings("ignore")
    torch.manual_seed(0)
    cfg = MM.tiny_model_config()
    cs, clip = calls(cfg, 21)
    max_tokens = 4 * clip
    ref_outs, ref_pres = run_reference(cfg, cs, max_tokens)
    orc_outs, orc_pres = run_oracle(cfg, cs, max_tokens)
    fx = {"geom": torch.tensor([cfg.num_layers, clip, len(cs), WSEED, ESEED, max_tokens])}
    for ci, c in enumerate(cs):
        for k in ("x", "t", "y", "mask", "kv_range", "drop"):
            fx[f"c{ci}_in_{k}"] = c[k]
        kw = c["kw"]
        fx[f"c{ci}_flags"] = torch.tensor([kw["range_num"], kw["denoising_range_num"], kw["slice_point"], int(kw["fwd_extra_1st_chunk"]),
                                          int(kw.get("distill_nearly_clean_chunk", False)), int(kw["update"])])
        rx, rcond, rmap, ry, rrope, rmeta = ref_pres[ci]
        ox, ocond, omap, oy, orope = orc_pres[ci]
        for nm, a, b in (("x", rx, ox), ("condition", rcond, ocond), ("condition_map", rmap, omap), ("y_xattn_flat", ry, oy), ("rope", rrope, orope)):
            assert a.shape == b.shape and torch.equal(a, b.to(a.dtype)), (ci, nm, "oracle != reference", float((a.float() - b.float()).abs().max()))
            fx[f"c{ci}_pre_{nm}"] = a
        assert rmeta.core_attn_params.np_q_range.tolist() == [[i * clip, (i + 1) * clip] for i in range(kw["denoising_range_num"])]
        assert torch.equal(ref_outs[ci], orc_outs[ci]), (ci, "model output: oracle != reference", float((ref_outs[ci] - orc_outs[ci]).abs().max()))
        fx[f"c{ci}_out"] = ref_outs[ci]
    path = os.path.join(GOLDEN_DIR, "magi_model_tiny.npz")
    save_npz(path, fx)
    print(f"wrote {path} ({os.path.getsize(path)} bytes); oracle == reference on every pre-processing tensor and model output (bit-exact)")


if __name__ == "__main__":
    main()
