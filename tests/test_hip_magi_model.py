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


class _OracleModel:
    """The bf16 restatement behind the same `forward` / `forward_dispatcher` surface (CPU): what `ChunkSchedule.run` drives in the
    reference, with `HipVideoDiTModel`'s dispatcher wiring (a mirror of dit_model.py:537-594) borrowed unbound."""

    def __init__(self, cfg, EW, Ws, config, max_tokens):
        from inferix_amd.magi.model import HipVideoDiTModel
        self.cfg, self.EW, self.Ws = cfg, EW, Ws
        self.model_config, self.engine_config, self.runtime_config = config.model_config, config.engine_config, config.runtime_config
        self.device, self.patch_size, self.t_patch_size = torch.device("cpu"), cfg.patch_size, cfg.t_patch_size
        L = cfg.layer
        self.caches = [MB.MagiLayerCache(max_tokens, L.num_query_groups, L.kv_channels) for _ in Ws]
        self._dispatch = HipVideoDiTModel.forward_dispatcher
        self._uncond = HipVideoDiTModel.generate_kv_range_for_uncondition

    def generate_kv_range_for_uncondition(self, x):
        return self._uncond(self, x)

    def forward(self, x, t, y, caption_dropout_mask=None, xattn_mask=None, kv_range=None, inference_params=None, **kw):
        cfg = self.cfg
        xs, cond, cmap, yf, rope, meta = MM.pre_process(self.EW, cfg, x, t, y, xattn_mask, kv_range, caption_dropout_mask,
                                                        range_num=kw["range_num"], denoising_range_num=kw["denoising_range_num"],
                                                        slice_point=kw["slice_point"])
        lm = MB.LayerMeta(q_ranges=[tuple(r) for r in meta["q_range"].tolist()], k_ranges=[tuple(r) for r in kv_range.tolist()],
                          cu_seqlens_q=meta["cu_seqlens_q"].tolist(), cu_seqlens_kv=meta["cu_seqlens_kv"].tolist(),
                          clip_token_nums=meta["clip_token_nums"], slice_point=kw["slice_point"],
                          update_kv_cache=bool(inference_params.update_kv_cache),
                          use_cache=bool(kw["fwd_extra_1st_chunk"]) or kw["slice_point"] > 0 or bool(kw.get("extract_prefix_video_feature", False)),
                          distill_nearly_clean_chunk=bool(kw.get("distill_nearly_clean_chunk", False)))
        h = xs
        for W, cache in zip(self.Ws, self.caches):
            h = MB.layer_forward(W, cfg.layer, h, cond, cmap, yf, rope, lm, cache)
        return MM.post_process(self.EW, cfg, h.float(), meta["H"], meta["W"])

    def forward_dispatcher(self, **kw):
        return self._dispatch(self, **kw)


def test_chunk_schedule_rollout_vs_oracle_model():
    """MAGI's whole denoising loop at tiny dimensions: 3 chunks x 2 latent frames, 8 steps in a window of 4 -> 12 forwards through
    `ChunkSchedule.run` + `forward_dispatcher` (clean-chunk prefix forwards, nearly-clean re-forwards, cache writes every step), the
    HIP model on the GPU against the bf16 restatement on the CPU from the same noise.  What is compared is the integrated velocity
    (x_final - x_noise); twelve bf16 forwards feed each other through the latents and the KV cache, hence a few e-2."""
    from inferix_amd.magi.model import HipVideoDiTModel
    from inferix_amd.magi.schedule import ChunkSchedule
    from inferix_amd.magi.types import InferenceParams
    cfg = MM.tiny_model_config()
    EW = MM.init_embedder_weights(cfg, 5)
    Ws = [MB.init_layer_weights(cfg.layer, 40 + li) for li in range(cfg.num_layers)]
    config = _config(cfg)
    config.runtime_config = SimpleNamespace(cfg_number=1, noise2clean_kvrange=[4, 3, 2, 2], clean_chunk_kvrange=1, clean_t=0.9999)
    config.engine_config.shortcut_mode = "8,16,16"
    config.engine_config.distill_nearly_clean_chunk_threshold = 0.3
    chunk_num, cw, Hl, Wl = 3, 2, 8, 12
    tokens = cw * (Hl // 2) * (Wl // 2)
    g = torch.Generator().manual_seed(77)
    x0 = torch.randn(1, cfg.in_channels, chunk_num * cw, Hl, Wl, generator=g)
    x = torch.cat([x0, x0], 0)
    y = torch.randn(2, chunk_num, cfg.caption_max_length, cfg.caption_channels, generator=g)
    masks = torch.zeros(2, chunk_num, cfg.caption_max_length)
    masks[0, :, :7] = 1
    masks[1, :, :2] = 1
    sch = ChunkSchedule(8, 4, chunk_num, cw)
    seen = []
    om = _OracleModel(cfg, EW, Ws, config, chunk_num * tokens)
    want = sch.run(om, x.clone(), y, masks, SimpleNamespace(update_kv_cache=False), on_forward=seen.append)
    assert len(seen) == 12 and any(p.fwd_extra_1st_chunk for p in seen)
    sd = dict(EW)
    for li, W in enumerate(Ws):
        sd.update({f"videodit_blocks.layers.{li}.{k}": v for k, v in W.items()})
    model = HipVideoDiTModel(config, "cuda")
    model.load_state_dict(sd)
    got = sch.run(model, x.clone().cuda(), y.cuda(), masks.cuda(), InferenceParams(1, chunk_num * tokens)).cpu()
    assert torch.equal(got[0], got[1]) and torch.isfinite(got).all()
    moved = float((want - x).norm() / x.norm())
    r = rel_l2(got - x, want - x)
    print(f"magi schedule rollout: 12 forwards, |x_final - x_noise| / |x_noise| = {moved:.3f}; HIP vs oracle on x_final - x_noise: {r:.3e}")
    assert r < 1e-2, r            # measured 3.3e-3
    # video continuation: a prefix of one whole chunk + one frame (extraction pass into the cache, the frame pasted over chunk 1's noise)
    prefix = torch.randn(1, cfg.in_channels, cw + 1, Hl, Wl, generator=g).repeat(2, 1, 1, 1, 1)
    sch_p = ChunkSchedule(8, 4, chunk_num, cw, chunk_offset=1)
    om_p = _OracleModel(cfg, EW, Ws, config, chunk_num * tokens)
    want_p = sch_p.run(om_p, x.clone(), y, masks, SimpleNamespace(update_kv_cache=False), prefix_video=prefix)
    model_p = HipVideoDiTModel(config, "cuda")
    model_p.load_state_dict(sd)
    got_p = sch_p.run(model_p, x.clone().cuda(), y.cuda(), masks.cuda(), InferenceParams(1, chunk_num * tokens), prefix_video=prefix.cuda()).cpu()
    assert torch.equal(got_p[:, :, :cw], x[:, :, :cw]) and torch.isfinite(got_p).all()
    rp = rel_l2(got_p[:, :, cw:] - x[:, :, cw:], want_p[:, :, cw:] - x[:, :, cw:])
    print(f"magi schedule rollout behind a prefix video: {sch_p.total_forward_step()} + 1 forwards; HIP vs oracle on x_final - x_noise: {rp:.3e}")
    assert rp < 1e-2, rp
    config.runtime_config.cfg_number = 2            # 1 and 3 are built; the reference raises for everything else as well
    with pytest.raises(NotImplementedError):
        model.forward_dispatcher(x=x.cuda(), timestep=None, y=None, mask=None, kv_range=None, inference_params=None)


def test_forward_dispatcher_vs_reference_golden():
    """`forward_dispatcher` in both guidance modes against tests/golden/magi_dispatch_tiny.npz — the REFERENCE's own dispatcher run on the
    CPU (oracle/gen_golden_magi_dispatch.py), every `self.forward` it makes recorded.  cfg_number = 3: three forwards per call (text +
    previous chunks without a cache write; null caption with the write; the denoising chunks as batch rows without a cache) over a
    sequence that walks the cache rule; cfg_number = 1: the nearly-clean re-forward.  Each component forward is held to the model's bf16
    floor (6.3-7.0e-3 measured against float64 layers in the test above: bound 1e-2); the guidance mix multiplies component errors by
    its scales (|1 - 1.5| + |1.5 - 7.5| + 7.5 = 14 at small t), so the mixed output is checked two ways: bit for bit against the
    reference's formula applied to THIS model's components, and against the reference's output (measured 5.0-6.7e-3, bound 2e-2)."""
    from inferix_amd.magi.model import HipVideoDiTModel
    from inferix_amd.magi.types import InferenceParams
    fx = golden("magi_dispatch_tiny.npz")
    clip, max_tokens, n3, n1, wseed, eseed = [int(v) for v in fx["geom"]]
    cfg = MM.tiny_model_config()
    sd = dict(MM.init_embedder_weights(cfg, eseed))
    for li in range(cfg.num_layers):
        sd.update({f"videodit_blocks.layers.{li}.{k}": v for k, v in MB.init_layer_weights(cfg.layer, wseed + li).items()})
    config = _config(cfg)
    scales = {k: fx[k].tolist() for k in ("cfg_t_range", "prev_chunk_scales", "text_scales")}
    model = HipVideoDiTModel(config, "cuda")
    model.load_state_dict(sd)
    seen = []
    fwd0 = model.forward

    depth = [0]

    def fwd1(*a, **k):
        depth[0] += 1                                  # the batched forward goes through its rows with nested calls: record the outer ones
        try:
            out = fwd0(*a, **k)
        finally:
            depth[0] -= 1
        if depth[0] == 0:
            seen.append(out.clone())
        return out
    model.forward = fwd1
    for tag, cfg_number, n in (("t", 3, n3), ("o", 1, n1)):
        model.runtime_config = config.runtime_config = SimpleNamespace(cfg_number=cfg_number, **scales)
        ip = InferenceParams(1, max_tokens)
        for ci in range(n):
            range_num, dn, sp, fe, di = [int(v) for v in fx[f"{tag}{ci}_flags"]]
            kw = dict(range_num=range_num, denoising_range_num=dn, slice_point=sp, fwd_extra_1st_chunk=bool(fe), chunk_width=1, num_steps=8,
                      distill_interval=1)
            if di:
                kw["distill_nearly_clean_chunk"] = True
            x, t, y, mask, kv = [fx[f"{tag}{ci}_in_{k}"].cuda() for k in ("x", "t", "y", "mask", "kv_range")]
            seen.clear()
            out = model.forward_dispatcher(x=x, timestep=t, y=y, mask=mask, kv_range=kv, inference_params=ip, **kw).cpu()
            assert len(seen) == int(fx[f"{tag}{ci}_n_forwards"])
            errs = []
            for fi, got in enumerate(seen):
                ref = fx[f"{tag}{ci}_fwd{fi}"]
                assert got.shape == ref.shape, (tag, ci, fi, got.shape, ref.shape)
                errs.append(rel_l2(got.cpu(), ref))
            want = fx[f"{tag}{ci}_out"]
            assert out.shape == want.shape and torch.equal(out[0], out[1])
            r = rel_l2(out, want)
            print(f"magi dispatcher cfg {cfg_number} call {ci}: component forwards vs reference {['%.2e' % e for e in errs]}; mixed output {r:.3e}")
            assert max(errs) < 1e-2, (tag, ci, errs)
            if cfg_number == 3:
                a, b, u = [s_.cpu() for s_ in seen]
                nd = dn - fe
                u = u.transpose(0, 1).reshape(1, -1, nd, *u.shape[3:])                  # batch rows back to chunks (chunk_width 1)
                tr, ps, ts = [torch.tensor(scales[k]) for k in ("cfg_t_range", "prev_chunk_scales", "text_scales")]
                pieces = []
                for c in range(nd):
                    idx = torch.searchsorted(tr - 1e-7, t[0, -nd:][c].cpu()) - 1
                    pieces.append((1 - ps[idx]) * u[:, :, c:c + 1] + (ps[idx] - ts[idx]) * b[:, :, -nd:][:, :, c:c + 1] + ts[idx] * a[:, :, -nd:][:, :, c:c + 1])
                mine = torch.cat([x[0:1, :, :-nd].cpu(), torch.cat(pieces, dim=2)], dim=2)
                assert torch.allclose(out[0:1], mine, rtol=1e-6, atol=1e-6), "the guidance mix is not the reference's formula on this model's components"
                assert r < 2e-2, (tag, ci, r)            # measured 5.0-6.7e-3: the component errors are correlated and do not add up to the 14x
            else:
                assert r < 1e-2, (tag, ci, r)


# ---- the whole model under context parallelism: 2 ranks on one GPU (gloo rendezvous, exchanges staged through the host) ----------------
def _model_cp_worker(rank, world, port, strategy, ret):
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
        torch.cuda.set_device(0)
        from inferix_amd.magi import context_parallel as cpl
        from inferix_amd.magi.model import HipVideoDiTModel
        from inferix_amd.magi.types import InferenceParams
        cpl.set_cp_group(dist.group.WORLD)
        fx = golden("magi_model_tiny.npz")
        n_layers, clip, n_calls, wseed, eseed, max_tokens = [int(v) for v in fx["geom"]]
        cfg = MM.tiny_model_config()
        sd = dict(MM.init_embedder_weights(cfg, eseed))
        for li in range(n_layers):
            sd.update({f"videodit_blocks.layers.{li}.{k}": v for k, v in MB.init_layer_weights(cfg.layer, wseed + li).items()})
        config = _config(cfg)
        config.engine_config.cp_size, config.engine_config.cp_strategy = world, strategy
        model = HipVideoDiTModel(config, "cuda")
        model.load_state_dict(sd)
        ip = InferenceParams(1, max_tokens)
        outs = []
        for ci in range(n_calls):
            range_num, dn, sp, fe, di, upd = [int(v) for v in fx[f"c{ci}_flags"]]
            kw = dict(range_num=range_num, denoising_range_num=dn, slice_point=sp, fwd_extra_1st_chunk=bool(fe), distill_nearly_clean_chunk=bool(di))
            args = [fx[f"c{ci}_in_{k}"].cuda() for k in ("x", "t", "y", "drop", "mask", "kv_range")]
            ip.update_kv_cache = bool(upd)
            outs.append(model(*args, inference_params=ip, **kw).cpu())
        torch.cuda.synchronize()
        ret[rank] = outs
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("strategy", ["cp_ulysses", "cp_shuffle_overlap"])
def test_model_under_context_parallelism_two_ranks_vs_reference_golden(strategy):
    """`HipVideoDiTModel.forward` with `engine_config.cp_size = 2` under both strategies: cp_pre_process shards the patch tokens (and the
    rope rows, the condition map, the cross-attention ranges), the layers exchange heads for tokens, cp_post_process gathers the head's
    output — every rank must hold the single-device model output of the golden (the reference's `VideoDiTModel.forward`), at the same
    bf16 floor as the unsharded test above."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    fx = golden("magi_model_tiny.npz")
    with mp.get_context("spawn").Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_model_cp_worker, args=(world, port, strategy, ret), nprocs=world, join=True)
        outs = [ret[r] for r in range(world)]
    for ci in range(int(fx["geom"][2])):
        ref = fx[f"c{ci}_out"]
        for r in range(world):
            got = outs[r][ci]
            assert got.shape == ref.shape
            e = rel_l2(got, ref)
            print(f"magi model under {strategy}, call {ci}, rank {r}: vs reference {e:.3e}")
            assert e < 1e-2, (strategy, ci, r, e)
        assert torch.equal(outs[0][ci], outs[1][ci]), "the ranks hold different gathered outputs"
