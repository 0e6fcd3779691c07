"""CPU: the oracle (oracle/wan_oracle.py) against the golden vectors generated from the
reference itself (oracle/gen_golden.py).  Bit-exact: both run the same PyTorch CPU ops in the
same order, so any difference is a restatement error."""
import torch

import wan_oracle as O
from fixture_io import golden, weights_checksum

BF = torch.bfloat16


def same(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.equal(a.double(), b.double()), (a.double() - b.double()).abs().max()


def test_ops_golden():
    fx = golden("ops.npz")
    same(fx["sin_out"], O.sinusoidal_embedding_1d(64, fx["sin_t"]))
    for dim in (44, 42):
        same(torch.view_as_real(fx[f"rope_params_{dim}"]), torch.view_as_real(O.rope_params(32, dim)))
    freqs = O.rope_freqs(128)
    x = fx["rope_x"]
    for sf in (0, 5):
        same(fx[f"rope_out_sf{sf}"], O.causal_rope_apply(x, (3, 4, 6), freqs, sf))
    for ws in (2, 4):
        for rk in (0, ws - 1):
            xl = x[:, :72 // ws].contiguous()
            same(fx[f"rope_chunk_ws{ws}_r{rk}"], O.causal_rope_apply(xl, (3, 4, 6), freqs, 3, ws, rk))
    xn, w, b = fx["norm_x"], fx["rms_w"], fx["ln_b"]
    same(fx["rms_out"], O.rms_norm(xn, w, 1e-6))
    same(fx["ln_out"], O.layer_norm(xn, 1e-6))
    same(fx["ln_affine_out"], O.layer_norm(xn, 1e-6, w, b))
    e = (fx["mod"].unsqueeze(1) + fx["e0"]).chunk(6, dim=2)
    same(fx["modulate_out"], O.modulate(O.layer_norm(xn, 1e-6), e[1], e[0], 3))
    same(fx["gate_out"], O.gated_residual(xn, fx["gate_y"], e[2], 3))
    mo, ml = O.merge_out_lse(fx["merge_out"], fx["merge_lse"], fx["merge_bo"], fx["merge_bl"])
    same(fx["merge_ro"], mo)
    same(fx["merge_rl"], ml)
    same(fx["attn_out"], O.attention(fx["attn_q"], fx["attn_k"], fx["attn_v"]))
    # the fp64 "math" attention is the accuracy yardstick: sdpa-bf16 must sit within bf16 noise of it
    ref = O.attention(fx["attn_q"], fx["attn_k"], fx["attn_v"], impl="math")
    assert (fx["attn_out"].double() - ref).abs().max() < 2e-2


def test_lse_merge_is_exact_split_kv():
    """merge(attn(K1), attn(K2)) == attn(K1 ++ K2): the identity CP relies on."""
    g = torch.Generator().manual_seed(0)
    q = torch.randn(1, 17, 2, 128, generator=g)
    k = torch.randn(1, 50, 2, 128, generator=g)
    v = torch.randn(1, 50, 2, 128, generator=g)
    full, lse = O.attention_with_lse(q, k, v)
    o1, l1 = O.attention_with_lse(q, k[:, :20], v[:, :20])
    o2, l2 = O.attention_with_lse(q, k[:, 20:], v[:, 20:])
    mo, ml = O.merge_out_lse(o1.float(), l1.transpose(1, 2).unsqueeze(-1).float(), o2,
                             l2.transpose(1, 2).unsqueeze(-1).float())
    assert (mo.double() - full).abs().max() < 1e-5
    assert (ml.squeeze(-1).transpose(1, 2).double() - lse).abs().max() < 1e-5


def test_layout_golden():
    fx = golden("layout.npz")
    cfg = O.tiny_config()
    idx = torch.arange(72 * 64, dtype=torch.float32).view(1, 72, 64)
    same(fx["unpatchify_idx"], O.unpatchify(idx, (3, 4, 6), cfg).to(torch.int32))
    tok = torch.arange(72 * 2, dtype=torch.float32).view(1, 72, 2)
    for cp in (2, 4):
        parts = [O.cp_scatter(tok, 3, cp, r) for r in range(cp)]
        for r in range(cp):
            same(fx[f"scatter_cp{cp}_r{r}"], parts[r].to(torch.int32))
        same(fx[f"gather_cp{cp}"], O.cp_gather_interleave(parts, 3).to(torch.int32))
    # patchify is the exact im2row of the Conv3d patch embedding
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(2, 16, 3, 8, 12, generator=g)
    w = torch.randn(32, 16, 1, 2, 2, generator=g)
    ref = torch.nn.functional.conv3d(lat, w, stride=(1, 2, 2)).flatten(2).transpose(1, 2)
    mine = O.patchify(lat, cfg) @ w.flatten(1).t()
    assert (ref - mine).abs().max() < 1e-4


def test_scheduler_golden():
    fx = golden("scheduler.npz")
    for shift, steps in ((5.0, [1000, 750, 500, 250]), (8.0, [1000, 757, 522])):
        tag = str(int(shift))
        s = O.FlowMatchSchedule(shift=shift)
        same(fx[f"sigmas_{tag}"], s.sigmas)
        same(fx[f"timesteps_{tag}"], s.timesteps)
        same(fx[f"warped_{tag}"], s.warp(steps))
        same(fx[f"an_out_{tag}"], s.add_noise(fx[f"an_x0_{tag}"], fx[f"an_eps_{tag}"], fx[f"an_t_{tag}"]))
        same(fx[f"f2x_out_{tag}"], O.flow_to_x0(fx[f"f2x_flow_{tag}"], fx[f"an_x0_{tag}"], fx[f"an_t_{tag}"], s))
    # SURVEY §3.2 [probed]: shift 5 warps [1000,750,500,250] to [1000, 937.5, 833.33, 625]
    w = O.FlowMatchSchedule(5.0).warp([1000, 750, 500, 250])
    assert torch.allclose(w, torch.tensor([1000.0, 937.5, 833.3333, 625.0]), atol=1e-3)


def test_block_real_dims_golden():
    fx = golden("block_real_dims.npz")
    cfg = O.WanConfig(num_layers=1, text_len=32, text_dim=64, freq_dim=64, latent_h=8, latent_w=12)
    W = O.init_weights(cfg, seed=3)
    assert weights_checksum(W) == int(fx["weights_checksum"])
    fs = cfg.frame_seqlen
    state = O.CacheState.allocate(cfg, 1, BF, cache_tokens=6 * fs)
    freqs = O.rope_freqs(cfg.head_dim)
    grid = (3, 4, 6)
    for b in range(2):
        out = O.block_forward(fx[f"x{b}"], fx[f"e0_{b}"], fx["context"], W, 0, cfg, grid, freqs, state, b * 3 * fs)
        same(fx[f"out{b}"], out)
    same(fx["cache_k"], state.layers[0].k[0, :6 * fs])
    same(fx["cache_v"], state.layers[0].v[0, :6 * fs])
    same(fx["cross_k"], state.cross[0].k[0])
    same(fx["cross_v"], state.cross[0].v[0])


def _rollout(name, cfg, nfb=3):
    fx = golden(name)
    W = O.init_weights(cfg, seed=0)
    assert weights_checksum(W) == int(fx["weights_checksum"])
    renoise = [fx[f"renoise_{i}"] for i in range(int(fx["num_renoise"]))]
    batch = fx["noise"].shape[0]
    state = O.CacheState.allocate(cfg, batch, BF, cache_tokens=int(fx["cache_tokens"]))
    rec = []
    out, state = O.inference(W, cfg, fx["noise"], list(fx["prompt_embeds"]), fx["steps"].tolist(),
                             renoise=renoise, shift=float(fx["shift"]), num_frame_per_block=nfb,
                             state=state, record=rec, initial_latent=fx.get("initial_latent"))
    same(fx["out"], out)
    # integer trace: (current_start, global_end, local_end) after every generator forward — bit-exact
    tr = fx["trace"]
    assert len(state.trace) == tr.shape[0]
    for row, s in zip(tr.tolist(), state.trace):
        assert row[1] == s.global_end and row[2] == s.local_end
        assert s.local_start == s.local_end - nfb * cfg.frame_seqlen
    le = state.layers[0].local_end
    same(fx["cache_k_layer0"], state.layers[0].k[0, :le])
    same(fx["cache_v_layer0"], state.layers[0].v[0, :le])
    same(fx["cache_k_last"], state.layers[-1].k[0, :le])
    # per-forward teacher-forced outputs
    n_prefill = 1 if "initial_latent" in fx else 0
    k = n_prefill
    for r in rec:
        if r["kind"] == "denoise":
            same(fx[f"call{k}_x_in"], r["x_in"])
            same(fx[f"call{k}_x0"], r["x0"])
        k += 1
    assert k == int(fx["num_calls"])
    return fx, state


def test_rollout_tiny_golden():
    _rollout("rollout_tiny.npz", O.tiny_config())


def test_rollout_local_attention_eviction_golden():
    fx, state = _rollout("rollout_tiny_local.npz", O.tiny_config(local_attn_size=6, sink_size=1))
    # rolling eviction really happened: global_end keeps growing, local_end saturates at the cache size
    assert state.layers[0].global_end == 4 * 72 and state.layers[0].local_end == 144
    assert any(s.evicted > 0 for s in state.trace)


def test_rollout_prefill_golden():
    _rollout("rollout_tiny_prefill.npz", O.tiny_config())


def test_rollout_batch2_golden():
    _rollout("rollout_tiny_b2.npz", O.tiny_config())


def test_kv_index_update_properties():
    # append, re-run in place, and the eviction arithmetic of causal_model.py:282-300
    s = O.kv_index_update(0, 0, 0, 72, 144, 6, 24)
    assert (s.local_start, s.local_end, s.global_end, s.evicted) == (0, 72, 72, 0)
    s = O.kv_index_update(72, 72, 0, 72, 144, 6, 24)       # re-run same block: overwrite in place
    assert (s.local_start, s.local_end, s.global_end, s.evicted) == (0, 72, 72, 0)
    s = O.kv_index_update(144, 144, 144, 72, 144, 6, 24)   # cache full: evict 72, keep 24 sink tokens
    assert (s.evicted, s.rolled, s.local_start, s.local_end, s.global_end) == (72, 48, 72, 144, 216)
    s = O.kv_index_update(216, 144, 144, 72, 144, 6, 24)   # re-run after eviction: same slots, no second roll
    assert (s.evicted, s.local_start, s.local_end, s.global_end) == (0, 72, 144, 216)


def test_causvid_golden():
    """CausVid twin (explicit kv_start/kv_end slots, x0-only generator, last step dropped), two segments with
    start_latents prefill: oracle == reference bit for bit."""
    fx = golden("causvid_tiny.npz")
    cfg = O.tiny_config(text_len=512)
    W = O.init_weights(cfg, seed=0)
    assert weights_checksum(W) == int(fx["weights_checksum"])
    for seg in range(2):
        renoise = [fx[f"seg{seg}_renoise_{i}"] for i in range(int(fx[f"seg{seg}_num_renoise"]))]
        rec = []
        out, st = O.causvid_inference(W, cfg, fx[f"seg{seg}_noise"], list(fx["prompt_embeds"]), fx["steps"].tolist(),
                                      renoise=renoise, shift=8.0, start_latents=fx.get(f"seg{seg}_start"),
                                      state=O.CacheState.allocate(cfg, 1, BF, cache_tokens=600), record=rec)
        same(fx[f"seg{seg}_out"], out)
        n = fx[f"seg{seg}_cache_k"].shape[0]
        same(fx[f"seg{seg}_cache_k"], st.layers[0].k[0, :n])
        same(fx[f"seg{seg}_cache_v"], st.layers[0].v[0, :n])
        n_tok = 3 * cfg.frame_seqlen
        assert [[r["block"] * n_tok, (r["block"] + 1) * n_tok] for r in rec] == fx[f"seg{seg}_slots"].tolist()
        assert [round(float(r["timestep"].flatten()[0]), 3) for r in rec] == [round(float(t), 3) for t in fx[f"seg{seg}_t"]]


def test_quantised_model_oracle_golden():
    """The quantised MODEL (BASELINE config 4; oracle/gen_golden_quant_model.py): `wan_oracle.linear_override` +
    `quant_oracle.model_hook` under the reference's exclusion dict (example/quantization/run_self_forcing_quantized.py:57-62)
    reproduce the committed fixture bit for bit, and the dict resolves as upstream: every nn.Linear quantised except
    text_embedding.* and head.head.  (DAX is not in the reference tree: the fixture comes from the oracle, parity with DAX unpinned.)"""
    import gen_golden_quant_model as G
    import quant_oracle as Q
    fx = golden("quant_model_tiny.npz")
    rfx, cfg, W = G.rollout_inputs()
    for fmt, nm in G.NAMES.items():
        log = []
        q, st = G.run_rollout(rfx, cfg, W, fmt, "sdpa", log)
        same(fx[f"out_{nm}"], q)
        assert [[s.local_start, s.local_end, s.global_end] for s in st.trace] == fx[f"trace_{nm}"].tolist()
        assert fx[f"trace_{nm}"][:, 1:].tolist() == golden("rollout_tiny.npz")["trace"][:, [2, 1]].tolist()   # quantisation moves no slot
        one = log[:log.index(("head.head", None)) + 1]
        assert {n for n, f in one if f is None} == {"text_embedding.0", "text_embedding.2", "head.head"}
        assert {n for n, f in one if f is not None} == (
            {f"blocks.{i}.{a}.{p}" for i in range(cfg.num_layers) for a in ("self_attn", "cross_attn") for p in "qkvo"}
            | {f"blocks.{i}.ffn.{j}" for i in range(cfg.num_layers) for j in (0, 2)}
            | {"time_embedding.0", "time_embedding.2", "time_projection.1"})
        assert all(f == fmt for _, f in one if f is not None)
    bfx, bcfg, bW = G.block_inputs()
    (o0, o1), _ = G.run_block(bfx, bcfg, bW, Q.FP8, "sdpa")
    same(fx["block_out0_fp8"], o0)
    same(fx["block_out1_fp8"], o1)
    # the hook is scoped: outside the `with` the restated model is the bf16 one again
    assert O._LINEAR_OVERRIDE is None
    assert Q.config_for("blocks.3.ffn.0", {"": 1, "blocks.3": None}) is None and Q.config_for("blocks.30.ffn.0", {"": 1, "blocks.3": None}) == 1


def test_block_720p_full_size_golden_first_block():
    """The oracle's CausVid block at BASELINE config 3's REAL size (10800 tokens, dim 1536, ffn 8960, explicit slots [0, 10800)) against
    the rows the reference's own block produced (tests/golden/block_720p_full_size.npz, oracle/gen_golden_block_720p.py): bit-exact.
    (Case 1, L = 75600, is re-checked by the generator itself; it is left out here to keep the CPU suite short.)"""
    import block_720p_inputs as BI
    fx = golden("block_720p_full_size.npz")
    cfg = BI.config()
    W = O.init_weights(cfg, seed=3)
    assert weights_checksum(W) == int(fx["weights_checksum"])
    d = BI.make(0)
    assert BI.checksum(d["x"]) == int(fx["c0_x_checksum"])
    n = BI.FRAMES * cfg.frame_seqlen
    state = O.CacheState.allocate(cfg, 1, BF, cache_tokens=n)
    grid = (BI.FRAMES, cfg.latent_h // 2, cfg.latent_w // 2)
    out = O.block_forward(d["x"], d["e0"], d["ctx"], W, 0, cfg, grid, O.rope_freqs(cfg.head_dim), state, 0, explicit=(0, n))
    sel = fx["sel"].long()
    same(fx["c0_out_rows"], out[0, sel])
    same(fx["c0_k_rows"], state.layers[0].k[0, sel])
    same(fx["c0_v_rows"], state.layers[0].v[0, sel])
