"""GPU parity tests of the host-side mirror (HipCausalWanModel / HipWanDiffusionWrapper /
CausalInferencePipeline) against the reference-generated golden vectors and the CPU oracle."""
import os
import sys
from types import SimpleNamespace

import pytest
import torch

import wan_oracle as O
from fixture_io import golden
from util import ulp_report, assert_bf16_parity, rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def build(cfg: O.WanConfig, W, **kw):
    from inferix_amd.wan import HipCausalWanModel, HipWanDiffusionWrapper
    m = HipCausalWanModel(patch_size=cfg.patch_size, text_len=cfg.text_len, in_dim=cfg.in_dim, dim=cfg.dim,
                          ffn_dim=cfg.ffn_dim, freq_dim=cfg.freq_dim, text_dim=cfg.text_dim, out_dim=cfg.out_dim,
                          num_heads=cfg.num_heads, num_layers=cfg.num_layers, local_attn_size=cfg.local_attn_size,
                          sink_size=cfg.sink_size, eps=cfg.eps, **kw)
    m.load_state_dict(W)
    return m


def test_block_real_dims_vs_reference_golden():
    """One transformer block at the real channel geometry (dim 1536, 12 heads, ffn 8960) against the
    reference's own CausalWanAttentionBlock outputs, two consecutive frame blocks (prefix growth)."""
    from inferix_amd import hip_ops as ops
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    fx = golden("block_real_dims.npz")
    cfg = O.WanConfig(num_layers=1, text_len=32, text_dim=64, freq_dim=64, latent_h=8, latent_w=12)
    m = build(cfg, O.init_weights(cfg, seed=3))
    fs, nf = cfg.frame_seqlen, 3
    n = nf * fs
    kvm, req = KVCacheManager("cuda"), [KVCacheRequest("r")]
    ad = m.blocks[0].kv_cache_manager
    ad.allocate_kv_cache(kv_cache_manager=kvm, kv_cache_request=req[0], sequence_length=6 * fs, dtype=BF)
    ad.allocate_crossattn_cache(kv_cache_manager=kvm, kv_cache_request=req[0], crossattn_length=cfg.text_len, dtype=BF)
    meta = {"global_end_index": torch.tensor([0]), "local_end_index": torch.tensor([0])}
    cmeta = {"is_init": False}
    ctx = fx["context"][0].cuda()
    # Noise floor of the comparison: the reference's CPU path uses bf16 SDPA; evaluating the SAME block with
    # exact (fp64) attention moves a third of the bf16 outputs by one ULP (rel-L2 ~2.6e-3).  The HIP block must
    # sit inside that band: this is the stated bf16 tolerance for a full block (north_star).
    W = O.init_weights(cfg, seed=3)
    st_exact = O.CacheState.allocate(cfg, 1, BF, cache_tokens=6 * fs)
    freqs = O.rope_freqs(cfg.head_dim)
    for b in range(2):
        exact = O.block_forward(fx[f"x{b}"], fx[f"e0_{b}"], fx["context"], W, 0, cfg, (3, 4, 6), freqs, st_exact,
                                b * n, attn_impl="math")
        floor_rel = rel_l2(exact, fx[f"out{b}"])
        x = fx[f"x{b}"][0].cuda().clone()
        El = (m.mod_all[0] + fx[f"e0_{b}"][0].cuda()).contiguous()          # [F, 6, dim]
        rope = ops.RopeGridSpec(m.freqs, b * nf, 4, 6)
        st = dict(B=1, N=n, F_=nf, fs=fs, rows_per_group=fs, rope=rope, sink_tokens=0, current_start=b * n, ctx=ctx)
        m._run_block(0, x, El, st, meta, cmeta, kvm, req)
        assert_bf16_parity(x, fx[f"out{b}"][0], max_ulp=4, max_mismatch_frac=0.5, rel=1.25 * floor_rel + 5e-4,
                           floor=1.0, what=f"block output #{b}")
        assert rel_l2(x.cpu(), exact[0]) <= 1.25 * floor_rel + 5e-4
        assert int(meta["local_end_index"]) == (b + 1) * n and int(meta["global_end_index"]) == (b + 1) * n
    raw = kvm.get_raw(req[0], "layer_0")
    assert_bf16_parity(raw[0, :2 * n, 0], fx["cache_k"], max_ulp=1, floor=1.0, what="cache K (post-RoPE)")
    assert_bf16_parity(raw[1, :2 * n, 0], fx["cache_v"], max_ulp=1, floor=0.05, what="cache V")
    craw = kvm.get_raw(req[0], "crossattn_layer_0")
    assert_bf16_parity(craw[0, :, 0], fx["cross_k"], what="cross K")
    assert_bf16_parity(craw[1, :, 0], fx["cross_v"], what="cross V")


def _pipeline(cfg, W, steps, shift, **model_kw):
    from inferix_amd.pipeline import CausalInferencePipeline
    from inferix_amd.wan import HipWanDiffusionWrapper
    m = build(cfg, W, **model_kw)
    gen = HipWanDiffusionWrapper(model=m, timestep_shift=shift)
    args = SimpleNamespace(denoising_step_list=list(steps), warp_denoising_step=True, num_frame_per_block=3,
                           independent_first_frame=False, context_noise=0, model_kwargs={},
                           frame_seq_length=cfg.frame_seqlen, kv_cache_tokens=None)
    return m, gen, args


def _run_rollout(name, cfg, paging=None, slack=1.25, eps=5e-4, pair=None, want_cache=False, pair_mode=None):
    from inferix_amd.core import DecodeMode
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    from inferix_amd.pipeline import CausalInferencePipeline
    fx = golden(name)
    W = O.init_weights(cfg, seed=0)
    m, gen, args = _pipeline(cfg, W, fx["steps"].tolist(), float(fx["shift"]))
    args.kv_cache_tokens = int(fx["cache_tokens"]) if cfg.local_attn_size == -1 else None
    args.pair_forwards = pair            # None: the pipeline's default; True: re-run + next block's first step enqueued as a pair
    m.pair_mode = pair_mode              # None: streams on one GPU; "lockstep": one chain over both forwards' rows
    if cfg.local_attn_size == -1:
        args.kv_cache_tokens = min(int(fx["cache_tokens"]), 21 * cfg.frame_seqlen)
    pe = fx["prompt_embeds"].cuda()
    pipe = CausalInferencePipeline(args, "cuda", generator=gen, text_encoder=lambda text_prompts: {"prompt_embeds": pe},
                                   vae=None)
    B = fx["noise"].shape[0]
    kvm = KVCacheManager("cuda")
    reqs = [KVCacheRequest(f"req{i}") for i in range(B)]
    # integer trace: layer 0's host-side index state after every forward, recorded by the MODEL in host order (a paired re-run /
    # first step does not pass through gen.forward; the first chain's layer 0 is enqueued before the second's: the sequential order)
    trace = []
    m.index_trace = trace
    if paging:
        # allocate first so paging can be enabled before the rollout starts
        pipe._initialize_kv_cache(kvm, reqs, BF)
        for l in range(cfg.num_layers):
            for r in reqs:
                kvm.enable_paging(r, f"layer_{l}", paging)
    renoise = [fx[f"renoise_{i}"] for i in range(int(fx["num_renoise"]))]
    init = fx["initial_latent"].cuda() if "initial_latent" in fx else None
    out = pipe.inference(noise=fx["noise"].cuda(), text_prompts=["x"] * B, kv_cache_manager=kvm, kv_cache_requests=reqs,
                         initial_latent=init, decode_mode=DecodeMode.NO_DECODE, free_cache_before_vae=False,
                         renoise=renoise)
    torch.cuda.synchronize()
    # (1) KV index state: bit-exact against the reference trace
    assert trace == [tuple(r) for r in fx["trace"].tolist()], "integer KV index trace differs from the reference"
    # (2) latents: chained bf16 forwards (15 for the tiny rollout).  Yardstick = the SAME rollout with exact (fp64) attention, generated
    # next to the reference's in oracle/gen_golden.py: the reference's bf16-SDPA result is `floor` away from it, and the HIP rollout has
    # to be as close to the exact one — and to the reference — as floor x 1.25 (+ eps), the rule of the block tests (round-2 verdict:
    # no fixed 1e-2).  A 1.5x regression of either distance fails.
    floor = rel_l2(fx["out"], fx["out_exact"])
    r_ref, r_exact = rel_l2(out.cpu(), fx["out"]), rel_l2(out.cpu(), fx["out_exact"])
    print(f"{name}: floor (reference vs exact attention) {floor:.3e}; HIP vs exact {r_exact:.3e}; HIP vs reference {r_ref:.3e}")
    assert r_exact <= slack * floor + eps, f"{name}: HIP vs exact-attention rollout {r_exact:.3e} > {slack} x floor {floor:.3e} + {eps}"
    assert r_ref <= slack * floor + eps, f"{name}: HIP vs reference rollout {r_ref:.3e} > {slack} x floor {floor:.3e} + {eps}"
    # (3) cache contents (logical view through the page table)
    le = int(fx["trace"][-1, 2])
    raw = kvm.get_raw(reqs[0], "layer_0")
    k_log, v_log = raw[0, :, 0], raw[1, :, 0]
    pt = kvm.page_table(reqs[0], "layer_0")
    if pt is not None:
        t = torch.arange(le)
        slot = (pt.host[t // pt.page_size].long() * pt.page_size + t % pt.page_size).cuda()
        k_log, v_log = k_log[slot], v_log[slot]
    # layer 0's cache rows are projections of the clean-context latents: their own floor (reference vs exact-attention rollout) is a
    # multiple of the latents'; same rule, against both
    for nm, got in (("K", k_log[:le].cpu()), ("V", v_log[:le].cpu())):
        ref_c, ex_c = fx[f"cache_{nm.lower()}_layer0"], fx[f"cache_{nm.lower()}_layer0_exact"]
        fl, r1, r2 = rel_l2(ref_c, ex_c), rel_l2(got, ref_c), rel_l2(got, ex_c)
        print(f"{name}: layer-0 cache {nm} after the rollout: floor {fl:.3e}; HIP vs exact {r2:.3e}; HIP vs reference {r1:.3e}")
        assert r1 <= slack * fl + eps and r2 <= slack * fl + eps, (nm, fl, r1, r2)
    if want_cache:
        caches = [kvm.get_raw(r, f"layer_{l}")[:, :le].clone() for r in reqs for l in range(cfg.num_layers)] if pt is None else \
                 [k_log[:le].clone(), v_log[:le].clone()]
        return out, fx, caches
    return out, fx


def test_rollout_tiny_vs_reference_golden():
    _run_rollout("rollout_tiny.npz", O.tiny_config())


def test_rollout_prefill_vs_reference_golden():
    _run_rollout("rollout_tiny_prefill.npz", O.tiny_config())


def test_rollout_batch2_vs_reference_golden():
    _run_rollout("rollout_tiny_b2.npz", O.tiny_config())


def test_rollout_local_attention_roll_and_page_table_agree():
    """Sink + rolling eviction: the physical shift kernel and the page-table rotation must give the SAME
    latents bit for bit, and both match the reference within the rollout tolerance."""
    cfg = O.tiny_config(local_attn_size=6, sink_size=1)
    out_roll, fx = _run_rollout("rollout_tiny_local.npz", cfg)
    out_page, _ = _run_rollout("rollout_tiny_local.npz", cfg, paging=cfg.frame_seqlen)
    assert torch.equal(out_roll, out_page), "page-table rotation and physical roll diverge"


@pytest.mark.parametrize("name,kw,paging", [("rollout_tiny.npz", {}, False), ("rollout_tiny_prefill.npz", {}, False),
                                            ("rollout_tiny_b2.npz", {}, False),
                                            ("rollout_tiny_local.npz", dict(local_attn_size=6, sink_size=1), False),
                                            ("rollout_tiny_local.npz", dict(local_attn_size=6, sink_size=1), True)])
@pytest.mark.parametrize("mode", ["streams", "lockstep"])
def test_rollout_with_paired_forwards_is_bit_identical(name, kw, paging, mode):
    """`pair_forwards`: the clean-context re-run of block b and the first denoising step of block b + 1 enqueued TOGETHER
    (HipCausalWanModel.forward_pair) — "streams": layer by layer on two streams; "lockstep": one launch chain over both forwards' rows
    (what a sequence-parallel rank runs).  Same arithmetic per row in the same per-cache order (and, without gemm_small_split, tiles
    whose summation order does not depend on the row count): latents AND every layer's cache rows must equal the sequential pipeline's
    BIT FOR BIT, the integer index trace must be the reference's, and the golden bounds hold — plain, prefill, batch 2, and sink +
    rolling eviction through the shift kernel and through the page table."""
    from inferix_amd import hip_ops as ops
    ops.set_option("gemm_small_split", 0)
    cfg = O.tiny_config(**kw)
    pg = cfg.frame_seqlen if paging else None
    seq, _, c_seq = _run_rollout(name, cfg, paging=pg, pair=False, want_cache=True)
    par, _, c_par = _run_rollout(name, cfg, paging=pg, pair=True, want_cache=True, pair_mode=mode)
    assert torch.equal(seq, par), f"{name}: paired forwards ({mode}) changed the latents ({rel_l2(par.cpu(), seq.cpu()):.3e})"
    assert len(c_seq) == len(c_par) and all(torch.equal(a, b) for a, b in zip(c_seq, c_par)), f"{name}: paired forwards ({mode}) changed cache rows"


def test_teacher_forced_forwards_vs_reference_golden():
    """Every generator forward of the tiny rollout with the reference's own inputs (per-forward parity:
    no error compounding through re-noising, SURVEY §7 hard part ii)."""
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    from inferix_amd.pipeline import CausalInferencePipeline
    fx = golden("rollout_tiny.npz")
    cfg = O.tiny_config()
    m, gen, args = _pipeline(cfg, O.init_weights(cfg, seed=0), fx["steps"].tolist(), 5.0)
    args.kv_cache_tokens = 21 * cfg.frame_seqlen
    pe = fx["prompt_embeds"].cuda()
    pipe = CausalInferencePipeline(args, "cuda", generator=gen, text_encoder=None, vae=None)
    kvm, reqs = KVCacheManager("cuda"), [KVCacheRequest("r")]
    pipe._initialize_kv_cache(kvm, reqs, BF)
    pipe._initialize_crossattn_cache(kvm, reqs, BF)
    worst = 0.0
    for i, (cs, ge, le) in enumerate(fx["trace"].tolist()):
        flow, x0 = gen(noisy_image_or_video=fx[f"call{i}_x_in"].cuda(), conditional_dict={"prompt_embeds": pe},
                       timestep=fx[f"call{i}_t"].cuda(), kv_cache_meta=pipe.kv_cache_meta,
                       crossattn_cache_meta=pipe.crossattn_cache_meta, current_start=cs, kv_cache_manager=kvm,
                       kv_cache_requests=reqs)
        assert (int(pipe.kv_cache_meta[0]["global_end_index"]), int(pipe.kv_cache_meta[-1]["local_end_index"])) == (ge, le)
        # yardstick per forward: the same forward (the reference's input, own cache evolution) with exact attention, generated next
        # to the reference's result in oracle/gen_golden.py; the reference sits `floor` from it and the HIP forward must be as close
        # to it — and to the reference — as 1.25 x floor + 5e-4 (round-3 verdict: no fixed 5e-3)
        for what, got in (("flow", flow), ("x0", x0)):
            ref, exact = fx[f"call{i}_{what}"], fx[f"call{i}_{what}_exact"]
            floor = rel_l2(ref, exact)
            r_ref, r_exact = rel_l2(got.cpu(), ref), rel_l2(got.cpu(), exact)
            worst = max(worst, r_ref / (1.25 * floor + 5e-4), r_exact / (1.25 * floor + 5e-4))
            assert r_ref <= 1.25 * floor + 5e-4 and r_exact <= 1.25 * floor + 5e-4, \
                f"forward {i} {what}: floor {floor:.3e}, HIP vs reference {r_ref:.3e}, HIP vs exact {r_exact:.3e}"
    print(f"teacher-forced forwards: worst distance / bound = {worst:.2f}")


def test_attention_registry_contract():
    """`collect_supported_attn()['HipPagedFA']` returns (out [B,L,H,D], lse [B,H,L]) like backends.py:36-76."""
    from inferix_amd.attention import attention, collect_supported_attn
    fx = golden("ops.npz")
    q, k, v = fx["attn_q"].cuda(), fx["attn_k"].cuda(), fx["attn_v"].cuda()
    fn = collect_supported_attn()["HipPagedFA"]
    out, lse = fn(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1))
    assert out.shape == q.shape and lse.shape == (1, 2, 40)
    ref, lse64 = O.attention_with_lse(fx["attn_q"], fx["attn_k"], fx["attn_v"])
    assert (out.cpu().double() - ref).abs().max() < 2e-2 and (lse.cpu().double() - lse64).abs().max() < 2e-3
    assert torch.equal(attention(q, k, v), out)
    with pytest.raises(NotImplementedError):
        fn(q, k, v, causal=True)


def test_attention_varlen_arguments():
    """The seam's `q_lens` / `k_lens` (flash_attention.py:42-150, round-2 verdict missing #3): sample b attends its first k_lens[b]
    keys — equal to dense attention over the truncated keys, for a batch whose samples have different lengths; a sample without keys
    gives zeros as flash-attn's varlen kernels do; `q_lens` other than Lq is refused (the reference's unflatten raises there)."""
    from inferix_amd.attention import attention
    g = torch.Generator().manual_seed(3)
    B, Lq, Lk, H, D = 3, 70, 200, 4, 128
    q = torch.randn(B, Lq, H, D, generator=g).to(torch.bfloat16)
    k = torch.randn(B, Lk, H, D, generator=g).to(torch.bfloat16)
    v = torch.randn(B, Lk, H, D, generator=g).to(torch.bfloat16)
    k_lens = torch.tensor([200, 37, 0])
    out = attention(q.cuda(), k.cuda(), v.cuda(), k_lens=k_lens, q_lens=torch.tensor([Lq] * B))
    assert out.shape == q.shape and out.dtype == torch.bfloat16
    for b, n in enumerate(k_lens.tolist()):
        if n == 0:
            assert not out[b].any()
            continue
        ref = O.attention_with_lse(q[b:b + 1], k[b:b + 1, :n], v[b:b + 1, :n])[0]
        assert (out[b:b + 1].cpu().double() - ref).abs().max() < 2e-2, b
        assert torch.equal(out[b:b + 1], attention(q[b:b + 1].cuda(), k[b:b + 1, :n].cuda(), v[b:b + 1, :n].cuda()))
    with pytest.raises(ValueError):
        attention(q.cuda(), k.cuda(), v.cuda(), q_lens=torch.tensor([Lq, 5, Lq]))
    with pytest.raises(ValueError):
        attention(q.cuda(), k.cuda(), v.cuda(), k_lens=torch.tensor([1, 2]))


_CAUSVID_SEQ: dict = {}        # segment -> latents of the sequential run (pair=False runs first): the paired run must reproduce them


@pytest.mark.parametrize("pair", [False, True])
def test_causvid_rollover_vs_reference_golden(pair):
    """CausVid (BASELINE config 3 mechanics): explicit slot addressing, dropped last step, start_latents prefill and
    the per-segment request swap, against latents/caches generated by the reference's own CausVid pipeline."""
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    from inferix_amd.pipeline import CausVidInferencePipeline
    from inferix_amd.wan import HipCausVidDiffusionWrapper
    fx = golden("causvid_tiny.npz")
    cfg = O.tiny_config(text_len=512)
    m = build(cfg, O.init_weights(cfg, seed=0))
    gen = HipCausVidDiffusionWrapper(model=m, timestep_shift=8.0)
    # `pair`: the clean-context re-run of a block enqueued layer-interleaved with the next block's first step (forward_pair, which does
    # not pass through gen.forward): the slot schedule is then read from the model's own record of layer 0
    args = SimpleNamespace(denoising_step_list=fx["steps"].tolist(), warp_denoising_step=True, num_frame_per_block=3,
                           frame_seq_length=cfg.frame_seqlen, kv_cache_tokens=600, pair_forwards=pair)
    pe = fx["prompt_embeds"].cuda()
    pipe = CausVidInferencePipeline(args, device="cuda", generator=gen,
                                    text_encoder=lambda text_prompts: {"prompt_embeds": pe}, vae=None)
    kvm = KVCacheManager("cuda")
    slots = []
    orig = gen.forward

    def rec(**kw):
        slots.append([kw["kv_start"], kw["kv_end"]])
        return orig(**kw)
    gen.forward = rec
    renoise = [[fx[f"seg{s}_renoise_{i}"] for i in range(int(fx[f"seg{s}_num_renoise"]))] for s in range(2)]
    # segment 1 of the golden starts from the reference's own segment-0 tail (teacher forced): run segments one by one
    start = None
    for seg in range(2):
        req = [KVCacheRequest(f"seg{seg}")]
        slots.clear()
        m.index_trace = []
        lat = pipe.inference(fx[f"seg{seg}_noise"].cuda(), ["x"], start, return_latents=False, kv_cache_manager=kvm,
                             kv_cache_requests=req, decode=False, renoise=renoise[seg])
        torch.cuda.synchronize()
        if pair:
            assert pipe._pairing() and len(slots) < len(fx[f"seg{seg}_slots"]), "the paired calls must not pass through gen.forward"
            got_slots = [[le - (ge - cs), le] for cs, ge, le in m.index_trace]      # explicit slots: (start, end, end) per forward
            assert [s_[1] for s_ in got_slots] == [s_[1] for s_ in fx[f"seg{seg}_slots"].tolist()], "slot schedule differs"
        else:
            assert slots == fx[f"seg{seg}_slots"].tolist(), "cache slot schedule differs from the reference"
        if pair and seg in _CAUSVID_SEQ:             # (the sequential run comes first in the parametrisation; absent under -k selection)
            assert torch.equal(lat, _CAUSVID_SEQ[seg]), "paired forwards changed the CausVid latents"
        elif not pair:
            _CAUSVID_SEQ[seg] = lat.clone()
        n = fx[f"seg{seg}_cache_k"].shape[0]
        raw = kvm.get_raw(req[0], "layer_0")
        # floor rule (round-3 verdict: no fixed 1e-2): the reference's own distance from the exact-attention evaluation of the
        # same segment (oracle/gen_golden.py::gen_causvid) is the yardstick, for the latents and for layer 0's cache rows
        for what, got, key in (("latents", lat.cpu(), "out"), ("cache K", raw[0, :n, 0].cpu(), "cache_k"), ("cache V", raw[1, :n, 0].cpu(), "cache_v")):
            ref, exact = fx[f"seg{seg}_{key}"], fx[f"seg{seg}_{key}_exact"]
            floor, r_ref, r_exact = rel_l2(ref, exact), rel_l2(got, ref), rel_l2(got, exact)
            print(f"causvid segment {seg} {what}: floor {floor:.3e}; HIP vs exact {r_exact:.3e}; HIP vs reference {r_ref:.3e}")
            assert r_ref <= 1.25 * floor + 5e-4 and r_exact <= 1.25 * floor + 5e-4, (seg, what, floor, r_ref, r_exact)
        pipe.clear_cache(kvm, req)
        kvm.free(req[0])
        start = fx["seg1_start"].cuda() if seg == 0 else None
    # and the rollover driver end to end (own tail as overlap): shapes, request lifecycle
    gen.forward = orig
    outs = pipe.rollover(["a", "b"], [fx["seg0_noise"].cuda(), fx["seg1_noise"].cuda()], kvm, overlap_frames=3)
    assert [tuple(o.shape) for o in outs] == [tuple(fx["seg0_noise"].shape), tuple(fx["seg1_noise"].shape)]
    assert torch.equal(outs[1][:, :3], outs[0][:, -3:])          # prefilled overlap frames are passed through
    assert list(kvm.request_to_kv_caches) == ["segment_1"]


@pytest.mark.parametrize("lat_h,lat_w,frames", [(90, 160, 3), (34, 58, 6)])
def test_other_resolutions_vs_oracle(lat_h, lat_w, frames):
    """Nothing is tied to 480p: the 720p latent grid of SURVEY §8d config 3 (90 x 160 -> 3600 tokens per frame, block of
    10800) and an odd-sized grid run through the same kernels (RoPE grid, per-frame modulation groups, KV slots, attention
    tiles with ragged edges) and match the CPU oracle on one block rollout (2 denoise steps + context re-run)."""
    from inferix_amd.core import DecodeMode
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    from inferix_amd.pipeline import CausalInferencePipeline
    cfg = O.tiny_config(latent_h=lat_h, latent_w=lat_w)
    W = O.init_weights(cfg, seed=0)
    m, gen, args = _pipeline(cfg, W, [1000, 500], 8.0)
    args.kv_cache_tokens = frames * cfg.frame_seqlen
    g = torch.Generator().manual_seed(lat_h)
    noise = torch.randn(1, frames, 16, lat_h, lat_w, generator=g).to(BF)
    pe = torch.zeros(1, cfg.text_len, cfg.text_dim)
    pe[:, :9] = torch.randn(1, 9, cfg.text_dim, generator=g)
    pe = pe.to(BF)
    eps = [torch.randn(3, 16, lat_h, lat_w, generator=g).to(BF) for _ in range(frames // 3)]
    pipe = CausalInferencePipeline(args, "cuda", generator=gen, text_encoder=lambda text_prompts: {"prompt_embeds": pe.cuda()},
                                   vae=None)
    out = pipe.inference(noise=noise.cuda(), text_prompts=["x"], kv_cache_manager=KVCacheManager("cuda"),
                         kv_cache_requests=[KVCacheRequest("r")], decode_mode=DecodeMode.NO_DECODE, renoise=list(eps))
    torch.cuda.synchronize()
    ref, _ = O.inference(W, cfg, noise, list(pe), [1000, 500], renoise=list(eps), shift=8.0, num_frame_per_block=3)
    exact, _ = O.inference(W, cfg, noise, list(pe), [1000, 500], renoise=list(eps), shift=8.0, num_frame_per_block=3, attn_impl="math")
    floor, r, rx = rel_l2(ref, exact), rel_l2(out.cpu(), ref), rel_l2(out.cpu(), exact)
    print(f"{lat_h}x{lat_w}: floor (oracle bf16 SDPA vs exact attention) {floor:.3e}; HIP vs exact {rx:.3e}; HIP vs oracle {r:.3e}")
    assert torch.isfinite(out.float()).all()
    assert r <= 1.25 * floor + 5e-4 and rx <= 1.25 * floor + 5e-4, f"{lat_h}x{lat_w}: floor {floor:.3e}, vs oracle {r:.3e}, vs exact {rx:.3e}"


def test_full_depth_30_layers_real_channels_vs_oracle():
    """The DEPTH of the BASELINE model: all 30 layers at the real channel geometry (dim 1536, 12 heads, ffn 8960, text 4096 / 512,
    freq 256) on a small latent grid, a two-block rollout (2 denoise steps + context re-run each; the second block's first step runs
    paired with the first block's re-run) against the CPU oracle.  Yardstick as everywhere: the oracle with the reference's bf16 SDPA sits
    `floor` from the oracle with exact attention; the HIP rollout may be 1.25 x floor + 5e-4 from either.  (bench.py reports the same
    comparison at the full 4680-token size inside `cpu_baseline.parity_vs_gpu`: 7.9e-3 on one run.)"""
    from inferix_amd.core import DecodeMode
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    from inferix_amd.pipeline import CausalInferencePipeline
    cfg = O.WanConfig(num_layers=30, latent_h=8, latent_w=12)
    W = O.init_weights(cfg, seed=5)
    m, gen, args = _pipeline(cfg, W, [1000, 500], 5.0)
    args.kv_cache_tokens = 6 * cfg.frame_seqlen
    g = torch.Generator().manual_seed(30)
    noise = torch.randn(1, 6, 16, cfg.latent_h, cfg.latent_w, generator=g).to(BF)
    pe = torch.zeros(1, cfg.text_len, cfg.text_dim)
    pe[:, :12] = torch.randn(1, 12, cfg.text_dim, generator=g)
    pe = pe.to(BF)
    eps = [torch.randn(3, 16, cfg.latent_h, cfg.latent_w, generator=g).to(BF) for _ in range(2)]
    pipe = CausalInferencePipeline(args, "cuda", generator=gen, text_encoder=lambda text_prompts: {"prompt_embeds": pe.cuda()}, vae=None)
    out = pipe.inference(noise=noise.cuda(), text_prompts=["x"], kv_cache_manager=KVCacheManager("cuda"),
                         kv_cache_requests=[KVCacheRequest("r")], decode_mode=DecodeMode.NO_DECODE, renoise=list(eps))
    torch.cuda.synchronize()
    ref, _ = O.inference(W, cfg, noise, list(pe), [1000, 500], renoise=list(eps), shift=5.0, num_frame_per_block=3)
    exact, _ = O.inference(W, cfg, noise, list(pe), [1000, 500], renoise=list(eps), shift=5.0, num_frame_per_block=3, attn_impl="math")
    floor, r, rx = rel_l2(ref, exact), rel_l2(out.cpu(), ref), rel_l2(out.cpu(), exact)
    print(f"30 layers, real channels: floor (oracle bf16 SDPA vs exact attention) {floor:.3e}; HIP vs exact {rx:.3e}; HIP vs oracle {r:.3e}")
    assert torch.isfinite(out.float()).all()
    assert r <= 1.25 * floor + 5e-4 and rx <= 1.25 * floor + 5e-4, f"30 layers: floor {floor:.3e}, vs oracle {r:.3e}, vs exact {rx:.3e}"


def test_wan_14b_channel_geometry_vs_oracle():
    """The 14B variant's channel geometry (dim 5120 = 40 heads x 128, ffn 13824) through the same kernels: one layer, a
    two-block rollout on a small latent grid against the CPU oracle."""
    from inferix_amd.core import DecodeMode
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    from inferix_amd.pipeline import CausalInferencePipeline
    cfg = O.tiny_config(dim=5120, ffn_dim=13824, num_heads=40, num_layers=1, freq_dim=256)
    W = O.init_weights(cfg, seed=2)
    m, gen, args = _pipeline(cfg, W, [1000, 500], 5.0)
    args.kv_cache_tokens = 6 * cfg.frame_seqlen
    g = torch.Generator().manual_seed(14)
    noise = torch.randn(1, 6, 16, cfg.latent_h, cfg.latent_w, generator=g).to(BF)
    pe = torch.zeros(1, cfg.text_len, cfg.text_dim)
    pe[:, :9] = torch.randn(1, 9, cfg.text_dim, generator=g)
    pe = pe.to(BF)
    eps = [torch.randn(3, 16, cfg.latent_h, cfg.latent_w, generator=g).to(BF) for _ in range(2)]
    pipe = CausalInferencePipeline(args, "cuda", generator=gen, text_encoder=lambda text_prompts: {"prompt_embeds": pe.cuda()},
                                   vae=None)
    out = pipe.inference(noise=noise.cuda(), text_prompts=["x"], kv_cache_manager=KVCacheManager("cuda"),
                         kv_cache_requests=[KVCacheRequest("r")], decode_mode=DecodeMode.NO_DECODE, renoise=list(eps))
    torch.cuda.synchronize()
    ref, _ = O.inference(W, cfg, noise, list(pe), [1000, 500], renoise=list(eps), shift=5.0, num_frame_per_block=3)
    exact, _ = O.inference(W, cfg, noise, list(pe), [1000, 500], renoise=list(eps), shift=5.0, num_frame_per_block=3, attn_impl="math")
    floor, r, rx = rel_l2(ref, exact), rel_l2(out.cpu(), ref), rel_l2(out.cpu(), exact)
    print(f"14B geometry: floor (oracle bf16 SDPA vs exact attention) {floor:.3e}; HIP vs exact {rx:.3e}; HIP vs oracle {r:.3e}")
    assert torch.isfinite(out.float()).all()
    assert r <= 1.25 * floor + 5e-4 and rx <= 1.25 * floor + 5e-4, f"14B geometry: floor {floor:.3e}, vs oracle {r:.3e}, vs exact {rx:.3e}"


@pytest.mark.parametrize("case", [0, 1])
def test_block_full_size_vs_reference_golden(case):
    """One block at the BASELINE size — 4680 tokens, dim 1536, 12 heads, ffn 8960 — over L = 4680 (case 0) and L = 32760 (case 1)
    cached keys, against rows of the reference's own CausalWanAttentionBlock output (tests/golden/block_full_size.npz, generated
    by oracle/gen_golden_block_full.py from the reference import; inputs regenerated from seeds).  Yardstick: the fixture's
    `exact_rows` = the same rows with exact (fp64) self-attention; the reference sits `floor` from them, the HIP block must
    not sit further than 1.25x that."""
    import block_full_inputs as BI
    from inferix_amd import hip_ops as ops
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    fx = golden("block_full_size.npz")
    cfg = BI.config()
    W = O.init_weights(cfg, seed=3)
    d = BI.make(case)
    assert BI.checksum(d["x"]) == int(fx[f"c{case}_x_checksum"]), "seeded inputs drifted from the generator's"
    start = int(fx[f"c{case}_start"])
    m = build(cfg, W)
    fs, nf = cfg.frame_seqlen, BI.FRAMES
    n = nf * fs
    kvm, req = KVCacheManager("cuda"), [KVCacheRequest("r")]
    ad = m.blocks[0].kv_cache_manager
    ad.allocate_kv_cache(kv_cache_manager=kvm, kv_cache_request=req[0], sequence_length=7 * n, dtype=BF)
    ad.allocate_crossattn_cache(kv_cache_manager=kvm, kv_cache_request=req[0], crossattn_length=cfg.text_len, dtype=BF)
    raw = kvm.get_raw(req[0], "layer_0")
    if start:
        assert (BI.checksum(d["prefix_k"]) ^ BI.checksum(d["prefix_v"])) == int(fx[f"c{case}_prefix_checksum"])
        raw[0, :start, 0].copy_(d["prefix_k"])
        raw[1, :start, 0].copy_(d["prefix_v"])
    meta = {"global_end_index": torch.tensor([start]), "local_end_index": torch.tensor([start])}
    cmeta = {"is_init": False}
    x = d["x"][0].cuda().clone()
    El = (m.mod_all[0] + d["e0"][0].cuda()).contiguous()
    rope = ops.RopeGridSpec(m.freqs, start // fs, cfg.latent_h // 2, cfg.latent_w // 2)
    st = dict(B=1, N=n, F_=nf, fs=fs, rows_per_group=fs, rope=rope, sink_tokens=0, current_start=start, ctx=d["ctx"][0].cuda())
    m._run_block(0, x, El, st, meta, cmeta, kvm, req)
    torch.cuda.synchronize()
    sel = fx["sel"].long()
    got = x[sel.cuda()].cpu()
    ref, exact, floor = fx[f"c{case}_out_rows"], fx[f"c{case}_exact_rows"], float(fx[f"c{case}_floor"])
    d_exact, d_ref = rel_l2(got, exact), rel_l2(got, ref)
    print(f"full-size block case {case} (L = {start + n}): reference-vs-exact {floor:.3e}, hip-vs-exact {d_exact:.3e}, hip-vs-reference {d_ref:.3e}")
    assert d_exact <= 1.25 * floor + 5e-4, (d_exact, floor)
    assert_bf16_parity(got, ref, max_ulp=4, max_mismatch_frac=0.5, rel=1.25 * floor + 5e-4, floor=1.0, what="full-size block rows", report=True)
    assert int(meta["local_end_index"]) == start + n and int(meta["global_end_index"]) == start + n
    # K = RoPE(bf16(bf16(rmsnorm(k)) * w)) behind a K = 1536 GEMM: three chained bf16 roundings; at this size a few in 10^4
    # elements land 2 ULP from the reference's (1 ULP holds on the small-grid fixture above)
    assert_bf16_parity(raw[0, start + sel.cuda(), 0], fx[f"c{case}_k_rows"], max_ulp=2, max_mismatch_frac=0.01, floor=1.0,
                       what="cache K rows (post-RoPE)", report=True)
    # V is the raw GEMM output: one rounding; 2 ULP covers a flip across a binade boundary (1 in 10^3 elements differs at all)
    assert_bf16_parity(raw[1, start + sel.cuda(), 0], fx[f"c{case}_v_rows"], max_ulp=2, max_mismatch_frac=0.01, floor=0.05,
                       what="cache V rows", report=True)
    if start:
        assert torch.equal(raw[0, :start, 0].cpu(), d["prefix_k"]), "the prefix must not be touched"


def test_config1_full_size_full_depth_vs_reference_golden():
    """BASELINE config 1 at FULL SIZE and FULL DEPTH, pinned to the reference itself (round-5 verdict, item 3): Self-Forcing 480p,
    block_size 3, one denoise step + the clean-context re-run over one block — 4680 tokens, all 30 layers of the 1.3B model, NO_DECODE —
    through `CausalInferencePipeline.inference` of the HIP path, against tests/golden/config1_full.npz = the output latents and K / V
    cache rows of the REFERENCE's own pipeline on the same seeded weights, noise and prompt (oracle/gen_golden_config1_full.py; the
    inputs are regenerated from seeds here, checksums in the fixture).  Yardstick as everywhere: the reference sits `floor` (7.9e-3)
    from the same rollout with exact attention; the HIP result may sit 1.25 x floor + 5e-4 from either.  bench.py asserts the same
    bound on its own config-1 run (`cpu_baseline.parity_vs_gpu`)."""
    from fixture_io import weights_checksum
    from gen_golden_config1_full import BLOCK, config1_inputs
    from inferix_amd.core import DecodeMode
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    from inferix_amd.pipeline import CausalInferencePipeline
    fx = golden("config1_full.npz")
    cfg, W, noise, pe = config1_inputs()
    assert weights_checksum(W) == int(fx["weights_checksum"]), "seeded weights drifted from the generator's"
    assert int(noise.view(torch.int16).to(torch.int64).sum()) == int(fx["noise_checksum"])
    assert int(pe.view(torch.int16).to(torch.int64).sum()) == int(fx["prompt_checksum"])
    m, gen, args = _pipeline(cfg, W, [1000], 5.0)
    del W
    args.kv_cache_tokens = 32760
    pipe = CausalInferencePipeline(args, "cuda", generator=gen, text_encoder=lambda text_prompts: {"prompt_embeds": pe.cuda()}, vae=None)
    kvm, reqs = KVCacheManager("cuda"), [KVCacheRequest("r")]
    out = pipe.inference(noise=noise.cuda(), text_prompts=["x"], kv_cache_manager=kvm, kv_cache_requests=reqs,
                         decode_mode=DecodeMode.NO_DECODE, free_cache_before_vae=False)
    torch.cuda.synchronize()
    ref, exact, floor = fx["out"], fx["out_exact"], float(fx["floor"])
    assert float(fx["oracle_maxdiff"]) == 0.0, "the CPU oracle is pinned to the reference at this size and depth too"
    r, rx = rel_l2(out.cpu(), ref), rel_l2(out.cpu(), exact)
    print(f"config 1, 4680 tokens x 30 layers: floor (reference vs exact attention) {floor:.3e}; HIP vs reference {r:.3e}; HIP vs exact {rx:.3e}; "
          f"{ulp_report(out.cpu(), ref)}")
    assert torch.isfinite(out.float()).all()
    bound = 1.25 * floor + 5e-4
    assert r <= bound and rx <= bound, f"config 1 full: floor {floor:.3e}, vs reference {r:.3e}, vs exact {rx:.3e}"
    # the cache the re-run left behind (clean-context K / V of the block), layers 0 / 15 / 29: layer 0's rows are projections of the
    # denoised latents (one GEMM + norm + RoPE behind them), the deeper ones carry the depth's noise — held to their own floors
    sel = fx["sel"].long()
    for l in fx["layers_sel"].tolist():
        raw = kvm.get_raw(reqs[0], f"layer_{l}")
        for name, idx in (("k", 0), ("v", 1)):
            got = raw[idx, sel.cuda(), 0].cpu()
            rr, ee = fx[f"{name}_rows_l{l}"], fx[f"{name}_rows_exact_l{l}"]
            fl = rel_l2(rr, ee)
            d_ref, d_ex = rel_l2(got, rr), rel_l2(got, ee)
            print(f"  cache {name.upper()} layer {l}: floor {fl:.3e}, HIP vs reference {d_ref:.3e}, vs exact {d_ex:.3e}; {ulp_report(got, rr)}")
            assert d_ref <= 1.25 * fl + 5e-4 and d_ex <= 1.25 * fl + 5e-4, (l, name, fl, d_ref, d_ex)


@pytest.mark.parametrize("case", [0, 1])
def test_block_720p_full_size_vs_reference_golden(case):
    """BASELINE config 3 at its REAL size through the whole block (round-4 verdict: the 720p geometry was covered piecewise only):
    one CausVid block — 10800 tokens (3 x 45 x 80), dim 1536, 12 heads, ffn 8960, caller-named cache slots — over L = 10800
    (case 0, slots [0, 10800)) and L = 75600 (case 1, slots [64800, 75600)) keys, against rows of the reference's own CausVid
    CausalWanAttentionBlock output (tests/golden/block_720p_full_size.npz, oracle/gen_golden_block_720p.py; inputs regenerated from
    seeds).  These are the launches that exist at this size only: the six GEMMs at 10800 rows (FFN down on the 192-token split
    tile), the row kernels at 10800 x 1536, the two-rounds attention schedule over 75600 keys.  Rule: 1.25 x floor + 5e-4."""
    import block_720p_inputs as BI
    from inferix_amd import hip_ops as ops
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    fx = golden("block_720p_full_size.npz")
    cfg = BI.config()
    W = O.init_weights(cfg, seed=3)
    d = BI.make(case)
    assert BI.checksum(d["x"]) == int(fx[f"c{case}_x_checksum"]), "seeded inputs drifted from the generator's"
    start = int(fx[f"c{case}_start"])
    m = build(cfg, W)
    fs, nf = cfg.frame_seqlen, BI.FRAMES
    n = nf * fs
    end = start + n
    kvm, req = KVCacheManager("cuda"), [KVCacheRequest("r")]
    ad = m.blocks[0].kv_cache_manager
    ad.allocate_kv_cache(kv_cache_manager=kvm, kv_cache_request=req[0], sequence_length=BI.BLOCKS * n, dtype=BF)
    ad.allocate_crossattn_cache(kv_cache_manager=kvm, kv_cache_request=req[0], crossattn_length=cfg.text_len, dtype=BF)
    raw = kvm.get_raw(req[0], "layer_0")
    raw.zero_()
    if start:
        assert (BI.checksum(d["prefix_k"]) ^ BI.checksum(d["prefix_v"])) == int(fx[f"c{case}_prefix_checksum"])
        raw[0, :start, 0].copy_(d["prefix_k"])
        raw[1, :start, 0].copy_(d["prefix_v"])
    meta = {"global_end_index": torch.tensor([0]), "local_end_index": torch.tensor([0])}      # untouched with explicit slots
    cmeta = {"is_init": False}
    x = d["x"][0].cuda().clone()
    El = (m.mod_all[0] + d["e0"][0].cuda()).contiguous()
    rope = ops.RopeGridSpec(m.freqs, start // fs, cfg.latent_h // 2, cfg.latent_w // 2)
    st = dict(B=1, N=n, F_=nf, fs=fs, rows_per_group=fs, rope=rope, sink_tokens=0, current_start=start, ctx=d["ctx"][0].cuda(),
              explicit_slots=(start, end))
    m._run_block(0, x, El, st, meta, cmeta, kvm, req)
    torch.cuda.synchronize()
    sel = fx["sel"].long()
    got = x[sel.cuda()].cpu()
    ref, exact, floor = fx[f"c{case}_out_rows"], fx[f"c{case}_exact_rows"], float(fx[f"c{case}_floor"])
    d_exact, d_ref = rel_l2(got, exact), rel_l2(got, ref)
    print(f"720p full-size block case {case} (L = {end}): reference-vs-exact {floor:.3e}, hip-vs-exact {d_exact:.3e}, hip-vs-reference {d_ref:.3e}")
    assert torch.isfinite(x.float()).all()
    assert d_exact <= 1.25 * floor + 5e-4, (d_exact, floor)
    assert_bf16_parity(got, ref, max_ulp=4, max_mismatch_frac=0.5, rel=1.25 * floor + 5e-4, floor=1.0, what="720p full-size block rows", report=True)
    assert int(meta["local_end_index"]) == 0 and int(meta["global_end_index"]) == 0
    assert_bf16_parity(raw[0, start + sel.cuda(), 0], fx[f"c{case}_k_rows"], max_ulp=2, max_mismatch_frac=0.01, floor=1.0,
                       what="720p cache K rows (post-RoPE)")
    assert_bf16_parity(raw[1, start + sel.cuda(), 0], fx[f"c{case}_v_rows"], max_ulp=2, max_mismatch_frac=0.01, floor=0.05,
                       what="720p cache V rows")
    if start:
        assert torch.equal(raw[0, :start, 0].cpu(), d["prefix_k"]), "the prefix must not be touched"
    assert not bool(raw[:, end:].any()), "slots behind kv_end must stay untouched"


def _heavy_block_run(m, d, cfg, n, fs):
    """Two consecutive 3-frame blocks of the heavy-tailed fixture through `m`'s layer 0; returns the outputs and the cache tensor."""
    from inferix_amd import hip_ops as ops
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    kvm, req = KVCacheManager("cuda"), [KVCacheRequest("r")]
    ad = m.blocks[0].kv_cache_manager
    ad.allocate_kv_cache(kv_cache_manager=kvm, kv_cache_request=req[0], sequence_length=2 * n, dtype=BF)
    ad.allocate_crossattn_cache(kv_cache_manager=kvm, kv_cache_request=req[0], crossattn_length=cfg.text_len, dtype=BF)
    meta = {"global_end_index": torch.tensor([0]), "local_end_index": torch.tensor([0])}
    cmeta = {"is_init": False}
    ctx = d["context"][0].cuda()
    outs = []
    for b in range(2):
        x = d[f"x{b}"][0].cuda().clone()
        El = (m.mod_all[0] + d[f"e0_{b}"][0].cuda()).contiguous()
        rope = ops.RopeGridSpec(m.freqs, b * 3, cfg.latent_h // 2, cfg.latent_w // 2)
        st = dict(B=1, N=n, F_=3, fs=fs, rows_per_group=fs, rope=rope, sink_tokens=0, current_start=b * n, ctx=ctx)
        m._run_block(0, x, El, st, meta, cmeta, kvm, req)
        outs.append(x.cpu())
    torch.cuda.synchronize()
    return outs, kvm.get_raw(req[0], "layer_0")


def test_block_heavy_tailed_statistics_vs_reference_golden():
    """Realistic magnitudes (round-5 verdict, item 4): one block at the real channel geometry with OUTLIER statistics generated through
    the reference (tests/golden/block_heavy_tail.npz, oracle/gen_golden_block_heavy.py): norm_q / norm_k channels x 50 (scores spanning
    hundreds of nats), FFN rows x 50, a x 100 text token, activations up to 2^8, one frame with 4x modulation.  The paths that depend
    on magnitudes — the attention kernel's lazy row maximum with its rescale branch, q pre-multiplied by scale * log2 e, the bf16
    gate x residual epilogues — against the reference's rows by the floor rule, on the output AND on the block's update `out - x`
    (the input's own 2^8 entries dominate the output's norm).  A debug counter proves the rescale branch ran."""
    import gen_golden_block_heavy as GH
    from fixture_io import weights_checksum
    from inferix_amd import hip_ops as ops
    fx = golden("block_heavy_tail.npz")
    cfg = GH.config()
    W = GH.heavy_weights(cfg)
    assert weights_checksum(W) == int(fx["weights_checksum"]), "seeded weights drifted from the generator's"
    d = GH.make_inputs(cfg)
    assert all(torch.equal(d[k], fx[k]) for k in ("context", "x0", "x1", "e0_0", "e0_1")), "seeded inputs drifted from the generator's"
    m = build(cfg, W)
    fs = cfg.frame_seqlen
    n = 3 * fs
    # twice: the kernel the shape-based choice picks for a 288-row launch (the four-wave kernel, which tracks the row maximum per 32-key
    # block), and the software-pipelined ping-pong kernel of the full-size launches (attn_variant 7: lazy maximum + rescale branch)
    for variant in (0, 7):
        ops.set_option("attn_variant", variant)
        ops.set_option("attn_debug_counters", 1)
        try:
            outs, raw = _heavy_block_run(m, d, cfg, n, fs)
            rescales = ops.get_option("attn_rescale_count")
        finally:
            ops.set_option("attn_debug_counters", 0)
            ops.set_option("attn_variant", 0)
        print(f"heavy-tailed block, attn_variant {variant}: {rescales} (wave, key tile) pairs took the lazy-maximum rescale branch")
        if variant == 7:
            assert rescales > 0, "the fixture is meant to drive the ping-pong attention kernel through its rescale branch"
        for b in range(2):
            ref, exact, x = fx[f"out{b}"][0], fx[f"out{b}_exact"][0], fx[f"x{b}"][0]
            assert torch.isfinite(outs[b].float()).all()
            floor, r, rx = rel_l2(ref, exact), rel_l2(outs[b], ref), rel_l2(outs[b], exact)
            upd = lambda t: t.double() - x.double()
            floor_u, r_u, rx_u = rel_l2(upd(ref), upd(exact)), rel_l2(upd(outs[b]), upd(ref)), rel_l2(upd(outs[b]), upd(exact))
            print(f"heavy-tailed block #{b}: output floor {floor:.3e}, HIP vs reference {r:.3e}, vs exact {rx:.3e}; update (out - x) floor {floor_u:.3e}, "
                  f"HIP vs reference {r_u:.3e}, vs exact {rx_u:.3e}; {ulp_report(outs[b], ref)}")
            assert r <= 1.25 * floor + 5e-4 and rx <= 1.25 * floor + 5e-4, (variant, b, floor, r, rx)
            assert r_u <= 1.25 * floor_u + 5e-4 and rx_u <= 1.25 * floor_u + 5e-4, (variant, b, floor_u, r_u, rx_u)
    k_rel, v_rel = rel_l2(raw[0, :2 * n, 0].cpu(), fx["cache_k"]), rel_l2(raw[1, :2 * n, 0].cpu(), fx["cache_v"])
    print(f"heavy-tailed cache rows vs reference: K rel-L2 {k_rel:.3e} ({ulp_report(raw[0, :2 * n, 0].cpu(), fx['cache_k'])}), V {v_rel:.3e}")
    assert k_rel <= 1e-3 and v_rel <= 1e-3


@pytest.mark.parametrize("which", ["fp8", "int8"])
def test_block_heavy_tailed_statistics_quantised_vs_quantised_oracle(which):
    """The same heavy-tailed block on the 8-bit path (per-token x per-channel dynamic quantisation of all ten linears: the e4m3 clamp
    and the per-token scales see rows whose maximum is 2^8 while most entries are O(1)) against the quantised-model oracle evaluated
    here on the CPU (`wan_oracle` with `quant_oracle`'s linear under the reference's exclusion dict; DAX itself: unpinned).  Floor rule
    on the output and on the update, floors from the quantised oracle's own bf16-SDPA vs exact-attention runs."""
    import gen_golden_block_heavy as GH
    import quant_oracle as Q
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_hip_quant import _quantised
    cfg = GH.config()
    W = GH.heavy_weights(cfg)
    d = GH.make_inputs(cfg)
    fs = cfg.frame_seqlen
    n = 3 * fs
    fmt = Q.FP8 if which == "fp8" else Q.INT8
    freqs = O.rope_freqs(cfg.head_dim)
    want = {}
    for impl in ("sdpa", "math"):
        st = O.CacheState.allocate(cfg, 1, BF, cache_tokens=2 * n)
        with O.linear_override(Q.model_hook(Q.reference_qconfig_dict(fmt))):
            want[impl] = [O.block_forward(d[f"x{b}"], d[f"e0_{b}"], d["context"], W, 0, cfg, (3, cfg.latent_h // 2, cfg.latent_w // 2), freqs,
                                          st, b * n, attn_impl=impl)[0] for b in range(2)]
    m, _ = _quantised(cfg, W, which)
    outs, _ = _heavy_block_run(m, d, cfg, n, fs)
    for b in range(2):
        ref, exact, x = want["sdpa"][b], want["math"][b].to(BF), d[f"x{b}"][0]
        assert torch.isfinite(outs[b].float()).all()
        upd = lambda t: t.double() - x.double()
        floor, r, rx = rel_l2(ref, exact), rel_l2(outs[b], ref), rel_l2(outs[b], exact)
        floor_u, r_u, rx_u = rel_l2(upd(ref), upd(exact)), rel_l2(upd(outs[b]), upd(ref)), rel_l2(upd(outs[b]), upd(exact))
        print(f"heavy-tailed {which} block #{b}: output floor {floor:.3e}, HIP vs q8 oracle {r:.3e}, vs exact {rx:.3e}; update floor {floor_u:.3e}, "
              f"HIP vs q8 oracle {r_u:.3e}, vs exact {rx_u:.3e}")
        assert r <= 1.25 * floor + 5e-4 and rx <= 1.25 * floor + 5e-4, (b, floor, r, rx)
        assert r_u <= 1.25 * floor_u + 5e-4 and rx_u <= 1.25 * floor_u + 5e-4, (b, floor_u, r_u, rx_u)
