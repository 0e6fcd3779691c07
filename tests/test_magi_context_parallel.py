"""CPU: MAGI context-parallel path (row a17).
 (1) the oracle restatement vs the golden produced by the REFERENCE's own functions under gloo (bit-exact);
 (2) the product module inferix_amd/magi/context_parallel.py under gloo with 4 ranks vs the same golden (bit-exact:
     the path is data movement + integer range arithmetic; attention inside the scheduler is an injected callable)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
import magi_cp_oracle as M  # noqa: E402
from fixture_io import golden  # noqa: E402


def _geom(fx):
    return [int(v) for v in fx["geom"].tolist()]


def test_oracle_matches_reference_golden():
    fx = golden("magi_cp.npz")
    CP, SEQ, B, DIM, ROPE, HQ, HK, HD = _geom(fx)
    sizes = fx["r0_sizes"].tolist()
    assert sizes == M.cp_split_sizes(SEQ, CP) == [10, 9, 9, 9]
    bs = [s * B for s in sizes]
    q_sh = [M.scatter(fx["in_q_full"], bs, r) for r in range(CP)]
    kv_sh = [M.scatter(fx["in_kv_full"], bs, r) for r in range(CP)]
    qa, kva = M.a2a_input_split(q_sh, bs), M.a2a_input_split(kv_sh, bs)
    back = M.a2a_output_split(qa, bs)
    fq, fk, fv = M.fused_qkv_communication(q_sh, [t[..., :HD].contiguous() for t in kv_sh],
                                           [t[..., HD:].contiguous() for t in kv_sh], bs)
    even = [SEQ // CP] * CP
    for r in range(CP):
        for name, mine in (("pre_x", M.scatter(fx["in_x"], sizes, r)), ("pre_cond", M.scatter(fx["in_condition_map"], sizes, r)),
                           ("pre_rope", M.scatter(fx["in_rope"], sizes, r)), ("q_a2a", qa[r]), ("kv_a2a", kva[r]),
                           ("q_back", back[r]), ("fused_q", fq[r]), ("fused_k", fk[r]), ("fused_v", fv[r])):
            assert torch.equal(fx[f"r{r}_{name}"], mine), (r, name)
        cr = M.cp_update_cross_attn_qkv_range(fx["cu_q"], fx["cu_k"], 7, B, sizes, r)
        assert torch.equal(fx[f"r{r}_xq_ranges"], cr.q_ranges) and torch.equal(fx[f"r{r}_xk_ranges"], cr.kv_ranges)
        assert torch.equal(fx[f"r{r}_xcu_q"], cr.cu_seqlens_q) and torch.equal(fx[f"r{r}_xcu_k"], cr.cu_seqlens_kv)
        assert int(fx[f"r{r}_xmax_q"]) == cr.max_seqlen_q and int(fx[f"r{r}_xmax_k"]) == cr.max_seqlen_kv
        assert torch.equal(fx[f"r{r}_post_x"], M.gather([M.scatter(fx["in_x"][:sum(even)], even, q) for q in range(CP)]))
    attn = lambda c, q, k, v: M.exact_attention(q, k, v).to(torch.bfloat16)
    for od in (1, -1):
        res = M.ulysses_attention(q_sh, kv_sh, bs, B, od, attn)
        for r in range(CP):
            assert torch.equal(fx[f"r{r}_sched_od{od}"], res[r]), (od, r)
    # the scheduler output equals plain attention over the gathered sequence, head-major channels (CP identity)
    full = M.exact_attention(fx["in_q_full"], fx["in_kv_full"][..., :HD], fx["in_kv_full"][..., HD:]).to(torch.bfloat16)
    got = torch.cat([fx[f"r{r}_sched_od1"] for r in range(CP)], dim=0)
    assert torch.equal(got, full.reshape(SEQ, B, HQ * HD))


def test_oracle_kv_cache_matches_reference_golden():
    fx = golden("magi_cp.npz")
    cache = M.MagiCacheOracle(int(fx["kvm_max_tokens"]), int(fx["kvm_hn"]), int(fx["kvm_hd"]))
    for i in range(int(fx["kvm_kv_calls"])):
        n, sp, upd, fe, di = fx[f"kvm_kv{i}_args"].tolist()
        k, v = cache.adjust(fx[f"kvm_kv{i}_in"], slice_point=sp, clip_token_nums=int(fx["kvm_clip"]),
                            update_kv_cache=bool(upd), fwd_extra_1st_chunk=bool(fe), distill_nearly_clean_chunk=bool(di))
        assert torch.equal(k, fx[f"kvm_kv{i}_k"]) and torch.equal(v, fx[f"kvm_kv{i}_v"]), i
    assert torch.equal(cache.mem, fx["kvm_kv_cache_final"])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from inferix_amd.magi import context_parallel as cp
        from inferix_amd.magi.types import ModelMetaArgs, PackedCrossAttnParams
        fx = golden("magi_cp.npz")
        CP, SEQ, B, DIM, ROPE, HQ, HK, HD = _geom(fx)
        cp.set_cp_group(dist.group.WORLD)
        bad = []

        def check(name, got):
            ref = fx[f"r{rank}_{name}"]
            if got.shape != ref.shape or not torch.equal(got, ref):
                bad.append(name)
        cross = PackedCrossAttnParams(cu_seqlens_q=fx["cu_q"].to(torch.int32), cu_seqlens_kv=fx["cu_k"].to(torch.int32),
                                      max_seqlen_q=20, max_seqlen_kv=7)
        x, cond, rope, pad, sizes, _, cross_p = cp.cp_pre_process(CP, "cp_ulysses", fx["in_x"], fx["in_condition_map"],
                                                                   fx["in_rope"], None, None, None, cross)
        assert pad == 0 and sizes == fx["r0_sizes"].tolist()
        check("pre_x", x), check("pre_cond", cond), check("pre_rope", rope)
        check("xq_ranges", cross_p.q_ranges), check("xk_ranges", cross_p.kv_ranges)
        check("xcu_q", cross_p.cu_seqlens_q), check("xcu_k", cross_p.cu_seqlens_kv)
        assert cross_p.max_seqlen_q == int(fx[f"r{rank}_xmax_q"]) and cross_p.max_seqlen_kv == 7
        bs = [s * B for s in sizes]
        off = sum(bs[:rank])
        q_loc = fx["in_q_full"][off:off + bs[rank]].contiguous()
        kv_loc = fx["in_kv_full"][off:off + bs[rank]].contiguous()
        qa, h = cp.all_to_all_input_split(q_loc, bs)
        h.wait()
        kva, h = cp.all_to_all_input_split(kv_loc, bs)
        h.wait()
        back, h = cp.all_to_all_output_split(qa.contiguous(), bs)
        h.wait()
        check("q_a2a", qa), check("kv_a2a", kva), check("q_back", back)
        fq, fk, fv = cp.fused_qkv_communication(q_loc, kv_loc[..., :HD].contiguous(), kv_loc[..., HD:].contiguous(), bs)
        check("fused_q", fq), check("fused_k", fk), check("fused_v", fv)
        attn = lambda q, k, v: M.exact_attention(q, k, v).to(torch.bfloat16).contiguous()
        split = lambda kv: tuple(t.contiguous() for t in torch.chunk(kv, 2, dim=-1))
        for od in (1, -1):
            core, _ = cp.UlyssesScheduler.get_attn_and_xattn_with_fused_kv_comm(
                lambda: q_loc, lambda: kv_loc, split, attn, lambda: None, od, B, CP, bs)
            check(f"sched_od{od}", core)
        core, _ = cp.UlyssesScheduler.get_attn_and_xattn_with_comm_overlap(
            lambda: q_loc, lambda: kv_loc[..., :HD].contiguous(), lambda: kv_loc[..., HD:].contiguous(), split, attn,
            lambda: None, 1, B, CP, bs)
        check("sched_od1", core)
        core, _ = cp.UlyssesScheduler.get_attn_and_xattn_with_fused_qkv_comm(
            lambda: (q_loc, kv_loc[..., :HD].contiguous(), kv_loc[..., HD:].contiguous()), split, attn, lambda: None,
            -1, B, CP, bs)
        check("sched_od-1", core)
        even = [SEQ // CP] * CP
        xe = cp.scatter_to_context_parallel_region(fx["in_x"][:sum(even)], even)
        meta = ModelMetaArgs(H=1, W=1, cp_pad_size=0, cp_split_sizes=even, slice_point=0, denoising_range_num=1, range_num=1,
                             extract_prefix_video_feature=False, fwd_extra_1st_chunk=False, distill_nearly_clean_chunk=False,
                             clip_token_nums=SEQ, enable_cuda_graph=False, core_attn_params=None, cross_attn_params=cross_p)
        check("post_x", cp.cp_post_process(CP, "cp_ulysses", xe, meta))
        try:
            cp.cp_pre_process(CP, "nope", fx["in_x"], fx["in_condition_map"], fx["in_rope"], None, {}, None, cross)
            bad.append("bad strategy did not raise")
        except ValueError:
            pass
        ret[rank] = bad
    finally:
        dist.destroy_process_group()


def test_product_context_parallel_gloo_world4_matches_reference_golden():
    world = 4
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert len(ret) == world
        for r in range(world):
            assert ret[r] == [], f"rank {r}: mismatching entries {ret[r]}"


def test_single_rank_is_identity():
    from inferix_amd.magi import context_parallel as cp
    x = torch.arange(12.).view(6, 1, 2)
    assert cp.cp_pre_process(1, "cp_ulysses", x, None, None, None, None, None, None)[0] is x
    assert cp.cp_post_process(1, "cp_ulysses", x, None) is x
    t, h = cp.all_to_all_input_split(x, None)
    h.wait()
    assert t is x


def _cso_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from inferix_amd.magi import context_parallel as cp
        from inferix_amd.magi.types import ModelMetaArgs, PackedCrossAttnParams
        fx = golden("magi_cso.npz")
        CP, DN, CHUNK, B, DIM, ROPE, HQ, HK, HD = [int(v) for v in fx["geom"].tolist()]
        cp.set_cp_group(dist.group.WORLD)
        bad = []

        def check(name, got):
            ref = fx[f"r{rank}_{name}"]
            if tuple(got.shape) != tuple(ref.shape) or not torch.equal(got.to(ref.dtype), ref):
                bad.append(name)
        cross = PackedCrossAttnParams(cu_seqlens_q=fx["cu_q"].to(torch.int32), cu_seqlens_kv=fx["cu_k"].to(torch.int32),
                                      max_seqlen_q=CHUNK, max_seqlen_kv=7)
        ardf = dict(denoising_range_num=DN, q_range=fx["ardf_q_range"].to(torch.int32), k_range=fx["ardf_k_range"].to(torch.int32),
                    max_seqlen_q=CHUNK, max_seqlen_k=DN * CHUNK)
        x, cond, rope, pad, sizes, core_p, cross_p = cp.cp_pre_process(CP, "cp_shuffle_overlap", fx["in_x"], fx["in_condition_map"],
                                                                         fx["in_rope"], None, ardf, None, cross)
        assert pad == int(fx[f"r{rank}_pad"]) == 6 and sizes == fx[f"r{rank}_sizes"].tolist() == [9, 9, 9, 9]
        check("pre_x", x), check("pre_cond", cond), check("pre_rope", rope)
        check("core_q_range", core_p.q_range), check("core_k_range", core_p.k_range)
        assert int(core_p.max_seqlen_q) == int(fx[f"r{rank}_core_max_q"]) and int(core_p.max_seqlen_k) == int(fx[f"r{rank}_core_max_k"])
        assert core_p.np_q_range.tolist() == core_p.q_range.tolist() and core_p.np_k_range.tolist() == core_p.k_range.tolist()
        check("xq_ranges", cross_p.q_ranges), check("xk_ranges", cross_p.kv_ranges)
        check("xcu_q", cross_p.cu_seqlens_q), check("xcu_k", cross_p.cu_seqlens_kv)
        assert cross_p.max_seqlen_q == int(fx[f"r{rank}_xmax_q"]) and cross_p.max_seqlen_kv == int(fx[f"r{rank}_xmax_k"])
        meta = ModelMetaArgs(H=1, W=1, cp_pad_size=pad, cp_split_sizes=sizes, slice_point=0, denoising_range_num=DN, range_num=DN,
                             extract_prefix_video_feature=False, fwd_extra_1st_chunk=False, distill_nearly_clean_chunk=False,
                             clip_token_nums=CHUNK, enable_cuda_graph=False, core_attn_params=core_p, cross_attn_params=cross_p)
        post = cp.cp_post_process(CP, "cp_shuffle_overlap", x, meta)
        check("post_x", post)
        if not torch.equal(post, fx["in_x"]):
            bad.append("gather(scatter(x)) != x")
        # the attention layer's exchange with exact attention injected (dit_module.py:1156-1188)
        bs = [s * B for s in sizes]
        q_loc, kv_loc = fx[f"r{rank}_in_q"], fx[f"r{rank}_in_kv"]
        kv, hkv = cp.cso_communication(kv_loc, CP, bs, "kv")
        helper = cp.CSOHelper(DN, CP, bs)
        qs, hq = helper.split_query_for_overlap(q_loc)
        hkv.wait()
        check("kv_a2a", kv)
        m = kv.shape[0] // (CP * DN)
        kvu = kv.view(CP, DN, m, *kv.shape[1:]).transpose(0, 1).reshape(DN, CP * m, *kv.shape[1:])[:, :CHUNK].flatten(0, 1).contiguous()
        check("kv_unpadded", kvu)
        key, value = [t.contiguous() for t in torch.chunk(kvu, 2, dim=-1)]
        hq.wait()
        check("q0_a2a", qs[0])
        kr = core_p.np_k_range

        def fattn(q, k, v, i):
            return M.exact_attention(q, k[kr[i, 0]:kr[i, 1]], v[kr[i, 0]:kr[i, 1]]).to(torch.bfloat16).contiguous()
        outs, h = helper.overlap(fattn, qs, key, value)
        h.wait()
        assert len(outs) == DN
        for i, o in enumerate(outs):
            check(f"overlap_out{i}", o)
        cat = torch.concat(outs, dim=0)                               # (dn cp sq b) hn hd -> (dn sq) b (cp hn hd)
        sq = cat.shape[0] // (DN * CP * B)
        core = cat.view(DN, CP, sq, B, *cat.shape[1:]).permute(0, 2, 3, 1, 4, 5).reshape(DN * sq, B, -1)
        check("core_attn_out", core)
        ret[rank] = bad
    finally:
        dist.destroy_process_group()


def test_product_context_shuffle_overlap_gloo_world4_matches_reference_golden():
    """cp_shuffle_overlap (the reference's second CP strategy): shuffled + padded scatter / gather, stretched query ranges,
    cross-attention ranges per (batch, chunk) window, the "kv" message, and CSOHelper's interleaving of query / output messages
    with the per-chunk attention — bit-exact against what the reference's own functions produced under gloo with 4 ranks."""
    world = 4
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_cso_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert len(ret) == world
        for r in range(world):
            assert ret[r] == [], f"rank {r}: mismatching entries {ret[r]}"


def test_kv_adapter_host_logic_matches_oracle_for_every_flag_combination():
    """The product adapter's placement rule (in-place rows / scratch tail / two-segment map) against the cache oracle on CPU, for
    every (slice_point, update_kv_cache, fwd_extra_1st_chunk, nearly_clean) the reference can reach — including forward_3cfg's first
    pass (slice_point = 0, update_kv_cache = False), where NOTHING is in place in front of the unstored rows and a map with
    seg_split == 0 would read as "no map" (round-2 advisor finding)."""
    from inferix_amd.magi.attention import MagiKVCacheManager
    from inferix_amd.magi.types import InferenceParams, ModelMetaArgs
    hn, hd, clip, cap = 2, 8, 4, 32
    g = torch.Generator().manual_seed(11)
    for sp in (0, 1, 2):
        for upd in (False, True):
            for fe in (False, True):
                for di in (False, True):
                    if sp == 0 and not fe:
                        continue                      # no cache involvement: fresh planes, nothing to place
                    mgr = MagiKVCacheManager(0, hn, hd, None)
                    ip = InferenceParams(1, cap, device="cpu")
                    oracle = M.MagiCacheOracle(cap, hn, hd)
                    if sp:                            # a clean prefix first
                        pre = torch.randn(sp * clip, hn, 2 * hd, generator=g).to(torch.bfloat16)
                        ip.update_kv_cache = True
                        m0 = ModelMetaArgs(H=1, W=1, cp_pad_size=0, cp_split_sizes=None, slice_point=0, denoising_range_num=1,
                                           range_num=1, extract_prefix_video_feature=False, fwd_extra_1st_chunk=True,
                                           distill_nearly_clean_chunk=False, clip_token_nums=clip, enable_cuda_graph=False,
                                           core_attn_params=None, cross_attn_params=None)
                        mgr.adjust_key_and_value_for_inference(pre, ip, m0)
                        oracle.adjust(pre, slice_point=0, clip_token_nums=clip, update_kv_cache=True, fwd_extra_1st_chunk=True)
                    kv = torch.randn(2 * clip, hn, 2 * hd, generator=g).to(torch.bfloat16)
                    ip.update_kv_cache = upd
                    meta = ModelMetaArgs(H=1, W=1, cp_pad_size=0, cp_split_sizes=None, slice_point=sp, denoising_range_num=1,
                                         range_num=1, extract_prefix_video_feature=False, fwd_extra_1st_chunk=fe,
                                         distill_nearly_clean_chunk=di, clip_token_nums=clip, enable_cuda_graph=False,
                                         core_attn_params=None, cross_attn_params=None)
                    handle = mgr.adjust_key_and_value_for_inference(kv, ip, meta)
                    kr, vr = oracle.adjust(kv, slice_point=sp, clip_token_nums=clip, update_kv_cache=upd,
                                           fwd_extra_1st_chunk=fe, distill_nearly_clean_chunk=di)
                    k, v = handle.materialize()
                    tag = (sp, upd, fe, di)
                    assert handle.kv_len == kr.shape[0], tag
                    assert torch.equal(k, kr) and torch.equal(v, vr), tag
                    # a view never carries a map that readers would take for "no map"
                    assert not (handle.view.seg_delta and not handle.view.seg_split), tag
                    raw = ip.kv_cache_manager.get_raw(ip.kv_cache_request, "layer_0")
                    rows = sp * clip + ((2 * clip - (clip if di else 0)) if upd else 0)      # what the rule has stored so far
                    assert torch.equal(raw[:, :rows, 0], oracle.mem.reshape(2, cap, hn, hd)[:, :rows]), tag
