#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/magi_cp.npz by running the REFERENCE's own MAGI context-parallel
functions (inferix/distributed/parallelism/context_parallel.py, kvcache_manager/model/magi_kv_cache_manager.py) on CPU:
4 gloo ranks in this container, `parallel_state` pointed at the world group.  Only data (inputs / outputs) is stored.

The attention arithmetic inside the scheduler is a callable the caller injects (flash-attn / magi_attention upstream,
absent here); the generator injects exact softmax attention so that the fixture pins the DATA MOVEMENT of the
scheduler: which (token, head) each rank attends with and where every output row lands.

usage (build container only; /root/reference must exist):  python oracle/gen_golden_magi.py
"""
from __future__ import annotations

import os
import sys
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refstub  # noqa: E402
from fixture_io import GOLDEN_DIR, load_npz, save_npz  # noqa: E402

CP = 4
SEQ, BATCH, DIM, ROPE = 37, 1, 16, 8          # 37 tokens -> uneven shards 10, 9, 9, 9
HQ, HK, HD = 8, 2, 16                          # 8 query heads, 2 kv heads (< cp: replicated x2), head_dim 16
CU_Q = [0, 20, 37]                             # two packed cross-attention segments
CU_K = [0, 5, 12]


def make_inputs():
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(SEQ, BATCH, DIM, generator=g).to(torch.bfloat16)
    cond = torch.arange(SEQ * BATCH, dtype=torch.int32) % 3
    rope = torch.randn(SEQ, ROPE, generator=g)
    q_full = torch.randn(SEQ * BATCH, HQ, HD, generator=g).to(torch.bfloat16)
    kv_full = torch.randn(SEQ * BATCH, HK, 2 * HD, generator=g).to(torch.bfloat16)
    return dict(x=x, condition_map=cond, rope=rope, q_full=q_full, kv_full=kv_full)


def exact_attn(q, k, v):
    import magi_cp_oracle as M
    return M.exact_attention(q, k, v).to(torch.bfloat16).contiguous()


def worker(rank: int, port: int, outdir: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=CP)
    _refstub.install()
    import importlib
    cpm = importlib.import_module("inferix.distributed.parallelism.context_parallel")
    types = importlib.import_module("inferix.core.types.inference")
    mpu = cpm.mpu
    mpu.get_cp_world_size = lambda: CP
    mpu.get_cp_rank = lambda: rank
    mpu.get_cp_group = lambda check_initialized=True: dist.group.WORLD
    inp = make_inputs()
    out = {}
    cross = types.PackedCrossAttnParams(
        q_ranges=None, kv_ranges=None, cu_seqlens_q=torch.tensor(CU_Q, dtype=torch.int32),
        cu_seqlens_kv=torch.tensor(CU_K, dtype=torch.int32), max_seqlen_q=20, max_seqlen_kv=7)
    # ---- cp_pre_process (cp_ulysses): split sizes, scatter, cross-attention ranges
    x, cond, rope, pad, sizes, core_p, cross_p = cpm.cp_pre_process(CP, "cp_ulysses", inp["x"], inp["condition_map"],
                                                                     inp["rope"], None, None, None, cross)
    assert pad == 0
    out.update(pre_x=x, pre_cond=cond, pre_rope=rope, sizes=torch.tensor(sizes), xq_ranges=cross_p.q_ranges,
               xk_ranges=cross_p.kv_ranges, xcu_q=cross_p.cu_seqlens_q, xcu_k=cross_p.cu_seqlens_kv,
               xmax_q=torch.tensor(cross_p.max_seqlen_q), xmax_k=torch.tensor(cross_p.max_seqlen_kv))
    bsizes = [s * BATCH for s in sizes]
    off = sum(bsizes[:rank])
    q_loc = inp["q_full"][off:off + bsizes[rank]].contiguous()
    kv_loc = inp["kv_full"][off:off + bsizes[rank]].contiguous()
    # ---- the two all-to-alls and the fused variant
    q_a2a, h = cpm.all_to_all_input_split(q_loc, bsizes)
    h.wait()
    kv_a2a, h = cpm.all_to_all_input_split(kv_loc, bsizes)
    h.wait()
    back, h = cpm.all_to_all_output_split(q_a2a.contiguous(), bsizes)
    h.wait()
    k_loc, v_loc = [t.contiguous() for t in torch.chunk(kv_loc, 2, dim=-1)]
    fq, fk, fv = cpm.fused_qkv_communication(q_loc, k_loc, v_loc, bsizes)
    out.update(q_a2a=q_a2a, kv_a2a=kv_a2a, q_back=back, fused_q=fq, fused_k=fk, fused_v=fv)
    # ---- scheduler, fused-kv variant, overlap degrees 1 and -1 (= q heads per kv head)
    for od in (1, -1):
        core, xo = cpm.UlyssesScheduler.get_attn_and_xattn_with_fused_kv_comm(
            lambda: q_loc, lambda: kv_loc, lambda kv: tuple(t.contiguous() for t in torch.chunk(kv, 2, dim=-1)),
            exact_attn, lambda: torch.zeros(1), od, BATCH, CP, bsizes)
        out[f"sched_od{od}"] = core
    # ---- cp_post_process (gloo's all_gather needs equal shards -> an evenly divisible prefix of x; NCCL/RCCL takes
    #      the uneven list as is)
    even = [SEQ // CP] * CP
    x_even = cpm.scatter_to_context_parallel_region(inp["x"][:sum(even)], even)
    meta = types.ModelMetaArgs(H=1, W=1, cp_pad_size=0, cp_split_sizes=even, slice_point=0, denoising_range_num=1,
                               range_num=1, extract_prefix_video_feature=False, fwd_extra_1st_chunk=False,
                               distill_nearly_clean_chunk=False, clip_token_nums=SEQ, enable_cuda_graph=False,
                               core_attn_params=None, cross_attn_params=cross_p)
    out["post_x"] = cpm.cp_post_process(CP, "cp_ulysses", x_even, meta)
    save_npz(os.path.join(outdir, f"rank{rank}.npz"), out)
    dist.barrier()
    dist.destroy_process_group()


def kv_manager_fixture():
    """Reference MagiKVCacheManager on CPU, one layer: three calls (store chunk 0; read prefix + store chunk 1 with the
    nearly-clean rule; read-only call)."""
    _refstub.install()
    import importlib
    import types as pytypes
    mk = importlib.import_module("inferix.kvcache_manager.model.magi_kv_cache_manager")
    types = importlib.import_module("inferix.core.types.inference")
    kvm_mod = importlib.import_module("inferix.kvcache_manager.kvcache_manager")
    hn, hd, clip, max_tokens = 2, 16, 6, 30
    mgr = mk.MagiKVCacheManager(0, hn, hd, pytypes.SimpleNamespace(kv_offload=False))
    ip = object.__new__(types.InferenceParams)
    ip.max_sequence_length, ip.max_batch_size, ip.sequence_len_offset = max_tokens, 1, 0
    ip.kv_cache_request = kvm_mod.KVCacheRequest(request_id="magi")
    ip.kv_cache_manager = kvm_mod.KVCacheManager(device="cpu")
    ip.key_value_memory_dict, ip.update_kv_cache = {}, False
    g = torch.Generator().manual_seed(77)
    out = {"hn": torch.tensor(hn), "hd": torch.tensor(hd), "clip": torch.tensor(clip), "max_tokens": torch.tensor(max_tokens)}
    calls = [dict(n=12, slice_point=0, update=True, fwd_extra=True, distill=False),
             dict(n=12, slice_point=2, update=True, fwd_extra=False, distill=True),
             dict(n=6, slice_point=3, update=False, fwd_extra=False, distill=False),
             dict(n=6, slice_point=0, update=False, fwd_extra=False, distill=False)]      # no cache involvement
    for i, c in enumerate(calls):
        kv = torch.randn(c["n"], hn, 2 * hd, generator=g).to(torch.bfloat16)
        meta = types.ModelMetaArgs(H=1, W=1, cp_pad_size=0, cp_split_sizes=None, slice_point=c["slice_point"],
                                   denoising_range_num=1, range_num=1, extract_prefix_video_feature=False,
                                   fwd_extra_1st_chunk=c["fwd_extra"], distill_nearly_clean_chunk=c["distill"],
                                   clip_token_nums=clip, enable_cuda_graph=False, core_attn_params=None,
                                   cross_attn_params=None)
        ip.update_kv_cache = c["update"]
        k, v = mgr.adjust_key_and_value_for_inference(kv, ip, meta)
        out.update({f"kv{i}_in": kv, f"kv{i}_k": k, f"kv{i}_v": v,
                    f"kv{i}_args": torch.tensor([c["n"], c["slice_point"], int(c["update"]), int(c["fwd_extra"]), int(c["distill"])])})
    # the reference allocates the cache with torch.empty (kvcache_manager.py:232-243): slots no call stored to hold whatever the
    # allocator returned.  Only the stored extent is part of the contract, so the never-written tail is zeroed in the fixture
    # (otherwise a regeneration differs from the committed file and from the oracle's zero-initialised cache).
    final = ip.kv_cache_manager.get_raw(ip.kv_cache_request, "layer_0").clone()
    written = 0
    for c in calls:
        if c["update"]:
            stored = c["n"] - clip if c["distill"] else c["n"]
            written = max(written, c["slice_point"] * clip + stored)
    final[:, written:] = 0
    out["kv_cache_final"] = final
    out["kv_cache_written"] = torch.tensor(written)
    out["kv_calls"] = torch.tensor(len(calls))
    return out


def check_oracle(fx):
    """The restatement must reproduce every reference output bit for bit."""
    import magi_cp_oracle as M
    inp = make_inputs()
    sizes = fx["r0_sizes"].tolist()
    assert sizes == M.cp_split_sizes(SEQ, CP)
    bs = [s * BATCH for s in sizes]
    q_sh = [M.scatter(inp["q_full"], bs, r) for r in range(CP)]
    kv_sh = [M.scatter(inp["kv_full"], bs, r) for r in range(CP)]
    qa, kva = M.a2a_input_split(q_sh, bs), M.a2a_input_split(kv_sh, bs)
    back = M.a2a_output_split(qa, bs)
    fq, fk, fv = M.fused_qkv_communication(q_sh, [t[..., :HD].contiguous() for t in kv_sh],
                                           [t[..., HD:].contiguous() for t in kv_sh], bs)
    for r in range(CP):
        for name, ref in (("pre_x", M.scatter(inp["x"], sizes, r)), ("pre_cond", M.scatter(inp["condition_map"], sizes, r)),
                          ("pre_rope", M.scatter(inp["rope"], sizes, r)), ("q_a2a", qa[r]), ("kv_a2a", kva[r]),
                          ("q_back", back[r]), ("fused_q", fq[r]), ("fused_k", fk[r]), ("fused_v", fv[r])):
            assert torch.equal(fx[f"r{r}_{name}"], ref), (r, name)
        cr = M.cp_update_cross_attn_qkv_range(torch.tensor(CU_Q), torch.tensor(CU_K), 7, BATCH, sizes, r)
        assert torch.equal(fx[f"r{r}_xq_ranges"], cr.q_ranges) and torch.equal(fx[f"r{r}_xk_ranges"], cr.kv_ranges)
        assert torch.equal(fx[f"r{r}_xcu_q"], cr.cu_seqlens_q) and torch.equal(fx[f"r{r}_xcu_k"], cr.cu_seqlens_kv)
        assert int(fx[f"r{r}_xmax_q"]) == cr.max_seqlen_q
        even = [SEQ // CP] * CP
        assert torch.equal(fx[f"r{r}_post_x"], M.gather([M.scatter(inp["x"][:sum(even)], even, q) for q in range(CP)]))
    for od in (1, -1):
        res = M.ulysses_attention(q_sh, kv_sh, bs, BATCH, od, lambda c, q, k, v: exact_attn(q, k, v))
        for r in range(CP):
            assert torch.equal(fx[f"r{r}_sched_od{od}"], res[r]), ("sched", od, r)
    cache = M.MagiCacheOracle(int(fx["kvm_max_tokens"]), int(fx["kvm_hn"]), int(fx["kvm_hd"]))
    for i in range(int(fx["kvm_kv_calls"])):
        n, sp, upd, fe, di = fx[f"kvm_kv{i}_args"].tolist()
        k, v = cache.adjust(fx[f"kvm_kv{i}_in"], slice_point=sp, clip_token_nums=int(fx["kvm_clip"]), update_kv_cache=bool(upd),
                            fwd_extra_1st_chunk=bool(fe), distill_nearly_clean_chunk=bool(di))
        assert torch.equal(k, fx[f"kvm_kv{i}_k"]) and torch.equal(v, fx[f"kvm_kv{i}_v"]), ("kv", i)
    assert torch.equal(cache.mem, fx["kvm_kv_cache_final"])
    print("oracle == reference on every MAGI CP fixture entry (bit-exact)")


def main():
    if not _refstub.available():
        raise SystemExit("reference tree not available: fixtures can only be generated in the build container")
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(worker, args=(29731, td), nprocs=CP, join=True)
        merged = {}
        for r in range(CP):
            for k, v in load_npz(os.path.join(td, f"rank{r}.npz")).items():
                merged[f"r{r}_{k}"] = v
    for k, v in kv_manager_fixture().items():
        merged[f"kvm_{k}"] = v
    merged.update({f"in_{k}": v for k, v in make_inputs().items()})
    merged["geom"] = torch.tensor([CP, SEQ, BATCH, DIM, ROPE, HQ, HK, HD])
    merged["cu_q"], merged["cu_k"] = torch.tensor(CU_Q), torch.tensor(CU_K)
    path = os.path.join(GOLDEN_DIR, "magi_cp.npz")
    save_npz(path, merged)
    check_oracle(load_npz(path))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
