#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/magi_cso.npz by running the REFERENCE's own `cp_shuffle_overlap` functions
(inferix/distributed/parallelism/context_parallel.py:30-88 scatter / gather with shuffle + padding, :135-226 cross-attention
ranges, :258-307 `cp_shuffle_overlap_process`, :604-665 `cso_communication` / `CSOHelper`) on CPU: 4 gloo ranks in this container,
`parallel_state` pointed at the world group.  Only data (inputs / outputs) is stored.

Geometry: 3 denoising chunks of 10 tokens over 4 ranks -> every chunk is padded to 12 (cp_pad_size 6), every rank holds 3 tokens
of EVERY chunk.  The attention inside `CSOHelper.overlap` is an injected callable upstream (flash-attn); the generator injects
exact softmax attention over the chunk's key range, so the fixture pins the data movement: which rows each rank attends with, in
which order the query / output messages travel, and where every output row lands.

usage (build container only; /root/reference must exist):  python oracle/gen_golden_magi_cso.py
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

CP, DN, CHUNK = 4, 3, 10
SEQ, BATCH, DIM, ROPE = DN * CHUNK, 1, 16, 8
HQ, HK, HD = 8, 2, 16                          # 8 query heads, 2 kv heads (< cp: replicated x2 by the "kv" message)
CU_Q = [0, 10, 20, 30]                         # one packed cross-attention segment per denoising chunk
CU_K = [0, 5, 12, 16]


def make_inputs():
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(SEQ, BATCH, DIM, generator=g).to(torch.bfloat16)
    cond = torch.arange(SEQ * BATCH, dtype=torch.int32) % 3
    rope = torch.randn(SEQ, ROPE, generator=g)
    return dict(x=x, condition_map=cond, rope=rope)


def local_qkv(rank: int, n_local: int):
    """What a rank's projections produce for its (shuffled, padded) rows: seeded per rank."""
    g = torch.Generator().manual_seed(100 + rank)
    q = torch.randn(n_local, HQ, HD, generator=g).to(torch.bfloat16)
    kv = torch.randn(n_local, HK, 2 * HD, generator=g).to(torch.bfloat16)
    return q, kv


def ardf_meta():
    q_range = torch.tensor([[i * CHUNK, (i + 1) * CHUNK] for i in range(DN)], dtype=torch.int32)
    k_range = torch.tensor([[0, (i + 1) * CHUNK] for i in range(DN)], dtype=torch.int32)
    return dict(denoising_range_num=DN, q_range=q_range, k_range=k_range, max_seqlen_q=CHUNK, max_seqlen_k=DN * CHUNK)


def worker(rank: int, port: int, outdir: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=CP)
    _refstub.install()
    import importlib
    from einops import rearrange
    cpm = importlib.import_module("inferix.distributed.parallelism.context_parallel")
    types = importlib.import_module("inferix.core.types.inference")
    mpu = cpm.mpu
    mpu.get_cp_world_size = lambda: CP
    mpu.get_cp_rank = lambda: rank
    mpu.get_cp_group = lambda check_initialized=True: dist.group.WORLD
    sys.path.insert(0, HERE)
    import magi_cp_oracle as M
    inp = make_inputs()
    out = {}
    cross = types.PackedCrossAttnParams(
        q_ranges=None, kv_ranges=None, cu_seqlens_q=torch.tensor(CU_Q, dtype=torch.int32),
        cu_seqlens_kv=torch.tensor(CU_K, dtype=torch.int32), max_seqlen_q=CHUNK, max_seqlen_kv=7)
    meta = ardf_meta()
    x, cond, rope, pad, sizes, core_p, cross_p = cpm.cp_pre_process(CP, "cp_shuffle_overlap", inp["x"], inp["condition_map"],
                                                                     inp["rope"], None, meta, None, cross)
    out.update(pre_x=x, pre_cond=cond, pre_rope=rope, pad=torch.tensor(pad), sizes=torch.tensor(sizes),
               core_q_range=core_p.q_range, core_k_range=core_p.k_range, core_max_q=torch.tensor(int(core_p.max_seqlen_q)),
               core_max_k=torch.tensor(int(core_p.max_seqlen_k)),
               xq_ranges=cross_p.q_ranges, xk_ranges=cross_p.kv_ranges, xcu_q=cross_p.cu_seqlens_q, xcu_k=cross_p.cu_seqlens_kv,
               xmax_q=torch.tensor(cross_p.max_seqlen_q), xmax_k=torch.tensor(cross_p.max_seqlen_kv))
    margs = types.ModelMetaArgs(H=1, W=1, cp_pad_size=pad, cp_split_sizes=sizes, slice_point=0, denoising_range_num=DN,
                                range_num=DN, extract_prefix_video_feature=False, fwd_extra_1st_chunk=False,
                                distill_nearly_clean_chunk=False, clip_token_nums=CHUNK, enable_cuda_graph=False,
                                core_attn_params=core_p, cross_attn_params=cross_p)
    out["post_x"] = cpm.cp_post_process(CP, "cp_shuffle_overlap", x, margs)            # gather(scatter(x)) == x
    # ---- the attention layer's exchange (dit_module.py:1156-1188) with exact attention injected
    bsizes = [s * BATCH for s in sizes]
    q_loc, kv_loc = local_qkv(rank, bsizes[rank])
    kv, hkv = cpm.cso_communication(kv_loc, CP, bsizes, "kv")
    helper = cpm.CSOHelper(DN, CP, bsizes)
    qs, hq = helper.split_query_for_overlap(q_loc)
    hkv.wait()
    out["kv_a2a"] = kv
    kv = rearrange(kv, "(cp dn sqb) hn nhd -> dn (cp sqb) hn nhd", dn=DN, cp=CP)[:, :CHUNK].flatten(0, 1).contiguous()
    out["kv_unpadded"] = kv
    key, value = [t.contiguous() for t in torch.chunk(kv, 2, dim=-1)]
    hq.wait()
    out["q0_a2a"] = qs[0]
    k_range = core_p.np_k_range

    def fattn(q, k, v, i):
        return M.exact_attention(q, k[k_range[i, 0]:k_range[i, 1]], v[k_range[i, 0]:k_range[i, 1]]).to(torch.bfloat16).contiguous()

    outs, handle = helper.overlap(fattn, qs, key, value)
    handle.wait()
    for i, o in enumerate(outs):
        out[f"overlap_out{i}"] = o
    out["core_attn_out"] = rearrange(torch.concat(outs, dim=0), "(dn cp sq b) hn hd -> (dn sq) b (cp hn hd)", cp=CP, b=BATCH, dn=DN)
    save_npz(os.path.join(outdir, f"rank{rank}.npz"), out)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if not os.path.isdir("/root/reference"):
        raise SystemExit("needs /root/reference (build container only)")
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(worker, args=(29741, td), nprocs=CP, join=True)
        fx = {}
        for r in range(CP):
            for k, v in load_npz(os.path.join(td, f"rank{r}.npz")).items():
                fx[f"r{r}_{k}"] = v
    inp = make_inputs()
    fx.update(in_x=inp["x"], in_condition_map=inp["condition_map"], in_rope=inp["rope"],
              geom=torch.tensor([CP, DN, CHUNK, BATCH, DIM, ROPE, HQ, HK, HD]), cu_q=torch.tensor(CU_Q), cu_k=torch.tensor(CU_K))
    meta = ardf_meta()
    fx.update(ardf_q_range=meta["q_range"], ardf_k_range=meta["k_range"])
    for r in range(CP):
        q, kv = local_qkv(r, int(fx[f"r{r}_sizes"][r]) * BATCH)
        fx[f"r{r}_in_q"], fx[f"r{r}_in_kv"] = q, kv
    path = os.path.join(GOLDEN_DIR, "magi_cso.npz")
    save_npz(path, fx)
    print("wrote", path, os.path.getsize(path), "bytes,", len(fx), "arrays")


if __name__ == "__main__":
    main()
