"""MAGI context parallelism (Ulysses): sequence split, head <-> sequence all-to-all, attention scheduling.

Mirror of the reference's `inferix/distributed/parallelism/context_parallel.py` for the `cp_ulysses` strategy — the same
function names, argument meaning and results — on torch.distributed, whose "nccl" backend is RCCL on ROCm.  The
all-to-all is the natural collective of the fully connected xGMI mesh (every pair of GPUs has its own link, so all
P-1 peer messages of a rank move concurrently); `cp_shuffle_overlap` (context_parallel.py:258-307, 604-665: every rank holds a
slice of every denoising chunk, queries and outputs of neighbouring chunks travel under the attention of the current one) is the
reference's strategy for PCIe-attached GPUs and is built for parity.

The process group is held by this module (`set_cp_group`), where the reference asks `parallel_state`
(distributed/parallel_state.py:498-503,620-634).  All functions also run on CPU tensors over gloo, which is how
tests/test_magi_context_parallel.py checks them against the reference's golden outputs with 4 ranks.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple, Union

import torch
import torch.distributed as dist

from .types import ModelMetaArgs, PackedCoreAttnParams, PackedCrossAttnParams

_CP_GROUP = None
_CP_EMULATED = None          # (world, rank): ONE rank of a cp-way run timed on one GPU, collectives replaced by copies of the same bytes


def set_cp_group(group) -> None:
    """Register the context-parallel process group (None = the default world group once initialised)."""
    global _CP_GROUP
    _CP_GROUP = group


def set_cp_emulation(world: Optional[int], rank: int = 0) -> None:
    """Benchmarking aid (bench.py `magi_cp8_emulated`): behave as rank `rank` of a `world`-way context-parallel run inside one
    process.  Every all-to-all becomes a device copy that moves the bytes this rank would receive (its own shard repeated in
    place of its peers'), so kernels see the real shapes and the copy stands in for the xGMI transfer; results are NOT those
    of a real run.  `None` switches it off."""
    global _CP_EMULATED
    _CP_EMULATED = None if not world or world <= 1 else (int(world), int(rank))


def get_cp_group():
    return _CP_GROUP if _CP_GROUP is not None else (dist.group.WORLD if dist.is_initialized() else None)


def get_cp_world_size() -> int:
    if _CP_EMULATED is not None:
        return _CP_EMULATED[0]
    g = get_cp_group()
    return dist.get_world_size(g) if g is not None else 1


def get_cp_rank() -> int:
    if _CP_EMULATED is not None:
        return _CP_EMULATED[1]
    g = get_cp_group()
    return dist.get_rank(g) if g is not None else 0


_EMU_IDX: dict = {}      # emulation only: (piece sizes, own piece, offset, device) -> row index of the stand-in gather


def _a2a(out: torch.Tensor, inp: torch.Tensor, out_sizes=None, in_sizes=None):
    """`dist.all_to_all_single` with two extras: under emulation a local copy of the same bytes; over gloo (CPU-only
    all-to-all: the 1-GPU test box runs 2 ranks on one device) device tensors are staged through the host."""
    if _CP_EMULATED is not None:
        world, rank = _CP_EMULATED
        o_sz = list(out_sizes) if out_sizes is not None else [out.shape[0] // world] * world
        i_sz = list(in_sizes) if in_sizes is not None else [inp.shape[0] // world] * world
        i_off = sum(i_sz[:rank])
        n0 = i_sz[rank]
        if all(v == n0 for v in o_sz) and out.shape[0] == world * n0 and out.is_contiguous():
            # equal pieces (the usual case): ONE broadcast copy writes the world x n0 rows a real all-to-all would deliver — the
            # same bytes into `out`, one launch like one collective, instead of eight slice copies (109 k copy launches per clip)
            out.view(world, n0, *out.shape[1:]).copy_(inp[i_off:i_off + n0].unsqueeze(0).expand(world, n0, *inp.shape[1:]))
            return FakeHandle()
        # unequal pieces (a chunk count that does not divide by the ranks: pieces differ by a row): peer r's piece := this rank's own
        # piece for itself, wrapped to its length — as ONE gather through a cached row index (was up to sixteen slice copies)
        key = (tuple(o_sz), n0, i_off, str(inp.device))
        idx = _EMU_IDX.get(key)
        if idx is None:
            rows = []
            for r in range(world):
                n = min(o_sz[r], n0)
                rows += list(range(i_off, i_off + n)) + list(range(i_off, i_off + o_sz[r] - n))
            idx = _EMU_IDX[key] = torch.tensor(rows, dtype=torch.long, device=inp.device)
        if out.is_contiguous() and out.shape[0] == idx.numel():
            torch.index_select(inp, 0, idx, out=out)
        else:
            out[:idx.numel()].copy_(inp.index_select(0, idx))
        return FakeHandle()
    group = get_cp_group()
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        host_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host_out, inp.cpu(), output_split_sizes=out_sizes, input_split_sizes=in_sizes, group=group)
        out.copy_(host_out)
        return FakeHandle()
    return dist.all_to_all_single(out, inp, output_split_sizes=out_sizes, input_split_sizes=in_sizes, group=group, async_op=True)


def divide(a: int, b: int) -> int:
    assert a % b == 0, f"{a} is not divisible by {b}"
    return a // b


class FakeHandle:
    def wait(self):
        pass


# ---------------------------------------------------------------------------------------------------------------
# split / scatter / gather                                   (context_parallel.py:30-88)
# ---------------------------------------------------------------------------------------------------------------
def scatter_to_context_parallel_region(input_: torch.Tensor, cp_split_sizes: Sequence[int], cp_shuffle_num: int = 1,
                                       cp_pad_size: int = 0) -> torch.Tensor:
    """This rank's rows of the first dimension.  Plain: `[sum(sizes[:rank]), +sizes[rank])` of the (zero-padded) sequence.
    Shuffled (`cp_shuffle_num` = dn > 1, cp_shuffle_overlap): the sequence is dn chunks; every chunk is zero-padded by pad/dn rows
    and every rank takes its window of EVERY chunk, chunk-major (context_parallel.py:30-54)."""
    if get_cp_world_size() == 1:
        return input_
    rank = get_cp_rank()
    if cp_shuffle_num > 1:
        dn = cp_shuffle_num
        m = divide(cp_split_sizes[rank], dn)                      # rows of one chunk on this rank
        lo = sum(divide(s, dn) for s in cp_split_sizes[:rank])
        c = divide(input_.shape[0], dn)                           # unpadded chunk length
        assert lo + m <= c + divide(cp_pad_size, dn)
        out = input_.new_zeros((dn, m) + tuple(input_.shape[1:]))
        hi = min(lo + m, c)
        if hi > lo:
            out[:, :hi - lo] = input_.reshape((dn, c) + tuple(input_.shape[1:]))[:, lo:hi]
        return out.reshape((dn * m,) + tuple(input_.shape[1:]))
    off = sum(cp_split_sizes[:rank])
    n = cp_split_sizes[rank]
    if cp_pad_size == 0:
        return input_[off:off + n].contiguous()
    out = input_.new_zeros((n,) + tuple(input_.shape[1:]))
    hi = min(off + n, input_.shape[0])
    if hi > off:
        out[:hi - off] = input_[off:hi]
    return out


def gather_from_context_parallel_region(input_: torch.Tensor, cp_split_sizes: Sequence[int], cp_shuffle_num: int = 1,
                                        cp_pad_size: int = 0) -> torch.Tensor:
    """Inverse of the scatter: one all-gather (shards may have different lengths), the shuffled form re-ordered chunk-major and
    every chunk's padding dropped (context_parallel.py:57-88)."""
    world = get_cp_world_size()
    if world == 1:
        return input_
    input_ = input_.contiguous()
    tail = tuple(input_.shape[1:])
    out = torch.empty((sum(cp_split_sizes),) + tail, dtype=input_.dtype, device=input_.device)
    if _CP_EMULATED is not None:                    # peers' shards := this rank's own rows (a device copy of the gathered bytes)
        for piece in torch.split(out, list(cp_split_sizes), dim=0):
            n = min(piece.shape[0], input_.shape[0])
            piece[:n].copy_(input_[:n])
            if n < piece.shape[0]:
                piece[n:].copy_(input_[:piece.shape[0] - n])
    elif input_.is_cuda and dist.get_backend(get_cp_group()) == "gloo":          # as _a2a: gloo moves host tensors
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather(list(torch.split(host, list(cp_split_sizes), dim=0)), input_.cpu(), group=get_cp_group())
        out.copy_(host)
    else:
        dist.all_gather(list(torch.split(out, list(cp_split_sizes), dim=0)), input_, group=get_cp_group())
    if cp_shuffle_num > 1:
        dn = cp_shuffle_num
        m = divide(cp_split_sizes[0], dn)
        assert all(s == cp_split_sizes[0] for s in cp_split_sizes), "shuffled shards are equal by construction"
        keep = world * m - divide(cp_pad_size, dn)
        out = out.reshape((world, dn, m) + tail).transpose(0, 1).reshape((dn, world * m) + tail)[:, :keep]
        return out.reshape((dn * keep,) + tail)
    return out[:out.shape[0] - cp_pad_size] if cp_pad_size else out


# ---------------------------------------------------------------------------------------------------------------
# cross-attention ranges under CP                             (context_parallel.py:135-216)
# ---------------------------------------------------------------------------------------------------------------
def cp_update_cross_attn_qkv_range(cross_attn_params: PackedCrossAttnParams, batch_size: int, cp_split_sizes: List[int],
                                   device, cp_shuffle_num: int = 1, cp_pad_size: int = 0) -> PackedCrossAttnParams:
    """Clip every packed query segment to this rank's token window(s), keep the key segment of the pieces that survive, and
    re-base the query ranges to the rank-local packed order.  One window per batch element, or — shuffled — one per (batch
    element, denoising chunk), each chunk padded by pad/dn rows, which shifts packed segment s by s * pad/dn.  Pure host-side
    integer work (context_parallel.py:135-226)."""
    rank = get_cp_rank()
    dn = max(int(cp_shuffle_num), 1)
    per = [divide(s, dn) for s in cp_split_sizes]
    total = sum(per)                                           # padded chunk (or sequence) length
    lo0, hi0 = sum(per[:rank]), sum(per[:rank + 1])
    shift = divide(cp_pad_size, dn)
    cq = [int(v) + i * shift for i, v in enumerate(cross_attn_params.cu_seqlens_q.tolist())]
    ck = cross_attn_params.cu_seqlens_kv.tolist()
    q_rows: List[List[int]] = []
    k_rows: List[List[int]] = []
    base = 0
    for b in range(batch_size):
        for j in range(dn):
            off = (b * dn + j) * total
            lo, hi = lo0 + off, hi0 + off
            piece_q, piece_k = [], []
            for sgm in range(len(cq) - 1):
                a, e = max(lo, cq[sgm]), min(hi, cq[sgm + 1])
                if a < e:
                    piece_q.append((a, e))
                    piece_k.append([ck[sgm], ck[sgm + 1]])
            first = min(a for a, _ in piece_q)
            rebased = [[a - first + base, e - first + base] for a, e in piece_q]
            base = rebased[-1][1]
            q_rows += rebased
            k_rows += piece_k
    q_ranges = torch.tensor(q_rows, dtype=torch.int32, device=device)
    kv_ranges = torch.tensor(k_rows, dtype=torch.int32, device=device)
    return PackedCrossAttnParams(q_ranges=q_ranges, kv_ranges=kv_ranges, cu_seqlens_q=torch.unique(q_ranges),
                                 cu_seqlens_kv=torch.unique(kv_ranges), max_seqlen_q=cp_split_sizes[rank],
                                 max_seqlen_kv=cross_attn_params.max_seqlen_kv)


# ---------------------------------------------------------------------------------------------------------------
# pre / post processing                                        (context_parallel.py:219-256, 309-376)
# ---------------------------------------------------------------------------------------------------------------
def cp_ulysses_process(cp_size: int, x: torch.Tensor, condition_map: torch.Tensor, rope: torch.Tensor,
                       xattn_mask_for_cuda_graph, cross_attn_params: PackedCrossAttnParams):
    seq_len, N, _ = x.shape
    assert seq_len == rope.size(0), f"seq_len: {seq_len} != rope.size(0): {rope.size(0)}"
    assert condition_map.size(0) == seq_len, f"condition_map.size(0): {condition_map.size(0)} != seq_len: {seq_len}"
    if xattn_mask_for_cuda_graph is not None:
        raise NotImplementedError("static-length cross-attention masks (CUDA-graph mode of the reference) are not built")
    cp_split_sizes = [seq_len // cp_size + (1 if r < seq_len % cp_size else 0) for r in range(cp_size)]
    x = scatter_to_context_parallel_region(x, cp_split_sizes)
    condition_map = scatter_to_context_parallel_region(condition_map, cp_split_sizes)
    rope = scatter_to_context_parallel_region(rope, cp_split_sizes)
    cross_attn_params = cp_update_cross_attn_qkv_range(cross_attn_params, N, cp_split_sizes, x.device)
    return x, condition_map, rope, cp_split_sizes, cross_attn_params


def cp_pre_process(cp_size: int, cp_strategy: str, x: torch.Tensor, condition_map: torch.Tensor, rope: torch.Tensor,
                   xattn_mask_for_cuda_graph, ardf_meta: Optional[dict], core_attn_params: Optional[PackedCoreAttnParams],
                   cross_attn_params: PackedCrossAttnParams):
    """-> (x, condition_map, rope, cp_pad_size, cp_split_sizes, core_attn_params, cross_attn_params), as upstream."""
    if cp_size == 1:
        return x, condition_map, rope, None, None, core_attn_params, cross_attn_params
    if cp_strategy == "cp_ulysses":
        x, condition_map, rope, sizes, cross_attn_params = cp_ulysses_process(cp_size, x, condition_map, rope,
                                                                              xattn_mask_for_cuda_graph, cross_attn_params)
        return x, condition_map, rope, 0, sizes, core_attn_params, cross_attn_params
    if cp_strategy == "cp_shuffle_overlap":
        return cp_shuffle_overlap_process(cp_size, x, condition_map, rope, xattn_mask_for_cuda_graph, ardf_meta, core_attn_params,
                                          cross_attn_params)
    raise ValueError(f"Invalid CP strategy: {cp_strategy}, expected cp_ulysses or cp_shuffle_overlap")


def cp_shuffle_overlap_process(cp_size: int, x: torch.Tensor, condition_map: torch.Tensor, rope: torch.Tensor,
                               xattn_mask_for_cuda_graph, ardf_meta: dict, core_attn_params: Optional[PackedCoreAttnParams],
                               cross_attn_params: PackedCrossAttnParams):
    """Context shuffle (context_parallel.py:258-307): every rank gets a slice of EVERY denoising chunk, so that all ranks have
    work in every attention range; chunks are padded to a multiple of cp and the query ranges stretched accordingly."""
    import math

    import numpy as np
    seq_len, N, _ = x.shape
    assert seq_len == rope.size(0), f"seq_len: {seq_len} != rope.size(0): {rope.size(0)}"
    assert condition_map.size(0) == seq_len, f"condition_map.size(0): {condition_map.size(0)} != seq_len: {seq_len}"
    if xattn_mask_for_cuda_graph is not None:
        raise NotImplementedError("static-length cross-attention masks (CUDA-graph mode of the reference) are not built")
    dn = int(ardf_meta["denoising_range_num"])
    chunk = divide(seq_len, dn)
    cp_pad_size = (cp_size - chunk % cp_size) * dn if chunk % cp_size else 0
    cp_split_sizes = [(seq_len + cp_pad_size) // cp_size] * cp_size
    x = scatter_to_context_parallel_region(x, cp_split_sizes, dn, cp_pad_size)
    condition_map = scatter_to_context_parallel_region(condition_map, cp_split_sizes, dn, cp_pad_size)
    rope = scatter_to_context_parallel_region(rope, cp_split_sizes, dn, cp_pad_size)
    g = math.gcd(seq_len, seq_len + cp_pad_size)
    num, den = (seq_len + cp_pad_size) // g, seq_len // g
    q_range = ardf_meta["q_range"] * num // den
    k_range = ardf_meta["k_range"]
    core_attn_params = PackedCoreAttnParams(q_range=q_range, k_range=k_range, np_q_range=np.asarray(q_range.cpu().numpy()),
                                            np_k_range=np.asarray(k_range.cpu().numpy()),
                                            max_seqlen_q=ardf_meta["max_seqlen_q"] * num // den,
                                            max_seqlen_k=ardf_meta["max_seqlen_k"])
    cross_attn_params = cp_update_cross_attn_qkv_range(cross_attn_params, N, cp_split_sizes, x.device, dn, cp_pad_size)
    return x, condition_map, rope, cp_pad_size, cp_split_sizes, core_attn_params, cross_attn_params


def cp_post_process(cp_size: int, cp_strategy: str, x: torch.Tensor, meta_args: ModelMetaArgs) -> torch.Tensor:
    if cp_size == 1:
        return x
    if cp_strategy == "cp_ulysses":
        return gather_from_context_parallel_region(x, meta_args.cp_split_sizes)
    if cp_strategy == "cp_shuffle_overlap":
        return gather_from_context_parallel_region(x, meta_args.cp_split_sizes, meta_args.denoising_range_num, meta_args.cp_pad_size)
    raise ValueError(f"Invalid CP strategy: {cp_strategy}, expected cp_ulysses or cp_shuffle_overlap")


# ---------------------------------------------------------------------------------------------------------------
# Ulysses all-to-all                                           (context_parallel.py:382-456)
# ---------------------------------------------------------------------------------------------------------------
def _heads_to_ranks(t: torch.Tensor, cp: int) -> torch.Tensor:
    """[seq, cp*hn, hd] -> [(cp seq), hn, hd]: destination-rank major, which is the send order of the collective.
    With fewer kv heads than ranks (cp % hn == 0) every head is first repeated cp/hn times (each rank gets a copy)."""
    hn = t.shape[1]
    if cp % hn == 0 and cp != hn:
        t = torch.repeat_interleave(t, cp // hn, dim=1)
    seq, heads, hd = t.shape
    return t.view(seq, cp, heads // cp, hd).permute(1, 0, 2, 3).reshape(cp * seq, heads // cp, hd).contiguous()


def all_to_all_input_split(tensor: torch.Tensor, cp_split_sizes: List[int]):
    """Scatter heads, gather sequence: (seq_r, cp*hn, hd) per rank -> (sum(seq), hn, hd).  Returns (tensor, handle)."""
    cp = get_cp_world_size()
    if cp == 1:
        return tensor, FakeHandle()
    assert cp_split_sizes is not None and tensor.is_contiguous()
    send = _heads_to_ranks(tensor, cp)
    out = torch.empty((sum(cp_split_sizes),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    handle = _a2a(out, send, out_sizes=list(cp_split_sizes))
    return out, handle


def all_to_all_output_split(tensor: torch.Tensor, cp_split_sizes: List[int]):
    """Scatter sequence, gather heads: (sum(seq), hn, hd) -> (cp * seq_r, hn, hd), source-rank major."""
    cp = get_cp_world_size()
    if cp == 1:
        return tensor, FakeHandle()
    assert cp_split_sizes is not None and tensor.is_contiguous()
    out = torch.empty((cp_split_sizes[get_cp_rank()] * cp,) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
    handle = _a2a(out, tensor, in_sizes=list(cp_split_sizes))
    return out, handle


def fused_qkv_communication(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cp_split_sizes: List[int]):
    """q, k and v in ONE all-to-all message (launch-bound first chunks)."""
    cp = get_cp_world_size()
    if cp == 1:
        return q, k, v
    assert cp_split_sizes is not None
    qs, ks, vs = _heads_to_ranks(q, cp), _heads_to_ranks(k, cp), _heads_to_ranks(v, cp)
    heads = [qs.shape[1], ks.shape[1], vs.shape[1]]
    send = torch.cat([qs, ks, vs], dim=1).contiguous()
    out = torch.empty((sum(cp_split_sizes),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    _a2a(out, send, out_sizes=list(cp_split_sizes)).wait()
    return torch.split(out, heads, dim=1)


# ---------------------------------------------------------------------------------------------------------------
# scheduler                                                    (context_parallel.py:459-598)
# ---------------------------------------------------------------------------------------------------------------
class UlyssesScheduler:
    """Order of projections, collectives and attention calls of one MAGI attention layer under cp_ulysses.
    `kv_cache_func(kv) -> (key, value)`; `core_attn_func(q, key, value) -> [S, hq', hd]` (the HIP range attention,
    inferix_amd/magi/attention.py); key/value may be any handle the two callables agree on (tensors upstream; a paged
    cache view here, so the prefix is never concatenated)."""

    @staticmethod
    def get_attn_and_xattn_with_comm_overlap(get_q_func: Callable, get_k_func: Callable, get_v_func: Callable,
                                             kv_cache_func: Callable, core_attn_func: Callable, cross_attn_func: Callable,
                                             overlap_degree: int, batch_size: int, cp_size: int,
                                             cp_split_sizes: Optional[List[int]] = None):
        """v, k, q projected in that order, each all-to-all launched as soon as its operand exists."""
        value, hv = all_to_all_input_split(get_v_func(), cp_split_sizes)
        key, hk = all_to_all_input_split(get_k_func(), cp_split_sizes)
        query, hq = all_to_all_input_split(get_q_func(), cp_split_sizes)
        hv.wait()
        hk.wait()
        key, value = kv_cache_func(torch.cat([key, value], dim=-1))
        hq.wait()
        return UlyssesScheduler.get_attn_and_xattn_base(query, key, value, core_attn_func, cross_attn_func, overlap_degree,
                                                        batch_size, cp_size, cp_split_sizes)

    @staticmethod
    def get_attn_and_xattn_with_fused_kv_comm(get_q_func: Callable, get_kv_func: Callable, kv_cache_func: Callable,
                                              core_attn_func: Callable, cross_attn_func: Callable, overlap_degree: int,
                                              batch_size: int, cp_size: int, cp_split_sizes: Optional[List[int]] = None):
        kv, hkv = all_to_all_input_split(get_kv_func(), cp_split_sizes)
        query, hq = all_to_all_input_split(get_q_func(), cp_split_sizes)
        hkv.wait()
        key, value = kv_cache_func(kv)
        hq.wait()
        return UlyssesScheduler.get_attn_and_xattn_base(query, key, value, core_attn_func, cross_attn_func, overlap_degree,
                                                        batch_size, cp_size, cp_split_sizes)

    @staticmethod
    def get_attn_and_xattn_with_fused_qkv_comm(get_qkv_func: Callable, kv_cache_func: Callable, core_attn_func: Callable,
                                               cross_attn_func: Callable, overlap_degree: int, batch_size: int,
                                               cp_size: int, cp_split_sizes: Optional[List[int]] = None):
        q, k, v = get_qkv_func()
        q, k, v = fused_qkv_communication(q, k, v, cp_split_sizes)
        k, v = kv_cache_func(torch.cat([k, v], dim=-1))
        return UlyssesScheduler.get_attn_and_xattn_base(q.contiguous(), k, v, core_attn_func, cross_attn_func,
                                                        overlap_degree, batch_size, cp_size, cp_split_sizes)

    @staticmethod
    def split_query_for_overlap(query: torch.Tensor, kv_head: int, overlap_degree: int) -> List[torch.Tensor]:
        q_seq, q_head, hd = query.shape
        if overlap_degree == -1:
            overlap_degree = q_head // kv_head
        else:
            assert overlap_degree <= q_head
        if overlap_degree == 1:
            return [query]
        if kv_head == 1:                                                     # MQA per rank (MAGI at cp = 8)
            return [c.contiguous() for c in query.chunk(overlap_degree, dim=1)]
        assert q_head % (overlap_degree * kv_head) == 0
        parts = query.reshape(q_seq, kv_head, -1, hd).chunk(overlap_degree, dim=2)
        return [p.reshape(q_seq, -1, hd).contiguous() for p in parts]

    @staticmethod
    def get_attn_and_xattn_base(query: torch.Tensor, key, value, core_attn_func: Callable, cross_attn_func: Callable,
                                overlap_degree: int, batch_size: int, cp_size: int,
                                cp_split_sizes: Optional[List[int]] = None, kv_head: Optional[int] = None):
        """Attention per query-head chunk; the output all-to-all of chunk i is in flight during the attention of chunk
        i+1 (RCCL runs on its own stream), the cross-attention covers the last one."""
        if kv_head is None:
            kv_head = key.shape[1] if isinstance(key, torch.Tensor) else key.kv_heads
        chunks = UlyssesScheduler.split_query_for_overlap(query, kv_head, overlap_degree)
        handle = None
        pending = None
        outs = []
        for qc in chunks:
            new = core_attn_func(qc, key, value)
            if handle is not None:
                handle.wait()
                outs.append(pending)
            pending, handle = all_to_all_output_split(new.contiguous(), cp_split_sizes)
        xattn_out = cross_attn_func()
        handle.wait()
        outs.append(pending)
        core = torch.cat(outs, dim=1)                                       # [(cp sq b), hn, hd]
        n, hn, hd = core.shape
        sq = n // (cp_size * batch_size)
        core = core.view(cp_size, sq, batch_size, hn, hd).permute(1, 2, 0, 3, 4).reshape(sq, batch_size, cp_size * hn * hd)
        return core.contiguous(), xattn_out


# ---------------------------------------------------------------------------------------------------------------
# context shuffle overlap: attention pipeline                  (context_parallel.py:604-665)
# ---------------------------------------------------------------------------------------------------------------
def cso_communication(input: torch.Tensor, cp_world_size: int, cp_split_sizes: List[int], comm_type: Optional[str] = None):
    """One all-to-all of equal pieces, rows grouped by destination rank.  `comm_type` "kv": `[rows, cp*hn, hd]` (heads repeated
    first when there are fewer kv heads than ranks) is re-ordered rank-major so that every rank receives all rows of its heads."""
    if cp_world_size == 1:
        return input, FakeHandle()
    assert cp_split_sizes is not None
    if comm_type == "kv":
        input = _heads_to_ranks(input, cp_world_size)
    input = input.contiguous()
    output = torch.empty_like(input)
    handle = _a2a(output, input, in_sizes=list(cp_split_sizes))
    return output, handle


class CSOHelper:
    """Order of the query / output messages of one attention layer under cp_shuffle_overlap: the queries of denoising chunk i+1
    travel (together with the finished output of chunk i-1) while chunk i is attended to; outputs come back in chunk order."""

    def __init__(self, cp_shuffle_num: int, cp_world_size: int, cp_split_sizes: List[int]):
        self.cp_shuffle_num = cp_shuffle_num
        self.cp_world_size = cp_world_size
        self.cp_split_sizes = [divide(x, cp_shuffle_num) for x in cp_split_sizes]

    def split_query_for_overlap(self, query: torch.Tensor):
        """`[(dn rows), (cp hn), hd]` -> dn messages `[(cp rows), hn, hd]` (destination-rank major); the first is sent at once."""
        dn, cp = self.cp_shuffle_num, self.cp_world_size
        rows, heads, hd = query.shape
        m = divide(rows, dn)
        q = query.view(dn, m, cp, divide(heads, cp), hd).permute(0, 2, 1, 3, 4).reshape(dn * cp * m, heads // cp, hd).contiguous()
        querys = list(torch.chunk(q, dn, dim=0))
        querys[0], handle_q = cso_communication(querys[0], cp, self.cp_split_sizes)
        return querys, handle_q

    def overlap(self, fattn: Callable, qs: List[torch.Tensor], k, v):
        """`fattn(q, k, v, i)` = attention of chunk i's queries (all rows of this rank's heads).  Returns the per-chunk outputs
        back in (source-rank, rows) order on their home ranks + the handle of the last message."""
        dn, cp = self.cp_shuffle_num, self.cp_world_size
        outs: List[torch.Tensor] = []
        handle_attn = FakeHandle()
        loop_var, loop_handle, o = None, None, None
        for i in range(dn):
            if dn == 1:
                q = qs[0]
            elif i == 0:
                q = qs[0]
                loop_var, loop_handle = cso_communication(qs[1], cp, self.cp_split_sizes)
            else:
                loop_handle.wait()
                if loop_var.numel() == qs[0].numel():
                    q = loop_var
                else:                                            # (next queries | finished output) travelled side by side
                    assert loop_var.numel() == qs[0].numel() * 2
                    q, ready_o = torch.chunk(loop_var, 2, dim=-1)
                    outs.append(ready_o)
                loop_var = torch.concat([qs[i + 1], o], dim=-1) if i < dn - 1 else o
                loop_var, loop_handle = cso_communication(loop_var, cp, self.cp_split_sizes)
            o = fattn(q, k, v, i)
            if i == dn - 1:
                if i != 0:
                    loop_handle.wait()
                    assert loop_var.numel() == qs[0].numel()
                    outs.append(loop_var)
                last_o, handle_attn = cso_communication(o, cp, self.cp_split_sizes)
                outs.append(last_o)
        return outs, handle_attn
