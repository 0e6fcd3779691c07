"""MAGI context parallelism (Ulysses): sequence split, head <-> sequence all-to-all, attention scheduling.

Mirror of the reference's `inferix/distributed/parallelism/context_parallel.py` for the `cp_ulysses` strategy — the same
function names, argument meaning and results — on torch.distributed, whose "nccl" backend is RCCL on ROCm.  The
all-to-all is the natural collective of the fully connected xGMI mesh (every pair of GPUs has its own link, so all
P-1 peer messages of a rank move concurrently); `cp_shuffle_overlap`, the reference's strategy for PCIe-attached
consumer GPUs (context_parallel.py:258-307), is not built and raises.

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


def _a2a(out: torch.Tensor, inp: torch.Tensor, out_sizes=None, in_sizes=None):
    """`dist.all_to_all_single` with two extras: under emulation a local copy of the same bytes; over gloo (CPU-only
    all-to-all: the 1-GPU test box runs 2 ranks on one device) device tensors are staged through the host."""
    if _CP_EMULATED is not None:
        world, rank = _CP_EMULATED
        o_sz = list(out_sizes) if out_sizes is not None else [out.shape[0] // world] * world
        i_sz = list(in_sizes) if in_sizes is not None else [inp.shape[0] // world] * world
        i_off = sum(i_sz[:rank])
        off = 0
        for r in range(world):                      # peer r's piece := this rank's own piece for itself (same size up to +-1 row)
            n = min(o_sz[r], i_sz[rank])
            out[off:off + n].copy_(inp[i_off:i_off + n])
            if n < o_sz[r]:
                out[off + n:off + o_sz[r]].copy_(inp[i_off:i_off + o_sz[r] - n])
            off += o_sz[r]
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
    """This rank's rows `[sum(sizes[:rank]), +sizes[rank])` of the first dimension."""
    if get_cp_world_size() == 1:
        return input_
    if cp_shuffle_num != 1 or cp_pad_size != 0:
        raise NotImplementedError("cp_shuffle_overlap (shuffled / padded split) is not built; use cp_ulysses")
    rank = get_cp_rank()
    off = sum(cp_split_sizes[:rank])
    return input_[off:off + cp_split_sizes[rank]].contiguous()


def gather_from_context_parallel_region(input_: torch.Tensor, cp_split_sizes: Sequence[int], cp_shuffle_num: int = 1,
                                        cp_pad_size: int = 0) -> torch.Tensor:
    """Rank-order concatenation of every rank's rows (one all-gather; shards may have different lengths)."""
    world = get_cp_world_size()
    if world == 1:
        return input_
    if cp_shuffle_num != 1 or cp_pad_size != 0:
        raise NotImplementedError("cp_shuffle_overlap (shuffled / padded gather) is not built; use cp_ulysses")
    input_ = input_.contiguous()
    out = torch.empty((sum(cp_split_sizes),) + tuple(input_.shape[1:]), dtype=input_.dtype, device=input_.device)
    dist.all_gather(list(torch.split(out, list(cp_split_sizes), dim=0)), input_, group=get_cp_group())
    return out


# ---------------------------------------------------------------------------------------------------------------
# cross-attention ranges under CP                             (context_parallel.py:135-216)
# ---------------------------------------------------------------------------------------------------------------
def cp_update_cross_attn_qkv_range(cross_attn_params: PackedCrossAttnParams, batch_size: int, cp_split_sizes: List[int],
                                   device, cp_shuffle_num: int = 1, cp_pad_size: int = 0) -> PackedCrossAttnParams:
    """Clip every packed query segment to this rank's token window (per batch element), keep the key segment of the
    pieces that survive, and re-base the query ranges to the rank-local packed order.  Pure host-side integer work."""
    if cp_shuffle_num != 1 or cp_pad_size != 0:
        raise NotImplementedError("cp_shuffle_overlap ranges are not built; use cp_ulysses")
    rank = get_cp_rank()
    total = sum(cp_split_sizes)
    lo0, hi0 = sum(cp_split_sizes[:rank]), sum(cp_split_sizes[:rank + 1])
    cq = cross_attn_params.cu_seqlens_q.tolist()
    ck = cross_attn_params.cu_seqlens_kv.tolist()
    q_rows: List[List[int]] = []
    k_rows: List[List[int]] = []
    base = 0
    for b in range(batch_size):
        lo, hi = lo0 + b * total, hi0 + b * total
        piece_q, piece_k = [], []
        for s in range(len(cq) - 1):
            a, e = max(lo, cq[s]), min(hi, cq[s + 1])
            if a < e:
                piece_q.append((a, e))
                piece_k.append([ck[s], ck[s + 1]])
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
        raise NotImplementedError("cp_shuffle_overlap targets PCIe consumer GPUs upstream; MI355X uses cp_ulysses over xGMI")
    raise ValueError(f"Invalid CP strategy: {cp_strategy}, expected cp_ulysses or cp_shuffle_overlap")


def cp_post_process(cp_size: int, cp_strategy: str, x: torch.Tensor, meta_args: ModelMetaArgs) -> torch.Tensor:
    if cp_size == 1:
        return x
    if cp_strategy == "cp_ulysses":
        return gather_from_context_parallel_region(x, meta_args.cp_split_sizes)
    if cp_strategy == "cp_shuffle_overlap":
        raise NotImplementedError("cp_shuffle_overlap is not built; use cp_ulysses")
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
