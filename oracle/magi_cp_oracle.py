"""TEST INFRASTRUCTURE — not product code.  CPU restatement of the reference's MAGI context-parallel path (§8 row a17).

Single-process simulation: a "collective" takes the list of every rank's tensor and returns the list of every rank's
result, so the data movement of torch.distributed is restated as explicit index arithmetic.  Each function cites the
reference lines it follows (paths relative to /root/reference/inferix/).  Pinned bit-exactly against the reference's
own functions run under gloo with 4 ranks (oracle/gen_golden_magi.py -> tests/golden/magi_cp.npz); only the attention
arithmetic itself is third-party there (flash-attn / magi_attention, absent) and is restated as exact softmax attention.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor


# ---------------------------------------------------------------------------------------------------------------
# split / scatter / gather            distributed/parallelism/context_parallel.py:30-88, 240-243
# ---------------------------------------------------------------------------------------------------------------
def cp_split_sizes(seq_len: int, cp: int) -> List[int]:
    """context_parallel.py:240-243: floor share, the first seq_len % cp ranks take one more token."""
    sizes = [seq_len // cp] * cp
    for i in range(seq_len % cp):
        sizes[i] += 1
    return sizes


def scatter(x: Tensor, sizes: Sequence[int], rank: int) -> Tensor:
    """scatter_to_context_parallel_region (cp_shuffle_num == 1 branch, :52-55): rows [offset, offset + sizes[rank])."""
    off = sum(sizes[:rank])
    return x[off:off + sizes[rank]].contiguous()


def gather(parts: Sequence[Tensor]) -> Tensor:
    """gather_from_context_parallel_region (:59-88, cp_shuffle_num == 1, no padding): rank-order concatenation."""
    return torch.cat(list(parts), dim=0)


# ---------------------------------------------------------------------------------------------------------------
# Ulysses all-to-all                      context_parallel.py:382-450
# ---------------------------------------------------------------------------------------------------------------
def _replicate_kv_heads(t: Tensor, cp: int) -> Tensor:
    """:397-398 / :441-443: with fewer kv heads than ranks (cp % hn == 0, cp != hn) every head is repeated cp/hn times."""
    hn = t.shape[1]
    if cp % hn == 0 and cp != hn:
        return torch.repeat_interleave(t, cp // hn, dim=1)
    return t


def a2a_input_split(per_rank: Sequence[Tensor], sizes: Sequence[int]) -> List[Tensor]:
    """all_to_all_input_split (:382-405).  Rank r holds [sizes[r], cp*hn, hd]; afterwards rank c holds the WHOLE sequence
    (rank-order concatenation of the shards) for head group c: [sum(sizes), hn, hd]."""
    cp = len(per_rank)
    per_rank = [_replicate_kv_heads(t, cp) for t in per_rank]
    hn = per_rank[0].shape[1] // cp
    out = []
    for c in range(cp):
        out.append(torch.cat([per_rank[r][:, c * hn:(c + 1) * hn] for r in range(cp)], dim=0).contiguous())
    return out


def a2a_output_split(per_rank: Sequence[Tensor], sizes: Sequence[int]) -> List[Tensor]:
    """all_to_all_output_split (:408-429).  Rank c holds [sum(sizes), hn, hd]; afterwards rank r holds, source-rank major,
    its own token range from every head group: [cp * sizes[r], hn, hd] (rows c*sizes[r] + j)."""
    cp = len(per_rank)
    offs = [sum(sizes[:r]) for r in range(cp)]
    return [torch.cat([per_rank[c][offs[r]:offs[r] + sizes[r]] for c in range(cp)], dim=0).contiguous()
            for r in range(cp)]


def fused_qkv_communication(q: Sequence[Tensor], k: Sequence[Tensor], v: Sequence[Tensor], sizes: Sequence[int]):
    """fused_qkv_communication (:432-456): the three input all-to-alls in one message; same result as three calls."""
    return a2a_input_split(q, sizes), a2a_input_split(k, sizes), a2a_input_split(v, sizes)


# ---------------------------------------------------------------------------------------------------------------
# range attention                          models/magi/dit/dit_module.py:975-1018 (flash_attn_func branch)
# ---------------------------------------------------------------------------------------------------------------
def exact_attention(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """softmax(q k^T / sqrt(d)) v in fp64, grouped-query: q [sq, hq, d], k/v [sk, hk, d], hq % hk == 0, query head h
    reads kv head h // (hq/hk) (flash_attn_func's GQA convention).  Returns fp64 [sq, hq, d]."""
    sq, hq, d = q.shape
    hk = k.shape[1]
    g = hq // hk
    kk = torch.repeat_interleave(k.double(), g, dim=1)
    vv = torch.repeat_interleave(v.double(), g, dim=1)
    s = torch.einsum("qhd,khd->hqk", q.double(), kk) / math.sqrt(d)
    p = torch.softmax(s, dim=-1)
    return torch.einsum("hqk,khd->qhd", p, vv)


def core_attention(query: Tensor, key: Tensor, value: Tensor, q_range, k_range, out_dtype=torch.bfloat16) -> Tensor:
    """core_attention with bs == 1 (:995-1016): for every denoising range i, queries [q_range[i,0], q_range[i,1])
    attend to keys [k_range[i,0], k_range[i,1]) — no mask inside a range; outputs concatenated in range order."""
    outs = []
    for (qs, qe), (ks, ke) in zip(q_range, k_range):
        outs.append(exact_attention(query[qs:qe], key[ks:ke], value[ks:ke]).to(out_dtype))
    return torch.cat(outs, dim=0)


# ---------------------------------------------------------------------------------------------------------------
# UlyssesScheduler                          context_parallel.py:462-598
# ---------------------------------------------------------------------------------------------------------------
def split_query_for_overlap(query: Tensor, kv_head: int, overlap_degree: int) -> List[Tensor]:
    """get_attn_and_xattn_base (:557-572): the query heads are processed in `overlap_degree` chunks so that the output
    all-to-all of chunk i overlaps the attention of chunk i+1."""
    q_seq, q_head, hd = query.shape
    if overlap_degree == -1:
        overlap_degree = q_head // kv_head
    if overlap_degree == 1:
        return [query]
    if kv_head == 1:
        return list(query.chunk(overlap_degree, dim=1))
    parts = query.reshape(q_seq, kv_head, -1, hd).chunk(overlap_degree, dim=2)
    return [p.reshape(q_seq, -1, hd) for p in parts]


def ulysses_attention(q_shards, kv_shards, sizes, batch_size, overlap_degree, attn_fn, kv_cache_fn=None):
    """get_attn_and_xattn_with_fused_kv_comm (:512-537) + base (:557-598) for every rank at once.
      q_shards[r]  [(sq_r b), cp*hq, hd]   kv_shards[r] [(sq_r b), cp*hk (or fewer), 2*hd]   sizes = batch_cp_split_sizes
      attn_fn(rank, q [S, hq', hd], k, v) -> [S, hq', hd];  kv_cache_fn(rank, kv [S, hk, 2hd]) -> (k, v)
    Returns per rank core_attn_out [sq_r, b, cp*hq*hd]   ('(cp sq b) hn hd -> sq b (cp hn hd)')."""
    cp = len(q_shards)
    kv = a2a_input_split(kv_shards, sizes)
    q = a2a_input_split(q_shards, sizes)
    outs_per_rank: List[List[Tensor]] = [[] for _ in range(cp)]
    n_chunks = None
    for c in range(cp):
        if kv_cache_fn is not None:
            k, v = kv_cache_fn(c, kv[c])
        else:
            k, v = torch.chunk(kv[c], 2, dim=-1)
        chunks = split_query_for_overlap(q[c], k.shape[1], overlap_degree)
        n_chunks = len(chunks)
        outs_per_rank[c] = [attn_fn(c, qc, k.contiguous(), v.contiguous()) for qc in chunks]
    result = [[] for _ in range(cp)]
    for i in range(n_chunks):
        back = a2a_output_split([outs_per_rank[c][i] for c in range(cp)], sizes)
        for r in range(cp):
            result[r].append(back[r])
    final = []
    for r in range(cp):
        o = torch.cat(result[r], dim=1)                         # [(cp sq_r b), hq_local, hd]
        sq = sizes[r] // batch_size
        o = o.reshape(cp, sq, batch_size, o.shape[1], o.shape[2]).permute(1, 2, 0, 3, 4)
        final.append(o.reshape(sq, batch_size, -1).contiguous())
    return final


# ---------------------------------------------------------------------------------------------------------------
# cross-attention ranges under CP           context_parallel.py:135-216 (cp_shuffle_num == 1, cp_pad_size == 0)
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class CrossRanges:
    q_ranges: Tensor
    kv_ranges: Tensor
    cu_seqlens_q: Tensor
    cu_seqlens_kv: Tensor
    max_seqlen_q: int
    max_seqlen_kv: int


def cp_update_cross_attn_qkv_range(cu_seqlens_q: Tensor, cu_seqlens_kv: Tensor, max_seqlen_kv: int, batch_size: int,
                                   sizes: Sequence[int], rank: int) -> CrossRanges:
    """Intersect every packed q segment [cu_q[s], cu_q[s+1]) with this rank's token window of each batch element;
    surviving pieces keep their kv segment and are re-based to the rank-local packed order."""
    total = sum(sizes)
    lo_base, hi_base = sum(sizes[:rank]), sum(sizes[:rank + 1])
    cq = cu_seqlens_q.tolist()
    ck = cu_seqlens_kv.tolist()
    q_all, k_all = [], []
    base_off = 0
    for i in range(batch_size):
        lo, hi = lo_base + i * total, hi_base + i * total
        qs, ks = [], []
        for s in range(len(cq) - 1):
            a, b = max(lo, cq[s]), min(hi, cq[s + 1])
            if a < b:
                qs.append([a, b])
                ks.append([ck[s], ck[s + 1]])
        m = min(a for a, _ in qs)
        qs = [[a - m + base_off, b - m + base_off] for a, b in qs]
        base_off = qs[-1][1]
        q_all += qs
        k_all += ks
    q_r = torch.tensor(q_all, dtype=torch.int32)
    k_r = torch.tensor(k_all, dtype=torch.int32)
    return CrossRanges(q_r, k_r, torch.unique(q_r), torch.unique(k_r), sizes[rank], max_seqlen_kv)


# ---------------------------------------------------------------------------------------------------------------
# KV cache prefix + append                   kvcache_manager/model/magi_kv_cache_manager.py:76-187
# ---------------------------------------------------------------------------------------------------------------
class MagiCacheOracle:
    """One layer's cache `(2, max_tokens, 1, hn, hd)` and the reference's prefix/append rule."""

    def __init__(self, max_tokens: int, hn: int, hd: int, max_batch_size: int = 1, dtype=torch.bfloat16):
        self.mem: Optional[Tensor] = None
        self.shape = (2, max_tokens, 1, hn, hd)
        self.b = max_batch_size
        self.dtype = dtype

    def adjust(self, key_and_value: Tensor, *, slice_point: int, clip_token_nums: int, update_kv_cache: bool,
               extract_prefix_video_feature=False, fwd_extra_1st_chunk=False, distill_nearly_clean_chunk=False):
        """adjust_key_and_value_for_inference (:150-187) -> (key, value) `[prefix + new, hn, hd]`.
        `_full_adjust_key_and_value` (:76-148): prefix = cache[0 : slice_point*clip_token_nums*B]; when update_kv_cache,
        the new rows (minus the last chunk under distill_nearly_clean_chunk) are stored at the end of the prefix."""
        hd = key_and_value.shape[-1] // 2
        if not (extract_prefix_video_feature or fwd_extra_1st_chunk or slice_point > 0):
            k, v = torch.chunk(key_and_value, 2, dim=-1)
            return k.contiguous(), v.contiguous()
        new = torch.stack([key_and_value[..., :hd], key_and_value[..., hd:]], dim=0)       # [2, n, hn, hd]
        if self.mem is None:
            self.mem = torch.zeros(self.shape, dtype=self.dtype)
        start = slice_point * clip_token_nums * self.b
        prefix = self.mem[:, :start, 0]
        if update_kv_cache:
            clip = new.shape[1] - clip_token_nums * self.b if distill_nearly_clean_chunk else new.shape[1]
            assert start + clip <= self.mem.shape[1]
            self.mem[:, start:start + clip, 0] = new[:, :clip]
        full = torch.cat([prefix, new], dim=1)
        return full[0].contiguous(), full[1].contiguous()
