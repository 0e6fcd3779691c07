"""Sequence (context) parallelism of the denoising step over the GPUs of one node.

Sharding follows the reference (causal_model.py:939-942): rank r owns, for EVERY frame, the hw-slice
`[r*fs/P, (r+1)*fs/P)` of the frame's `fs` tokens, so all ranks have the same work at every block index and
every non-attention op is token-local.  The exchange is re-designed for xGMI:

  reference  : KV cache sharded (tokens/ring x heads/ulysses); per layer Ulysses all-to-all x4 + ring p2p of
               the WHOLE cached prefix (attention/distributed.py:183-208,610-706); 12 heads do not divide 8.
  this module: KV cache REPLICATED per GPU (6 GB of 288 GB) in the single-GPU token order; per layer ONE
               RCCL all-gather of the NEW block's post-RoPE K and V (2*N/P*dim bf16 per rank, direct on the
               fully connected xGMI mesh), scattered into cache slots with the (frame, rank, hw/P) interleave;
               attention is purely local: N/P queries over the full prefix.  Overlap: the all-gather runs on a
               side HIP stream while the main stream attends to the OLD prefix [0, local_start) — split-KV +
               LSE merge (ifx_lse_merge) — so only the new-block part waits for the collective.
  Page-table / slot indices therefore stay bit-identical to the 1-GPU run (same kv_index_update inputs).

`SequenceParallelExchange` holds only torch.distributed + index logic and runs on CPU tensors with gloo
(tests/test_sequence_parallel.py, world_size 2); `attach_sequence_parallel` wires it into HipCausalWanModel.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


class SequenceParallelExchange:
    """Collective + index arithmetic of the K/V exchange (device-agnostic)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._slot_cache: Dict[Tuple, torch.Tensor] = {}
        self._flat_ok = True

    def token_slots(self, local_start: int, frames: int, fs: int, device, page_table: Optional[torch.Tensor] = None,
                    page_size: int = 1) -> torch.Tensor:
        """Physical cache slot of gathered row (rank r, local token j = f*hw_local + i):
        logical token local_start + f*fs + r*hw_local + i  (the single-GPU (frame, hw) order), then through
        the page table if there is one.  Returned flattened in all-gather order [P * frames * hw_local]."""
        key = (local_start, frames, fs, str(device), None if page_table is None else page_table.data_ptr(), page_size)
        if page_table is None and key in self._slot_cache:
            return self._slot_cache[key]
        P = self.world
        if fs % P != 0:
            raise ValueError(f"a frame's {fs} tokens do not divide over {P} sequence-parallel ranks")
        hw_local = fs // P
        r = torch.arange(P).view(P, 1, 1)
        f = torch.arange(frames).view(1, frames, 1)
        i = torch.arange(hw_local).view(1, 1, hw_local)
        logical = (local_start + f * fs + r * hw_local + i).reshape(-1)
        if page_table is not None:
            pt = page_table.cpu().long()
            logical = pt[logical // page_size] * page_size + logical % page_size
        slots = logical.to(device)
        if page_table is None:
            self._slot_cache[key] = slots
        return slots

    def all_gather_rows(self, local: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[n, c] per rank -> [P*n, c] in rank order (one collective)."""
        if out is None:
            out = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                              device=local.device)
        local = local.contiguous()
        if self._flat_ok:
            try:
                dist.all_gather_into_tensor(out, local, group=self.group)
                return out
            except (RuntimeError, NotImplementedError):      # backend without a flat all-gather (some gloo builds)
                self._flat_ok = False
        dist.all_gather(list(out.view(self.world, *local.shape).unbind(0)), local, group=self.group)
        return out

    def exchange_new_block(self, kv_local: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                           local_start: int, frames: int, fs: int, page_table=None, page_size: int = 1,
                           gathered: Optional[torch.Tensor] = None) -> None:
        """kv_local `[N/P, 2, heads, head_dim]` (this rank's post-RoPE K and raw V of the new block) ->
        every rank's cache holds all N new tokens at slots [local_start, local_start + N) in (frame, hw) order."""
        g = self.all_gather_rows(kv_local, gathered)                   # [P*N/P, 2, H, D]
        slots = self.token_slots(local_start, frames, fs, k_cache.device, page_table, page_size)
        k_cache.index_copy_(0, slots, g[:, 0])
        v_cache.index_copy_(0, slots, g[:, 1])

    def gather_head(self, y_local: torch.Tensor, batch: int, frames: int) -> torch.Tensor:
        """Head output `[B*F*hw_local, c]` per rank -> `[B*F*fs, c]` in (b, f, rank, hw_local) order
        (all_gather + 'b (cp f hw) c -> b (f cp hw) c', causal_model.py:1008-1022)."""
        P = self.world
        n_local, c = y_local.shape
        hw_local = n_local // (batch * frames)
        g = self.all_gather_rows(y_local)                               # [P, B, F, hw_local, c]
        g = g.view(P, batch, frames, hw_local, c).permute(1, 2, 0, 3, 4)
        return g.reshape(batch * frames * P * hw_local, c)


class LoopbackExchange(SequenceParallelExchange):
    """One rank of a `world`-way shard without a process group: the gather replicates the local rows `world`
    times (a device copy instead of the collective).  Used by `bench.py --emulate-sp P` to time one rank's compute
    on a single GPU; results are not a valid clip."""

    def __init__(self, world: int, rank: int = 0):
        self.group = None
        self.world = world
        self.rank = rank
        self._slot_cache = {}
        self._flat_ok = True

    def all_gather_rows(self, local, out=None):
        if out is None:
            out = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                              device=local.device)
        out.view(self.world, *local.shape).copy_(local.unsqueeze(0).expand(self.world, *local.shape))
        return out


class HipSequenceParallel:
    """GPU side: what HipCausalWanModel calls per layer when world_size > 1."""

    def __init__(self, group=None, overlap: bool = True, exchange: Optional[SequenceParallelExchange] = None):
        self.ex = exchange if exchange is not None else SequenceParallelExchange(group)
        self.overlap = overlap
        self.comm_stream: Optional[torch.cuda.Stream] = None
        self._buf: Dict[Tuple, torch.Tensor] = {}

    def _scratch(self, name, shape, dtype, device):
        key = (name, tuple(shape), dtype)
        b = self._buf.get(key)
        if b is None:
            b = torch.empty(*shape, dtype=dtype, device=device)
            self._buf[key] = b
        return b

    def self_attention(self, model, l, b, view, qkv_b, q_out, a_out, w, rope, current_start, g_end, l_end,
                       sink_tokens, mgr, req, name):
        from . import hip_ops as ops
        from .wan.causal_model import kv_index_update
        P = self.ex.world
        n_local = qkv_b.shape[0]
        N = n_local * P
        H, hd, d = model.num_heads, model.head_dim, model.dim
        dev = qkv_b.device
        fs = rope.height * rope.width
        frames = n_local // rope.hw_local
        step = kv_index_update(g_end, l_end, current_start, N, view.k.shape[0], model.local_attn_size, sink_tokens)
        if step.evicted:
            model._evict(mgr, req, name, view, step)
            view = model._kv_view(mgr, req, name)
        # this rank's K / V of the new block -> staging (laid out as a 1-page cache), q -> q_out
        stage = self._scratch("kv_stage", (2, n_local, H, hd), torch.bfloat16, dev)
        ops.rmsnorm_rope_kv_append(qkv_b, w["nq"], w["nk"], model.eps, rope, ops.KvCacheView(stage[0], stage[1]), 0,
                                   d, q_out=q_out)
        gathered = self._scratch("kv_gather", (P, 2, n_local, H, hd), torch.bfloat16, dev)
        qv = q_out.view(n_local, H, hd)
        av = a_out.view(n_local, H, hd)
        have_prefix = step.local_start > 0

        def exchange():
            # one collective: [2, n_local, H, D] per rank -> [P, 2, n_local, H, D]; one kernel scatters K and V rows
            # to their cache slots (through the page table when there is one)
            self.ex.all_gather_rows(stage.view(2 * n_local, H * hd), gathered.view(P * 2 * n_local, H * hd))
            ops.kv_scatter_shards(gathered, P, frames, rope.hw_local, fs, step.local_start, view)

        if self.overlap and have_prefix and dev.type == "cuda":
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
            self.comm_stream.wait_stream(main)
            with torch.cuda.stream(self.comm_stream):
                exchange()
            # main stream: attend to the old prefix while the collective is in flight; both launches write fp32 partials
            # into one workspace and a single merge rounds to bf16 once
            s1 = ops.attention_split_plan(n_local, H, step.local_start)
            s2 = ops.attention_split_plan(n_local, H, step.local_end - step.local_start)
            cap = s1 + s2
            ws = ops.attention_workspace(qv, cap)
            u1 = ops.attention_partial(qv, view, step.local_start, 0, s1, ws, 0, cap, tag="attn_self")
            main.wait_stream(self.comm_stream)
            u2 = ops.attention_partial(qv, view, step.local_end, step.local_start, s2, ws, u1, cap, tag="attn_self")
            ops.attention_merge(ws, cap, u1 + u2, av)
        else:
            exchange()
            ops.attention(qv, view, step.local_end, out=av, tag="attn_self")
        return step

    def gather_head(self, y_local: torch.Tensor, batch: int, frames: int) -> torch.Tensor:
        return self.ex.gather_head(y_local, batch, frames)


def attach_sequence_parallel(model, group=None, overlap: bool = True,
                             exchange: Optional[SequenceParallelExchange] = None) -> HipSequenceParallel:
    """Enable sequence parallelism on a HipCausalWanModel whose ParallelConfig has world_size > 1."""
    sp = HipSequenceParallel(group, overlap, exchange)
    pc = model.parallel_config
    if pc.world_size != sp.ex.world or pc.rank != sp.ex.rank:
        raise ValueError(f"ParallelConfig (rank {pc.rank}/{pc.world_size}) does not match the process group "
                         f"(rank {sp.ex.rank}/{sp.ex.world})")
    model.cp = sp
    return sp
