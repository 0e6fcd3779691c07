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


class PeerStoreExchange:
    """Control plane of the exchange WITHOUT a collective: every rank stores its K/V rows of the new block straight into the cache
    slots of every peer's replicated cache (`ifx_rmsnorm_rope_kv_push`), ordered by two per-layer flags in fine-grained peer memory
    (`ifx_peer_signal` / `ifx_peer_wait`; protocol in csrc/ifx_peer.hip).  HIP IPC handles of the caches and flag blocks travel once
    per cache tensor through `all_gather_object` on `group` (any backend); nothing else uses the process group per layer.
    `emulate_world` > 0: one rank of that many, no peers — the rows are stored to `world` destinations that are all this rank's own
    cache (the launch and the outbound bytes of a real rank; nothing arrives), for `bench.py --emulate-sp`."""
    MAX_LAYERS = 64
    READY, DONE = 0, 1

    def __init__(self, group=None, timeout_ms: int = 20000, emulate_world: int = 0, emulate_rank: int = 0):
        from . import hip_ops as ops
        from . import _hip
        self.group = group
        self.emulated = emulate_world > 0
        self.world = emulate_world if self.emulated else dist.get_world_size(group)
        self.rank = emulate_rank if self.emulated else dist.get_rank(group)
        if self.world > _hip.IFX_MAX_PEERS:
            raise ValueError(f"peer-store exchange is built for up to {_hip.IFX_MAX_PEERS} ranks, got {self.world}")
        self.timeout_ms = timeout_ms
        self._opened: Dict[bytes, int] = {}                 # IPC handle -> base address mapped into this process
        self._views: Dict[Tuple, Tuple[List[int], List[int]]] = {}
        self._view_handles: Dict[Tuple, set] = {}           # allocation id -> the peer IPC handles its addresses lie in
        self._handle_refs: Dict[bytes, int] = {}            # IPC handle -> number of remembered views (+ the flag blocks) using it
        self._epoch = [0] * self.MAX_LAYERS
        self.status = torch.zeros(1, dtype=torch.int32, device="cuda")
        try:
            self.flags = ops.PeerBuffer(self.MAX_LAYERS * 2 * _hip.IFX_MAX_PEERS * 4, fine_grained=True)
        except Exception:                                   # noqa: BLE001 — reported through the collective below, on every rank
            self.flags = None
        if self.emulated:
            if self.flags is None:
                raise RuntimeError("peer-store exchange: could not allocate the flag block")
            self.peer_flags = [self.flags.ptr]
        else:
            pinned: set = set()
            self.peer_flags = self._exchange_addresses(self.flags.ptr if self.flags is not None else None, pinned)
            for h in pinned:                                # the flag blocks stay mapped for the life of the exchange
                self._handle_refs[h] = self._handle_refs.get(h, 0) + 1

    # ---- addresses ----
    def _exchange_addresses(self, ptr: Optional[int], used: Optional[set] = None) -> List[int]:
        """This rank's device address `ptr` -> the same buffer of every rank, mapped into this process (rank order).  Every rank
        runs the same two collectives whatever fails locally, and all ranks raise together: a rank must never sit in a collective
        its peers have left (the caller falls back to the all-gather exchange).  `used` collects the peer handles touched."""
        from . import hip_ops as ops
        mine = None
        if ptr is not None:
            try:
                mine = ops.peer_export(ptr)
            except Exception as exc:                        # noqa: BLE001
                mine = None
                self._last_error = str(exc)
        everyone: List = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self.group)
        out: List[int] = []
        ok = all(e is not None for e in everyone)
        if ok:
            try:
                for p, (handle, offset) in enumerate(everyone):
                    if p == self.rank:
                        out.append(ptr)
                        continue
                    base = self._opened.get(handle)
                    if base is None:
                        base = ops.peer_open(handle)
                        self._opened[handle] = base
                    if used is not None:
                        used.add(handle)
                    out.append(base + offset)
            except Exception as exc:                        # noqa: BLE001
                ok = False
                self._last_error = str(exc)
        votes: List = [None] * self.world
        dist.all_gather_object(votes, bool(ok), group=self.group)
        if not all(votes):
            bad = [r for r, v in enumerate(votes) if not v]
            raise RuntimeError(f"peer-store exchange: IPC export / mapping failed on rank(s) {bad}"
                               + (f" ({self._last_error})" if getattr(self, "_last_error", None) else ""))
        return out

    @staticmethod
    def view_geometry(view) -> Tuple:
        """What `push` assumes is IDENTICAL on every rank, because it addresses the peers' slots through this rank's own view:
        slot count, heads, head size, page size and the page table's contents (crc of the host copy)."""
        import zlib
        pt = getattr(view, "page_table", None)
        crc = 0 if pt is None else zlib.crc32(pt.detach().cpu().numpy().tobytes())
        return (int(view.k.shape[0]), int(view.k.shape[1]), int(view.k.shape[2]), int(getattr(view, "page_size", 1)), crc)

    def cache_addresses(self, view, ident: Optional[Tuple] = None) -> Tuple[List[int], List[int]]:
        """(K base, V base) of this (request, layer) cache on every rank; exchanged the first time an ALLOCATION is seen (a
        collective: every rank reaches it at the same call).  `ident` = `KVCacheManager.allocation_id(req, layer)` — (request, layer,
        (manager serial, generation)), on which the ranks of an SPMD run agree.  Keying by data pointers (round 2) was wrong twice over: a freed cache's
        address can come back on one rank and not on its peers (stale mapping -> stores into freed peer memory), and ranks that
        disagree on hit / miss deadlock in the exchange.  Without an `ident` (the self test's scratch cache) nothing is remembered.
        Raises — on every rank together — if the ranks' cache geometries or page tables differ."""
        got = self._views.get(ident) if ident is not None else None
        if got is None:
            kp, vp = view.k.data_ptr(), view.v.data_ptr()
            if self.emulated:
                got = ([kp], [vp])
            else:
                geo: List = [None] * self.world
                dist.all_gather_object(geo, self.view_geometry(view), group=self.group)
                if any(g != geo[0] for g in geo):
                    raise RuntimeError(f"peer-store exchange: ranks disagree on the cache geometry / page table of {ident}: {geo}")
                used: set = set()
                got = (self._exchange_addresses(kp, used), self._exchange_addresses(vp, used))
            if ident is not None:
                for old in [k for k in self._views if k[:2] == ident[:2]]:      # earlier allocations of this (request, layer)
                    self._drop(old)
                self._views[ident] = got
                if not self.emulated:
                    self._view_handles[ident] = used
                    for h in used:
                        self._handle_refs[h] = self._handle_refs.get(h, 0) + 1
        return got

    def _drop(self, key: Tuple) -> None:
        del self._views[key]
        for h in self._view_handles.pop(key, ()):
            self._handle_refs[h] = self._handle_refs.get(h, 1) - 1

    def forget(self, request_id: str, layer_name: Optional[str] = None) -> None:
        """Drop the address book of a freed (request, layer) — `KVCacheManager.free_layer` — or of a whole request (`free`,
        `layer_name` None); harmless if there is none.  Only THAT layer's entry goes (dropping the request's whole book on a
        single layer's release forced an all_gather_object + IPC re-exchange for every other layer in the middle of the next forward).
        On a whole-request release the peer mappings no remembered view uses any more are closed (after a device synchronise: a
        kernel in flight may still store through them); the caching allocator's segments they name can then be returned by the peers."""
        for k in [k for k in self._views if k[0] == request_id and (layer_name is None or k[1] == layer_name)]:
            self._drop(k)
        if layer_name is None and not self.emulated:
            dead = [h for h, n in self._handle_refs.items() if n <= 0 and h in self._opened]
            if dead:
                from . import hip_ops as ops
                torch.cuda.synchronize()
                for h in dead:
                    ops.peer_close(self._opened.pop(h))
                    del self._handle_refs[h]

    # ---- per layer ----
    def _index(self, layer: int, kind: int) -> int:
        from . import _hip
        return (layer * 2 + kind) * _hip.IFX_MAX_PEERS

    def push(self, layer: int, kv_rows: torch.Tensor, wk: torch.Tensor, eps: float, rope, view, local_start: int, frame_tokens: int,
             dim: int, ident: Optional[Tuple] = None) -> int:
        """On the current stream: ready -> wait for every peer's ready -> store the rows everywhere -> done.  Returns the epoch."""
        from . import hip_ops as ops
        if layer >= self.MAX_LAYERS:
            raise ValueError(f"layer {layer}: the flag block holds {self.MAX_LAYERS} layers")
        ks, vs = self.cache_addresses(view, ident)
        self._epoch[layer] += 1
        e = self._epoch[layer]
        hw_local = rope.hw_local
        if self.emulated:
            # the launch a real rank issues: ONE kernel storing its rows to `world` destinations (all of them this rank's own cache
            # here, so the peers' slots stay unwritten — timing only, bench.py marks the line INVALID)
            ops.rmsnorm_rope_kv_push(kv_rows, wk, eps, rope, ks * self.world, vs * self.world, view, local_start, frame_tokens,
                                     hw_local, self.rank * hw_local, dim)
            return e
        ops.peer_signal(self.peer_flags, self._index(layer, self.READY) + self.rank, e)
        ops.peer_wait(self.flags.ptr + 4 * self._index(layer, self.READY), self.world, e, self.timeout_ms, self.status)
        ops.rmsnorm_rope_kv_push(kv_rows, wk, eps, rope, ks, vs, view, local_start, frame_tokens, hw_local, self.rank * hw_local, dim)
        ops.peer_signal(self.peer_flags, self._index(layer, self.DONE) + self.rank, e)
        return e

    def wait_done(self, layer: int, epoch: int) -> None:
        """The current stream waits until every rank's rows of `epoch` are in this rank's cache."""
        from . import hip_ops as ops
        if not self.emulated:
            ops.peer_wait(self.flags.ptr + 4 * self._index(layer, self.DONE), self.world, epoch, self.timeout_ms, self.status)

    def self_test(self, rounds: int = 3) -> bool:
        """Ready / push / done rounds through a small scratch cache, with the CONSUMER in the position the attention kernel has in a
        layer: a device copy enqueued right behind `wait_done` on the same stream, no host synchronisation in between (a stale
        cache line of the previous round would show).  Checked on every rank and agreed on over the process group: True only if
        every rank saw every rank's rows in every round.  Collective; a False means "use the all-gather"."""
        from . import hip_ops as ops
        if self.emulated:
            return True
        ok = True
        try:
            H, hd, rows = 2, 128, 4
            d = H * hd
            dev = self.status.device
            kc = torch.zeros(self.world * rows, H, hd, dtype=torch.bfloat16, device=dev)
            vc = torch.zeros_like(kc)
            view = ops.KvCacheView(kc, vc)
            addr_ok = True
            try:
                ks, vs = self.cache_addresses(view)
            except RuntimeError:                               # raised on every rank together (see _exchange_addresses)
                addr_ok = False
            if not addr_ok:
                return False
            wk = torch.ones(d, dtype=torch.bfloat16, device=dev)
            layer = self.MAX_LAYERS - 1
            snaps = []
            for rnd in range(rounds):
                kv = torch.full((rows, 2 * d), float(self.rank + 1 + 8 * rnd), dtype=torch.bfloat16, device=dev)
                self._epoch[layer] += 1
                e = self._epoch[layer]
                ops.peer_signal(self.peer_flags, self._index(layer, self.READY) + self.rank, e)
                ops.peer_wait(self.flags.ptr + 4 * self._index(layer, self.READY), self.world, e, self.timeout_ms, self.status)
                ops.rmsnorm_rope_kv_push(kv, wk, 1e-6, None, ks, vs, view, 0, self.world * rows, rows, self.rank * rows, d)
                ops.peer_signal(self.peer_flags, self._index(layer, self.DONE) + self.rank, e)
                self.wait_done(layer, e)
                snaps.append(vc.clone())                       # the consumer kernel, stream-ordered behind the wait
            torch.cuda.synchronize()
            ok = int(self.status.item()) == 0
            for rnd, snap in enumerate(snaps):
                want = (torch.arange(1, self.world + 1, dtype=torch.float32, device=dev) + 8 * rnd).repeat_interleave(rows)
                ok = ok and torch.equal(snap[:, 0, 0].float(), want) and torch.equal(snap[:, H - 1, hd - 1].float(), want)
        except Exception:                                      # noqa: BLE001 — any failure here means "do not use this path"
            ok = False
        votes: List = [None] * self.world
        dist.all_gather_object(votes, bool(ok), group=self.group)
        self.status.zero_()
        return all(votes)

    def check(self) -> None:
        """Raise — on EVERY rank together — if a wait gave up on any rank (one small MAX all-reduce of the status words; synchronises:
        call at a point that synchronises anyway).  A rank that timed out alone and raised alone would leave its peers in the next
        collective."""
        words = [int(self.status.item())]
        if not self.emulated and dist.is_initialized():
            # every rank's own status word (1 + index of the peer it gave up on, 0 = fine): a MAX all-reduce would name the highest
            # awaited index seen ANYWHERE, not who waited for whom
            everyone = [torch.zeros_like(self.status) for _ in range(self.world)]
            dist.all_gather(everyone, self.status, group=self.group)
            words = [int(t.item()) for t in everyone]
        if any(words):
            self.status.zero_()
            pairs = ", ".join(f"rank {r} gave up waiting for rank {w - 1}" for r, w in enumerate(words) if w)
            raise RuntimeError(f"peer-store exchange: a wait timed out after {self.timeout_ms} ms ({pairs})")

    def close(self) -> None:
        from . import hip_ops as ops
        torch.cuda.synchronize()
        for base in self._opened.values():
            ops.peer_close(base)
        self._opened.clear()
        self._views.clear()
        self._view_handles.clear()
        self._handle_refs.clear()
        if self.flags is not None:
            self.flags.free()


def _lab_emulated_check() -> bool:
    import os
    return os.environ.get("IFX_SP_LAB_CHECK") == "1"


def _comm_priority() -> int:
    """Priority of the exchange streams (lab switch IFX_SP_COMM_PRIORITY: 0 = default, -1 = high).  The exchange launches are short and
    everything behind them waits for the peers' rows; with a second launch chain on the chip their workgroups otherwise queue behind the
    other chain's long kernels."""
    import os
    return int(os.environ.get("IFX_SP_COMM_PRIORITY", "0"))


def _lab_single_attention() -> bool:
    import os
    return os.environ.get("IFX_SP_SINGLE_ATTN", "") == "1"


def _split_div():
    import os
    v = os.environ.get("IFX_SP_SPLIT_DIV", "")
    if not v:
        return (1, 1)
    a, _, b = v.partition(",")
    return (max(1, int(a)), max(1, int(b or a)))


class HipSequenceParallel:
    """GPU side: what HipCausalWanModel calls per layer when world_size > 1.

    `begin` takes the K/V-only projection of the layer's rows and starts the exchange on a side stream; the caller then computes the
    q projection (which overlaps the exchange) and calls `finish`.  Two exchanges: one RCCL all-gather + `ifx_kv_scatter_shards`
    (default), or direct peer stores (`peer` = a PeerStoreExchange)."""

    def __init__(self, group=None, overlap: bool = True, exchange: Optional[SequenceParallelExchange] = None,
                 peer: Optional[PeerStoreExchange] = None, kv_first: Optional[bool] = None):
        self.ex = exchange if exchange is not None else SequenceParallelExchange(group)
        self.peer = peer
        # K/V projection first = the exchange starts one GEMM earlier, at the price of two projection launches and a separate
        # q norm/RoPE kernel (12 us per layer at 585 rows).  Worth it in front of a collective (~100 us of launch + transfer), not in
        # front of the peer stores (two flag round trips + 3.6 MB per link), which the prefix attention covers from the second block on.
        self.kv_first = (peer is None) if kv_first is None else kv_first
        self.overlap = overlap
        # the peer-store waits report a timeout through a status word; reading it is a host synchronisation (+ one small collective so
        # that every rank raises together).  Once per forward would drain the launch queue 35 times per clip: the word is sticky, every
        # rank runs the same collectives whether or not a wait gave up, so the check runs every `check_every` forwards (default: a
        # block's five) and once more when the pipeline finishes a clip (`check_now`).
        import os
        self.check_every = max(1, int(os.environ.get("IFX_SP_CHECK_EVERY", "5")))
        if peer is not None and not peer.emulated and dist.is_available() and dist.is_initialized() and self.ex.world > 1:
            # `peer.check()` is a collective: ranks that disagree on the cadence would issue it at different forwards and deadlock
            # (ADVICE r5).  The environment is per process, so the value is validated over the group once, here.
            seen = [None] * self.ex.world
            dist.all_gather_object(seen, self.check_every, group=self.ex.group)
            if len(set(seen)) != 1:
                raise ValueError(f"IFX_SP_CHECK_EVERY differs between ranks {seen}: the peer-store status check is a collective")
        self._forwards_since_check = 0
        self.comm_stream: Optional[torch.cuda.Stream] = None      # chain 0's side stream (kept under this name for the tests / tools)
        self._comm_streams: Dict[int, torch.cuda.Stream] = {}     # chain -> side stream (HipCausalWanModel.forward_pair runs two chains)
        self._buf: Dict[Tuple, torch.Tensor] = {}
        if peer is not None and (peer.world != self.ex.world or peer.rank != self.ex.rank):
            raise ValueError("peer-store exchange and collective exchange disagree on (rank, world)")

    def _scratch(self, name, shape, dtype, device):
        key = (name, tuple(shape), dtype)
        b = self._buf.get(key)
        if b is None:
            b = torch.empty(*shape, dtype=dtype, device=device)
            self._buf[key] = b
        return b

    def begin(self, model, l, view, kv_rows, w, rope, current_start, g_end, l_end, sink_tokens, mgr, req, name) -> dict:
        """kv_rows `[N/P, >= 2*dim]` = (k | v) of this rank's rows (any row stride).  Starts the exchange; returns the state `finish` needs."""
        from . import hip_ops as ops
        from .wan.causal_model import kv_index_update
        P = self.ex.world
        n_local = kv_rows.shape[0]
        N = n_local * P
        H, hd, d = model.num_heads, model.head_dim, model.dim
        dev = kv_rows.device
        fs = rope.height * rope.width
        frames = n_local // rope.hw_local
        step = kv_index_update(g_end, l_end, current_start, N, view.k.shape[0], model.local_attn_size, sink_tokens)
        if step.evicted:
            model._evict(mgr, req, name, view, step)
            view = model._kv_view(mgr, req, name)
        have_prefix = step.local_start > 0
        ident = mgr.allocation_id(req, name) if hasattr(mgr, "allocation_id") else None
        if self.peer is not None:
            if hasattr(mgr, "add_free_listener"):           # the manager's free / free_layer drop this request's address book
                mgr.add_free_listener(self.peer.forget)
            try:                                            # first use of an allocation: handles travel (collective, main stream)
                self.peer.cache_addresses(view, ident)
            except RuntimeError as exc:                     # raised on every rank together: all fall back to the collective
                import warnings
                warnings.warn(f"peer-store exchange disabled, using the all-gather: {exc}")
                self.peer = None
        side = self.overlap and dev.type == "cuda"
        chain = int(getattr(model, "_chain", 0))
        st = dict(step=step, view=view, n_local=n_local, side=side, have_prefix=have_prefix, layer=l, epoch=0, comm=None)
        main = torch.cuda.current_stream(dev) if side else None
        if side:
            comm = self._comm_streams.get(chain)
            if comm is None:
                comm = self._comm_streams[chain] = torch.cuda.Stream(device=dev, priority=_comm_priority())
                if chain == 0:
                    self.comm_stream = comm
            comm.wait_stream(main)
            st["comm"] = comm

        def exchange():
            if self.peer is not None:
                st["epoch"] = self.peer.push(l, kv_rows, w["nk"], model.eps, rope, view, step.local_start, fs, d, ident)
                return
            # this rank's K / V -> staging (a dense 1-shard cache), one collective [2, n_local, H, D] -> [P, 2, n_local, H, D], one
            # kernel scatters K and V rows to their cache slots (through the page table when there is one)
            stage = self._scratch(("kv_stage", chain), (2, n_local, H, hd), torch.bfloat16, dev)
            gathered = self._scratch(("kv_gather", chain), (P, 2, n_local, H, hd), torch.bfloat16, dev)
            sview = ops.KvCacheView(stage[0], stage[1])
            ops.rmsnorm_rope_kv_push(kv_rows, w["nk"], model.eps, rope, [stage[0].data_ptr()], [stage[1].data_ptr()], sview, 0,
                                     rope.hw_local, rope.hw_local, 0, d)
            self.ex.all_gather_rows(stage.view(2 * n_local, H * hd), gathered.view(P * 2 * n_local, H * hd))
            ops.kv_scatter_shards(gathered, P, frames, rope.hw_local, fs, step.local_start, view)

        if side:
            with torch.cuda.stream(st["comm"]):
                exchange()
        else:
            exchange()
        return st

    def finish(self, model, st: dict, q_out, a_out) -> "KVIndexStep":
        """q_out `[N/P, dim]` post-norm/RoPE queries -> a_out: attention over the old prefix while the exchange is in flight, then over
        the new block; both launches write fp32 partials into one workspace and a single merge rounds to bf16 once."""
        from . import hip_ops as ops
        step, view, n_local = st["step"], st["view"], st["n_local"]
        H, hd = model.num_heads, model.head_dim
        qv = q_out.view(n_local, H, hd)
        av = a_out.view(n_local, H, hd)
        dev = q_out.device
        scale = getattr(model, "_attn_scale", 0.0)    # ln 2 when q carries scale * log2(e) (hip_ops.attn_q_prescale), else the default

        def arrived():
            # the wait for the peers' rows goes on the EXCHANGE stream, behind this rank's own push: the main stream then has one
            # cross-stream dependency in front of the attention over the new block instead of that dependency + a wait launch of its own
            # (one dependent launch less per layer on the critical path of a real rank; the emulated rank has no wait kernels at all)
            if st["side"]:
                if self.peer is not None:
                    with torch.cuda.stream(st["comm"]):
                        self.peer.wait_done(st["layer"], st["epoch"])
                torch.cuda.current_stream(dev).wait_stream(st["comm"])
            elif self.peer is not None:
                self.peer.wait_done(st["layer"], st["epoch"])

        if st["side"] and st["have_prefix"] and _lab_single_attention():
            # LAB ONLY (IFX_SP_SINGLE_ATTN=1; wrong ordering on real peers): ONE split launch over the prefix and the new block without
            # waiting for the exchange — what a launch that waits for the peers' rows in-kernel could save at best
            s = ops.attention_split_plan(n_local, H, step.local_end)
            ws = ops.attention_workspace(qv, s)
            u = ops.attention_partial(qv, view, step.local_end, 0, s, ws, 0, s, scale=scale, tag="attn_self")
            ops.attention_merge(ws, s, u, av)
            arrived()
        elif st["side"] and st["have_prefix"]:
            s1 = ops.attention_split_plan(n_local, H, step.local_start)
            s2 = ops.attention_split_plan(n_local, H, step.local_end - step.local_start)
            div = _split_div()
            if div != (1, 1):                          # lab (IFX_SP_SPLIT_DIV="a,b"): fewer key chunks = fewer fp32 partials for the merge
                s1, s2 = max(1, -(-s1 // div[0])), max(1, -(-s2 // div[1]))
            cap = s1 + s2
            ws = ops.attention_workspace(qv, cap)
            u1 = ops.attention_partial(qv, view, step.local_start, 0, s1, ws, 0, cap, scale=scale, tag="attn_self")
            arrived()
            u2 = ops.attention_partial(qv, view, step.local_end, step.local_start, s2, ws, u1, cap, scale=scale, tag="attn_self")
            ops.attention_merge(ws, cap, u1 + u2, av)
        else:
            arrived()
            ops.attention(qv, view, step.local_end, scale=scale, out=av, tag="attn_self")
        return step

    def self_attention(self, model, l, b, view, qkv_b, q_out, a_out, w, rope, current_start, g_end, l_end,
                       sink_tokens, mgr, req, name):
        """Fused-projection form (8-bit linears): qkv_b `[N/P, 3*dim]` already computed."""
        from . import hip_ops as ops
        d = model.dim
        st = self.begin(model, l, view, qkv_b[:, d:], w, rope, current_start, g_end, l_end, sink_tokens, mgr, req, name)
        ops.rmsnorm_rope_kv_append(qkv_b[:, :d], w["nq"], None, model.eps, rope, None, 0, d, q_out=q_out)
        return self.finish(model, st, q_out, a_out)

    def gather_head(self, y_local: torch.Tensor, batch: int, frames: int) -> torch.Tensor:
        out = self.ex.gather_head(y_local, batch, frames)
        self._forwards_since_check += 1
        if self._forwards_since_check >= self.check_every:
            self.check_now()
        return out

    def check_now(self) -> None:
        """Raise on every rank together if a peer-store wait gave up since the last check (collective + host synchronisation; a no-op
        without a peer-store exchange).  The pipelines call it at the end of a clip."""
        self._forwards_since_check = 0
        if self.peer is None:
            return
        if not self.peer.emulated:
            self.peer.check()
        elif _lab_emulated_check():
            int(self.peer.status.item())       # LAB (IFX_SP_LAB_CHECK=1): the host synchronisation of a real rank's check, without the collective

    def preflight(self, model, height: int, width: int, frames: int = 3, reps: int = 3) -> dict:
        """First contact with the interconnect, before anything is timed (collective: every rank calls it).  ONE layer's exchange of a
        synthetic block — rank-specific K/V rows of `frames` frames of `height x width` tokens — goes through the collective path
        (stage, all-gather, scatter) into one scratch cache and, when a peer-store exchange is attached, through the peer stores
        into a second one, `reps` times each, with a consumer (a device copy) enqueued right behind on the same stream.  Checked:
        every slot of a cache was written (K rows are RMS-normalised: never all-zero), the two caches of a rank are bit-identical,
        and the caches of all ranks are bit-identical (crc32 over the group).  FAILS CLOSED: a peer-store mismatch drops that path on
        every rank (the all-gather is used); an inconsistent all-gather raises on every rank.  Returns what the bench line reports:
        ranks seen, per-exchange microseconds of each path, the verdicts."""
        import zlib
        from . import hip_ops as ops
        P, rank = self.ex.world, self.ex.rank
        H, hd, d = model.num_heads, model.head_dim, model.dim
        fs = height * width
        if fs % P:
            raise ValueError(f"a frame's {fs} tokens do not divide over {P} ranks")
        hw_local, dev = fs // P, torch.device("cuda", torch.cuda.current_device())
        n_local, N = frames * hw_local, frames * fs
        g = torch.Generator(device=dev).manual_seed(4321 + rank)
        kv_rows = torch.randn(n_local, 2 * d, generator=g, device=dev).to(torch.bfloat16)
        wk = torch.ones(d, dtype=torch.bfloat16, device=dev)
        rope = ops.RopeGridSpec(model.freqs, 0, height, width, rank * hw_local, hw_local)
        report = {"ranks": P, "frames": frames, "rows_per_rank": n_local}

        def gathered_equal(value) -> bool:
            seen = [None] * P
            if self.ex.group is not None or dist.is_initialized():
                dist.all_gather_object(seen, value, group=self.ex.group)
            else:
                seen = [value] * P
            return all(v == seen[0] for v in seen)

        def run(path: str):
            kc = torch.zeros(N, H, hd, dtype=torch.bfloat16, device=dev)
            vc = torch.zeros_like(kc)
            view = ops.KvCacheView(kc, vc)
            us, snap = [], None
            for r in range(reps):
                kc.zero_(), vc.zero_()
                if dist.is_initialized():
                    dist.barrier(group=self.ex.group)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if path == "peer":
                    ep = self.peer.push(PeerStoreExchange.MAX_LAYERS - 2, kv_rows, wk, model.eps, rope, view, 0, fs, d,
                                        ("__preflight__", "kv", 0))
                    self.peer.wait_done(PeerStoreExchange.MAX_LAYERS - 2, ep)
                else:
                    stage = torch.empty(2, n_local, H, hd, dtype=torch.bfloat16, device=dev)
                    gathered = torch.empty(P, 2, n_local, H, hd, dtype=torch.bfloat16, device=dev)
                    sview = ops.KvCacheView(stage[0], stage[1])
                    ops.rmsnorm_rope_kv_push(kv_rows, wk, model.eps, rope, [stage[0].data_ptr()], [stage[1].data_ptr()], sview, 0,
                                             hw_local, hw_local, 0, d)
                    self.ex.all_gather_rows(stage.view(2 * n_local, H * hd), gathered.view(P * 2 * n_local, H * hd))
                    ops.kv_scatter_shards(gathered, P, frames, hw_local, fs, 0, view)
                snap = (kc.clone(), vc.clone())          # the consumer, stream-ordered behind the exchange
                e1.record()
                torch.cuda.synchronize()
                us.append(e0.elapsed_time(e1) * 1e3)
            k, v = snap
            filled = bool((k.view(N, -1).float().abs().amax(1) > 0).all().item())
            crc = zlib.crc32(k.cpu().view(torch.int16).numpy().tobytes()) ^ zlib.crc32(v.cpu().view(torch.int16).numpy().tobytes())
            return snap, sorted(us)[len(us) // 2], filled, crc

        (ag_k, ag_v), ag_us, ag_filled, ag_crc = run("allgather")
        report["allgather_us"] = round(ag_us, 1)
        report["allgather_ok"] = gathered_equal((ag_filled, ag_crc)) and ag_filled
        if not gathered_equal(report["allgather_ok"]) or not report["allgather_ok"]:
            raise RuntimeError(f"sequence-parallel preflight: the all-gather exchange left different caches on different ranks "
                               f"(rank {rank}: filled={ag_filled}, crc={ag_crc:#x})")
        if self.peer is not None and not self.peer.emulated:
            ok = True
            try:
                (pk, pv), p_us, p_filled, p_crc = run("peer")
                self.peer.check()
                ok = p_filled and torch.equal(pk, ag_k) and torch.equal(pv, ag_v)
                report["peer_store_us"] = round(p_us, 1)
            except RuntimeError as exc:                  # a timed-out wait raises on every rank together (PeerStoreExchange.check)
                ok = False
                report["peer_store_error"] = str(exc)[:200]
            ok = gathered_equal(bool(ok)) and ok
            report["peer_store_ok"] = bool(ok)
            if not ok:                                   # every rank sees the same votes: all of them drop the path
                self.peer.forget("__preflight__")
                self.peer, self.kv_first = None, True
            else:
                self.peer.forget("__preflight__")
        return report


def attach_sequence_parallel(model, group=None, overlap: bool = True,
                             exchange: Optional[SequenceParallelExchange] = None, peer: Optional[PeerStoreExchange] = None,
                             kv_first: Optional[bool] = None) -> HipSequenceParallel:
    """Enable sequence parallelism on a HipCausalWanModel whose ParallelConfig has world_size > 1."""
    sp = HipSequenceParallel(group, overlap, exchange, peer, kv_first)
    # a rank's launches have 4680 / P rows: the GEMM tile choice may split K inside a workgroup for them (row-count dependent bits,
    # which the default choice avoids; the row count of a rank is fixed by P).  The option is scoped to THIS model's forwards
    # (HipCausalWanModel.forward: hip_ops.option_scope, which reads the library's own value back on entry: ifx_get_option): every other
    # GEMM enqueued by this thread OUTSIDE those forwards — a second, unsharded model, the umT5 encoder, the VAE — keeps its
    # row-count-invariant summation order, and so does whatever ANOTHER host thread enqueues meanwhile: the library keeps the option per
    # host thread (ifx_core.hip).
    sp.gemm_small_split = True
    pc = model.parallel_config
    if pc.world_size != sp.ex.world or pc.rank != sp.ex.rank:
        raise ValueError(f"ParallelConfig (rank {pc.rank}/{pc.world_size}) does not match the process group "
                         f"(rank {sp.ex.rank}/{sp.ex.world})")
    model.cp = sp
    return sp
