"""Sequence-parallel exchange, world_size 2 / 4 on CPU (gloo): after `exchange_new_block` every rank's
replicated cache equals the single-device cache (same slots, same bytes); `gather_head` reproduces the
reference's '(cp f hw) -> (f cp hw)' interleave (golden: tests/golden/layout.npz)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import wan_oracle as O
from fixture_io import golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, paged, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from inferix_amd.sequence_parallel import SequenceParallelExchange
        ex = SequenceParallelExchange()
        frames, fs, H, D = 3, 24, 2, 8
        hw_local = fs // world
        g = torch.Generator().manual_seed(0)
        cap = 6 * fs
        full_k = torch.randn(cap, H, D, generator=g)          # what a single device would hold
        full_v = torch.randn(cap, H, D, generator=g)
        local_start = 2 * fs + (fs if paged else 0)
        n = frames * fs
        # single-device new block, token order (frame, hw)
        new_k = torch.randn(n, H, D, generator=g)
        new_v = torch.randn(n, H, D, generator=g)
        ref_k, ref_v = full_k.clone(), full_v.clone()
        ref_k[local_start:local_start + n] = new_k
        ref_v[local_start:local_start + n] = new_v
        # this rank's shard: hw-slice `rank` of every frame (causal_model.py:939-942)
        mine_k = new_k.view(frames, fs, H, D)[:, rank * hw_local:(rank + 1) * hw_local].reshape(-1, H, D)
        mine_v = new_v.view(frames, fs, H, D)[:, rank * hw_local:(rank + 1) * hw_local].reshape(-1, H, D)
        kv_local = torch.stack([mine_k, mine_v], dim=1)
        kc, vc = full_k.clone(), full_v.clone()
        pt = None
        if paged:
            ps = fs
            perm = torch.tensor([3, 0, 5, 1, 4, 2], dtype=torch.int32)
            t = torch.arange(cap)
            slot = perm[t // ps].long() * ps + t % ps
            kc, vc = torch.zeros_like(full_k), torch.zeros_like(full_v)
            kc[slot], vc[slot] = full_k, full_v
            pt = perm
            ex.exchange_new_block(kv_local, kc, vc, local_start, frames, fs, pt, ps)
            ok = torch.equal(kc[slot], ref_k) and torch.equal(vc[slot], ref_v)
        else:
            ex.exchange_new_block(kv_local, kc, vc, local_start, frames, fs)
            ok = torch.equal(kc, ref_k) and torch.equal(vc, ref_v)
        # slots are the single-device ones: logical token index == position in the (frame, hw) order
        slots = ex.token_slots(local_start, frames, fs, "cpu")
        ok = ok and sorted(slots.tolist()) == list(range(local_start, local_start + n))
        # head gather against the reference golden
        fx = golden("layout.npz")
        if world in (2, 4):
            part = fx[f"scatter_cp{world}_r{rank}"][0].float()          # [F*hw_local, 2]
            back = ex.gather_head(part, 1, 3)
            ok = ok and torch.equal(back, fx[f"gather_cp{world}"][0].float())
            tok = torch.arange(72 * 2, dtype=torch.float32).view(72, 2)
            ok = ok and torch.equal(back, tok)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,paged", [(2, False), (4, False), (2, True)])
def test_exchange_equals_single_device_cache(world, paged):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), paged, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def _launcher_worker(rank, world, port, ret):
    """What the reference's `run_self_forcing.py:49-70` does on every rank, with gloo standing in for nccl: ParallelConfig from
    (LOCAL_RANK, RANK, world size, --ulysses_size, --ring_size), then the pipeline sizes its caches from it."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
        from inferix_amd.kvcache_manager.model import SelfForcingKVCacheManagerFactory
        from inferix_amd.pipeline import CausalInferencePipeline
        from inferix_amd.schedulers import FlowMatchScheduler
        from inferix_amd.sequence_parallel import SequenceParallelExchange, attach_sequence_parallel
        from inferix_amd.wan import ParallelConfig
        pc = ParallelConfig(local_rank=rank, rank=rank, world_size=dist.get_world_size(), ulysses_size=1, ring_size=world)
        ok = (pc.ring_size, pc.ulysses_size, pc.world_size, pc.rank) == (world, 1, world, rank)
        heads, hd, layers, fs = 2, 8, 2, 24
        sched = FlowMatchScheduler(shift=5.0, sigma_min=0.0, extra_one_step=True)
        sched.set_timesteps(1000, training=True)
        model = SimpleNamespace(num_layers=layers, local_attn_size=-1, parallel_config=pc, cp=None, text_len=8,
                                blocks=[SimpleNamespace(kv_cache_manager=SelfForcingKVCacheManagerFactory.create_manager(i, heads, hd))
                                        for i in range(layers)])
        gen = SimpleNamespace(model=model, parallel_config=pc, get_scheduler=lambda: sched)
        args = SimpleNamespace(denoising_step_list=[1000, 500], warp_denoising_step=True, num_frame_per_block=3,
                               frame_seq_length=fs, kv_cache_tokens=6 * fs)
        pipe = CausalInferencePipeline(args, "cpu", generator=gen, parallel_config=pc)
        mgr, req = KVCacheManager("cpu"), KVCacheRequest("req_0")
        pipe._initialize_kv_cache(mgr, [req], torch.float32)
        # replicated, full size on every rank: NOT tokens / ring x heads / ulysses (reference self_forcing_kv_cache_manager.py:45-57)
        ok = ok and tuple(mgr.get_raw(req, "layer_0").shape) == (2, 6 * fs, 1, heads, hd)
        # the attach accepts the launcher's config (rank / world are checked against the process group, the degrees are not)
        sp = attach_sequence_parallel(model, exchange=SequenceParallelExchange())
        ok = ok and model.cp is sp and sp.ex.world == world and sp.gemm_small_split
        # and the exchange at that degree fills the manager's cache exactly as one device would
        frames, local_start, n = 3, 2 * fs, 3 * fs
        g = torch.Generator().manual_seed(3)
        new_k, new_v = torch.randn(n, heads, hd, generator=g), torch.randn(n, heads, hd, generator=g)
        hw = fs // world
        mine = torch.stack([new_k.view(frames, fs, heads, hd)[:, rank * hw:(rank + 1) * hw].reshape(-1, heads, hd),
                            new_v.view(frames, fs, heads, hd)[:, rank * hw:(rank + 1) * hw].reshape(-1, heads, hd)], dim=1)
        t = mgr.get_raw(req, "layer_0")
        t.zero_()
        kc, vc = t[0].view(-1, heads, hd), t[1].view(-1, heads, hd)
        sp.ex.exchange_new_block(mine, kc, vc, local_start, frames, fs)
        ok = ok and torch.equal(kc[local_start:local_start + n], new_k) and torch.equal(vc[local_start:local_start + n], new_v)
        # allocation identity: a second manager reusing the request id names a different allocation, identically on all ranks
        mgr2 = KVCacheManager("cpu")
        pipe._initialize_kv_cache(mgr2, [req], torch.float32)
        a, b = mgr.allocation_id(req, "layer_0"), mgr2.allocation_id(req, "layer_0")
        both = [None] * world
        dist.all_gather_object(both, (a, b))
        ok = ok and a != b and a[:2] == b[:2] and all(x == both[0] for x in both)
        seen = []
        mgr2.add_free_listener(seen.append)
        mgr2.free(req)
        ok = ok and seen == ["req_0"]
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_reference_launcher_parallel_config_two_ranks():
    """`self_forcing.sh` defaults (ULYSSES_SIZE=1, RING_SIZE=2) through `run_self_forcing.py:58-67`'s ParallelConfig call on two gloo
    ranks: accepted, cache replicated at full size, sequence-parallel exchange attached at degree world_size."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_launcher_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_peer_store_timeout_check_runs_once_per_block_and_at_the_end_of_a_clip():
    """The peer-store waits report a timeout through a sticky status word; reading it synchronises the host and runs a small collective.
    `HipSequenceParallel` does that every `check_every` forwards (default 5 = a block) and when the pipeline finishes a clip
    (`check_now`), not once per forward — every rank runs the same collectives either way, so the cadence cannot split the group."""
    from inferix_amd.sequence_parallel import HipSequenceParallel, LoopbackExchange

    class FakePeer:
        emulated, world, rank = False, 2, 0

        def __init__(self):
            self.checks = 0

        def check(self):
            self.checks += 1

    peer = FakePeer()
    sp = HipSequenceParallel(exchange=LoopbackExchange(2, 0), peer=peer)
    assert sp.check_every == 5
    y = torch.zeros(3 * 4, 2)
    for i in range(12):
        sp.gather_head(y, 1, 3)
        assert peer.checks == (i + 1) // 5
    sp.check_now()                                   # what the pipelines call before they hand a clip out
    assert peer.checks == 3
    for _ in range(4):
        sp.gather_head(y, 1, 3)
    assert peer.checks == 3, "check_now restarts the count"
    sp.gather_head(y, 1, 3)
    assert peer.checks == 4
    sp.peer = None
    sp.check_now()                                   # no peer-store exchange: nothing to read
