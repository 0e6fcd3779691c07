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
