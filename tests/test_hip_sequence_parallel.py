"""GPU: the sequence-parallel model path with 2 ranks (both on cuda:0, gloo transport so that it runs on the
1-GPU test box; the RCCL path differs only in the backend string) reproduces the single-device golden rollout:
same integer KV trace, same latents within the rollout tolerance, with and without compute/comm overlap."""
import os
import socket
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, overlap, name, ret, exchange="allgather", backend="gloo", quant=None, degrees=None, second_call=False):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":                  # one GPU per rank over RCCL (only where the box has them; world 1 = the 1-GPU twin)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import wan_oracle as O
        from fixture_io import golden
        from util import rel_l2
        from inferix_amd.core import DecodeMode
        from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
        from inferix_amd.pipeline import CausalInferencePipeline
        from inferix_amd.sequence_parallel import attach_sequence_parallel
        from inferix_amd.wan import HipCausalWanModel, HipWanDiffusionWrapper, ParallelConfig
        torch.cuda.set_device(rank if backend == "nccl" else 0)
        fx = golden(name)
        cfg = O.tiny_config(local_attn_size=6, sink_size=1) if "local" in name else O.tiny_config()
        if degrees is None:
            pc = ParallelConfig(rank=rank, world_size=world)
        else:       # exactly the call of the reference's launcher (example/self_forcing/run_self_forcing.py:58-67; self_forcing.sh: RING_SIZE=2)
            pc = ParallelConfig(local_rank=rank, rank=rank, world_size=world, ulysses_size=degrees[0], ring_size=degrees[1])
        m = HipCausalWanModel(patch_size=cfg.patch_size, text_len=cfg.text_len, in_dim=cfg.in_dim, dim=cfg.dim,
                              ffn_dim=cfg.ffn_dim, freq_dim=cfg.freq_dim, text_dim=cfg.text_dim, out_dim=cfg.out_dim,
                              num_heads=cfg.num_heads, num_layers=cfg.num_layers, local_attn_size=cfg.local_attn_size,
                              sink_size=cfg.sink_size, eps=cfg.eps, parallel_config=pc)
        m.load_state_dict(O.init_weights(cfg, seed=0))
        peer = None
        if exchange == "peer":
            from inferix_amd.sequence_parallel import PeerStoreExchange
            peer = PeerStoreExchange(timeout_ms=10000)
        sp = attach_sequence_parallel(m, overlap=overlap, peer=peer)
        # the bench's preflight (first contact with the interconnect): both exchanges into scratch caches, compared bit for bit on
        # every rank and across ranks, before the model's own caches see either
        pf = sp.preflight(m, cfg.latent_h // cfg.patch_size[1], cfg.latent_w // cfg.patch_size[2], frames=3)
        assert pf["allgather_ok"] and pf["allgather_us"] > 0 and pf["ranks"] == world, pf
        if exchange == "peer":
            assert pf.get("peer_store_ok") is True and sp.peer is not None, pf
        gen = HipWanDiffusionWrapper(model=m, timestep_shift=float(fx["shift"]), parallel_config=pc)
        single = None
        if quant:
            # the 8-bit linears under sequence parallelism (the split K/V-first projection used to bypass them): the same
            # quantised model WITHOUT sharding, on this rank's device, is what the sharded rollout has to reproduce
            from inferix_amd.quant import get_dynamic_fp8_per_token_act_per_channel_weight_qconfig, quantize_dynamic
            qc = get_dynamic_fp8_per_token_act_per_channel_weight_qconfig()
            excl = {"": qc, "text_embedding": None, "proj_out": None, "head": None}
            quantize_dynamic(gen, excl)
            assert m.quantized_linears == cfg.num_layers * 8
            from inferix_amd import hip_ops as _ops
            bf16_shapes, lin0 = set(), _ops.linear

            def lin1(x, w, *a, **k):
                bf16_shapes.add(tuple(w.shape))
                return lin0(x, w, *a, **k)
            _ops.linear = lin1                       # every bf16 GEMM of this process from here on is on record
            m1 = HipCausalWanModel(patch_size=cfg.patch_size, text_len=cfg.text_len, in_dim=cfg.in_dim, dim=cfg.dim, ffn_dim=cfg.ffn_dim,
                                   freq_dim=cfg.freq_dim, text_dim=cfg.text_dim, out_dim=cfg.out_dim, num_heads=cfg.num_heads,
                                   num_layers=cfg.num_layers, local_attn_size=cfg.local_attn_size, sink_size=cfg.sink_size, eps=cfg.eps)
            m1.load_state_dict(O.init_weights(cfg, seed=0))
            single = HipWanDiffusionWrapper(model=m1, timestep_shift=float(fx["shift"]))
            quantize_dynamic(single, excl)
        args = SimpleNamespace(denoising_step_list=fx["steps"].tolist(), warp_denoising_step=True,
                               num_frame_per_block=3, independent_first_frame=False, context_noise=0,
                               frame_seq_length=cfg.frame_seqlen, kv_cache_tokens=21 * cfg.frame_seqlen)
        pe = fx["prompt_embeds"].cuda()
        pipe = CausalInferencePipeline(args, "cuda", generator=gen,
                                       text_encoder=lambda text_prompts: {"prompt_embeds": pe}, vae=None,
                                       parallel_config=pc)
        trace = []
        shapes, idents = set(), set()
        # recorded where the MODEL leaves layer 0 (not around gen.forward): a sequence-parallel pipeline enqueues the clean-context re-run
        # of a block together with the next block's first step (forward_pair, which does not pass through gen.forward); the first chain's
        # layer 0 precedes the second's on the host, so the records come in the order of the sequential calls
        run0 = m._run_block
        index_trace = m.index_trace = []

        def rec(l, xact, El, st, meta, cmeta, mgr, rq):
            run0(l, xact, El, st, meta, cmeta, mgr, rq)
            if l == 0:
                shapes.add(tuple(mgr.get_raw(rq[0], "layer_0").shape))
                del trace[:]
                trace.extend(list(r) for r in index_trace)
            if peer is not None:
                idents.update(k[2] for k in peer._views)
        m._run_block = rec
        assert pipe._pairing() or os.environ.get("IFX_PAIR_FORWARDS", "") in ("0", "false", "off"), "the pipeline pairs its forwards by default"
        renoise = [fx[f"renoise_{i}"] for i in range(int(fx["num_renoise"]))]
        mgr1 = KVCacheManager("cuda")
        out = pipe.inference(noise=fx["noise"].cuda(), text_prompts=["x"], kv_cache_manager=mgr1,
                             kv_cache_requests=[KVCacheRequest("r")], decode_mode=DecodeMode.NO_DECODE,
                             renoise=renoise)
        torch.cuda.synchronize()
        ok_trace = trace == fx["trace"].tolist()
        # floor rule of the single-device rollout tests (tests/test_hip_model.py::_run_rollout): the reference's own distance from the
        # exact-attention rollout is the yardstick; `r` = the worse of (HIP vs reference, HIP vs exact) as a fraction of 1.25 x floor + 5e-4
        floor = rel_l2(fx["out"], fx["out_exact"])
        r_ref, r_exact = rel_l2(out.cpu(), fx["out"]), rel_l2(out.cpu(), fx["out_exact"])
        print(f"rank {rank} {name} ({exchange}): floor {floor:.3e}; sharded HIP vs exact {r_exact:.3e}; vs reference {r_ref:.3e}")
        r = r_ref if quant else max(r_ref, r_exact) / (1.25 * floor + 5e-4)
        if degrees is not None:      # the replicated cache: every rank holds the full-size tensor whatever the degrees say
            assert shapes == {(2, 21 * cfg.frame_seqlen, 1, cfg.num_heads, cfg.dim // cfg.num_heads)}, shapes
        if second_call:
            # the pipelines make a NEW manager per call and reuse the request id; the peer address book must not hand the
            # second call the first call's (freed, still IPC-mapped) caches.  Same request id, fresh manager, caches of the first call
            # dropped in between: the rollout has to come out again, bit for bit.
            del trace[:]
            del index_trace[:]
            idents.clear()
            mgr1.free(KVCacheRequest("r"))                       # (inference() has already dropped the layers: free_cache_before_vae)
            del mgr1
            torch.cuda.empty_cache()
            mgr2 = KVCacheManager("cuda")
            out2 = pipe.inference(noise=fx["noise"].cuda(), text_prompts=["x"], kv_cache_manager=mgr2,
                                  kv_cache_requests=[KVCacheRequest("r")], decode_mode=DecodeMode.NO_DECODE, renoise=renoise)
            torch.cuda.synchronize()
            assert trace == fx["trace"].tolist()
            assert torch.equal(out2, out), f"second call through a fresh manager differs: {rel_l2(out2.cpu(), out.cpu()):.3e}"
            if peer is not None:
                # the address book was keyed by the live manager's allocations only, and the layers' release at the end of
                # inference() (free_layer -> free listener -> PeerStoreExchange.forget) has emptied it
                assert idents and all(i[0] == mgr2.serial for i in idents), (idents, mgr2.serial)
                assert not peer._views, "KVCacheManager.free_layer did not drop the peer address book"
        if single is not None:
            _ops.linear = lin0
            block_shapes = {(3 * cfg.dim, cfg.dim), (2 * cfg.dim, cfg.dim), (cfg.ffn_dim, cfg.dim), (cfg.dim, cfg.ffn_dim)}
            assert not (bf16_shapes & block_shapes), f"block linears ran in bf16 under sequence parallelism: {bf16_shapes & block_shapes}"
            pipe1 = CausalInferencePipeline(args, "cuda", generator=single, text_encoder=lambda text_prompts: {"prompt_embeds": pe}, vae=None)
            out1 = pipe1.inference(noise=fx["noise"].cuda(), text_prompts=["x"], kv_cache_manager=KVCacheManager("cuda"),
                                   kv_cache_requests=[KVCacheRequest("q1")], decode_mode=DecodeMode.NO_DECODE, renoise=renoise)
            torch.cuda.synchronize()
            ret[rank] = (ok_trace, r, rel_l2(out.cpu(), out1.cpu()), rel_l2(out1.cpu(), fx["out"]))
        else:
            ret[rank] = (ok_trace, r)
        if peer is not None:
            peer.check()
            dist.barrier()
            peer.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("name", ["rollout_tiny.npz", "rollout_tiny_local.npz"])
def test_sequence_parallel_rollout_matches_single_device_golden(overlap, name):
    world = 2
    # (a SPAWNED manager: forking the pytest process — HIP runtime, streams and events alive in it — for the manager server has
    #  crashed in the child's garbage collector; the workers themselves are spawned by mp.spawn already)
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), overlap, name, ret), nprocs=world, join=True)
    for rank in range(world):
        ok_trace, r = ret[rank]
        assert ok_trace, f"rank {rank}: KV index trace differs from the single-device reference trace"
        assert r <= 1.0, f"rank {rank}: rollout at {r:.2f} x the bound (1.25 x floor + 5e-4)"


@pytest.mark.parametrize("overlap,name", [(True, "rollout_tiny.npz"), (True, "rollout_tiny_local.npz"), (False, "rollout_tiny.npz")])
def test_sequence_parallel_peer_store_rollout(overlap, name):
    """The exchange without a collective: two PROCESSES (both on cuda:0) store their K/V rows into each other's cache through HIP IPC
    mappings, ordered by the ready / done flags — same golden, same tolerance as the all-gather path."""
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), overlap, name, ret, "peer"), nprocs=world, join=True)
    for rank in range(world):
        ok_trace, r = ret[rank]
        assert ok_trace, f"rank {rank}: KV index trace differs from the single-device reference trace"
        assert r <= 1.0, f"rank {rank}: rollout at {r:.2f} x the bound (1.25 x floor + 5e-4)"


@pytest.mark.parametrize("exchange", ["allgather", "peer"])
def test_sequence_parallel_with_quantised_linears_matches_the_unsharded_quantised_model(exchange):
    """FP8 per-token x per-channel linears under 2-way sequence parallelism, both exchanges (the all-gather one defaults to the
    K/V-projection-first split, which must stay on the quantised weights): per-token activation scales and per-channel weight scales
    make every row independent of the sharding, so the sharded rollout has to land on the unsharded quantised one at the bf16 floor of
    a rollout (the attention key splits differ), and no block linear may run on the bf16 weights (every bf16 GEMM's weight shape is recorded)."""
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), True, "rollout_tiny.npz", ret, exchange, "gloo", "fp8"), nprocs=world, join=True)
    for rank in range(world):
        ok_trace, r, vs_single, single_vs_bf16 = ret[rank]
        print(f"rank {rank} ({exchange}): sharded fp8 vs bf16 golden {r:.3e}, unsharded fp8 vs golden {single_vs_bf16:.3e}, sharded vs unsharded {vs_single:.3e}")
        assert ok_trace and r < 0.15 and single_vs_bf16 < 0.15
        assert single_vs_bf16 > 3e-3, "the unsharded model does not look quantised"
        assert vs_single < 5e-3, (rank, vs_single, single_vs_bf16)      # measured 2.8e-3: the attention splits differ, at the bf16 floor of a rollout


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="one GPU per rank over RCCL / xGMI: needs >= 2 GPUs (the driver's multi-GPU node)")
@pytest.mark.parametrize("exchange", ["allgather", "peer"])
def test_sequence_parallel_rollout_over_rccl(exchange):
    """The same rollout with one GPU per rank: the all-gather over RCCL, the peer stores over HIP IPC between DEVICES, the preflight in
    front.  Runs only where two GPUs are visible (round-2 verdict: nothing had executed over RCCL)."""
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), True, "rollout_tiny.npz", ret, exchange, "nccl"), nprocs=world, join=True)
    for rank in range(world):
        ok_trace, r = ret[rank]
        assert ok_trace and r <= 1.0, (rank, ok_trace, r)


@pytest.mark.parametrize("exchange,degrees", [("allgather", (1, 2)), ("peer", (1, 2)), ("peer", (2, 1))])
def test_reference_launcher_parallel_config_runs_the_sequence_parallel_exchange(exchange, degrees):
    """The reference's stock launch line (`self_forcing.sh`: ULYSSES_SIZE=1 RING_SIZE=2 -> `ParallelConfig(local_rank, rank,
    world_size, ulysses_size, ring_size)`, run_self_forcing.py:58-67) on two ranks: both degrees map onto the sequence-parallel
    exchange with a replicated full-size cache, the rollout equals the single-device golden, and a second call through a fresh
    manager with the same request id (what the pipelines do) reproduces it bit for bit."""
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), True, "rollout_tiny.npz", ret, exchange, "gloo", None, degrees, True), nprocs=world, join=True)
    for rank in range(world):
        ok_trace, r = ret[rank]
        assert ok_trace and r <= 1.0, (rank, ok_trace, r)


@pytest.mark.parametrize("overlap,name", [(True, "rollout_tiny.npz"), (False, "rollout_tiny.npz"), (True, "rollout_tiny_local.npz")])
def test_sequence_parallel_rollout_over_rccl_one_rank_twin(overlap, name):
    """1-GPU twin of the RCCL rollout test (round-3 verdict #8).  RCCL refuses two ranks on one device (tools/probe_rccl_one_gpu.py:
    "Duplicate GPU detected", with or without NCCL_IGNORE_DUPLICATE_GPU), so the RCCL group here has ONE rank: the model runs its
    sequence-parallel route — K/V-first projection, `dist.all_gather_into_tensor` on the RCCL backend from the side stream, the scatter
    into the replicated cache, prefix / new-block attention split + merge, the head gather, the preflight — against the single-device
    golden (with and without overlap, with the rolling window).  Not covered: bytes between two devices."""
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(1, _free_port(), overlap, name, ret, "allgather", "nccl"), nprocs=1, join=True)
    ok_trace, r = ret[0]
    assert ok_trace and r <= 1.0, (ok_trace, r)


def _push_worker(rank, world, port, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from inferix_amd import hip_ops as ops
        from inferix_amd.sequence_parallel import PeerStoreExchange
        from inferix_amd.wan import components as C
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        H, hd, frames, gh, gw = 12, 128, 3, 6, 8
        d, fs = H * hd, gh * gw
        hw_local = fs // world
        n_local, N = frames * hw_local, frames * fs
        slots, local_start, eps = 3 * N, N + 7, 1e-6
        g = torch.Generator().manual_seed(11)
        qkv_full = torch.randn(N, 3 * d, generator=g).to(torch.bfloat16).to(dev)       # single-device rows, (frame, hw) order
        wq = (1 + 0.1 * torch.randn(d, generator=g)).to(torch.bfloat16).to(dev)
        wk = (1 + 0.1 * torch.randn(d, generator=g)).to(torch.bfloat16).to(dev)
        freqs = C.rope_table(hd).to(dev)
        # what a single device writes
        kc = torch.zeros(slots, H, hd, dtype=torch.bfloat16, device=dev)
        vc = torch.zeros_like(kc)
        rope1 = ops.RopeGridSpec(freqs, 2, gh, gw, 0, fs)
        ops.rmsnorm_rope_kv_append(qkv_full, wq, wk, eps, rope1, ops.KvCacheView(kc, vc), local_start, d)
        # this rank's rows: per frame the hw slice [rank * hw_local, (rank + 1) * hw_local)
        mine = qkv_full.view(frames, fs, 3 * d)[:, rank * hw_local:(rank + 1) * hw_local].reshape(n_local, 3 * d).contiguous()
        kp = torch.zeros(slots, H, hd, dtype=torch.bfloat16, device=dev)
        vp = torch.zeros_like(kp)
        view = ops.KvCacheView(kp, vp)
        px = PeerStoreExchange(timeout_ms=10000)
        ropeP = ops.RopeGridSpec(freqs, 2, gh, gw, rank * hw_local, hw_local)
        ok = True
        for it in range(3):                      # several epochs through the same flags
            kp.zero_()
            vp.zero_()
            torch.cuda.synchronize()
            dist.barrier()
            e = px.push(5, mine[:, d:], wk, eps, ropeP, view, local_start, fs, d)      # strided (k | v) view of the fused rows
            px.wait_done(5, e)
            torch.cuda.synchronize()
            px.check()
            ok = ok and torch.equal(kp, kc) and torch.equal(vp, vc)
            dist.barrier()
        ret[rank] = ok
        px.close()
    finally:
        dist.destroy_process_group()


def test_peer_store_push_two_processes_bit_exact():
    """`ifx_rmsnorm_rope_kv_push` through IPC mappings: after ready / push / done, each of the two processes' caches holds exactly
    the bytes one device writes with `ifx_rmsnorm_rope_kv_append` for the whole block."""
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_push_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)
