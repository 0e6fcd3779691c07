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


def _worker(rank, world, port, overlap, name, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
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
        torch.cuda.set_device(0)
        fx = golden(name)
        cfg = O.tiny_config(local_attn_size=6, sink_size=1) if "local" in name else O.tiny_config()
        pc = ParallelConfig(rank=rank, world_size=world)
        m = HipCausalWanModel(patch_size=cfg.patch_size, text_len=cfg.text_len, in_dim=cfg.in_dim, dim=cfg.dim,
                              ffn_dim=cfg.ffn_dim, freq_dim=cfg.freq_dim, text_dim=cfg.text_dim, out_dim=cfg.out_dim,
                              num_heads=cfg.num_heads, num_layers=cfg.num_layers, local_attn_size=cfg.local_attn_size,
                              sink_size=cfg.sink_size, eps=cfg.eps, parallel_config=pc)
        m.load_state_dict(O.init_weights(cfg, seed=0))
        attach_sequence_parallel(m, overlap=overlap)
        gen = HipWanDiffusionWrapper(model=m, timestep_shift=float(fx["shift"]), parallel_config=pc)
        args = SimpleNamespace(denoising_step_list=fx["steps"].tolist(), warp_denoising_step=True,
                               num_frame_per_block=3, independent_first_frame=False, context_noise=0,
                               frame_seq_length=cfg.frame_seqlen, kv_cache_tokens=21 * cfg.frame_seqlen)
        pe = fx["prompt_embeds"].cuda()
        pipe = CausalInferencePipeline(args, "cuda", generator=gen,
                                       text_encoder=lambda text_prompts: {"prompt_embeds": pe}, vae=None,
                                       parallel_config=pc)
        trace = []
        orig = gen.forward

        def rec(**kw):
            out = orig(**kw)
            meta = kw["kv_cache_meta"][0]
            trace.append([int(kw["current_start"]), int(meta["global_end_index"]), int(meta["local_end_index"])])
            return out
        gen.forward = rec
        renoise = [fx[f"renoise_{i}"] for i in range(int(fx["num_renoise"]))]
        out = pipe.inference(noise=fx["noise"].cuda(), text_prompts=["x"], kv_cache_manager=KVCacheManager("cuda"),
                             kv_cache_requests=[KVCacheRequest("r")], decode_mode=DecodeMode.NO_DECODE,
                             renoise=renoise)
        torch.cuda.synchronize()
        ok_trace = trace == fx["trace"].tolist()
        r = rel_l2(out.cpu(), fx["out"])
        ret[rank] = (ok_trace, r)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("name", ["rollout_tiny.npz", "rollout_tiny_local.npz"])
def test_sequence_parallel_rollout_matches_single_device_golden(overlap, name):
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), overlap, name, ret), nprocs=world, join=True)
    for rank in range(world):
        ok_trace, r = ret[rank]
        assert ok_trace, f"rank {rank}: KV index trace differs from the single-device reference trace"
        assert r < 1e-2, f"rank {rank}: rollout rel-L2 {r:.3e}"
