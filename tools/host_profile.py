"""Host-side cost of one rank of an sp-P shard (P=1: the single-GPU path): wall time per clip, the time the Python side needs
to ENQUEUE a clip (no GPU wait: measured with the GPU kept behind by a long sleep kernel is not needed — the launches are
asynchronous, so `enqueue` is the time until `clip()` returns), and a cProfile of the enqueue.  `python tools/host_profile.py 8`."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    prof = len(sys.argv) > 2 and sys.argv[2] == "profile"
    from inferix_amd.core import DecodeMode
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    from inferix_amd.wan import ParallelConfig
    device = torch.device("cuda:0")
    pc = ParallelConfig(rank=0, world_size=P, local_rank=0) if P > 1 else None
    model, gen, pipe = bench.build_pipeline(device, pc)
    if P > 1:
        from inferix_amd.sequence_parallel import LoopbackExchange, PeerStoreExchange, attach_sequence_parallel
        peer = PeerStoreExchange(emulate_world=P) if os.environ.get("SP_EXCHANGE", "peer") == "peer" else None     # as bench.py --sp-exchange peer
        attach_sequence_parallel(model, exchange=LoopbackExchange(P, 0), peer=peer)
    g = torch.Generator().manual_seed(0)
    noise = torch.randn(1, bench.FRAMES, *bench.LATENT, generator=g).to(torch.bfloat16).to(device)
    kvm, reqs = KVCacheManager(device), [KVCacheRequest("bench")]

    def clip():
        return pipe.inference(noise=noise, text_prompts=["synthetic"], kv_cache_manager=kvm, kv_cache_requests=reqs,
                              decode_mode=DecodeMode.NO_DECODE, free_cache_before_vae=False)
    clip()
    torch.cuda.synchronize()
    enq, wall = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        clip()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3)
        wall.append((t2 - t0) * 1e3)
    print(f"sp{P}: enqueue {min(enq):.1f} ms/clip, wall {min(wall):.1f} ms/clip")
    if prof:
        pr = cProfile.Profile()
        pr.enable()
        clip()
        pr.disable()
        torch.cuda.synchronize()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
        print(s.getvalue()[:12000])


if __name__ == "__main__":
    main()
