"""Which torch (non-library) kernels one clip launches, by call site: torch.profiler over ONE clip of the emulated sequence-parallel rank
(or P = 1), aten ops that launch a kernel grouped by the innermost inferix_amd / bench frame.  `python tools/glue_profile.py 8`."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    from inferix_amd.core import DecodeMode
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    from inferix_amd.wan import ParallelConfig
    device = torch.device("cuda:0")
    pc = ParallelConfig(rank=0, world_size=P, local_rank=0) if P > 1 else None
    model, gen, pipe = bench.build_pipeline(device, pc)
    if P > 1:
        from inferix_amd.sequence_parallel import LoopbackExchange, PeerStoreExchange, attach_sequence_parallel
        attach_sequence_parallel(model, exchange=LoopbackExchange(P, 0), peer=PeerStoreExchange(emulate_world=P))
    g = torch.Generator().manual_seed(0)
    noise = torch.randn(1, bench.FRAMES, *bench.LATENT, generator=g).to(torch.bfloat16).to(device)
    kvm, reqs = KVCacheManager(device), [KVCacheRequest("bench")]

    def clip():
        return pipe.inference(noise=noise, text_prompts=["synthetic"], kv_cache_manager=kvm, kv_cache_requests=reqs,
                              decode_mode=DecodeMode.NO_DECODE, free_cache_before_vae=False)
    clip()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        clip()
        torch.cuda.synchronize()
    by_site = collections.Counter()
    by_op = collections.Counter()
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
            continue
        if not ev.kernels:
            continue
        site = next((f for f in ev.stack if "inferix_amd" in f or "bench.py" in f), "?")
        by_site[(site.split("/")[-1][:90], ev.name)] += 1
        by_op[ev.name] += 1
    print("kernel-launching aten ops per clip:", sum(by_op.values()))
    for (site, op), n in by_site.most_common(45):
        print(f"{n:6d}  {op:28s} {site}")


if __name__ == "__main__":
    main()
