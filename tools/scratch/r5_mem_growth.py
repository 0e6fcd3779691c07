"""Does device memory grow over many paired clips?  (side-stream allocations, record_stream, per-layer events, memo tables)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from inferix_amd.core import DecodeMode
from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
dev = torch.device("cuda", 0)
model, gen, pipe = bench.build_pipeline(dev)
noise = torch.randn(1, bench.FRAMES, *bench.LATENT, generator=torch.Generator().manual_seed(0)).to(torch.bfloat16).to(dev)
kvm, reqs = KVCacheManager(dev), [KVCacheRequest("m")]
for i in range(30):
    pipe.inference(noise=noise, text_prompts=["x"], kv_cache_manager=kvm, kv_cache_requests=reqs, decode_mode=DecodeMode.NO_DECODE,
                   free_cache_before_vae=False)
    if i % 5 == 4:
        torch.cuda.synchronize()
        print(f"clip {i + 1}: allocated {torch.cuda.memory_allocated() / 2**20:.1f} MiB, reserved {torch.cuda.memory_reserved() / 2**20:.1f} MiB, "
              f"peak {torch.cuda.max_memory_allocated() / 2**20:.1f} MiB", flush=True)
