#!/usr/bin/env python3
"""Run the full 480p 21-frame clip several times with identical seeds and compare the latents bit for bit
(run-to-run determinism of the whole path at real sizes)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from inferix_amd.core import DecodeMode  # noqa: E402
from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
model, gen, pipe = bench.build_pipeline(dev)
g = torch.Generator().manual_seed(0)
noise = torch.randn(1, bench.FRAMES, *bench.LATENT, generator=g).to(torch.bfloat16).to(dev)
kvm, reqs = KVCacheManager(dev), [KVCacheRequest("det")]
hs = []
for i in range(n):
    torch.manual_seed(1234)
    out = pipe.inference(noise=noise, text_prompts=["x"], kv_cache_manager=kvm, kv_cache_requests=reqs,
                         decode_mode=DecodeMode.NO_DECODE, free_cache_before_vae=False)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    hs.append(hashlib.md5(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:10])
print("clip hashes:", hs, "-> deterministic" if len(set(hs)) == 1 else "-> NOT deterministic")
