#!/usr/bin/env python3
"""Run single legs of bench.py on one GPU without the timed headline region (development aid):
    python tools/run_bench_leg.py streaming [blocks]     # streaming_steady_leg
    python tools/run_bench_leg.py config1                # cpu_baseline + config1_gpu + the hard parity check against config1_full.npz
    python tools/run_bench_leg.py batch [B]              # batch_leg
    python tools/run_bench_leg.py decode                 # per_block_decode_leg + vae_decode_leg
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "streaming"
device = torch.device("cuda:0")
torch.cuda.set_device(0)
model, gen, pipe = bench.build_pipeline(device)
if which == "streaming":
    res = bench.streaming_steady_leg(model, gen, device, blocks=int(sys.argv[2]) if len(sys.argv) > 2 else 30)
elif which == "config1":
    res = bench.cpu_baseline(30, gpu_leg=lambda W: bench.config1_gpu(model, gen, device, W))
elif which == "batch":
    res = bench.batch_leg(model, gen, device, int(sys.argv[2]) if len(sys.argv) > 2 else 2)
elif which == "decode":
    from inferix_amd.core import DecodeMode
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    g = torch.Generator().manual_seed(0)
    noise = torch.randn(1, bench.FRAMES, *bench.LATENT, generator=g).to(torch.bfloat16).to(device)
    kvm, reqs = KVCacheManager(device), [KVCacheRequest("bench")]
    res = {"per_block_decode": bench.per_block_decode_leg(lambda cb: pipe.inference(
        noise=noise, text_prompts=["synthetic"], kv_cache_manager=kvm, kv_cache_requests=reqs, decode_mode=DecodeMode.NO_DECODE,
        free_cache_before_vae=False, block_callback=cb)), "vae_decode": bench.vae_decode_leg()}
else:
    raise SystemExit(__doc__)
print(json.dumps(res))
