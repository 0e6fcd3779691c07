#!/usr/bin/env python3
"""One eviction form of the streaming leg, denoise only, `blocks` blocks (lab: run under rocprofv3 to compare the two forms per kernel).
    python tools/scratch/stream_form.py page|shift [blocks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from types import SimpleNamespace
import torch
import bench
from inferix_amd.core import DecodeMode
from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
from inferix_amd.pipeline import CausalInferencePipeline

form = sys.argv[1]
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 10
device = torch.device("cuda:0")
torch.cuda.set_device(0)
model, gen, _ = bench.build_pipeline(device)
model.local_attn_size, model.sink_size = 21, 3
args = SimpleNamespace(denoising_step_list=bench.STEPS_LIST, warp_denoising_step=True, num_frame_per_block=bench.BLOCK,
                       independent_first_frame=False, context_noise=0, frame_seq_length=1560, kv_cache_tokens=None)
pe = torch.zeros(1, 512, 4096)
pe[:, :40] = torch.randn(1, 40, 4096, generator=torch.Generator().manual_seed(1))
pe = pe.to(torch.bfloat16).to(device)
noise = torch.randn(1, blocks * bench.BLOCK, *bench.LATENT, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).to(device)
pipe = CausalInferencePipeline(args, device, generator=gen, text_encoder=lambda text_prompts: {"prompt_embeds": pe}, vae=None)
kvm, reqs = KVCacheManager(device), [KVCacheRequest("stream")]
pipe._initialize_kv_cache(kvm, reqs, torch.bfloat16)
if form == "page":
    for l in range(model.num_layers):
        kvm.enable_paging(reqs[0], f"layer_{l}", 1560)
for rep in range(2):
    marks = []
    host = []
    def cb(block_latent, block_index):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e); host.append(time.perf_counter())
    e0 = torch.cuda.Event(enable_timing=True); e0.record(); t0 = time.perf_counter()
    pipe.inference(noise=noise, text_prompts=["synthetic"], kv_cache_manager=kvm, kv_cache_requests=reqs,
                   decode_mode=DecodeMode.NO_DECODE, free_cache_before_vae=False, block_callback=cb)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ms = [a.elapsed_time(b) for a, b in zip([e0] + marks[:-1], marks)]
    print(form, "rep", rep, "ms/block", [round(m, 1) for m in ms], "host enqueue s", round(t1 - t0, 3), "wall s", round(t2 - t0, 3),
          "host per block ms", [round((b - a) * 1e3, 1) for a, b in zip([t0] + host[:-1], host)])
