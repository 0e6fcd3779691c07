#!/usr/bin/env python3
"""bench.py — Self-Forcing 480p block-diffusion denoising throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one batch of synthetic input: a 21-latent-frame clip
(1 x 21 x 16 x 60 x 104), block_size 3 -> 7 blocks x (4 denoise steps + 1 clean-context re-run) = 35
generator forwards of the 30-layer Wan2.1-1.3B causal DiT with a growing paged KV prefix
(4680 ... 32760 keys).  Text encoder and VAE are outside the path (prompt embeddings are synthetic,
decode_mode = NO_DECODE; the per-block callback receives latents).  Inputs/weights are resident in HBM
before the timed region.

N > 1: the clip is sharded along the per-frame spatial token axis (sequence/context parallel, RCCL K/V
all-gather per layer) -> fixed total work, "scaling": "strong".

Prints ONE JSON line on rank 0 (driver contract) carrying `roofline` (block-causal attention kernel,
MFMA-bound) and `cpu_baseline` (CPU oracle timed on the host cores, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required by RCCL on this driver stack
import torch  # noqa: E402

# gfx950 peaks (MI355X_MICROARCH.md): dense bf16 MFMA ~2.5 PFLOP/s, HBM3E 8 TB/s
PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBPS = 8000.0

WAN_1_3B = dict(patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=1536, ffn_dim=8960, freq_dim=256,
                text_dim=4096, out_dim=16, num_heads=12, num_layers=30, eps=1e-6)
LATENT = (16, 60, 104)         # 480p: 480x832 pixels / 8
FRAMES, BLOCK = 21, 3
STEPS_LIST = [1000, 750, 500, 250]


def build_pipeline(device, parallel_config=None, num_layers=None):
    from inferix_amd.pipeline import CausalInferencePipeline
    from inferix_amd.wan import HipCausalWanModel, HipWanDiffusionWrapper
    from inferix_amd.wan.synthetic import synthetic_state_dict
    cfg = dict(WAN_1_3B)
    if num_layers:
        cfg["num_layers"] = num_layers
    model = HipCausalWanModel(**cfg, parallel_config=parallel_config, device=device)
    model.load_state_dict(synthetic_state_dict(model, seed=0))
    gen = HipWanDiffusionWrapper(model=model, timestep_shift=5.0, parallel_config=parallel_config)
    args = SimpleNamespace(denoising_step_list=STEPS_LIST, warp_denoising_step=True, num_frame_per_block=BLOCK,
                           independent_first_frame=False, context_noise=0, frame_seq_length=1560,
                           kv_cache_tokens=32760)
    g = torch.Generator().manual_seed(1)
    pe = torch.zeros(1, 512, 4096)
    pe[:, :40] = torch.randn(1, 40, 4096, generator=g)
    pe = pe.to(torch.bfloat16).to(device)
    pipe = CausalInferencePipeline(args, device, generator=gen, text_encoder=lambda text_prompts: {"prompt_embeds": pe},
                                   vae=None, parallel_config=parallel_config)
    return model, gen, pipe


def cpu_baseline(budget_layers: int = 4):
    """The CPU oracle (port of the reference's CPU/PyTorch path, parity-pinned to it) on this box's host cores:
    `budget_layers` real-size layers of one denoise forward at the prefix of block 0, 3 and 6 (L_kv = 4680, 18720,
    32760; about 10-20 s of CPU work).  The clip = 7 blocks x 5 forwards x 30 layers is then priced with the
    per-layer time interpolated linearly in the block index between the measured prefixes."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import wan_oracle as O
    cfg = O.WanConfig(num_layers=budget_layers)
    W = O.init_weights(cfg, seed=0)
    g = torch.Generator().manual_seed(0)
    fs = cfg.frame_seqlen
    n = BLOCK * fs
    nblk = FRAMES // BLOCK
    x0 = torch.randn(1, n, cfg.dim, generator=g).to(torch.bfloat16)
    e0 = (torch.randn(1, BLOCK, 6, cfg.dim, generator=g) * 0.5).to(torch.bfloat16)
    ctx = torch.randn(1, cfg.text_len, cfg.dim, generator=g).to(torch.bfloat16)
    freqs = O.rope_freqs(cfg.head_dim)
    state = O.CacheState.allocate(cfg, 1, torch.bfloat16, cache_tokens=nblk * n)
    for l in state.layers:                      # a filled prefix (values are irrelevant for the timing)
        l.k.normal_(generator=g)
        l.v.normal_(generator=g)
    per_layer = {}
    total = 0.0
    with torch.no_grad():
        for b in (0, nblk // 2, nblk - 1):
            for l in state.layers:
                l.global_end = l.local_end = b * n
            x = x0
            t0 = time.perf_counter()
            for i in range(budget_layers):
                x = O.block_forward(x, e0, ctx, W, i, cfg, (BLOCK, 30, 52), freqs, state, b * n)
            dt = time.perf_counter() - t0
            total += dt
            per_layer[b] = dt / budget_layers
    bs = sorted(per_layer)

    def interp(b):
        lo = max(v for v in bs if v <= b)
        hi = min(v for v in bs if v >= b)
        return per_layer[lo] if lo == hi else per_layer[lo] + (per_layer[hi] - per_layer[lo]) * (b - lo) / (hi - lo)
    clip_s = sum(interp(b) for b in range(nblk)) * (len(STEPS_LIST) + 1) * 30
    return {"value": FRAMES / clip_s, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{budget_layers} of 30 layers of one generator forward at block 0 / {nblk // 2} / {nblk - 1} "
                      f"(N = 4680 queries, L_kv = 4680 / {(nblk // 2 + 1) * n} / {nblk * n}; real Wan-1.3B dims, bf16 CPU "
                      f"oracle), {total:.1f} s measured; clip = 7 blocks x 5 forwards x 30 layers, per-layer time "
                      "interpolated linearly in the block index",
            "ms_per_layer_forward_by_block": {str(k): round(v * 1e3, 1) for k, v in per_layer.items()},
            "host_cpus": os.cpu_count()}


def pmc_traffic(shards):
    """HBM bytes per self-attention launch from the committed PMC pass (profiles/pmc_traffic.json: FETCH_SIZE /
    WRITE_SIZE of `rocprofv3 --pmc` on tools/pmc_micro.py at the clip's MEAN prefix L = 18720, N = 4680 — traffic
    is linear in L, so that launch is the clip average).  FETCH_SIZE is doubled per the gfx950 correction of
    MI355X_MICROARCH.md §HBM.  Counters cannot be collected inside this process; null when absent / sharded."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if shards != 1 or not os.path.exists(path):
        return {"traffic": None}
    with open(path) as f:
        t = json.load(f)["attn_self"]
    b = (2 * t["fetch_size_kib"] + t["write_size_kib"]) * 1024.0
    return {"traffic": round(b), "traffic_unit": "B/launch", "traffic_source": t["source"],
            "algorithmic_bytes_per_launch": 2 * (2 * 4680 * 1536 + 2 * 18720 * 1536)}


def vae_decode_leg():
    """The other half of the per-block latency, measured AFTER the timed region and not part of `value`:
    the Wan2.1 VAE decoder (dim 96, synthetic weights) on this GPU, (a) as the streaming pipeline calls it — one block of 3
    latent frames from a clean cache -> 9 video frames — and (b) a whole 21-frame clip -> 81 frames."""
    import time
    from inferix_amd import hip_ops as ops
    from inferix_amd.vae import HipWanVAEWrapper, synthetic_decoder_state_dict
    vae = HipWanVAEWrapper(synthetic_decoder_state_dict(seed=0))
    lat = torch.randn(1, FRAMES, *LATENT, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).cuda()

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, out

    blk_ms, blk = timed(lambda: vae.decode_to_pixel(lat[:, :BLOCK], use_cache=True, chunk_size=1), 3)
    clip_ms, vid = timed(lambda: vae.decode_to_pixel(lat, use_cache=True, chunk_size=BLOCK), 2)
    t = ops.KernelTimer(names=("conv3d",))
    ops.set_kernel_timer(t)
    vae.decode_to_pixel(lat, use_cache=True, chunk_size=BLOCK)
    ops.set_kernel_timer(None)
    cs = t.summary()["conv3d"]
    return {"workload": f"Wan2.1 VAE decoder (dim 96), latent {LATENT[1]}x{LATENT[2]} -> {vid.shape[-2]}x{vid.shape[-1]} px, bf16, "
                        "synthetic weights; includes the fp32 [-1,1] pixel hand-off",
            "ms_per_block_streaming": round(blk_ms, 2), "video_frames_per_block": int(blk.shape[1]),
            "ms_per_clip": round(clip_ms, 1), "video_frames_per_clip": int(vid.shape[1]),
            "video_frames_per_s": round(vid.shape[1] / clip_ms * 1e3, 1),
            "conv_launches": cs["launches"], "conv_tflops": round(cs["flops"] / (cs["ms"] * 1e-3) / 1e12, 1),
            "conv_frac_of_bf16_peak": round(cs["flops"] / (cs["ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)}


def per_block_decode_leg(clip_with_callback):
    """BASELINE config 2 end to end (PER_BLOCK): the same clip with every finished block handed to the VAE decoder the way the
    streaming pipeline does (`decode_to_pixel(block, use_cache=True, chunk_size=1)` from the block callback).  Measured AFTER the
    timed region; `value` keeps SURVEY §8d's definition (generator calls only)."""
    import time
    from inferix_amd.vae import HipWanVAEWrapper, synthetic_decoder_state_dict
    vae = HipWanVAEWrapper(synthetic_decoder_state_dict(seed=0))
    frames = []

    def cb(block_latent, block_index):
        frames.append(vae.decode_to_pixel(block_latent, use_cache=True, chunk_size=1))

    # (decoding block b on a second HIP stream while block b + 1 is denoised was measured too: 1494.8 vs 1498.9 ms per clip — both
    #  workloads fill the chip, the streams serialise)
    clip_with_callback(cb)
    torch.cuda.synchronize()
    frames.clear()
    t0 = time.perf_counter()
    clip_with_callback(cb)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    n_video = sum(int(f.shape[1]) for f in frames)
    del vae
    torch.cuda.empty_cache()
    return {"workload": "config 2 end to end: 7 blocks x (4 denoise + 1 context forwards + per-block VAE decode to 480x832 pixels)",
            "ms_per_clip": round(ms, 1), "latent_frames_per_s": round(FRAMES / ms * 1e3, 3), "video_frames": n_video,
            "video_frames_per_s": round(n_video / ms * 1e3, 1)}


def text_encoder_leg():
    """Time-to-first-block component of a prompt switch, measured AFTER the timed region and not part of
    `value`: umT5-XXL encoder (24 layers, dim 4096, synthetic weights generated on the device) on one 512-token prompt."""
    import time
    from inferix_amd.t5 import HipWanTextEncoder, synthetic_t5_state_dict
    enc = HipWanTextEncoder(synthetic_t5_state_dict(device="cuda"), None)
    ids = torch.randint(1, 256384, (1, 512), generator=torch.Generator().manual_seed(0))
    mask = torch.zeros(1, 512, dtype=torch.long)
    mask[:, :60] = 1
    enc.encode_ids(ids, mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = enc.encode_ids(ids, mask)["prompt_embeds"]
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    flops = 24 * (2.0 * 512 * 4096 * (4 * 4096 + 2 * 10240) + 2.0 * 512 * 10240 * 4096 + 4.0 * 64 * 512 * 512 * 64)
    del enc
    torch.cuda.empty_cache()
    return {"workload": "umT5-XXL encoder (24 layers, dim 4096, 64 heads, ffn 10240), one prompt padded to 512 tokens, bf16, "
                        "synthetic weights; tokenizer not included", "ms_per_prompt": round(ms, 2),
            "tflops": round(flops / ms / 1e9, 1), "output": list(out.shape)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers", type=int, default=0, help="debug: fewer layers (makes the result INVALID)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quant", choices=["none", "fp8", "int8"], default="none",
                    help="BASELINE config 4: dynamic per-token x per-channel 8-bit linears (not the headline dtype)")
    ap.add_argument("--kernel-breakdown", action="store_true", help="extra untimed clip with every kernel timed")
    ap.add_argument("--no-decode-leg", action="store_true", help="skip the VAE decode / text encoder measurements after the timed region")
    ap.add_argument("--emulate-sp", type=int, default=0, metavar="P",
                    help="debug: time ONE rank of a P-way sequence-parallel run on one GPU, the K/V all-gather replaced "
                         "by a device copy (makes the result INVALID)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    # IFX_BENCH_SHARE_GPU=1 (test only): all ranks on cuda:0 over gloo, to exercise the N>1 code path on a 1-GPU box
    share = os.environ.get("IFX_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    pc = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)     # "nccl" is RCCL on ROCm
        from inferix_amd.sequence_parallel import attach_sequence_parallel
        from inferix_amd.wan import ParallelConfig
        pc = ParallelConfig(rank=rank, world_size=world, local_rank=local_rank)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    if a.emulate_sp > 1:
        assert world == 1, "--emulate-sp is a single-process debugging mode"
        from inferix_amd.wan import ParallelConfig
        pc = ParallelConfig(rank=0, world_size=a.emulate_sp, local_rank=0)

    from inferix_amd import hip_ops as ops
    from inferix_amd.core import DecodeMode
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    model, gen, pipe = build_pipeline(device, pc, a.layers or None)
    if world > 1:
        attach_sequence_parallel(model, dist.group.WORLD)
    elif a.emulate_sp > 1:
        from inferix_amd.sequence_parallel import LoopbackExchange, attach_sequence_parallel
        attach_sequence_parallel(model, exchange=LoopbackExchange(a.emulate_sp, 0))
    if a.quant != "none":
        from inferix_amd import quant as Qz
        qc = (Qz.get_dynamic_fp8_per_token_act_per_channel_weight_qconfig() if a.quant == "fp8"
              else Qz.get_dynamic_int8_per_token_act_per_channel_weight_qconfig())
        Qz.quantize_dynamic(gen, {"": qc, "text_embedding": None, "proj_out": None, "head": None})
    g = torch.Generator().manual_seed(0)
    noise = torch.randn(1, FRAMES, *LATENT, generator=g).to(torch.bfloat16).to(device)
    kvm = KVCacheManager(device)
    reqs = [KVCacheRequest("bench")]
    fwd_ms = []                    # per generator forward, filled only when asked

    def clip():
        return pipe.inference(noise=noise, text_prompts=["synthetic"], kv_cache_manager=kvm, kv_cache_requests=reqs,
                              decode_mode=DecodeMode.NO_DECODE, free_cache_before_vae=False)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        clip()
    timer = ops.KernelTimer(names=("attn_self",))
    ops.set_kernel_timer(timer)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = clip()
    sync()
    dt = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(out.float()).all()

    ks = timer.summary().get("attn_self", dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
    attn_tflops = ks["flops"] / (ks["ms"] * 1e-3) / 1e12 if ks["ms"] else 0.0
    forwards = a.steps * (FRAMES // BLOCK) * (len(STEPS_LIST) + 1)

    # per-forward latency by block index (one extra untimed clip, events around each generator call)
    per_block_ms = []
    evs = []
    orig = gen.forward

    def timed_forward(**kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig(**kw)
        e.record()
        evs.append((kw["current_start"] // (BLOCK * 1560), s, e))
        return r
    gen.forward = timed_forward
    breakdown = None
    if a.kernel_breakdown:
        allt = ops.KernelTimer(names=("attn_self", "attn_cross", "gemm", "gemm_q8", "quant_per_token", "layernorm",
                                      "rmsnorm_rope_append"))
        ops.set_kernel_timer(allt)
    clip()
    torch.cuda.synchronize()
    if a.kernel_breakdown:
        ops.set_kernel_timer(None)
        breakdown = {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                         "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] and v["flops"] else None,
                         "gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] else None}
                     for k, v in allt.summary().items()}
    gen.forward = orig
    nblk = FRAMES // BLOCK
    for b in range(nblk):
        ms = [s.elapsed_time(e) for (bi, s, e) in evs if bi == b]
        per_block_ms.append(round(sum(ms) / len(ms), 3))
    denoise_ms = [s.elapsed_time(e) for i, (bi, s, e) in enumerate(evs) if i % (len(STEPS_LIST) + 1) != len(STEPS_LIST)]

    if rank == 0:
        res = {
            "metric": "latent_frames_per_sec",
            "value": round(FRAMES * a.steps / dt, 4),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "bf16" if a.quant == "none" else f"{a.quant} linears (per-token x per-channel) + bf16 attention",
            "data": "synthetic",
            "config": {"workload": "Self-Forcing 480p bf16 (Wan2.1-T2V-1.3B causal DiT, 30 layers), block_size=3, "
                                   "21 latent frames = 7 blocks x (4 denoise + 1 context) generator forwards, paged KV "
                                   "prefix 4680..32760 keys; `value` counts the generator calls (SURVEY 8d), the per-block VAE decode of config 2 "
                                   "is measured beside it (`per_block_decode`, `vae_decode`), the text encoder in `text_encoder`",
                       "batch": 1, "latent": [FRAMES, *LATENT], "denoising_step_list": STEPS_LIST,
                       "timestep_shift": 5.0, "parallelism": f"sp{world}" if world > 1 else "single",
                       "layers": model.num_layers},
            "ms_per_denoise_step": round(sum(denoise_ms) / len(denoise_ms), 3),
            "ms_per_forward_by_block": per_block_ms,
            "generator_forwards_timed": forwards,
            "roofline": {"kernel": "ifx::attn_fwd_pp_kernel (block-causal paged flash attention, self-attention launches)", "bound": "mfma",
                         "achieved": round(attn_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(attn_tflops / PEAK_BF16_TFLOPS, 4), "traffic": None,
                         "launches": ks["launches"], "avg_launch_ms": round(ks["ms"] / max(ks["launches"], 1), 4),
                         "algorithmic_flops_per_launch": "4*N*L*d, N=4680/n_gpus, d=1536, L=(b+1)*4680",
                         "algorithmic_gbps": round(ks["bytes"] / (ks["ms"] * 1e-3) / 1e9, 1) if ks["ms"] else None},
        }
        if a.layers:
            res["config"]["INVALID"] = "debug run with fewer layers"
        if a.emulate_sp > 1:
            res["config"]["INVALID"] = f"one rank of sp{a.emulate_sp} emulated on one GPU, collective replaced by a copy"
        res["roofline"].update(pmc_traffic(world if world > 1 else max(a.emulate_sp, 1)))
        if breakdown:
            res["kernel_breakdown"] = breakdown
        if world == 1 and not a.no_decode_leg and a.emulate_sp <= 1:
            res["per_block_decode"] = per_block_decode_leg(lambda cb: pipe.inference(
                noise=noise, text_prompts=["synthetic"], kv_cache_manager=kvm, kv_cache_requests=reqs,
                decode_mode=DecodeMode.NO_DECODE, free_cache_before_vae=False, block_callback=cb))
            res["vae_decode"] = vae_decode_leg()
        if world == 1 and not a.no_decode_leg and a.emulate_sp <= 1:
            res["text_encoder"] = text_encoder_leg()
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
