"""Per-block VAE decode on one MI355X at the Self-Forcing 480p geometry: latent blocks of 3 frames, 60 x 104, Wan2.1 VAE
decoder (dim 96), synthetic weights.  Prints ms per block, output video frames/s and the conv TFLOP/s.

    python tools/bench_vae.py [--blocks 7] [--frames-per-call 3] [--detail]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=7)
    ap.add_argument("--frames-per-call", type=int, default=3)
    ap.add_argument("--h", type=int, default=60)
    ap.add_argument("--w", type=int, default=104)
    ap.add_argument("--dim", type=int, default=96)
    ap.add_argument("--detail", action="store_true", help="per-op event timing (adds launch gaps)")
    a = ap.parse_args()
    from inferix_amd import hip_ops as ops
    from inferix_amd.vae import HipWanVAEWrapper, synthetic_decoder_state_dict
    W = synthetic_decoder_state_dict(dim=a.dim, seed=1)
    vae = HipWanVAEWrapper(W, dim=a.dim, max_frames_per_call=a.frames_per_call)
    g = torch.Generator().manual_seed(0)
    latent = torch.randn(1, 3 * a.blocks, 16, a.h, a.w, generator=g).to(torch.bfloat16).cuda()

    def run_clip(timer=None):
        vae.model.clear_cache()
        ops.set_kernel_timer(timer)
        times = []
        for b in range(a.blocks):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            z = latent[:, 3 * b:3 * b + 3].permute(0, 2, 1, 3, 4)
            out = vae.model.cached_decode(z)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        ops.set_kernel_timer(None)
        return times, out

    run_clip()                                            # warm-up (allocations, first-launch costs)
    times, out = run_clip()
    frames_out = 1 + 4 * (3 * a.blocks - 1)
    res = {"workload": f"Wan2.1 VAE decode, {a.blocks} blocks x 3 latent frames {a.h}x{a.w} -> {out.shape[-2]}x{out.shape[-1]} px",
           "ms_per_block": [round(t, 2) for t in times], "ms_per_clip": round(sum(times), 1),
           "video_frames_per_s": round(frames_out / sum(times) * 1e3, 1),
           "latent_frames_per_s": round(3 * a.blocks / sum(times) * 1e3, 2), "frames_per_call": a.frames_per_call}
    if a.detail:
        # per-shape conv timing (events around every launch)
        recs = []
        orig = ops.conv3d_cl

        def timed_conv(x, in_slots, w, bias, *, kt, ks, y, out_slots, upsample=False, residual=None):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            r = orig(x, in_slots, w, bias, kt=kt, ks=ks, y=y, out_slots=out_slots, upsample=upsample, residual=residual)
            e_.record()
            ho, wo = y.shape[1], y.shape[2]
            cin = x.shape[1] * 32 if x.dim() == 5 else x.shape[3]            # planar frame rings: [slots, cin/32, h, w, 32]
            key = f"k{kt}x{ks}x{ks}{'u' if upsample else ''} {cin}->{w.shape[2]} @{ho}x{wo} t{len(out_slots)}"
            recs.append((key, s_, e_, 2.0 * len(out_slots) * ho * wo * w.shape[2] * cin * w.shape[0]))
            return r

        import inferix_amd.vae as vmod
        vmod.ops.conv3d_cl = timed_conv
        run_clip()
        vmod.ops.conv3d_cl = orig
        torch.cuda.synchronize()
        agg = {}
        for key, s_, e_, fl in recs:
            d = agg.setdefault(key, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += s_.elapsed_time(e_)
            d[2] += fl
        print(f"{'conv shape':44s} {'calls':>5s} {'ms':>8s} {'us/call':>8s} {'TFLOP/s':>8s}", file=sys.stderr)
        for key, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"{key:44s} {d[0]:5d} {d[1]:8.2f} {d[1] / d[0] * 1e3:8.1f} {d[2] / d[1] / 1e9:8.1f}", file=sys.stderr)
        t = ops.KernelTimer(names=("conv3d", "rmsnorm_cl"))
        run_clip(t)
        s = t.summary()
        for k, d in s.items():
            res[k] = {"launches": d["launches"], "ms": round(d["ms"], 1),
                      "TFLOP/s": round(d["flops"] / d["ms"] / 1e9, 1) if d["flops"] else None,
                      "GB/s": round(d["bytes"] / d["ms"] / 1e6, 1) if d["bytes"] else None}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
