#!/usr/bin/env python3
"""A/B of the two `ifx_conv3d_cl` kernels on the decoder's dominant shapes: the persistent ping-pong kernel (conv_variant 0) against
the lock-step kernel of round 1 (conv_variant 1) — time per launch by events over `reps` launches, TFLOP/s, and BIT-IDENTITY of the two
outputs (same K order, same epilogue rounding: any difference is a bug).

    python tools/bench_conv.py [--reps 10] [--shapes main|all] [--variants 0,1]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from inferix_amd import hip_ops as ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--shapes", default="main")
ap.add_argument("--variants", default="0,1")
ap.add_argument("--cl", action="store_true", help="channels-last input frames (default: the planar layout of the frame rings)")
a = ap.parse_args()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
# (cin, cout, h, w, t, kt, upsample, residual)
SHAPES = [(96, 96, 480, 832, 12, 3, 0, 1), (192, 192, 240, 416, 12, 3, 0, 1), (384, 384, 120, 208, 6, 3, 0, 1),
          (384, 384, 60, 104, 3, 3, 0, 1), (192, 96, 240, 416, 12, 1, 1, 0), (96, 3, 480, 832, 12, 3, 0, 0)]
if a.shapes == "all":
    SHAPES += [(384, 192, 120, 208, 12, 1, 1, 0), (192, 384, 120, 208, 6, 3, 0, 0), (96, 96, 480, 832, 1, 3, 0, 1), (384, 384, 60, 104, 1, 3, 0, 1),
               (32, 384, 60, 104, 3, 3, 0, 0), (96, 96, 50, 70, 2, 3, 0, 1), (192, 96, 25, 35, 2, 1, 1, 0)]
variants = [int(v) for v in a.variants.split(",")]
print(f"{'shape':44s}" + "".join(f"  v{v}: us/launch  TFLOP/s" for v in variants) + "   bits")
for cin, cout, h, w_, t, kt, ups, has_res in SHAPES:
    ho, wo = (2 * h, 2 * w_) if ups else (h, w_)
    ring = rnd(t + kt - 1, h, w_, cin)
    ring_cl = ring
    if not a.cl and not ups:
        ring = ops.to_planar(ring)                        # as the decoder's frame rings hand the frames over (not the upsample convs' inputs)
    wt = (rnd(kt * 9, cin // 32, cout, 32) * (kt * 9 * cin) ** -0.5).contiguous()
    b = rnd(cout)
    res = rnd(t, ho, wo, cout) if has_res else None
    slots = list(range(t + kt - 1))
    if kt == 3:
        slots[0] = -1                                     # a zero frame in front of the stream, as in the first chunk
    outs, line = [], f"k{kt}x3x3{'u' if ups else ' '} {cin:3d}->{cout:3d} @{ho}x{wo} t{t}".ljust(44)
    flops = 2.0 * t * ho * wo * cout * cin * kt * 9
    for v in variants:
        ops.set_option("conv_variant", v)
        y = torch.full((t, ho, wo, cout), float("nan"), dtype=torch.bfloat16, device=dev)
        run = lambda: ops.conv3d_cl(ring, slots, wt, b, kt=kt, ks=3, y=y, out_slots=list(range(t)), upsample=bool(ups), residual=res)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        line += f"  {us:12.1f} {flops / us * 1e-6:8.1f}"
        outs.append(y)
    if ring is not ring_cl:                               # the channels-last form of the same launch: the same bits
        ycl = torch.empty_like(outs[0])
        ops.conv3d_cl(ring_cl, slots, wt, b, kt=kt, ks=3, y=ycl, out_slots=list(range(t)), upsample=bool(ups), residual=res)
        outs.append(ycl)
    same = all(torch.equal(outs[0].view(torch.int16), o.view(torch.int16)) for o in outs[1:])
    nan = any(bool(torch.isnan(o.float()).any()) for o in outs)
    print(line + ("   identical" if same else "   DIFFERENT") + ("  NaN!" if nan else ""), flush=True)
    del ring, ring_cl, res, outs, y
ops.set_option("conv_variant", 0)
