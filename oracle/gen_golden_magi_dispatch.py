#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/magi_dispatch_tiny.npz: `VideoDiTModel.forward_dispatcher` of the REFERENCE
(inferix/models/magi/dit/dit_model.py:399-596) run on CPU at tiny dimensions, both guidance modes:
  cfg_number = 3   forward_3cfg (text + previous chunks without a cache write / previous chunks with the null caption and the cache write /
                   the denoising chunks as batch rows without a cache) and the per-chunk scale mix; three calls that walk the cache
                   (chunks 0, 1; the clean copy of chunk 0 in front of chunk 1: fwd_extra_1st_chunk; chunks 1, 2 behind the cached chunk 0)
  cfg_number = 1   the distilled dispatch with the nearly-clean re-forward (one call on a fresh cache)
The model is the one gen_golden_magi_model.py builds (same weights, same fp32 cast-up around the embedders).  Every `self.forward` the
dispatcher makes is recorded, so the test can hold each component to the bf16 floor and the mix to the reference's arithmetic.
Two things of the reference cannot run without a GPU and are stood in for here: `Tensor.cuda()` on the three small scale tensors (identity)
and `generate_kv_range_for_uncondition`'s device string (the same integers on the CPU).

usage (build container only; /root/reference must exist):  python oracle/gen_golden_magi_dispatch.py
"""
from __future__ import annotations

import os
import sys
import warnings
from types import SimpleNamespace

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refstub  # noqa: E402
import gen_golden_magi_model as G  # noqa: E402
import magi_model_oracle as MM  # noqa: E402
from fixture_io import GOLDEN_DIR, save_npz  # noqa: E402

SCALES = dict(cfg_t_range=[0.0, 0.0217, 0.1, 0.3, 0.999], prev_chunk_scales=[1.5, 1.5, 1.5, 1.0, 1.0], text_scales=[7.5, 7.5, 7.5, 0.0, 0.0])


def dispatch_calls(cfg, clip, seed):
    g = torch.Generator().manual_seed(seed)
    Hl, Wl, Lc, Cc = 8, 12, cfg.caption_max_length, cfg.caption_channels

    def mk(dn, t, kv, caps, **kw):
        x0 = torch.randn(1, cfg.in_channels, dn, Hl, Wl, generator=g)                 # chunk_width = 1 latent frame
        y = torch.randn(2 * dn, 1, Lc, Cc, generator=g)
        mask = torch.zeros(2 * dn, 1, Lc)
        for r, c in enumerate(caps + (2,) * dn):                                        # conditional rows, then the null-caption rows
            mask[r, :, :c] = 1
        return dict(x=torch.cat([x0, x0], 0), t=torch.tensor([t, t]), y=y, mask=mask, kv_range=torch.tensor(kv, dtype=torch.int32),
                    kw=dict(kw, chunk_width=1, num_steps=8, distill_interval=1))
    # the cache rule caches nothing at slice_point 0 without the extra clean chunk (magi_kv_cache_manager.py:100-118): the sequence is
    # chunks 0, 1 from scratch / the clean copy of chunk 0 in front of chunk 1 (writes the cache) / chunks 1, 2 behind the cached chunk 0
    three = [mk(2, [0.05, 0.01], [(0, clip), (0, 2 * clip)], (7, 5), range_num=2, denoising_range_num=2, slice_point=0, fwd_extra_1st_chunk=False),
             mk(2, [0.9999, 0.4], [(0, clip), (0, 2 * clip)], (12, 3), range_num=2, denoising_range_num=2, slice_point=0, fwd_extra_1st_chunk=True),
             mk(2, [0.7, 0.15], [(0, 2 * clip), (0, 3 * clip)], (6, 11), range_num=3, denoising_range_num=2, slice_point=1, fwd_extra_1st_chunk=False)]
    one = [mk(2, [0.6, 0.2], [(0, clip), (0, 2 * clip)], (9, 4), range_num=2, denoising_range_num=2, slice_point=0, fwd_extra_1st_chunk=False,
              distill_nearly_clean_chunk=True)]
    return three, one


def main():
    if not _refstub.available():
        raise SystemExit("reference tree not available: fixtures can only be generated in the build container")
    warnings.filterwarnings("ignore")
    torch.manual_seed(0)
    cfg = MM.tiny_model_config()
    _, clip = G.calls(cfg, 21)
    max_tokens = 4 * clip
    model, make_ip = G.build_reference(cfg, max_tokens)
    torch.Tensor.cuda = lambda self, *a, **k: self
    model.generate_kv_range_for_uncondition = lambda ux: torch.tensor(
        [[b * clip * ux.shape[2], (b + 1) * clip * ux.shape[2]] for b in range(ux.shape[0])], dtype=torch.int32)
    recorded = []
    fwd0 = model.forward

    def fwd1(*a, **k):
        out = fwd0(*a, **k)
        recorded.append(out.clone())
        return out
    model.forward = fwd1
    three, one = dispatch_calls(cfg, clip, 33)
    fx = {"geom": torch.tensor([clip, max_tokens, len(three), len(one), G.WSEED, G.ESEED])}
    for k, v in SCALES.items():
        fx[k] = torch.tensor(v)
    for tag, cfg_number, calls in (("t", 3, three), ("o", 1, one)):
        model.runtime_config = SimpleNamespace(cfg_number=cfg_number, **SCALES)
        ip = make_ip()
        for ci, c in enumerate(calls):
            recorded.clear()
            with torch.no_grad():
                out = model.forward_dispatcher(c["x"], c["t"], c["y"], c["mask"], c["kv_range"], ip, **dict(c["kw"]))
            for k in ("x", "t", "y", "mask", "kv_range"):
                fx[f"{tag}{ci}_in_{k}"] = c[k]
            kw = c["kw"]
            fx[f"{tag}{ci}_flags"] = torch.tensor([kw["range_num"], kw["denoising_range_num"], kw["slice_point"], int(kw["fwd_extra_1st_chunk"]),
                                                  int(kw.get("distill_nearly_clean_chunk", False))])
            fx[f"{tag}{ci}_out"] = out
            fx[f"{tag}{ci}_n_forwards"] = torch.tensor(len(recorded))
            for fi, r in enumerate(recorded):
                fx[f"{tag}{ci}_fwd{fi}"] = r
            print(tag, ci, "forwards", [tuple(r.shape) for r in recorded], "out", tuple(out.shape), float(out.abs().mean()))
    path = os.path.join(GOLDEN_DIR, "magi_dispatch_tiny.npz")
    save_npz(path, fx)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
