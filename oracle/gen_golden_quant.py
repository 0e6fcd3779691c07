#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/quant_fp8.npz from the ONE quantisation routine that is in the reference
tree: MAGI's static-scale FP8 linears (inferix/models/magi/dit/dit_module.py:367-490).

  div_clamp_to(x, scale)                  :367-387   x / scale (fp32) -> clamp +-448 -> bf16 -> e4m3fn  (reference code, as is)
  PerTensorQuantizedFp8Linear.forward     :448-462   divisor = input_scale [in]; bmm_fp8(xq, W^T, input_scale, weight_scale)
  PerChannelQuantizedFp8Linear.forward    :480-490   divisor = smooth_scale [1, in]

`bmm_fp8` is flashinfer (un-vendored): the stand-in of oracle/_refstub.install_magi restates its published definition
(fp32 accumulate x A_scale x B_scale -> bf16), so the LINEAR outputs are pinned to that restatement, while the
quantise-clamp-cast bytes are the reference's own arithmetic.  The DAX dynamic scheme (a15) stays "parity unpinned"; what
this fixture adds is that the cast step of the HIP quantisers is checked against reference-produced bytes.

usage (build container only):  python oracle/gen_golden_quant.py
"""
from __future__ import annotations

import os
import sys
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refstub  # noqa: E402
import magi_block_oracle as MB  # noqa: E402
from fixture_io import GOLDEN_DIR, save_npz  # noqa: E402

BF = torch.bfloat16


def tie_rows(K: int) -> torch.Tensor:
    """Values whose quotient by a power-of-two scale sits on / next to e4m3 rounding boundaries, including the cases where
    the bf16 intermediate decides the direction (a value just above an e4m3 tie rounds to the tie in bf16, and then to even)."""
    grid = torch.arange(0, 256, dtype=torch.uint8).view(torch.float8_e4m3fn).float()
    grid = grid[torch.isfinite(grid) & (grid >= 0)].sort().values          # non-negative e4m3 values
    mids = (grid[:-1] + grid[1:]) / 2                                       # exact ties between neighbours
    vals = torch.cat([grid, mids, mids * (1 + 2.0 ** -9), mids * (1 - 2.0 ** -9), mids * (1 + 2.0 ** -7), mids * (1 - 2.0 ** -7),
                      torch.tensor([448.0, 449.0, 464.0, 480.0, 1000.0, 2.0 ** -10, 2.0 ** -9 * 1.5])])
    vals = torch.cat([vals, -vals])
    reps = (K + vals.numel() - 1) // vals.numel()
    return vals.repeat(reps)[:K]


def main():
    if not _refstub.available():
        raise SystemExit("reference tree not available: fixtures can only be generated in the build container")
    warnings.filterwarnings("ignore")
    dm = _refstub.import_magi_dit()
    g = torch.Generator().manual_seed(2024)
    rows, K, N = 96, 1024, 384
    x = (torch.randn(rows, K, generator=g) * 2.0).to(BF)
    x[0] = 0
    x[1, 5] = 3000.0                                                       # saturates
    x[2] = (tie_rows(K) * 0.25).to(BF)                                     # scale 0.25 below: quotients on the e4m3 ties
    x[3] = (tie_rows(K).flip(0) * 0.25).to(BF)
    fx = {"x": x}
    # ---- per-channel divisors (what both MAGI linears pass to div_clamp_to) and a scalar divisor
    div_vec = (torch.rand(K, generator=g) * 0.05 + 0.005).float()
    div_vec[: K // 2] = 0.25
    div_one = torch.tensor([0.25], dtype=torch.float32)
    for name, d in (("vec", div_vec), ("one", div_one)):
        q = dm.div_clamp_to(x, d)
        assert q.dtype == torch.float8_e4m3fn
        fx[f"div_{name}"] = d
        fx[f"q_{name}"] = q.view(torch.uint8)
        assert torch.equal(q.view(torch.uint8), MB.div_clamp_to(x, d).view(torch.uint8))
    # the bf16 intermediate matters: count the bytes a direct fp32 -> e4m3 cast would get differently (documented in the test)
    direct = torch.clamp(x.float() / div_vec, -448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    fx["double_rounding_diffs"] = torch.tensor(int((direct != fx["q_vec"]).sum()))
    # ---- the two linears
    w = torch.randn(N, K, generator=g) / 32.0
    w_scale = (w.abs().max() / 448.0).reshape(1)
    wq = torch.clamp(w / w_scale, -448, 448).to(torch.float8_e4m3fn)
    in_scale = torch.tensor([0.03], dtype=torch.float32)
    pt = dm.PerTensorQuantizedFp8Linear(K, N)
    pt.weight.data = wq.reshape(1, N, K).clone()
    pt.weight_scale.data = w_scale.clone()
    pt.input_scale.data = in_scale.expand(K).clone()                       # [in_features], every entry the tensor's scale
    pc = dm.PerChannelQuantizedFp8Linear(K, N)
    pc.weight.data = wq.reshape(1, N, K).clone()
    pc.weight_scale.data = w_scale.clone()
    pc.input_scale.data = in_scale.clone()
    pc.smooth_scale.data = div_vec.reshape(1, K).clone()
    xin = x.reshape(4, 24, K)                                              # prefix dims as in [sq, b, h]
    with torch.no_grad():
        y_pt, y_pc = pt(xin), pc(xin)
    fx.update(wq=wq.view(torch.uint8), w_scale=w_scale, in_scale=in_scale, y_per_tensor=y_pt, y_per_channel=y_pc)
    assert torch.equal(y_pt, MB.fp8_static_linear(xin, wq, w_scale, in_scale.expand(K), in_scale.expand(K)))
    assert torch.equal(y_pc, MB.fp8_static_linear(xin, wq, w_scale, in_scale, div_vec.reshape(1, K)))
    path = os.path.join(GOLDEN_DIR, "quant_fp8.npz")
    save_npz(path, fx)
    print(f"wrote {path} ({os.path.getsize(path)} bytes); {int(fx['double_rounding_diffs'])} bytes depend on the bf16 "
          "intermediate; oracle == reference (bit-exact)")


if __name__ == "__main__":
    main()
