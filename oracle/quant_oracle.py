"""TEST INFRASTRUCTURE — CPU restatement of the dynamic 8-bit linear scheme (include/inferix_hip.h,
`ifx_quant_per_token` / `ifx_gemm_q8`).

The reference delegates this arithmetic to DAX (github.com/RiseAI-Sys/DAX, cloned unpinned by
example/quantization/README.md:19-24; call sites example/quantization/run_self_forcing_quantized.py:19-23,47-64),
which is NOT in /root/reference and not installable here: **parity with DAX is unpinned**.  What is pinned is the
conventional scheme both sides of THIS repository implement (per-token activation scale = row abs-max / QMAX,
per-channel weight scale, QMAX 448 for OCP e4m3fn / 127 for int8, round-to-nearest-even, fp32 (fp8) or exact
int32 (int8) accumulation, dequantisation by the outer product of scales, bias added before the bf16 rounding),
and the quantise-clamp-cast step agrees with the one quantisation routine that IS in the reference tree, MAGI's
`div_clamp_to` (inferix/models/magi/dit/dit_module.py:367-387: x / scale, clamp +-448, cast to e4m3fn).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

FP8, INT8 = 0, 1


def qmax(fmt: int) -> float:
    return 448.0 if fmt == FP8 else 127.0


def quantize_rows(x: torch.Tensor, fmt: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """[rows, K] -> (dequantisable values as float32 [rows, K] holding exactly the 8-bit grid points, scale [rows])."""
    xf = x.float()
    amax = xf.abs().amax(dim=-1)
    s = torch.where(amax > 0, amax / qmax(fmt), torch.ones_like(amax))
    v = (xf / s[:, None]).clamp(-qmax(fmt), qmax(fmt))
    q = v.to(torch.float8_e4m3fn).float() if fmt == FP8 else torch.round(v)
    return q, s


def quantized_bytes(x: torch.Tensor, fmt: int) -> torch.Tensor:
    """The byte image the kernel must produce (bit-exact check of ifx_quant_per_token)."""
    q, _ = quantize_rows(x, fmt)
    if fmt == FP8:
        return q.to(torch.float8_e4m3fn).view(torch.uint8)
    return q.to(torch.int8).view(torch.uint8)


def linear_q8(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], fmt: int) -> torch.Tensor:
    """bf16( (xq @ wq^T) * (s_a ⊗ s_w) + bias ), accumulation in float64 (exact for int8, reference for fp8)."""
    xq, sa = quantize_rows(x, fmt)
    wq, sw = quantize_rows(w, fmt)
    acc = (xq.double() @ wq.double().t()).float()
    y = acc * (sa[:, None] * sw[None, :])
    if bias is not None:
        y = y + bias.float()
    return y.to(torch.bfloat16)


# --------------------------------------------------------------------------
# the quantised MODEL: which linears, under the reference's exclusion dict
# --------------------------------------------------------------------------
def reference_qconfig_dict(fmt: int) -> dict:
    """The dict of example/quantization/run_self_forcing_quantized.py:57-62: every nn.Linear of the transformer is quantised
    (the empty key) except the modules named `text_embedding`, `proj_out` (no such module in the Wan DiT) and `head`."""
    return {"": fmt, "text_embedding": None, "proj_out": None, "head": None}


def config_for(name: str, qconfig_dict: dict):
    """Module-name resolution of a `qconfig_dict`: the longest key that is the module's name or one of its ancestors wins, the
    empty key is the default (torch.ao.quantization's propagate rule, which DAX's `quantize_dynamic(module, qconfig_dict)`
    signature follows; DAX itself is not in the tree — parity unpinned)."""
    best, best_len = qconfig_dict.get("", None), -1
    for k, v in qconfig_dict.items():
        if k and (name == k or name.startswith(k + ".")) and len(k) > best_len:
            best, best_len = v, len(k)
    return best


def model_hook(qconfig_dict: dict, log: Optional[list] = None):
    """`wan_oracle.linear_override` callback = the model after `quantize_dynamic(transformer, qconfig_dict)`: a Linear whose
    resolved config is a format runs `linear_q8` on its 2-D rows, one whose config is None stays `F.linear`.
    Quantised with the reference's dict: blocks.*.self_attn.{q,k,v,o}, blocks.*.cross_attn.{q,k,v,o}, blocks.*.ffn.{0,2},
    time_embedding.{0,2}, time_projection.1 (all nn.Linear: models/self_forcing/causal_model.py:125-128,378-379,615-618); kept: text_embedding.{0,2}, head.head;
    patch_embedding is a Conv3d and is not a Linear."""
    def fn(x, weight, bias, name):
        fmt = config_for(name, qconfig_dict)
        if log is not None:
            log.append((name, fmt))
        if fmt is None:
            return None
        y = linear_q8(x.reshape(-1, x.shape[-1]), weight, bias, fmt)
        return y.view(*x.shape[:-1], weight.shape[0]).to(x.dtype)
    return fn
