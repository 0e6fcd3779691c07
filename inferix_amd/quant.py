"""Dynamic 8-bit linears: the `quantize_dynamic(module, qconfig_dict)` surface the reference's quantization
examples use (example/quantization/run_self_forcing_quantized.py:19-23,47-64; DAX, un-vendored), backed by
`ifx_quant_per_token` + `ifx_gemm_q8` (fp8 e4m3 / int8 MFMA).  Scheme: include/inferix_hip.h.

    from inferix_amd.quant import quantize_dynamic, get_dynamic_fp8_per_token_act_per_channel_weight_qconfig
    quantize_dynamic(pipeline.generator.model,
                     {"": get_dynamic_fp8_per_token_act_per_channel_weight_qconfig(),
                      "text_embedding": None, "proj_out": None, "head": None})

The empty key is the default for every linear; named prefixes override it (None = keep bf16), as in DAX's dict.
Weights are quantised once here (per output channel); activations per token at every call on the device.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _hip


@dataclass(frozen=True)
class QConfig:
    fmt: int
    name: str

    @property
    def qmax(self) -> float:
        return 448.0 if self.fmt == _hip.IFX_Q_FP8_E4M3 else 127.0


def get_dynamic_fp8_per_token_act_per_channel_weight_qconfig() -> QConfig:
    return QConfig(_hip.IFX_Q_FP8_E4M3, "dynamic_fp8_e4m3_per_token_act_per_channel_weight")


def get_dynamic_int8_per_token_act_per_channel_weight_qconfig() -> QConfig:
    return QConfig(_hip.IFX_Q_INT8, "dynamic_int8_per_token_act_per_channel_weight")


def quantize_weight(w: torch.Tensor, qc: QConfig) -> Tuple[torch.Tensor, torch.Tensor]:
    """[N, K] bf16 -> (bytes [N, K] uint8, scale [N] fp32): per OUTPUT channel abs-max / QMAX.  The rows of an
    nn.Linear weight are its output channels, so this is the activation quantiser (`ifx_quant_per_token`) run once
    on the weight matrix: one rule, one kernel, bit-identical to the oracle on both operands."""
    from . import hip_ops as ops
    return ops.quant_per_token(w.contiguous(), qc.fmt)


def _config_for(name: str, qconfig_dict: Dict[str, Optional[QConfig]]) -> Optional[QConfig]:
    best, best_len = qconfig_dict.get("", None), -1
    for k, v in qconfig_dict.items():
        if k and (name == k or name.startswith(k + ".") or ("." + k + ".") in ("." + name + ".")) and len(k) > best_len:
            best, best_len = v, len(k)
    return best


# (reference module name, key in the packed per-block weight dict)
_BLOCK_LINEARS = (("self_attn.qkv", "qkv"), ("self_attn.o", "o"), ("cross_attn.q", "cq"), ("cross_attn.k", "ck"),
                  ("cross_attn.v", "cv"), ("cross_attn.o", "co"), ("ffn.0", "f0"), ("ffn.2", "f2"))


def quantize_dynamic(module, qconfig_dict: Dict[str, Optional[QConfig]]):
    """Quantise, in place, the linears of a HipCausalWanModel (or a wrapper / pipeline holding one).  Linears whose
    K is not a multiple of 128 stay bf16 (kernel constraint), as do the O(frames)-row timestep MLPs that run as
    PyTorch glue; `text_embedding`, `proj_out`, `head` are excluded by the caller's dict as upstream."""
    model = module
    for attr in ("generator", "model"):
        while hasattr(model, attr) and not hasattr(model, "blocks"):
            model = getattr(model, attr)
    if not hasattr(model, "blocks") or not hasattr(model, "mod_all"):
        raise TypeError("quantize_dynamic expects a HipCausalWanModel (or a wrapper/pipeline around one)")
    if model.mod_all is None:
        raise RuntimeError("load the bf16 weights before quantising")
    n = 0
    for i, blk in enumerate(model.blocks):
        for ref_name, key in _BLOCK_LINEARS:
            names = ([f"blocks.{i}.self_attn.{c}" for c in "qkv"] if key == "qkv" else [f"blocks.{i}.{ref_name}"])
            qcs = [_config_for(nm, qconfig_dict) for nm in names]
            qc = qcs[0]
            if qc is None or any(q != qc for q in qcs):
                continue
            w = blk.w[key + "_w"]
            if w.shape[1] % 128:
                continue
            blk.w[key + "_q"], blk.w[key + "_s"] = quantize_weight(w, qc)
            blk.w[key + "_fmt"] = qc.fmt
            n += 1
    model.quantized_linears = n
    return module
