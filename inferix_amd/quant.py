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
    act: str = "per_token"          # dynamic activation scale: one per token row | "per_tensor": one for the whole tensor
    weight: str = "per_channel"     # weight scale: one per output channel | "per_tensor"

    @property
    def qmax(self) -> float:
        return 448.0 if self.fmt == _hip.IFX_Q_FP8_E4M3 else 127.0


def get_dynamic_fp8_per_token_act_per_channel_weight_qconfig() -> QConfig:
    return QConfig(_hip.IFX_Q_FP8_E4M3, "dynamic_fp8_e4m3_per_token_act_per_channel_weight")


def get_dynamic_int8_per_token_act_per_channel_weight_qconfig() -> QConfig:
    return QConfig(_hip.IFX_Q_INT8, "dynamic_int8_per_token_act_per_channel_weight")


def get_dynamic_fp8_per_tensor_act_per_tensor_weight_qconfig() -> QConfig:
    return QConfig(_hip.IFX_Q_FP8_E4M3, "dynamic_fp8_e4m3_per_tensor_act_per_tensor_weight", "per_tensor", "per_tensor")


def get_dynamic_int8_per_tensor_act_per_tensor_weight_qconfig() -> QConfig:
    return QConfig(_hip.IFX_Q_INT8, "dynamic_int8_per_tensor_act_per_tensor_weight", "per_tensor", "per_tensor")


def get_dynamic_fp8_per_tensor_act_per_channel_weight_qconfig() -> QConfig:
    return QConfig(_hip.IFX_Q_FP8_E4M3, "dynamic_fp8_e4m3_per_tensor_act_per_channel_weight", "per_tensor", "per_channel")


def get_dynamic_int8_per_tensor_act_per_channel_weight_qconfig() -> QConfig:
    return QConfig(_hip.IFX_Q_INT8, "dynamic_int8_per_tensor_act_per_channel_weight", "per_tensor", "per_channel")


def quantize_activation(x: torch.Tensor, qc: QConfig) -> Tuple[torch.Tensor, torch.Tensor]:
    """Activation bytes + the per-row scale vector `ifx_gemm_q8` takes (a per-tensor scale is that scalar repeated)."""
    from . import hip_ops as ops
    return ops.quant_per_tensor(x, qc.fmt) if qc.act == "per_tensor" else ops.quant_per_token(x, qc.fmt)


class StaticFp8Linear:
    """MAGI's static-scale FP8 linears on `ifx_quant_static` + `ifx_gemm_q8` (fp8 MFMA):
      PerTensorQuantizedFp8Linear  (inferix/models/magi/dit/dit_module.py:434-462): divisor = input_scale `[in_features]`
      PerChannelQuantizedFp8Linear (:465-490):                                        divisor = smooth_scale `[1, in_features]`
    y = bf16( (div_clamp_to(x, divisor) @ W_fp8^T) * input_scale * weight_scale ).  `weight` holds e4m3 bytes `[out, in]`
    (the checkpoint's `(1, out, in)` float8 tensor viewed as uint8)."""

    def __init__(self, weight: torch.Tensor, weight_scale: torch.Tensor, input_scale: torch.Tensor, divisor: torch.Tensor):
        w = weight.view(torch.uint8) if weight.dtype != torch.uint8 else weight
        self.weight = w.reshape(-1, w.shape[-1]).contiguous()
        self.out_features, self.in_features = self.weight.shape
        dev = self.weight.device
        self.w_scale = weight_scale.reshape(-1)[:1].float().to(dev).expand(self.out_features).contiguous()
        self.in_scale = input_scale.reshape(-1)[:1].float().to(dev).contiguous()
        self.divisor = divisor.reshape(-1).float().to(dev).contiguous()
        assert self.divisor.numel() in (1, self.in_features)
        self._sx: Optional[torch.Tensor] = None          # per-row activation scale (one value repeated), kept between calls

    def _row_scale(self, rows: int) -> torch.Tensor:
        if self._sx is None or self._sx.shape[0] != rows:
            self._sx = self.in_scale.expand(rows).contiguous()
        return self._sx

    def __call__(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, **epilogue) -> torch.Tensor:
        """`out`: optional `[rows, out_features]` destination (any row stride: a column block of a wider buffer)."""
        from . import hip_ops as ops
        x2 = x.reshape(-1, self.in_features)
        q = ops.quant_static(x2, self.divisor, _hip.IFX_Q_FP8_E4M3, via_bf16=True)
        y = self.matmul(q, out=out, **epilogue)
        return y if out is not None else y.view(*x.shape[:-1], self.out_features)

    def matmul(self, q: torch.Tensor, out: Optional[torch.Tensor] = None, **epilogue) -> torch.Tensor:
        """The GEMM half on bytes some producer already quantised with THIS linear's divisor (`q` `[rows, in_features]` uint8, any
        row stride): `ops.layernorm_quant_static` for the linears behind a LayerNorm, `matmul_quant_out` of the linear in front."""
        from . import hip_ops as ops
        return ops.linear_q8(q, self._row_scale(q.shape[0]), self.weight, self.w_scale, None, _hip.IFX_Q_FP8_E4M3, out=out, **epilogue)

    def matmul_quant_out(self, q: torch.Tensor, nxt: "StaticFp8Linear", epilogue: int) -> torch.Tensor:
        """GEMM + GELU whose result is quantised for `nxt` in the epilogue: bytes `[rows, out_features]`, = nxt's own quantiser
        applied to `self.matmul(q, epilogue=...)`, bit for bit."""
        from . import hip_ops as ops
        assert nxt.in_features == self.out_features and nxt.divisor.numel() == self.out_features
        return ops.linear_q8_quant_out(q, self._row_scale(q.shape[0]), self.weight, self.w_scale, _hip.IFX_Q_FP8_E4M3, nxt.divisor,
                                       epilogue=epilogue, via_bf16=True)


def quantize_weight(w: torch.Tensor, qc: QConfig) -> Tuple[torch.Tensor, torch.Tensor]:
    """[N, K] bf16 -> (bytes [N, K] uint8, scale [N] fp32): per OUTPUT channel abs-max / QMAX.  The rows of an
    nn.Linear weight are its output channels, so this is the activation quantiser (`ifx_quant_per_token`) run once
    on the weight matrix: one rule, one kernel, bit-identical to the oracle on both operands."""
    from . import hip_ops as ops
    if qc.weight == "per_tensor":
        return ops.quant_per_tensor(w.contiguous(), qc.fmt)          # the tensor's scale repeated per output channel
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


# linears outside the blocks: (reference module name, key prefix in the model's `g` dict)
_GLOBAL_LINEARS = (("time_embedding.0", "time0"), ("time_embedding.2", "time2"), ("time_projection.1", "tproj"),
                   ("text_embedding.0", "text0"), ("text_embedding.2", "text2"), ("head.head", "head"))


def quantize_dynamic(module, qconfig_dict: Dict[str, Optional[QConfig]]):
    """Quantise, in place, the linears of a HipCausalWanModel (or a wrapper / pipeline holding one): every nn.Linear of the
    reference's transformer that the dict does not exclude — the eight per block and the three timestep-MLP linears; with the
    example's dict (`text_embedding`, `proj_out`, `head`: None) exactly the set `oracle/quant_oracle.py::model_hook` quantises.
    Linears whose K is not a multiple of 128 stay bf16 (kernel constraint; none in the Wan configs)."""
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
            blk.w[key + "_act"] = qc.act
            n += 1
    # the timestep MLPs (time_embedding.{0,2}, time_projection.1: nn.Linear children of the transformer, causal_model.py:615-618)
    # fall under the dict's default like every other Linear; text_embedding / head are the caller's exclusions, the patch
    # embedding is a Conv3d upstream and is no Linear
    ng = 0
    for ref_name, key in _GLOBAL_LINEARS:
        qc = _config_for(ref_name, qconfig_dict)
        w = model.g.get(key + "_w")
        if qc is None or w is None or w.shape[1] % 128:
            continue
        model.g[key + "_q"], model.g[key + "_s"] = quantize_weight(w, qc)
        model.g[key + "_fmt"], model.g[key + "_act"] = qc.fmt, qc.act
        ng += 1
    model.quantized_linears = n
    model.quantized_global_linears = ng
    getattr(model, "_temb_cache", {}).clear()        # memoised modulation tables were computed with the bf16 timestep MLPs
    return module


def dequantize(module):
    """Undo `quantize_dynamic` in place: drop the 8-bit copies, the bf16 weights (never removed) serve again."""
    model = module
    for attr in ("generator", "model"):
        while hasattr(model, attr) and not hasattr(model, "blocks"):
            model = getattr(model, attr)
    for d in [blk.w for blk in model.blocks] + [model.g]:
        for k in [k for k in d if k.endswith(("_q", "_s", "_fmt", "_act"))]:
            del d[k]
    model.quantized_linears = model.quantized_global_linears = 0
    getattr(model, "_temb_cache", {}).clear()
    return module
