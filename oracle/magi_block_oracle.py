"""TEST INFRASTRUCTURE — CPU restatement of one MAGI transformer layer (the denoise-step compute of BASELINE config 5).

Follows, function by function, `inferix/models/magi/dit/dit_module.py` of the reference:
  TransformerLayer.forward / attn_post_process / attn_linear_proj / gating_and_mlp      :1243-1319
  FullyParallelAttention.forward (cp_strategy "none" and the per-rank view of "cp_ulysses") :1087-1195
  get_q / get_k / get_v / get_xqkv, core_attention, cross_attention                       :902-1085
  CustomLayerNormLinear, CustomMLP, FusedLayerNorm, AdaModulateLayer, softcap             :180-201,326-364,393-431,496-557
  bias_modulate_add + range_mod (the Triton kernel's indexing)                            :204-313
  per-tensor / per-channel static FP8 linears + div_clamp_to                              :367-490
and `_high_precision_promoter` (dit_model.py:620-637: q/k layer norms — not the `_xattn` ones — and the two post
norms are fp32 modules; everything else is `params_dtype` = bf16).

Third-party arithmetic the reference calls and that is NOT in its tree (flash-attn rotary / attention, flashinfer
bmm_fp8 / silu_and_mul; all unpinned) is restated from the published definitions exactly as oracle/_refstub.install_magi
does for the golden generator — for those steps parity is pinned to that restatement only.

Weights: a dict with the reference layer's own state-dict names (`self_attention.linear_qkv.q.weight`, ...).
Pinned by tests/golden/magi_block*.npz (oracle/gen_golden_magi_block.py runs the reference's TransformerLayer on CPU;
this file reproduces every stored tensor bit for bit — tests/test_magi_block_oracle.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

BF = torch.bfloat16


@dataclass
class MagiLayerConfig:
    hidden_size: int = 3072
    ffn_hidden_size: int = 12288
    num_attention_heads: int = 24
    num_query_groups: int = 8
    kv_channels: int = 128
    layernorm_epsilon: float = 1e-6
    apply_layernorm_1p: bool = True
    cond_hidden_ratio: float = 0.25
    xattn_cond_hidden_ratio: float = 1.0
    cond_gating_ratio: float = 1.0
    gated_linear_unit: bool = False

    @property
    def q_size(self) -> int:
        return self.kv_channels * self.num_attention_heads

    @property
    def kv_size(self) -> int:
        return self.kv_channels * self.num_query_groups

    @property
    def cond_size(self) -> int:
        return int(self.hidden_size * self.cond_hidden_ratio)

    @property
    def xattn_size(self) -> int:
        return int(self.hidden_size * self.xattn_cond_hidden_ratio)


def tiny_config() -> MagiLayerConfig:
    return MagiLayerConfig(hidden_size=256, ffn_hidden_size=512, num_attention_heads=4, num_query_groups=2)


FP32_PARAMS = ("self_attention.q_layernorm.", "self_attention.k_layernorm.", "self_attn_post_norm.", "mlp_post_norm.")


def param_shapes(cfg: MagiLayerConfig) -> Dict[str, Tuple[int, ...]]:
    h, q, kv, f, hd = cfg.hidden_size, cfg.q_size, cfg.kv_size, cfg.ffn_hidden_size, cfg.kv_channels
    fc1 = 2 * f if cfg.gated_linear_unit else f
    return {
        "ada_modulate_layer.proj.0.weight": (int(h * cfg.cond_gating_ratio * 2), cfg.cond_size),
        "ada_modulate_layer.proj.0.bias": (int(h * cfg.cond_gating_ratio * 2),),
        "self_attention.linear_qkv.layer_norm.weight": (h,), "self_attention.linear_qkv.layer_norm.bias": (h,),
        "self_attention.linear_qkv.q.weight": (q, h), "self_attention.linear_qkv.qx.weight": (q, h),
        "self_attention.linear_qkv.k.weight": (kv, h), "self_attention.linear_qkv.v.weight": (kv, h),
        "self_attention.linear_kv_xattn.weight": (2 * kv, cfg.xattn_size),
        "self_attention.linear_proj.weight": (h, 2 * q),
        "self_attention.q_layernorm.weight": (hd,), "self_attention.q_layernorm.bias": (hd,),
        "self_attention.q_layernorm_xattn.weight": (hd,), "self_attention.q_layernorm_xattn.bias": (hd,),
        "self_attention.k_layernorm.weight": (hd,), "self_attention.k_layernorm.bias": (hd,),
        "self_attention.k_layernorm_xattn.weight": (hd,), "self_attention.k_layernorm_xattn.bias": (hd,),
        "self_attn_post_norm.weight": (h,), "self_attn_post_norm.bias": (h,),
        "mlp.layer_norm.weight": (h,), "mlp.layer_norm.bias": (h,),
        "mlp.linear_fc1.weight": (fc1, h), "mlp.linear_fc2.weight": (h, f),
        "mlp_post_norm.weight": (h,), "mlp_post_norm.bias": (h,),
    }


FP8_PER_TENSOR = ("self_attention.linear_qkv.q", "self_attention.linear_qkv.qx", "self_attention.linear_qkv.k",
                  "self_attention.linear_qkv.v", "mlp.linear_fc1")            # PerTensorQuantizedFp8Linear (dit_module.py:413, :526)
FP8_PER_CHANNEL = ("self_attention.linear_proj", "mlp.linear_fc2")           # PerChannelQuantizedFp8Linear (:867, :539)


def layer_is_fp8(layer_number: int, num_layers: int) -> bool:
    """Which layers of an fp8_quant model hold FP8 linears: all but the first and the last (dit_module.py:410, :526, :864-866)."""
    return layer_number != 0 and layer_number != num_layers - 1


def to_fp8_layer_weights(W: Dict[str, torch.Tensor], seed: int) -> Dict[str, torch.Tensor]:
    """The parameter set of a layer built with `engine_config.fp8_quant` (layers other than the first and the last): the seven
    linears hold e4m3 weights `[1, out, in]` + `weight_scale [1]` + `input_scale` (`[in]` per-tensor form / `[1]` per-channel
    form) + `smooth_scale [1, in]` (per-channel form).  Synthetic scales: weight_scale = amax / 448; activation divisors
    around 0.03 with +-20 % spread per input channel, so that x / divisor stays well inside +-448 for unit-scale activations."""
    g = torch.Generator().manual_seed(seed + 7919)
    out = dict(W)
    for name in FP8_PER_TENSOR + FP8_PER_CHANNEL:
        w = W[name + ".weight"].float()
        ws = (w.abs().max() / 448.0).reshape(1)
        out[name + ".weight"] = (w / ws).to(torch.float8_e4m3fn).unsqueeze(0)
        out[name + ".weight_scale"] = ws.float()
        spread = 0.03 * (1.0 + 0.2 * (2 * torch.rand(w.shape[1], generator=g) - 1))
        if name in FP8_PER_TENSOR:
            out[name + ".input_scale"] = spread.float()
        else:
            out[name + ".input_scale"] = torch.tensor([0.03])
            out[name + ".smooth_scale"] = spread.float().unsqueeze(0)
    return out


def init_layer_weights(cfg: MagiLayerConfig, seed: int, fp8: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights (no checkpoint exists offline): matrices ~ N(0, 1/fan_in) so that activations keep unit
    scale through the layer, norm weights ~ N(0, 0.1) (they are `1 + w` under apply_layernorm_1p), biases ~ N(0, 0.1).
    The golden generator loads exactly these into the reference module."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape in param_shapes(cfg).items():
        if len(shape) == 2:
            t = torch.randn(shape, generator=g) / math.sqrt(shape[1])
        elif name.endswith("layer_norm.weight"):                      # nn.LayerNorm (no 1p): weights around 1
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = 0.1 * torch.randn(shape, generator=g)
        W[name] = t.float() if name.startswith(FP32_PARAMS) else t.to(BF)
    return to_fp8_layer_weights(W, seed) if fp8 else W


# ---------------------------------------------------------------------------------------------------------------------
def fused_layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, cfg: MagiLayerConfig) -> torch.Tensor:
    """FusedLayerNorm.forward (dit_module.py:358-360): `weight + 1` in the PARAMETER dtype, then F.layer_norm."""
    w = w + 1 if cfg.apply_layernorm_1p else w
    return F.layer_norm(x, (x.shape[-1],), w, b, cfg.layernorm_epsilon)


def softcap(x: torch.Tensor, cap: float) -> torch.Tensor:
    return (cap * torch.tanh(x.float() / cap)).to(x.dtype)                        # dit_module.py:363-364


def apply_rotary(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """flash_attn.layers.rotary.apply_rotary_emb, non-interleaved (published definition; third party).
    x `[b, s, hn, hd]`, cos / sin `[s, hd/2]`."""
    ro = cos.shape[-1] * 2
    c, s = cos.float()[None, :, None, :], sin.float()[None, :, None, :]
    x1, x2 = x[..., : ro // 2].float(), x[..., ro // 2: ro].float()
    return torch.cat([x1 * c - x2 * s, x1 * s + x2 * c, x[..., ro:].float()], dim=-1).to(x.dtype)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """softmax(q k^T / sqrt(d)) v, grouped-query heads, `[s, h, d]` (flash_attn_func's definition, via SDPA)."""
    rep = q.shape[1] // k.shape[1]
    if rep > 1:
        k, v = k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1)
    o = F.scaled_dot_product_attention(q.transpose(0, 1)[None], k.transpose(0, 1)[None], v.transpose(0, 1)[None],
                                       scale=1.0 / math.sqrt(q.shape[-1]))
    return o[0].transpose(0, 1).contiguous()


def div_clamp_to(x: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """dit_module.py:367-387: x / scale in fp32, clamp to +-448, round to bf16, THEN cast to e4m3fn (two roundings)."""
    v = torch.clamp(x.float() / scale.float(), -448.0, 448.0).bfloat16()
    return v.to(torch.float8_e4m3fn)


def fp8_static_linear(x: torch.Tensor, wq: torch.Tensor, weight_scale: torch.Tensor, input_scale: torch.Tensor,
                      divisor: torch.Tensor) -> torch.Tensor:
    """PerTensorQuantizedFp8Linear (divisor = input_scale `[in]`) / PerChannelQuantizedFp8Linear (divisor = smooth_scale
    `[1, in]`) forward (dit_module.py:448-490) with flashinfer's bmm_fp8 restated as fp32 accumulate x A_scale x B_scale."""
    xq = div_clamp_to(x, divisor)
    acc = xq.reshape(-1, xq.shape[-1]).float() @ wq.reshape(-1, wq.shape[-1]).float().t()
    y = (acc * input_scale.flatten()[0] * weight_scale.flatten()[0]).to(BF)
    return y.reshape(x.shape[:-1] + (wq.shape[-2],))


# ---------------------------------------------------------------------------------------------------------------------
class MagiLayerCache:
    """MagiKVCacheManager's rule (magi_kv_cache_manager.py:76-187) for one layer, batch 1: the prefix `[0, slice_point*clip)`
    is read from the cache, the new rows are appended behind it for this forward, and — when `update_kv_cache` — stored
    (all of them, or all but the last clip under `distill_nearly_clean_chunk`)."""

    def __init__(self, max_tokens: int, kv_heads: int, hd: int):
        self.k = torch.zeros(max_tokens, kv_heads, hd, dtype=BF)
        self.v = torch.zeros(max_tokens, kv_heads, hd, dtype=BF)

    def adjust(self, k: torch.Tensor, v: torch.Tensor, *, slice_point: int, clip: int, update: bool, use_cache: bool,
               distill: bool) -> Tuple[torch.Tensor, torch.Tensor]:
        if not use_cache:
            return k, v
        start = slice_point * clip
        n = k.shape[0]
        if update:
            stored = n - clip if distill else n
            self.k[start:start + stored] = k[:stored]
            self.v[start:start + stored] = v[:stored]
        return torch.cat([self.k[:start], k]), torch.cat([self.v[:start], v])


@dataclass
class LayerMeta:
    """The fields of ModelMetaArgs / InferenceParams the layer reads (core/types/inference.py:72-101), batch 1."""
    q_ranges: Sequence[Tuple[int, int]]
    k_ranges: Sequence[Tuple[int, int]]
    cu_seqlens_q: Sequence[int]
    cu_seqlens_kv: Sequence[int]
    clip_token_nums: int
    slice_point: int = 0
    update_kv_cache: bool = False
    use_cache: bool = False               # extract_prefix_video_feature or fwd_extra_1st_chunk or slice_point > 0
    distill_nearly_clean_chunk: bool = False


def layer_forward(W: Dict[str, torch.Tensor], cfg: MagiLayerConfig, x: torch.Tensor, condition: torch.Tensor,
                  condition_map: torch.Tensor, y_xattn_flat: torch.Tensor, rope: torch.Tensor, meta: LayerMeta,
                  cache: Optional[MagiLayerCache] = None, taps: Optional[dict] = None) -> torch.Tensor:
    """TransformerLayer.forward for batch 1: x `[s, 1, h]` bf16, condition `[1, ranges, cond]` bf16, condition_map `[s, 1]`
    int, y_xattn_flat `[y_tokens, xattn]` bf16, rope `[s, hd]` fp32 = (sin | cos).  `taps` (optional dict) receives the
    intermediate tensors the golden fixture also stores."""
    p = "self_attention."
    eps, hd = cfg.layernorm_epsilon, cfg.kv_channels
    s_len, bsz, h = x.shape
    assert bsz == 1, "the cached MAGI path runs batch 1 (3-cfg folds ranges into the batch upstream)"
    tap = (lambda k, v: taps.__setitem__(k, v)) if taps is not None else (lambda k, v: None)

    def lin(t: torch.Tensor, name: str) -> torch.Tensor:         # nn.Linear, or the static-scale FP8 linear of an fp8_quant layer
        if name + ".weight_scale" in W:
            div = W[name + ".smooth_scale"] if name + ".smooth_scale" in W else W[name + ".input_scale"]
            return fp8_static_linear(t, W[name + ".weight"], W[name + ".weight_scale"], W[name + ".input_scale"], div)
        return F.linear(t, W[name + ".weight"])
    sin_emb, cos_emb = rope.tensor_split(2, -1)                                          # :1097
    # ---- CustomLayerNormLinear.forward_ln + the four projections (:415-431)
    hln = F.layer_norm(x, (h,), W[p + "linear_qkv.layer_norm.weight"], W[p + "linear_qkv.layer_norm.bias"], eps)
    tap("hln", hln)

    def qk(name: str, ln: str) -> torch.Tensor:                                          # get_q / get_k (:902-934)
        t = lin(hln, p + f"linear_qkv.{name}")
        t = t.reshape(s_len, bsz, -1, hd).float()
        t = fused_layer_norm(t, W[p + ln + ".weight"], W[p + ln + ".bias"], cfg)
        t = apply_rotary(t.transpose(0, 1).contiguous(), cos_emb, sin_emb).to(BF)
        return t.transpose(0, 1).reshape(s_len * bsz, -1, hd).contiguous()               # "b sq hn hd -> (sq b) hn hd"
    q, k = qk("q", "q_layernorm"), qk("k", "k_layernorm")
    v = lin(hln, p + "linear_qkv.v").reshape(s_len * bsz, -1, hd).contiguous()
    tap("q", q), tap("k", k), tap("v", v)
    if cache is not None:
        key, value = cache.adjust(k, v, slice_point=meta.slice_point, clip=meta.clip_token_nums,
                                  update=meta.update_kv_cache, use_cache=meta.use_cache,
                                  distill=meta.distill_nearly_clean_chunk)
    else:
        key, value = k, v
    # ---- core_attention, bs == 1 branch (:989-1015)
    core = torch.cat([attention(q[qs:qe], key[ks:ke], value[ks:ke])
                      for (qs, qe), (ks, ke) in zip(meta.q_ranges, meta.k_ranges)], dim=0)
    core = core.reshape(s_len, bsz, -1)
    tap("core", core)
    # ---- cross_attention / get_xqkv (:954-970, :1047-1085)
    qx = lin(hln, p + "linear_qkv.qx").transpose(0, 1).reshape(bsz * s_len, -1, hd)
    qx = fused_layer_norm(qx, W[p + "q_layernorm_xattn.weight"], W[p + "q_layernorm_xattn.bias"], cfg)
    kvx = torch.cat([torch.matmul(y_xattn_flat, w.t()) for w in torch.chunk(W[p + "linear_kv_xattn.weight"], 8, dim=0)],
                    dim=1)
    kvx = kvx.view(y_xattn_flat.shape[0], -1, 2 * hd)
    kx, vx = torch.split(kvx, hd, dim=-1)
    kx = fused_layer_norm(kx, W[p + "k_layernorm_xattn.weight"], W[p + "k_layernorm_xattn.bias"], cfg)
    cq, ck = list(meta.cu_seqlens_q), list(meta.cu_seqlens_kv)
    xo = torch.cat([attention(qx[cq[i]:cq[i + 1]], kx[ck[i]:ck[i + 1]], vx[ck[i]:ck[i + 1]].contiguous())
                    for i in range(len(cq) - 1)], dim=0)
    xo = xo.reshape(bsz, s_len, -1).transpose(0, 1).contiguous()                          # "(b sq) hn hd -> sq b (hn hd)"
    tap("qx", qx), tap("kx", kx), tap("xattn", xo)
    # ---- attn_linear_proj (:1281-1295): "sq b (n hn hd) -> sq b (hn n hd)", n = 2, hn = 8 HARD-CODED
    a = torch.cat([core, xo], dim=2)
    a = a.reshape(s_len, bsz, 2, 8, -1).transpose(2, 3).reshape(s_len, bsz, -1)
    a = lin(a, p + "linear_proj")
    tap("proj", a)
    # ---- gating_and_mlp (:1297-1319)
    gate = F.linear(F.silu(condition), W["ada_modulate_layer.proj.0.weight"], W["ada_modulate_layer.proj.0.bias"])
    gate = softcap(gate, 1.0)
    gate_msa, gate_mlp = gate.chunk(2, dim=-1)
    tap("gate", gate)

    def bias_modulate_add(t: torch.Tensor, residual: torch.Tensor, g: torch.Tensor, norm: str) -> torch.Tensor:   # :295-313
        rows = t.float().transpose(0, 1).flatten(0, 1)                                   # (b s) rows, as range_mod_triton
        gv = g.float().flatten(0, 1)[condition_map.transpose(0, 1).flatten(0, 1).long()]
        t = (rows * gv).reshape(bsz, s_len, h).transpose(0, 1)
        t = fused_layer_norm(t, W[norm + ".weight"], W[norm + ".bias"], cfg)
        return (t + residual.float()).to(BF)
    hs = bias_modulate_add(a, x, gate_msa, "self_attn_post_norm")
    tap("attn_res", hs)
    m = F.layer_norm(hs, (h,), W["mlp.layer_norm.weight"], W["mlp.layer_norm.bias"], eps)  # CustomMLP.forward (:545-557)
    m = lin(m, "mlp.linear_fc1")
    if cfg.gated_linear_unit:
        d = m.shape[-1] // 2
        m = F.silu(m[..., :d]) * m[..., d:]                                              # flashinfer silu_and_mul (published)
    else:
        m = F.gelu(m)
    m = lin(m, "mlp.linear_fc2")
    tap("mlp", m)
    return bias_modulate_add(m, hs, gate_mlp, "mlp_post_norm")


def exact_layer_forward(W, cfg, x, condition, condition_map, y_xattn_flat, rope, meta, cache=None):
    """The same layer evaluated in float64 with NO intermediate rounding: the distance of the reference's own bf16 result
    from this is the rounding-noise floor parity tests measure the HIP path against.  `cache` (a MagiLayerCache) is read,
    never written."""
    c64 = None
    if cache is not None:
        c64 = MagiLayerCache(1, 1, 1)
        c64.k, c64.v = cache.k.double(), cache.v.double()
    return _layer_forward_f64({k: v.double() for k, v in W.items()}, cfg, x.double(), condition.double(), condition_map,
                              y_xattn_flat.double(), rope.double(), meta, c64)


def _layer_forward_f64(W, cfg, x, condition, condition_map, y, rope, meta, cache):
    p = "self_attention."
    eps, hd = cfg.layernorm_epsilon, cfg.kv_channels
    s_len, bsz, h = x.shape
    sin_emb, cos_emb = rope.tensor_split(2, -1)
    one = 1.0 if cfg.apply_layernorm_1p else 0.0

    def ln(t, w, b, plus=0.0):
        return F.layer_norm(t, (t.shape[-1],), w + plus, b, eps)

    def rot(t):
        ro = cos_emb.shape[-1] * 2
        c, s = cos_emb[None, :, None, :], sin_emb[None, :, None, :]
        x1, x2 = t[..., : ro // 2], t[..., ro // 2: ro]
        return torch.cat([x1 * c - x2 * s, x1 * s + x2 * c, t[..., ro:]], dim=-1)     # partial rotary: the tail passes through

    def att(q, k, v):
        rep = q.shape[1] // k.shape[1]
        k, v = k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1)
        sc = torch.einsum("qhd,khd->hqk", q, k) / math.sqrt(hd)
        return torch.einsum("hqk,khd->qhd", sc.softmax(-1), v)
    hln = ln(x, W[p + "linear_qkv.layer_norm.weight"], W[p + "linear_qkv.layer_norm.bias"])

    def qk(name, lnn):
        t = F.linear(hln, W[p + f"linear_qkv.{name}.weight"]).reshape(s_len, bsz, -1, hd)
        t = ln(t, W[p + lnn + ".weight"], W[p + lnn + ".bias"], one)
        return rot(t.transpose(0, 1)).transpose(0, 1).reshape(s_len * bsz, -1, hd)
    q, k = qk("q", "q_layernorm"), qk("k", "k_layernorm")
    v = F.linear(hln, W[p + "linear_qkv.v.weight"]).reshape(s_len * bsz, -1, hd)
    if cache is not None and meta.use_cache:
        start = meta.slice_point * meta.clip_token_nums
        key, value = torch.cat([cache.k[:start], k]), torch.cat([cache.v[:start], v])
    else:
        key, value = k, v
    core = torch.cat([att(q[a:b], key[c:d], value[c:d]) for (a, b), (c, d) in zip(meta.q_ranges, meta.k_ranges)]).reshape(s_len, bsz, -1)
    qx = F.linear(hln, W[p + "linear_qkv.qx.weight"]).transpose(0, 1).reshape(bsz * s_len, -1, hd)
    qx = ln(qx, W[p + "q_layernorm_xattn.weight"], W[p + "q_layernorm_xattn.bias"], one)
    kvx = F.linear(y, W[p + "linear_kv_xattn.weight"]).view(y.shape[0], -1, 2 * hd)
    kx, vx = kvx[..., :hd], kvx[..., hd:]
    kx = ln(kx, W[p + "k_layernorm_xattn.weight"], W[p + "k_layernorm_xattn.bias"], one)
    cq, ck = list(meta.cu_seqlens_q), list(meta.cu_seqlens_kv)
    xo = torch.cat([att(qx[cq[i]:cq[i + 1]], kx[ck[i]:ck[i + 1]], vx[ck[i]:ck[i + 1]]) for i in range(len(cq) - 1)])
    xo = xo.reshape(bsz, s_len, -1).transpose(0, 1)
    a = torch.cat([core, xo], dim=2).reshape(s_len, bsz, 2, 8, -1).transpose(2, 3).reshape(s_len, bsz, -1)
    a = F.linear(a, W[p + "linear_proj.weight"])
    gate = torch.tanh(F.linear(F.silu(condition), W["ada_modulate_layer.proj.0.weight"], W["ada_modulate_layer.proj.0.bias"]))
    g_msa, g_mlp = gate.chunk(2, dim=-1)

    def bma(t, res, g, norm):
        gv = g.flatten(0, 1)[condition_map.transpose(0, 1).flatten(0, 1).long()]
        t = (t.transpose(0, 1).flatten(0, 1) * gv).reshape(bsz, s_len, h).transpose(0, 1)
        return ln(t, W[norm + ".weight"], W[norm + ".bias"], one) + res
    hs = bma(a, x, g_msa, "self_attn_post_norm")
    m = F.linear(ln(hs, W["mlp.layer_norm.weight"], W["mlp.layer_norm.bias"]), W["mlp.linear_fc1.weight"])
    if cfg.gated_linear_unit:
        d = m.shape[-1] // 2
        m = F.silu(m[..., :d]) * m[..., d:]
    else:
        m = F.gelu(m)
    return bma(F.linear(m, W["mlp.linear_fc2.weight"]), hs, g_mlp, "mlp_post_norm")


# ---------------------------------------------------------------------------------------------------------------------
def fixture_geometry(fx) -> Tuple[MagiLayerConfig, int, int, int, int, int]:
    """(config, layers, clip, calls, weight seed, cache tokens) of a tests/golden/magi_block_*.npz fixture."""
    h, f, hq, hk, hd, n_layers, clip, n_calls, wseed, max_tokens = [int(v) for v in fx["geom"]]
    cfg = MagiLayerConfig(hidden_size=h, ffn_hidden_size=f, num_attention_heads=hq, num_query_groups=hk, kv_channels=hd)
    return cfg, n_layers, clip, n_calls, wseed, max_tokens


def fixture_call(fx, ci: int) -> Tuple[Dict[str, torch.Tensor], LayerMeta]:
    """Inputs and LayerMeta of forward `ci` of a fixture."""
    inp = {k: fx[f"c{ci}_in_{k}"] for k in ("x", "condition", "condition_map", "y", "rope")}
    clip, sp, upd, use, dis = [int(v) for v in fx[f"c{ci}_meta_flags"]]
    meta = LayerMeta(q_ranges=[tuple(r) for r in fx[f"c{ci}_meta_q_ranges"].tolist()],
                     k_ranges=[tuple(r) for r in fx[f"c{ci}_meta_k_ranges"].tolist()],
                     cu_seqlens_q=fx[f"c{ci}_meta_cu_q"].tolist(), cu_seqlens_kv=fx[f"c{ci}_meta_cu_kv"].tolist(),
                     clip_token_nums=clip, slice_point=sp, update_kv_cache=bool(upd), use_cache=bool(use),
                     distill_nearly_clean_chunk=bool(dis))
    return inp, meta
