"""TEST INFRASTRUCTURE — CPU restatement of the MAGI `VideoDiTModel` around its layer stack (BASELINE config 5 as a model step).

Follows `inferix/models/magi/dit/dit_model.py` of the reference:
  forward_pre_process / get_embedding_and_meta   :111-330   x_embedder (a Conv3d whose stride is its kernel: a patchify GEMM), rope table,
                                                            TimestepEmbedder, CaptionEmbedder, condition / condition_map, y_xattn_flat,
                                                            cross-attention and core-attention ranges
  forward / TransformerBlock.forward            :337-362, dit_module.py:1361-1390   the layers, then the final LayerNorm on .float()
  forward_post_process / unpatchify             :97-107,332-351   FinalLinear in fp32, (T H W) N (pT pH pW C) -> N C T H W
and `inferix/models/magi/dit/dit_module.py`: TimestepEmbedder :53-106, CaptionEmbedder :109-160, FinalLinear :163-177,
build_fourier_pos_embed / build_rotary_pos_embed / LearnableRotaryEmbeddingCat :602-776.

Precision, as the reference runs it on a GPU: the embedders and the final linear are fp32 modules evaluated under
`torch.autocast("cuda", dtype=torch.float32)` — their linear / conv inputs are cast UP to fp32 (the one place a narrower value enters
is `t_freq.to(params_dtype)` inside TimestepEmbedder: the sinusoid is rounded to bf16 first); x, condition and y_xattn_flat are
rounded to bf16 behind them; the final LayerNorm is an fp32 module on an fp32 input (`_high_precision_promoter`, :620-637).

The layers themselves are oracle/magi_block_oracle.py.  Pinned by tests/golden/magi_model_tiny.npz (oracle/gen_golden_magi_model.py
runs the reference's VideoDiTModel on CPU with the fp32 autocast emulated; this file reproduces every stored tensor bit for bit).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

import magi_block_oracle as MB

BF = torch.bfloat16


@dataclass
class MagiModelConfig:
    layer: MB.MagiLayerConfig
    num_layers: int = 34
    patch_size: int = 2
    t_patch_size: int = 1
    in_channels: int = 16
    out_channels: int = 16
    caption_channels: int = 4096
    caption_max_length: int = 800
    x_rescale_factor: float = 1.0
    half_channel_vae: bool = False
    frequency_embedding_size: int = 256


def tiny_model_config() -> MagiModelConfig:
    return MagiModelConfig(layer=MB.tiny_config(), num_layers=3, caption_channels=64, caption_max_length=12, x_rescale_factor=0.5)


def embedder_shapes(cfg: MagiModelConfig) -> Dict[str, Tuple[int, ...]]:
    L = cfg.layer
    h, cond, xat = L.hidden_size, L.cond_size, L.xattn_size
    cin = cfg.in_channels * (2 if cfg.half_channel_vae else 1)
    return {"x_embedder.weight": (h, cin, cfg.t_patch_size, cfg.patch_size, cfg.patch_size),
            "t_embedder.mlp.0.weight": (cond, cfg.frequency_embedding_size), "t_embedder.mlp.0.bias": (cond,),
            "t_embedder.mlp.2.weight": (cond, cond), "t_embedder.mlp.2.bias": (cond,),
            "y_embedder.null_caption_embedding": (cfg.caption_max_length, cfg.caption_channels),
            "y_embedder.y_proj_xattn.0.weight": (xat, cfg.caption_channels), "y_embedder.y_proj_xattn.0.bias": (xat,),
            "y_embedder.y_proj_adaln.0.weight": (cond, cfg.caption_channels), "y_embedder.y_proj_adaln.0.bias": (cond,),
            "rope.bands": (L.hidden_size // L.num_attention_heads // 8,),      # LearnableRotaryEmbeddingCat(hidden // heads), dit_model.py:75-77
            "videodit_blocks.final_layernorm.weight": (h,), "videodit_blocks.final_layernorm.bias": (h,),
            "final_linear.linear.weight": (cfg.patch_size * cfg.patch_size * cfg.t_patch_size * cfg.out_channels, h)}


def default_bands(dim: int, temperature: float = 10000.0) -> torch.Tensor:
    """freq_bands(dim // 8, step=1) (dit_module.py:594-599, in_pixels=False)."""
    n = dim // 8
    exp = torch.arange(0, n, 1, dtype=torch.int64).to(torch.float32) / n
    return 1.0 / (temperature ** exp)


def init_embedder_weights(cfg: MagiModelConfig, seed: int) -> Dict[str, torch.Tensor]:
    """Seeded weights of everything outside the layers: fp32, as `_high_precision_promoter` (dit_model.py:620-637) makes the embedders,
    the rope bands, the final LayerNorm and the final linear."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape in embedder_shapes(cfg).items():
        if name == "rope.bands":
            W[name] = default_bands(cfg.layer.hidden_size // cfg.layer.num_attention_heads) * (1.0 + 0.05 * torch.randn(shape, generator=g))     # "learnable": not the default
        elif name.endswith("final_layernorm.weight"):
            W[name] = 0.1 * torch.randn(shape, generator=g)            # zero-centred gamma (apply_layernorm_1p): weight + 1 is applied
        elif name.endswith(".bias"):
            W[name] = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("null_caption_embedding"):
            W[name] = torch.randn(shape, generator=g)
        else:
            fan_in = math.prod(shape[1:])
            W[name] = torch.randn(shape, generator=g) * fan_in ** -0.5
    return W


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0, rescale: float = 1000.0) -> torch.Tensor:
    """TimestepEmbedder.timestep_embedding (dit_module.py:76-95): cos first, then sin."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None] * rescale
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def rope_table(bands: torch.Tensor, shape: List[int], ref_shape: List[float]) -> torch.Tensor:
    """LearnableRotaryEmbeddingCat.get_embed (dit_module.py:760-776, in_pixels=False): [T*H*W, 2 * 3 * dim/8 ... ] = (sin | cos)."""
    t = [torch.arange(s, dtype=torch.int64).to(torch.float32) for s in shape]
    t[1] = t[1] - (shape[1] - 1) / 2
    t[2] = t[2] - (shape[2] - 1) / 2
    tr = []
    for x, f, r in zip(t, shape, ref_shape):
        if f == 1:
            assert r == 1
            tr.append(x)
        else:
            tr.append(x / (f - 1) * (r - 1))
    grid = torch.stack(torch.meshgrid(*tr, indexing="ij"), dim=-1).unsqueeze(-1)
    pos = grid * bands
    n = shape[0] * shape[1] * shape[2]
    return torch.cat([pos.sin().reshape(n, -1), pos.cos().reshape(n, -1)], -1)


def pre_process(W: Dict[str, torch.Tensor], cfg: MagiModelConfig, x: torch.Tensor, t: torch.Tensor, y: torch.Tensor,
                xattn_mask: torch.Tensor, kv_range: torch.Tensor, caption_dropout_mask: torch.Tensor, *, range_num: int,
                denoising_range_num: int, slice_point: int = 0):
    """forward_pre_process for one rank without context parallelism.  Returns (x [S, N, h] bf16, condition [N, ranges, cond] bf16,
    condition_map [S, N] int64, y_xattn_flat [tokens, xattn] bf16, rope [S, hd] fp32, meta dict)."""
    L = cfg.layer
    x = (x * cfg.x_rescale_factor)
    if cfg.half_channel_vae:
        x = torch.cat([x, x], dim=1)
    x, t, y = x.float(), t.float(), y.float()
    pt, p = cfg.t_patch_size, cfg.patch_size
    xe = F.conv3d(x, W["x_embedder.weight"].float(), stride=(pt, p, p))                       # [N, h, T, H, W]
    N, _, T, H, Wd = xe.shape
    frame_in_range = T // denoising_range_num
    t_total = T + frame_in_range * slice_point
    rescale = math.sqrt((H * Wd) / (16 * 16))
    rope = rope_table(W["rope.bands"].float(), [t_total, H, Wd], [t_total, H / rescale, Wd / rescale])[-(T * H * Wd):]
    tf = timestep_embedding(t.flatten(), cfg.frequency_embedding_size).to(BF).float()       # `.to(self.data_type)`, then the fp32 autocast
    te = F.linear(F.silu(F.linear(tf, W["t_embedder.mlp.0.weight"], W["t_embedder.mlp.0.bias"])),
                  W["t_embedder.mlp.2.weight"], W["t_embedder.mlp.2.bias"]).reshape(N, denoising_range_num, -1)
    y_xattn = F.silu(F.linear(y, W["y_embedder.y_proj_xattn.0.weight"], W["y_embedder.y_proj_xattn.0.bias"]))    # [N*R, 1, Lc, xattn]
    # CaptionEmbedder.forward at inference (:149-160): the AdaLN branch does not see the caption at all — one of the two last rows of
    # the learned null caption, picked by the per-sample dropout flag (forward_3cfg passes [False] / [True], dit_model.py:410-429)
    null = W["y_embedder.null_caption_embedding"]
    cap = torch.where(caption_dropout_mask[:, None, None], null[None, -1, :], null[None, -2, :])                  # [N, 1, C]
    y_adaln = F.linear(cap.float(), W["y_embedder.y_proj_adaln.0.weight"], W["y_embedder.y_proj_adaln.0.bias"])  # [N, 1, cond]
    mask = xattn_mask.squeeze(1).squeeze(1)                                                   # [N*R, Lc]
    condition = te + y_adaln.squeeze(1).unsqueeze(1)
    assert condition.shape[:2] == (N, denoising_range_num), condition.shape
    per = (T * H * Wd) // denoising_range_num
    cmap = torch.repeat_interleave(torch.arange(N * denoising_range_num), per).reshape(N, -1).transpose(0, 1).contiguous()
    y_flat = torch.masked_select(y_xattn.squeeze(1), mask.unsqueeze(-1).bool()).reshape(-1, y_xattn.shape[-1])
    y_index = mask.reshape(mask.shape[0], -1).sum(-1)
    clip = H * Wd * frame_in_range
    cu_q = torch.tensor([0] + [clip] * denoising_range_num * N).cumsum(-1).to(torch.int32)
    cu_k = torch.cat([y_index.new_zeros(1), y_index]).to(torch.int64).cumsum(-1).to(torch.int32)
    meta = dict(H=H, W=Wd, clip_token_nums=clip, slice_point=slice_point, range_num=range_num, denoising_range_num=denoising_range_num,
                q_range=torch.stack([cu_q[:-1], cu_q[1:]], 1), k_range=kv_range, cu_seqlens_q=cu_q, cu_seqlens_kv=cu_k)
    xs = xe.to(BF).permute(2, 3, 4, 0, 1).reshape(T * H * Wd, N, -1).contiguous()             # "N C T H W -> (T H W) N C"
    return xs, condition.to(BF), cmap, y_flat.to(BF), rope, meta


def post_process(W: Dict[str, torch.Tensor], cfg: MagiModelConfig, hidden: torch.Tensor, H: int, Wd: int) -> torch.Tensor:
    """final LayerNorm on .float() (TransformerBlock.forward :1386-1388, an fp32 module), FinalLinear in fp32,
    unpatchify, undo the rescale (forward_post_process :332-351)."""
    L = cfg.layer
    hn = MB.fused_layer_norm(hidden.float(), W["videodit_blocks.final_layernorm.weight"], W["videodit_blocks.final_layernorm.bias"], L)
    o = F.linear(hn.float(), W["final_linear.linear.weight"])                                 # [S, N, pT pH pW C]
    S, N, _ = o.shape
    pt, p, C = cfg.t_patch_size, cfg.patch_size, cfg.out_channels
    T = S // (H * Wd)
    o = o.reshape(T, H, Wd, N, pt, p, p, C).permute(3, 7, 0, 4, 1, 5, 2, 6).reshape(N, C, T * pt, H * p, Wd * p).contiguous()
    if cfg.half_channel_vae:
        o = o[:, :16]
    return o / cfg.x_rescale_factor
