"""Host-side pieces of the Wan DiT that are tables or glue, not kernels: sinusoidal timestep
embedding, RoPE frequency tables, (un)patchify index maps.  Semantics follow
inferix/models/wan_base/components.py:11-51 and causal_model.py:634-641,1196-1219 of the reference."""
from __future__ import annotations

import math
from typing import Tuple

import torch


def sinusoidal_embedding_1d(dim: int, position: torch.Tensor) -> torch.Tensor:
    """[len(position), dim] float64: cos block then sin block, frequencies 10000^(-i/half)."""
    if dim % 2:
        raise ValueError("sinusoidal embedding needs an even dim")
    half = dim // 2
    pos = position.to(torch.float64)
    freq = torch.pow(10000, -torch.arange(half, dtype=torch.float64, device=pos.device) / half)
    ang = pos[:, None] * freq[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=1)


def rope_params(max_seq_len: int, dim: int, theta: float = 10000.0) -> torch.Tensor:
    """complex128 [max_seq_len, dim/2]: exp(i * pos * theta^(-2j/dim))."""
    if dim % 2:
        raise ValueError("rope dim must be even")
    inv = 1.0 / torch.pow(theta, torch.arange(0, dim, 2, dtype=torch.float64) / dim)
    ang = torch.outer(torch.arange(max_seq_len), inv)
    return torch.polar(torch.ones_like(ang), ang)


def rope_table(head_dim: int, max_len: int = 1024) -> torch.Tensor:
    """[max_len, head_dim/2, 2] float64 (cos, sin): temporal | height | width column groups with
    d - 4*(d//6), 2*(d//6), 2*(d//6) real dims (44/42/42 for head_dim 128) — what the fused
    RMSNorm+RoPE kernel indexes by (position, pair)."""
    d = head_dim
    f = torch.cat([rope_params(max_len, d - 4 * (d // 6)), rope_params(max_len, 2 * (d // 6)),
                   rope_params(max_len, 2 * (d // 6))], dim=1)
    return torch.view_as_real(f).contiguous()


def patchify(latent: torch.Tensor, patch_size: Tuple[int, int, int]) -> torch.Tensor:
    """[B, C, F, H, W] -> [B * F' * H' * W', C*pt*ph*pw]: the im2row of the stride==kernel Conv3d patch
    embedding (causal_model.py:609-610,917-920), columns in Conv3d weight order (c, pt, ph, pw)."""
    b, c, f, h, w = latent.shape
    pt, ph, pw = patch_size
    x = latent.reshape(b, c, f // pt, pt, h // ph, ph, w // pw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7)
    return x.reshape(b * (f // pt) * (h // ph) * (w // pw), c * pt * ph * pw)


def unpatchify(x: torch.Tensor, batch: int, grid: Tuple[int, int, int], patch_size: Tuple[int, int, int],
               out_dim: int) -> torch.Tensor:
    """[B*N, out_dim*prod(patch)] -> [B, C, F, H, W] ('fhwpqrc->cfphqwr', causal_model.py:1196-1219)."""
    f, h, w = grid
    pt, ph, pw = patch_size
    u = x.reshape(batch, f, h, w, pt, ph, pw, out_dim).permute(0, 7, 1, 4, 2, 5, 3, 6)
    return u.reshape(batch, out_dim, f * pt, h * ph, w * pw)


def rope_axis_split(head_dim: int) -> Tuple[int, int, int]:
    c = head_dim // 2
    return c - 2 * (c // 3), c // 3, c // 3
