"""TEST INFRASTRUCTURE — seeded inputs of the CausVid 720p full-size block fixture (tests/golden/block_720p_full_size.npz): shared by
the generator (oracle/gen_golden_block_720p.py, which runs the reference's CausVid block on them) and the GPU test, which
regenerates them instead of storing ~450 MB of tensors.  Same construction as oracle/block_full_inputs.py (480p), at BASELINE
config 3's geometry: 3 frames of 45 x 80 patches = 10800 tokens, dim 1536, 12 heads, ffn 8960."""
from __future__ import annotations

from typing import Dict

import torch

import wan_oracle as O

BF = torch.bfloat16
FRAMES, LAT_H, LAT_W = 3, 90, 160          # CausVid 720p block: 3 frames of 45 x 80 patches = 10800 tokens
SEL = torch.cat([torch.arange(0, 64), torch.arange(5300, 5364), torch.arange(10736, 10800)])     # stored output rows
BLOCKS = 7                                  # case 1 is the seventh block of a clip: L = 75600 keys


def config() -> O.WanConfig:
    return O.WanConfig(num_layers=1, text_len=512, text_dim=64, freq_dim=64, latent_h=LAT_H, latent_w=LAT_W)


def make(case: int) -> Dict[str, torch.Tensor]:
    """case 0: first block (kv slots [0, 10800), L = 10800); case 1: seventh block (kv slots [64800, 75600), the 64800 cached
    tokens in front of them seeded, standing for the six earlier blocks, L = 75600)."""
    cfg = config()
    g = torch.Generator().manual_seed(2000 + case)
    n = FRAMES * cfg.frame_seqlen
    d = dict(x=torch.randn(1, n, cfg.dim, generator=g).to(BF),
             e0=(torch.randn(1, FRAMES, 6, cfg.dim, generator=g) * 0.5).to(BF),
             ctx=torch.randn(1, cfg.text_len, cfg.dim, generator=g).to(BF))
    d["ctx"][:, 40:] = d["ctx"][:, 40:41]          # padded prompt: identical rows behind the 40 real tokens
    start = 0 if case == 0 else (BLOCKS - 1) * n
    d["kv_start"] = torch.tensor(start)
    if start:
        d["prefix_k"] = torch.randn(start, cfg.num_heads, cfg.head_dim, generator=g).to(BF)
        d["prefix_v"] = torch.randn(start, cfg.num_heads, cfg.head_dim, generator=g).to(BF)
    return d


def checksum(t: torch.Tensor) -> int:
    return int(t.contiguous().view(torch.int16).to(torch.int64).sum().item())
