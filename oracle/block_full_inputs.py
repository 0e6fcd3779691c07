"""TEST INFRASTRUCTURE — seeded inputs of the full-size block fixture (tests/golden/block_full_size.npz): shared by the generator
(oracle/gen_golden_block_full.py, which runs the reference on them) and the GPU test, which regenerates them instead of storing
~200 MB of tensors.  torch's CPU generator is deterministic for a given torch build; the fixture stores checksums to catch drift."""
from __future__ import annotations

from typing import Dict

import torch

import wan_oracle as O

BF = torch.bfloat16
FRAMES, LAT_H, LAT_W = 3, 60, 104          # Self-Forcing 480p block: 3 frames of 30 x 52 patches = 4680 tokens
SEL = torch.cat([torch.arange(0, 64), torch.arange(2300, 2364), torch.arange(4616, 4680)])     # stored output rows


def config() -> O.WanConfig:
    return O.WanConfig(num_layers=1, text_len=512, text_dim=64, freq_dim=64, latent_h=LAT_H, latent_w=LAT_W)


def make(case: int) -> Dict[str, torch.Tensor]:
    """case 0: first block (current_start 0, L = 4680); case 1: seventh block (current_start 28080, prefix of 28080 cached tokens
    with seeded contents standing for the six earlier blocks, L = 32760)."""
    cfg = config()
    g = torch.Generator().manual_seed(1000 + case)
    n = FRAMES * cfg.frame_seqlen
    d = dict(x=torch.randn(1, n, cfg.dim, generator=g).to(BF),
             e0=(torch.randn(1, FRAMES, 6, cfg.dim, generator=g) * 0.5).to(BF),
             ctx=torch.randn(1, cfg.text_len, cfg.dim, generator=g).to(BF))
    d["ctx"][:, 40:] = d["ctx"][:, 40:41]          # padded prompt: identical rows behind the 40 real tokens (text_embedding of zeros)
    start = 0 if case == 0 else 6 * n
    d["current_start"] = torch.tensor(start)
    if start:
        d["prefix_k"] = torch.randn(start, cfg.num_heads, cfg.head_dim, generator=g).to(BF)
        d["prefix_v"] = torch.randn(start, cfg.num_heads, cfg.head_dim, generator=g).to(BF)
    return d


def checksum(t: torch.Tensor) -> int:
    return int(t.contiguous().view(torch.int16).to(torch.int64).sum().item())
