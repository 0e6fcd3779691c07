"""Key ranges of MAGI's chunked denoising: which cached chunks every denoising chunk of a forward attends to.

Mirror of `SampleTransport.generate_default_kvrange`, `generate_noise2clean_kvrange`, `generate_kvrange_for_denoising_video` and
`generate_kvrange_for_prefix_video` (inferix/pipeline/magi/video_generate.py:373-529) as functions of the few numbers the methods
read from the transport object — the ranges are what `ModelMetaArgs.core_attn_params.k_range` carries into the attention layer
(`inferix_amd/magi/dit.py`), in tokens.  Pure integer work on the host; results are int32 `[batch * ranges, 2]` as upstream.
tests/test_magi_kv_ranges.py checks them against a golden the reference's own methods produced.
"""
from __future__ import annotations

from typing import List, Sequence

import torch


def chunk_token_nums(chunk_width: int, latent_h: int, latent_w: int, patch_size: int) -> int:
    """Tokens of one chunk: chunk_width latent frames of (H / patch) x (W / patch) patches (video_generate.py:362-371)."""
    return chunk_width * (latent_h // patch_size) * (latent_w // patch_size)


def _ranges(rows: List[List[int]], tokens: int, device) -> torch.Tensor:
    return (torch.tensor(rows, dtype=torch.int64) * tokens).to(torch.int32).to(device)


def generate_default_kvrange(tokens: int, slice_point: int, denoising_range_num: int, batch_size: int = 1,
                             device="cpu") -> torch.Tensor:
    """Causal default (:456-468): denoising chunk j sees every chunk from 0 up to itself."""
    range_num = slice_point + denoising_range_num
    rows = [[b * range_num, b * range_num + slice_point + j + 1] for b in range(batch_size) for j in range(denoising_range_num)]
    return _ranges(rows, tokens, device)


def generate_noise2clean_kvrange(tokens: int, slice_point: int, denoising_range_num: int, noise2clean_kvrange: Sequence[int],
                                 clean_chunk_kvrange: int, denoise_step_of_each_chunk: Sequence[int], num_steps: int,
                                 batch_size: int = 1, device="cpu") -> torch.Tensor:
    """Windowed schedule (:470-511): a chunk that has done s of `num_steps` steps is in stage s // (num_steps / stages) and looks
    back `noise2clean_kvrange[stage]` chunks (itself included); a finished chunk looks back `clean_chunk_kvrange` (-1: the last
    entry of the schedule)."""
    assert len(denoise_step_of_each_chunk) == denoising_range_num and len(noise2clean_kvrange) > 0
    if clean_chunk_kvrange == -1:
        clean_chunk_kvrange = noise2clean_kvrange[-1]
    assert num_steps % len(noise2clean_kvrange) == 0
    per_stage = num_steps // len(noise2clean_kvrange)
    look_back = [clean_chunk_kvrange if s == num_steps else noise2clean_kvrange[s // per_stage] for s in denoise_step_of_each_chunk]
    range_num = slice_point + denoising_range_num
    rows = []
    for b in range(batch_size):
        for j in range(denoising_range_num):
            end = slice_point + j + 1
            rows.append([b * range_num + max(0, end - look_back[j]), b * range_num + end])
    return _ranges(rows, tokens, device)


def generate_kvrange_for_denoising_video(tokens: int, slice_point: int, denoising_range_num: int,
                                         denoise_step_of_each_chunk: Sequence[int], num_steps: int, noise2clean_kvrange: Sequence[int],
                                         clean_chunk_kvrange: int, batch_size: int = 1, device="cpu") -> torch.Tensor:
    """The dispatcher (:513-529): the windowed schedule when the runtime config has one, else the causal default."""
    if len(noise2clean_kvrange) == 0:
        return generate_default_kvrange(tokens, slice_point, denoising_range_num, batch_size, device)
    return generate_noise2clean_kvrange(tokens, slice_point, denoising_range_num, noise2clean_kvrange, clean_chunk_kvrange,
                                        denoise_step_of_each_chunk, num_steps, batch_size, device)


def generate_kvrange_for_prefix_video(tokens: int, range_num: int, noise2clean_kvrange: Sequence[int], clean_chunk_kvrange: int,
                                      batch_size: int = 1, device="cpu") -> torch.Tensor:
    """Clean-feature extraction of a prefix video (:373-389): chunk j looks back `clean_chunk_kvrange` chunks (or the schedule's
    last entry, or 8 without either)."""
    if clean_chunk_kvrange != -1:
        prev = clean_chunk_kvrange
    elif len(noise2clean_kvrange) > 0:
        prev = noise2clean_kvrange[-1]
    else:
        prev = 8
    rows = [[b * range_num + max(0, j + 1 - prev), b * range_num + j + 1] for b in range(batch_size) for j in range(range_num)]
    return _ranges(rows, tokens, device)
