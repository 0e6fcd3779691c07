"""Argument bundles of the MAGI attention path — field-for-field mirrors of the reference's dataclasses
(inferix/core/types/inference.py:51-101) so call sites read the same."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from ..kvcache_manager import KVCacheManager, KVCacheRequest


@dataclass(frozen=True)
class PackedCoreAttnParams:
    q_range: torch.Tensor
    k_range: torch.Tensor
    np_q_range: np.ndarray
    np_k_range: np.ndarray
    max_seqlen_q: int
    max_seqlen_k: int


@dataclass(frozen=True)
class PackedCrossAttnParams:
    q_ranges: Optional[torch.Tensor] = None
    kv_ranges: Optional[torch.Tensor] = None
    cu_seqlens_q: Optional[torch.Tensor] = None
    cu_seqlens_kv: Optional[torch.Tensor] = None
    max_seqlen_q: Optional[int] = None
    max_seqlen_kv: Optional[int] = None


@dataclass(frozen=True)
class ModelMetaArgs:
    H: int
    W: int
    cp_pad_size: int
    cp_split_sizes: Optional[List[int]]
    slice_point: int
    denoising_range_num: int
    range_num: int
    extract_prefix_video_feature: bool
    fwd_extra_1st_chunk: bool
    distill_nearly_clean_chunk: bool
    clip_token_nums: int
    enable_cuda_graph: bool
    core_attn_params: Optional[PackedCoreAttnParams]
    cross_attn_params: Optional[PackedCrossAttnParams]


class InferenceParams:
    """inference.py:90-101: per-request cache handle.  The manager lives on the given device (the reference takes
    torch.cuda.current_device())."""

    def __init__(self, max_batch_size: int, max_sequence_length: int, device=None):
        self.max_sequence_length = max_sequence_length
        self.max_batch_size = max_batch_size
        self.sequence_len_offset = 0
        self.kv_cache_request = KVCacheRequest(request_id="magi")
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.kv_cache_manager = KVCacheManager(device=device)
        self.key_value_memory_dict = {}
        self.update_kv_cache = False
