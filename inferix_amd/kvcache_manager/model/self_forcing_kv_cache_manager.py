"""Per-layer Self-Forcing adapter over KVCacheManager — same API as the reference's
inferix/kvcache_manager/model/self_forcing_kv_cache_manager.py:8-217 (layer names `layer_{i}` /
`crossattn_layer_{i}`, block_size = 1, self cache `sequence_length // ring_size` tokens x
`heads // ulysses_size` heads, cross cache `crossattn_length` tokens)."""
from __future__ import annotations

from typing import List, Optional

import torch

from ..kvcache_manager import KVCacheManager, KVCacheRequest, KVCacheRequestSpec, KVCacheSpec


class SelfForcingKVCacheManager:
    def __init__(self, layer_number: int, num_query_groups_per_partition: int,
                 hidden_size_per_attention_head: int, enable_kv_offload: bool = False):
        self.layer_number = layer_number
        self.num_query_groups_per_partition = num_query_groups_per_partition
        self.hidden_size_per_attention_head = hidden_size_per_attention_head
        self.enable_kv_offload = enable_kv_offload

    # names ------------------------------------------------------------------
    @property
    def self_name(self) -> str:
        return f"layer_{self.layer_number}"

    @property
    def cross_name(self) -> str:
        return f"crossattn_layer_{self.layer_number}"

    def _spec(self, heads: int, dtype: torch.dtype) -> KVCacheSpec:
        return KVCacheSpec(num_kv_heads=heads, head_size=self.hidden_size_per_attention_head, dtype=dtype,
                           kv_offload=self.enable_kv_offload, use_mla=False)

    # allocation ---------------------------------------------------------------
    def allocate_kv_cache(self, kv_cache_manager: KVCacheManager, kv_cache_request: KVCacheRequest,
                          sequence_length: int, dtype: torch.dtype, ulysses_size: int = 1, ring_size: int = 1) -> None:
        kv_cache_manager.allocate_slots(kv_cache_request, KVCacheRequestSpec(
            num_tokens=sequence_length // ring_size, block_size=1,
            specs={self.self_name: self._spec(self.num_query_groups_per_partition // ulysses_size, dtype)}))

    def allocate_crossattn_cache(self, kv_cache_manager: KVCacheManager, kv_cache_request: KVCacheRequest,
                                 crossattn_length: int, dtype: torch.dtype) -> None:
        kv_cache_manager.allocate_slots(kv_cache_request, KVCacheRequestSpec(
            num_tokens=crossattn_length, block_size=1,
            specs={self.cross_name: self._spec(self.num_query_groups_per_partition, dtype)}))

    def reset_kv_cache(self, kv_cache_manager, kv_cache_request, device) -> None:   # indices live in kv_cache_meta
        return None

    def reset_crossattn_cache(self, kv_cache_manager, kv_cache_request) -> None:    # flag lives in crossattn_cache_meta
        return None

    # access ---------------------------------------------------------------------
    def get_kv_cache(self, kv_cache_manager: KVCacheManager, kv_cache_request: KVCacheRequest) -> torch.Tensor:
        """(2, tokens, heads, head_dim) view (block_size dim squeezed), as the reference returns."""
        return kv_cache_manager.get(kv_cache_request, self.self_name).squeeze(2)

    def set_kv_cache(self, kv_cache_manager: KVCacheManager, kv_cache_request: KVCacheRequest, start_index: int,
                     k_data: torch.Tensor, v_data: torch.Tensor) -> None:
        kv = torch.stack([k_data, v_data], dim=0).unsqueeze(2)
        kv_cache_manager.set(kv_cache_request, self.self_name, start_index, kv.shape[1], kv)

    def get_crossattn_cache(self, kv_cache_manager: KVCacheManager, kv_cache_request: KVCacheRequest) -> torch.Tensor:
        return kv_cache_manager.get(kv_cache_request, self.cross_name)

    def set_crossattn_cache(self, kv_cache_manager: KVCacheManager, kv_cache_request: KVCacheRequest,
                            k_data: torch.Tensor, v_data: torch.Tensor) -> None:
        kv = torch.stack([k_data, v_data], dim=0).unsqueeze(2)
        kv_cache_manager.set(kv_cache_request, self.cross_name, 0, kv.shape[1], kv)

    def clear_cache(self, kv_cache_manager: KVCacheManager, kv_cache_request: KVCacheRequest) -> None:
        for name in (self.self_name, self.cross_name):
            if name in kv_cache_manager.layers(kv_cache_request):
                kv_cache_manager.free_layer(kv_cache_request, name)

    def get_cache_size(self, kv_cache_manager: KVCacheManager, kv_cache_request: KVCacheRequest) -> Optional[int]:
        if self.is_cached(kv_cache_manager, kv_cache_request):
            return kv_cache_manager.get_raw(kv_cache_request, self.self_name).numel()
        return None

    def is_cached(self, kv_cache_manager: KVCacheManager, kv_cache_request: KVCacheRequest) -> bool:
        return self.self_name in kv_cache_manager.layers(kv_cache_request)


class SelfForcingKVCacheManagerFactory:
    @staticmethod
    def create_manager(layer_number: int, num_query_groups_per_partition: int = 12,
                       hidden_size_per_attention_head: int = 128,
                       enable_kv_offload: bool = False) -> SelfForcingKVCacheManager:
        return SelfForcingKVCacheManager(layer_number, num_query_groups_per_partition,
                                         hidden_size_per_attention_head, enable_kv_offload)

    @staticmethod
    def create_managers(num_layers: int, num_query_groups_per_partition: int = 12,
                        hidden_size_per_attention_head: int = 128,
                        enable_kv_offload: bool = False) -> List[SelfForcingKVCacheManager]:
        return [SelfForcingKVCacheManagerFactory.create_manager(i, num_query_groups_per_partition,
                                                               hidden_size_per_attention_head, enable_kv_offload)
                for i in range(num_layers)]
