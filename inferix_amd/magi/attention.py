"""MAGI core attention on MI355X: range attention over an in-place KV cache, grouped-query heads.

Replaces, for the Ulysses path of `FullyParallelAttention.forward` (inferix/models/magi/dit/dit_module.py:1087-1195):
  * `MagiKVCacheManager.adjust_key_and_value_for_inference` / `_full_adjust_key_and_value`
    (inferix/kvcache_manager/model/magi_kv_cache_manager.py:76-187).  Upstream reads the clean prefix out of the cache,
    concatenates it with the new keys/values and returns fresh tensors every layer, every step.  Here the new rows are
    written into the same allocation as the prefix and attention reads the whole range IN PLACE through the paged view
    of `ifx_attn_fwd_paged`: rows that the reference's rule stores go to their final slots, rows it does not store
    (read-only calls, the nearly-clean last chunk) go to a scratch tail of the allocation that the page table maps
    behind the prefix.  The stored cache contents are bit-identical to upstream's.
  * `core_attention` (dit_module.py:975-1018): per denoising range i, queries [q_range[i]) attend keys [k_range[i]),
    no mask inside a range, grouped-query heads (MAGI-4.5B at cp = 8: 3 query heads on 1 kv head per rank).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch

from .. import hip_ops as ops
from ..kvcache_manager import KVCacheRequestSpec, KVCacheSpec
from .types import InferenceParams, ModelMetaArgs


@dataclass
class MagiKvHandle:
    """What `adjust_key_and_value_for_inference` hands to `core_attention`: a cache view and the number of valid
    logical keys (prefix + new rows)."""
    view: ops.KvCacheView
    kv_len: int
    kv_heads: int

    def materialize(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(key, value) `[kv_len, hn, hd]` as upstream returns them (tests / debugging; the hot path never calls this)."""
        t = torch.arange(self.kv_len, device=self.view.k.device)
        if self.view.page_table is not None:
            ps = self.view.page_size
            t = self.view.page_table.long()[t // ps] * ps + t % ps
        elif self.view.seg_split:
            t = torch.where(t >= self.view.seg_split, t + self.view.seg_delta, t)
        return self.view.k[t].contiguous(), self.view.v[t].contiguous()


class MagiKVCacheManager:
    """Per-layer adapter over `KVCacheManager` with the reference's constructor and method names."""

    def __init__(self, layer_number: int, num_query_groups_per_partition: int, hidden_size_per_attention_head: int,
                 engine_config=None):
        self.layer_number = layer_number
        self.num_query_groups_per_partition = num_query_groups_per_partition
        self.hidden_size_per_attention_head = hidden_size_per_attention_head
        self.engine_config = engine_config
        self._scratch_rows = 0
        self._maps: Dict[Tuple[int, int, int], torch.Tensor] = {}

    @property
    def layer_name(self) -> str:
        return f"layer_{self.layer_number}"

    def allocate_key_value_memory(self, inference_params: InferenceParams, sequence_length: int, batch_size: int,
                                  dtype: torch.dtype, scratch_rows: int = 0) -> None:
        """`sequence_length` cache tokens as upstream (block_size 1) + `scratch_rows` unstored rows behind them."""
        if getattr(self.engine_config, "kv_offload", False):
            raise NotImplementedError("kv_offload parks the cache in host memory for 24 GB GPUs; the HIP kernels read it "
                                      "in place from HBM (288 GB): build the engine config with kv_offload=False")
        spec = KVCacheRequestSpec(num_tokens=sequence_length + scratch_rows, block_size=1, specs={
            self.layer_name: KVCacheSpec(num_kv_heads=self.num_query_groups_per_partition,
                                         head_size=self.hidden_size_per_attention_head, dtype=dtype, kv_offload=False,
                                         use_mla=False)})
        inference_params.kv_cache_manager.allocate_slots(inference_params.kv_cache_request, spec)
        self._scratch_rows = scratch_rows

    def is_cached(self, inference_params: InferenceParams) -> bool:
        return self.layer_name in inference_params.kv_cache_manager.layers(inference_params.kv_cache_request)

    def clear_cache(self, inference_params: InferenceParams) -> None:
        inference_params.kv_cache_manager.free_layer(inference_params.kv_cache_request, self.layer_name)
        self._maps.clear()

    def get_cache_size(self, inference_params: InferenceParams) -> Optional[int]:
        if not self.is_cached(inference_params):
            return None
        return inference_params.kv_cache_manager.get_raw(inference_params.kv_cache_request, self.layer_name).numel()

    # -----------------------------------------------------------------------------------------------------------
    def _token_map(self, device, start: int, stored: int, unstored: int, capacity: int) -> Optional[torch.Tensor]:
        """logical key j -> physical row: prefix and stored rows are in place, unstored rows live in the scratch tail."""
        if unstored == 0:
            return None
        key = (start, stored, unstored)
        m = self._maps.get(key)
        if m is None:
            m = torch.cat([torch.arange(start + stored, dtype=torch.int32),
                           torch.arange(capacity, capacity + unstored, dtype=torch.int32)]).to(device)
            self._maps[key] = m
        return m

    def prepare_append(self, n: int, hn: int, hd: int, dtype: torch.dtype, device, inference_params: Optional[InferenceParams],
                       meta_args: ModelMetaArgs) -> Tuple[MagiKvHandle, Tuple[int, int, int]]:
        """Where the `n` new key / value rows of this forward go, decided BEFORE they are computed, so that the kernel that
        produces them (ifx_magi_head_prep) writes them in place: returns the handle attention reads and `(row0, split, row1)` —
        row r lands in plane row `row0 + r` if `r < split` (the rows the reference's rule stores: final cache slots) and in
        `row1 + r - split` otherwise (the scratch tail behind the cache).  Without cache involvement the planes are fresh."""
        use_cache = inference_params is not None and (meta_args.extract_prefix_video_feature or
                                                      meta_args.fwd_extra_1st_chunk or meta_args.slice_point > 0)
        if not use_cache:
            k = torch.empty(n, hn, hd, dtype=dtype, device=device)
            v = torch.empty(n, hn, hd, dtype=dtype, device=device)
            return MagiKvHandle(ops.KvCacheView(k, v), n, hn), (0, n, 0)
        ip = inference_params
        B = ip.max_batch_size
        if not self.is_cached(ip):
            self.allocate_key_value_memory(ip, ip.max_sequence_length, B, dtype, scratch_rows=max(self._scratch_rows, n))
        raw = ip.kv_cache_manager.get_raw(ip.kv_cache_request, self.layer_name)     # (2, tokens, 1, hn, hd)
        capacity = ip.max_sequence_length
        if raw.shape[1] - capacity < n:                  # scratch tail too small for this forward: grow it (rare)
            keep = raw[:, :capacity].clone()
            self.clear_cache(ip)
            self.allocate_key_value_memory(ip, capacity, B, dtype, scratch_rows=n)
            raw = ip.kv_cache_manager.get_raw(ip.kv_cache_request, self.layer_name)
            raw[:, :capacity].copy_(keep)
        start = meta_args.slice_point * meta_args.clip_token_nums * B
        stored = 0
        if ip.update_kv_cache:
            stored = n - meta_args.clip_token_nums * B if meta_args.distill_nearly_clean_chunk else n
            assert start + stored <= capacity, "KV cache overflow"
        kc, vc = raw[0, :, 0], raw[1, :, 0]
        # logical keys [0, start + stored) are in place, the n - stored unstored rows sit in the scratch tail at [capacity, ...):
        # a two-segment map (no per-token table: the attention kernel's paged address path does one dependent load per key row)
        unstored = n - stored
        if unstored and start + stored == 0:
            # nothing in place in front of the unstored rows (forward_3cfg's first pass: update_kv_cache=False with
            # slice_point = chunk_start - 1 = 0): seg_split == 0 means "no map" to every reader, so the view is simply BASED at the
            # scratch tail — logical key j == row j of the tail, rows written at (row0, split, row1) = (0, 0, 0) of that view
            return MagiKvHandle(ops.KvCacheView(kc[capacity:], vc[capacity:]), n, hn), (0, 0, 0)
        view = ops.KvCacheView(kc, vc, None, 1, start + stored if unstored else 0, capacity - (start + stored) if unstored else 0)
        return MagiKvHandle(view, start + n, hn), (start, stored, capacity)

    def adjust_key_and_value_for_inference(self, key_and_value: torch.Tensor, inference_params: Optional[InferenceParams],
                                           meta_args: ModelMetaArgs) -> MagiKvHandle:
        """`key_and_value` `[n, hn, 2*hd]` (K | V on the last dim, after the all-to-all).  Stores what the reference's
        rule stores and returns the handle attention reads: logical keys = prefix `[0, slice_point*clip*B)` + the n new."""
        n, hn, hd2 = key_and_value.shape
        hd = hd2 // 2
        handle, (row0, split, row1) = self.prepare_append(n, hn, hd, key_and_value.dtype, key_and_value.device,
                                                          inference_params, meta_args)
        kc, vc = handle.view.k, handle.view.v
        if (key_and_value.is_cuda and hd == 128 and key_and_value.dtype == torch.bfloat16 and key_and_value.is_contiguous()
                and kc.is_contiguous() and vc.is_contiguous() and kc.dim() == 3):
            ops.kv_split_rows(key_and_value, kc, vc, row0, split, row1)          # one launch (round 5) for the four copies below
            return handle
        if split:
            kc[row0:row0 + split].copy_(key_and_value[:split, :, :hd])
            vc[row0:row0 + split].copy_(key_and_value[:split, :, hd:])
        if n - split:
            kc[row1:row1 + n - split].copy_(key_and_value[split:, :, :hd])
            vc[row1:row1 + n - split].copy_(key_and_value[split:, :, hd:])
        return handle


def core_attention(query: torch.Tensor, key, value, bs: int, meta_args: ModelMetaArgs,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Range attention (dit_module.py:975-1018, bs == 1 path): `query` `[(sq b), hq, hd]`; `key` a `MagiKvHandle`
    (`value` is then ignored) or plain `[sk, hk, hd]` tensors.  One `ifx_attn_fwd_paged` launch per denoising range,
    keys read in place; grouped-query heads are resolved inside the kernel."""
    if bs != 1:
        raise NotImplementedError("MAGI runs the cached path with batch 1 (3-cfg converts ranges to batch upstream)")
    handle = key if isinstance(key, MagiKvHandle) else MagiKvHandle(ops.KvCacheView(key.contiguous(), value.contiguous()),
                                                                     key.shape[0], key.shape[1])
    q_range = meta_args.core_attn_params.np_q_range
    k_range = meta_args.core_attn_params.np_k_range
    out = torch.empty_like(query) if out is None else out
    for i in range(meta_args.denoising_range_num):
        qs, qe = int(q_range[i, 0]), int(q_range[i, 1])
        ks, ke = int(k_range[i, 0]), int(k_range[i, 1])
        if ke > handle.kv_len:
            raise ValueError(f"k_range[{i}] = [{ks}, {ke}) exceeds the {handle.kv_len} available keys")
        ops.attention(query[qs:qe], handle.view, ke, out=out[qs:qe], kv_start=ks, tag="attn_magi")
    return out
