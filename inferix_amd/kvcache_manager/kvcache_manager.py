"""KV-cache manager with the reference's Python API (inferix/kvcache_manager/kvcache_manager.py:21-244),
backed by device-resident HIP allocations that the attention kernel reads IN PLACE.

Same classes, method names, tensor layout and error behaviour as the reference:
one tensor per (request, layer) shaped `(coef, num_blocks, block_size, kv_heads, head_size)`
with `coef = 2` (K, V) or 1 (MLA); `allocate_slots` on an existing layer raises ValueError
(`:103-109`), `free` of an unknown request raises KeyError (`:117`).

MI355X-first differences (all behind the same API):
  * caches live in HBM (288 GB): `kv_offload=True` is honoured (pinned host tensor, as the
    reference does for 16-24 GB GPUs) but nothing in this package requests it;
  * `get()` of a device-resident layer returns the tensor itself (`.to(device)` is the identity),
    so models borrow views and kernels read pages in place: no stack / copy-back per layer;
  * optional per-layer page table (`page_table()` / `rotate_pages()`): logical token t lives in
    physical slot `table[t // page_size] * page_size + t % page_size`, which turns the sink +
    rolling eviction shift (causal_model.py:287-292) into an O(pages) table rotation.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import Callable, Dict, KeysView, List, Optional, Sequence, Union

import torch


def cdiv(a: int, b: int) -> int:
    return -(-a // b)


def align(a: int, b: int) -> int:
    return cdiv(a, b) * b


def get_dtype_size(dtype: torch.dtype) -> int:
    return torch.empty((), dtype=dtype).element_size()


@dataclass(frozen=True)
class KVCacheRequest:
    request_id: str


@dataclass(frozen=True)
class KVCacheSpec:
    num_kv_heads: int
    head_size: int
    dtype: torch.dtype
    kv_offload: bool
    use_mla: bool


@dataclass
class KVCacheRequestSpec:
    num_tokens: int
    block_size: int
    specs: Dict[str, KVCacheSpec]


@dataclass(frozen=True)
class KVCacheTensorSpec:
    size: int
    num_tokens: int
    num_blocks: int
    block_size: int
    spec: KVCacheSpec


@dataclass
class KVCaches:
    tensors: Dict[str, torch.Tensor]
    specs: Dict[str, KVCacheTensorSpec]
    page_tables: Dict[str, "PageTable"] = field(default_factory=dict)
    views: Dict[str, object] = field(default_factory=dict)     # per layer: zero-copy K / V views built by the HIP model


@dataclass
class PageTable:
    """logical page -> physical page of one layer's cache (int32, host master + device copy)."""
    page_size: int
    host: torch.Tensor
    device: torch.Tensor

    def sync(self) -> None:
        self.device.copy_(self.host, non_blocking=True)


class KVCacheManager:
    # managers constructed so far in this process.  The pipelines build a fresh manager per call and reuse request ids ("req_0",
    # "stream_req_0": pipeline/self_forcing.py), so an allocation is only named uniquely together with its manager's serial — which,
    # like the per-layer generation, is a pure function of the call sequence and therefore the same on every rank of an SPMD run.
    _serials = itertools.count(1)

    def __init__(self, device: Union[str, torch.device, int]):
        self.device = torch.device(device)
        self.serial = next(KVCacheManager._serials)
        self._free_listeners: List[tuple] = []   # (fn, takes_layer): called on `free` / `free_layer` (peer-mapping owners)
        self.offload_device = torch.device("cpu")
        self.request_to_kv_caches: Dict[str, KVCaches] = {}
        self._generation: Dict[tuple, int] = {}      # (request id, layer) -> how many times that layer has been allocated

    # ---- allocation -------------------------------------------------------
    def allocate_slots(self, req: KVCacheRequest, spec: KVCacheRequestSpec) -> KVCaches:
        num_blocks = cdiv(spec.num_tokens, spec.block_size)
        tokens = align(spec.num_tokens, spec.block_size)
        caches = self.request_to_kv_caches.setdefault(req.request_id, KVCaches(tensors={}, specs={}))
        for name in spec.specs:
            if name in caches.tensors or name in caches.specs:
                raise ValueError(f"Layer {name} already exists")
        for name, s in spec.specs.items():
            coef = 1 if s.use_mla else 2
            caches.specs[name] = KVCacheTensorSpec(
                size=coef * tokens * s.num_kv_heads * s.head_size * get_dtype_size(s.dtype),
                num_tokens=tokens, num_blocks=num_blocks, block_size=spec.block_size, spec=s)
            self._generation[(req.request_id, name)] = self._generation.get((req.request_id, name), 0) + 1
            caches.tensors[name] = torch.empty(
                (coef, num_blocks, spec.block_size, s.num_kv_heads, s.head_size), dtype=s.dtype,
                device=self.offload_device if s.kv_offload else self.device,
                pin_memory=bool(s.kv_offload) and torch.cuda.is_available())
        return caches

    def free(self, req: KVCacheRequest) -> None:
        del self.request_to_kv_caches[req.request_id]
        self._notify_free(req.request_id)

    def add_free_listener(self, fn: Callable[..., None]) -> None:
        """`fn(request_id)` runs when a request is freed, `fn(request_id, layer_name)` when ONE of its layers is (listeners that take a
        single argument get `fn(request_id)` for both): the sequence-parallel peer-store exchange drops its IPC address book for exactly
        that (request, layer) there (inferix_amd.sequence_parallel.PeerStoreExchange.forget)."""
        if any(f == fn for f, _ in self._free_listeners):
            return
        import inspect
        try:
            params = list(inspect.signature(fn).parameters.values())
            takes_layer = len(params) >= 2 or any(p.kind == p.VAR_POSITIONAL for p in params)
        except (TypeError, ValueError):          # builtins without a signature: the one-argument form
            takes_layer = False
        self._free_listeners.append((fn, takes_layer))

    def _notify_free(self, request_id: str, layer_name: Optional[str] = None) -> None:
        for fn, takes_layer in list(self._free_listeners):
            if layer_name is not None and takes_layer:
                fn(request_id, layer_name)
            else:
                fn(request_id)

    def free_layer(self, req: KVCacheRequest, layer_name: str) -> None:
        c = self.request_to_kv_caches[req.request_id]
        del c.tensors[layer_name]
        del c.specs[layer_name]
        c.page_tables.pop(layer_name, None)
        c.views.pop(layer_name, None)
        self._notify_free(req.request_id, layer_name)

    def allocation_id(self, req: KVCacheRequest, layer_name: str) -> tuple:
        """(request id, layer, (manager serial, generation)): names ONE allocation of a layer's cache in this process.  A data pointer
        does not — the caching allocator hands a freed address out again — and neither does (request, layer, generation) alone: the
        pipelines make a new manager per call with the same request ids.  Serial and generation are pure functions of the call
        sequence, so the ranks of an SPMD run agree on them (what inferix_amd.sequence_parallel keys its peer mappings by)."""
        if layer_name not in self.request_to_kv_caches[req.request_id].tensors:
            raise KeyError(layer_name)
        return (req.request_id, layer_name, (self.serial, self._generation[(req.request_id, layer_name)]))

    # ---- lookup -----------------------------------------------------------
    def layers(self, req: KVCacheRequest) -> Union[KeysView[str], Sequence[str]]:
        c = self.request_to_kv_caches.get(req.request_id)
        return c.tensors.keys() if c is not None else ()

    def get_raw(self, req: KVCacheRequest, layer_name: str) -> torch.Tensor:
        return self.request_to_kv_caches[req.request_id].tensors[layer_name]

    def layer_spec(self, req: KVCacheRequest, layer_name: str) -> KVCacheTensorSpec:
        return self.request_to_kv_caches[req.request_id].specs[layer_name]

    def get(self, req: KVCacheRequest, layer_name: str) -> torch.Tensor:
        return self.get_raw(req, layer_name).to(self.device)

    def get_range_raw(self, req: KVCacheRequest, layer_name: str, start: int, length: int) -> torch.Tensor:
        return self.get_raw(req, layer_name)[:, start:start + length, ...]

    def get_range(self, req: KVCacheRequest, layer_name: str, start: int, length: int) -> torch.Tensor:
        t = self.get_raw(req, layer_name)
        if self.layer_spec(req, layer_name).spec.use_mla:
            return t[0:1, start:start + length, ...].to(self.device, non_blocking=True)
        out = torch.empty((t.shape[0], length, *t.shape[2:]), dtype=t.dtype, device=self.device)
        out.copy_(t[:, start:start + length, ...], non_blocking=True)
        return out

    def select(self, req: KVCacheRequest, layer_name: str, block_indices: List[int]) -> torch.Tensor:
        return self.get_raw(req, layer_name)[:, block_indices, ...].to(self.device)

    def set(self, req: KVCacheRequest, layer_name: str, start: int, size: int, new_kv: torch.Tensor) -> None:
        assert len(new_kv) == 2
        t = self.get_raw(req, layer_name)
        t[0, start:start + size, ...] = new_kv[0]
        if not self.layer_spec(req, layer_name).spec.use_mla:
            t[1, start:start + size, ...] = new_kv[1]

    # ---- paging (extension; identity when never requested) ------------------
    def page_table(self, req: KVCacheRequest, layer_name: str) -> Optional[PageTable]:
        return self.request_to_kv_caches[req.request_id].page_tables.get(layer_name)

    def enable_paging(self, req: KVCacheRequest, layer_name: str, page_size: int) -> PageTable:
        c = self.request_to_kv_caches[req.request_id]
        spec = c.specs[layer_name]
        if spec.num_tokens % page_size:
            raise ValueError(f"page_size {page_size} does not divide cache tokens {spec.num_tokens}")
        host = torch.arange(spec.num_tokens // page_size, dtype=torch.int32)
        pt = PageTable(page_size, host, host.to(self.device))
        c.page_tables[layer_name] = pt
        return pt

    def rotate_pages(self, req: KVCacheRequest, layer_name: str, sink_pages: int, evicted_pages: int,
                     rolled_pages: int) -> None:
        """Logical shift `[sink+evicted, sink+evicted+rolled) -> [sink, sink+rolled)`; the evicted
        physical pages are recycled behind the rolled span (where the new block will be written)."""
        pt = self.page_table(req, layer_name)
        a, e, r = sink_pages, evicted_pages, rolled_pages
        h = pt.host
        seg = torch.cat([h[a + e:a + e + r], h[a:a + e]]).clone()
        h[a:a + e + r] = seg
        pt.sync()
