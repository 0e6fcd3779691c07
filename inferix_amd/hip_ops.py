"""Torch-tensor front end of the C-ABI (inferix_amd/_hip.py): pointer/shape marshalling only.

Every function enqueues one HIP kernel of libinferix_hip.so on torch's CURRENT
stream and returns immediately.  Tensors must live on the GPU; there is no CPU
path here by design (the CPU restatement lives in oracle/, for tests only).
"""
from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass
from typing import Sequence, Optional, Tuple

import torch

from . import _hip

BF16 = torch.bfloat16


class KernelTimer:
    """Optional per-launch timing with events recorded on the launch stream (torch's current stream, the one
    every kernel here is enqueued on).  Used by bench.py for the roofline figures; off by default."""

    def __init__(self, names=("attn_self",)):
        self.names = set(names)
        self.records = []            # (name, start_event, end_event, flops, bytes)

    def wants(self, name: str) -> bool:
        return name in self.names

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, s, e, fl, by in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += fl
            d["bytes"] += by
        return out


_timer: Optional[KernelTimer] = None


def set_kernel_timer(t: Optional[KernelTimer]) -> None:
    global _timer
    _timer = t


class _Timing:
    def __init__(self, name, flops, nbytes):
        self.name, self.flops, self.nbytes = name, flops, nbytes

    def __enter__(self):
        self.s = torch.cuda.Event(enable_timing=True)
        self.e = torch.cuda.Event(enable_timing=True)
        self.s.record()

    def __exit__(self, *a):
        self.e.record()
        _timer.records.append((self.name, self.s, self.e, self.flops, self.nbytes))


class _NotTimed:
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOT_TIMED = _NotTimed()


def _timed(name, flops=0.0, nbytes=0.0):
    """Context around one launch: a shared no-op unless a KernelTimer asked for `name` (the wrappers sit on the host's critical
    path when a sequence-parallel rank issues ~17 k launches of 5-50 us per clip)."""
    if _timer is None or name not in _timer.names:
        return _NOT_TIMED
    return _Timing(name, flops, nbytes)


_raw_stream = torch._C._cuda_getCurrentRawStream        # the HIP stream torch would launch on, without building a Stream object
_cur_device = torch._C._cuda_getDevice


def _stream() -> int:
    return _raw_stream(_cur_device())


def _dev(t: torch.Tensor, name: str, dtype=BF16) -> int:
    if not t.is_cuda:
        raise _hip.HipKernelError(f"{name}: tensor is on {t.device}; inferix_amd ops run on the GPU only")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.dim() and t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")
    return t.data_ptr()


def _rows2d(t: torch.Tensor, name: str) -> Tuple[int, int, int]:
    """(rows, cols, row_stride) of a tensor viewed as a row-major matrix with uniform row stride."""
    if t.dim() == 1:
        return 1, t.shape[0], t.shape[0]
    if t.dim() > 2:
        if t.is_contiguous():
            t = t.flatten(0, -2)
        else:
            # the caller passes the ORIGINAL tensor's data pointer: a reshape that had to copy would pair that pointer with the
            # copy's pitch, so only true views (uniformly strided rows) are accepted
            try:
                t = t.view(-1, t.shape[-1])
            except RuntimeError as e:
                raise ValueError(f"{name}: tensor of shape {tuple(t.shape)} / strides {t.stride()} is not a matrix of "
                                 "uniformly strided rows; make it contiguous first") from e
    return t.shape[0], t.shape[1], t.stride(0)


@dataclass
class KvCacheView:
    """One (request, layer) of the paged KV cache: `k`,`v` `[num_slots, heads, head_dim]` bf16 views into
    the KVCacheManager tensor + optional page table (int32, device) — mirrors `ifx_kv_view`."""
    k: torch.Tensor
    v: torch.Tensor
    page_table: Optional[torch.Tensor] = None
    page_size: int = 1
    seg_split: int = 0          # two-segment map (no table): logical tokens >= seg_split live seg_delta slots further on
    seg_delta: int = 0

    def struct(self) -> _hip.KvView:
        # checked and marshalled once per (k, v, page table): a layer's view is reused by every forward
        key = (self.k.data_ptr(), self.v.data_ptr(), None if self.page_table is None else self.page_table.data_ptr(), self.page_size,
               self.seg_split, self.seg_delta)
        memo = self.__dict__.get("_memo")
        if memo is not None and memo[0] == key:
            return memo[1]
        assert self.k.shape == self.v.shape and self.k.dim() == 3 and self.k.is_contiguous() and self.v.is_contiguous()
        pt = 0
        if self.page_table is not None:
            pt = _dev(self.page_table, "page_table", torch.int32)
        assert not (self.seg_split and self.page_table is not None), "a view has a page table or a two-segment map, not both"
        st = _hip.KvView(_dev(self.k, "kv.k"), _dev(self.v, "kv.v"), pt, int(self.page_size),
                         self.k.shape[0], self.k.shape[1], self.k.shape[2], int(self.seg_split), int(self.seg_delta))
        self.__dict__["_memo"] = (key, st)
        return st

    @staticmethod
    def from_manager_tensor(t: torch.Tensor, page_table=None, page_size=1) -> "KvCacheView":
        """`t` is the reference-layout manager tensor (2, num_blocks, block_size, kv_heads, head_dim)
        (kvcache_manager.py:222-244); K and V are zero-copy views of it."""
        assert t.dim() == 5 and t.shape[0] == 2 and t.is_contiguous()
        return KvCacheView(t[0].flatten(0, 1), t[1].flatten(0, 1), page_table, page_size)


@dataclass
class RopeGridSpec:
    freqs: torch.Tensor          # [max_pos, head_dim/2, 2] float64 on device (cos, sin)
    start_frame: int
    height: int
    width: int
    hw_offset: int = 0
    hw_local: Optional[int] = None
    q_scale: float = 0.0         # rmsnorm_rope_kv_append: q is multiplied by this before its rounding to bf16 (0 = 1); see ATTN_Q_PRESCALE

    def struct(self) -> _hip.RopeGrid:
        hl = self.hw_local if self.hw_local is not None else self.height * self.width
        return _hip.RopeGrid(_dev(self.freqs, "rope.freqs", torch.float64), self.freqs.shape[0],
                             int(self.start_frame), int(self.height), int(self.width), int(self.hw_offset), int(hl),
                             float(self.q_scale))


def attn_q_prescale(head_dim: int) -> Tuple[float, float]:
    """(q_scale, attention scale) of the exponent fast path: q carries softmax_scale * log2(e) from the kernel that produces it
    (`RopeGridSpec.q_scale`) and the attention is called with scale = ln 2 — softmax(q k^T / sqrt(d)) as before, with ONE rounding
    of q to bf16 either way, and no scale-FMA per score in the attention loop (ifx_attn_pp.hip, FR = 7)."""
    return (head_dim ** -0.5) * 1.4426950408889634, 0.6931471805599453


def layernorm(x: torch.Tensor, eps: float, *, gamma=None, beta=None, mod=None, shift_slot=0, scale_slot=1,
              rows_per_group: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm / affine LayerNorm / AdaLN-modulated LayerNorm over the last dim (ifx_layernorm)."""
    lib = _hip.load()
    rows, dim, ld = _rows2d(x, "x")
    assert ld == dim, "x rows must be dense"
    out = torch.empty_like(x) if out is None else out
    if mod is not None:
        mode = _hip.IFX_LN_MODULATE
        assert mod.dim() == 3 and mod.shape[-1] == dim and mod.is_contiguous() and rows_per_group
        assert rows <= mod.shape[0] * rows_per_group
        args = (None, None, _dev(mod, "mod"), mod.shape[1], shift_slot, scale_slot, rows_per_group)
    elif gamma is not None:
        mode = _hip.IFX_LN_AFFINE
        args = (_dev(gamma, "gamma"), _dev(beta, "beta"), None, 0, 0, 0, 1)
    else:
        mode = _hip.IFX_LN_PLAIN
        args = (None, None, None, 0, 0, 0, 1)
    with _timed("layernorm", 0.0, 4.0 * rows * dim):
        _hip.check(lib.ifx_layernorm(_dev(x, "x"), _dev(out, "out"), rows, dim, eps, mode, *args, _stream()),
                   "ifx_layernorm")
    return out


def layernorm_quant(x: torch.Tensor, eps: float, fmt: int, *, gamma=None, beta=None, mod=None, shift_slot=0, scale_slot=1,
                    rows_per_group: Optional[int] = None, q: Optional[torch.Tensor] = None,
                    scale: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """`quant_per_token(layernorm(x, ...), fmt)` in one pass (ifx_layernorm_quant): bytes `[rows, dim]` uint8 + scale `[rows]` fp32,
    bit-identical to the two calls."""
    lib = _hip.load()
    rows, dim, ld = _rows2d(x, "x")
    assert ld == dim, "x rows must be dense"
    q = torch.empty(rows, dim, dtype=torch.uint8, device=x.device) if q is None else q
    scale = torch.empty(rows, dtype=torch.float32, device=x.device) if scale is None else scale
    if mod is not None:
        mode = _hip.IFX_LN_MODULATE
        assert mod.dim() == 3 and mod.shape[-1] == dim and mod.is_contiguous() and rows_per_group
        assert rows <= mod.shape[0] * rows_per_group
        args = (None, None, _dev(mod, "mod"), mod.shape[1], shift_slot, scale_slot, rows_per_group)
    elif gamma is not None:
        mode = _hip.IFX_LN_AFFINE
        args = (_dev(gamma, "gamma"), _dev(beta, "beta"), None, 0, 0, 0, 1)
    else:
        mode = _hip.IFX_LN_PLAIN
        args = (None, None, None, 0, 0, 0, 1)
    with _timed("layernorm", 0.0, 3.0 * rows * dim):
        _hip.check(lib.ifx_layernorm_quant(_dev(x, "x"), _dev(q, "q", torch.uint8), q.stride(0), _dev(scale, "scale", torch.float32),
                                           rows, dim, eps, mode, *args, int(fmt), _stream()), "ifx_layernorm_quant")
    return q, scale


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _hip.load()
    rows, dim, ldx = _rows2d(x, "x")
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device) if out is None else out
    _, _, ldy = _rows2d(out, "out")
    _hip.check(lib.ifx_rmsnorm(_dev(x, "x"), ldx, _dev(out, "out"), ldy, _dev(w, "w"), rows, dim, eps, _stream()),
               "ifx_rmsnorm")
    return out


def rmsnorm_rope_kv_append(qkv: torch.Tensor, wq: torch.Tensor, wk: Optional[torch.Tensor], eps: float,
                           rope: Optional[RopeGridSpec], kv: Optional[KvCacheView], local_start: int,
                           dim: int, q_out: Optional[torch.Tensor] = None, v_in_place: bool = False) -> torch.Tensor:
    """q_out = RoPE(RMSNorm(q)*wq); cache[local_start + r] <- (RoPE(RMSNorm(k)*wk), v).  `v_in_place`: the projection has already
    written the V rows into their cache slots (`linear(..., out2=)`): q and K only (the V columns of `qkv` are not read)."""
    lib = _hip.load()
    rows, _, ld = _rows2d(qkv, "qkv")
    q_out = torch.empty(rows, dim, dtype=BF16, device=qkv.device) if q_out is None else q_out
    rs = rope.struct() if rope is not None else None
    if v_in_place:
        assert rs is not None and kv is not None, "v_in_place needs a rope grid (it carries the flag) and a cache view"
        rs.flags = 1
    ks = kv.struct() if kv is not None else None
    nb = 2.0 * rows * dim * ((4 if v_in_place else 6) if kv is not None else 2)      # read q,k,v + write q,K,V  |  read q + write q
    with _timed("rmsnorm_rope_append", 0.0, nb):
        _hip.check(lib.ifx_rmsnorm_rope_kv_append(
            _dev(qkv, "qkv"), ld, _dev(q_out, "q_out"), _dev(wq, "wq"), _dev(wk, "wk") if wk is not None else None,
            C.byref(rs) if rs is not None else None, C.byref(ks) if ks is not None else None,
            int(local_start), rows, dim, eps, _stream()), "ifx_rmsnorm_rope_kv_append")
    return q_out


_OPTIONS: dict = {}          # key -> value last seen in the library (set_option / option_scope refresh it from ifx_get_option)


def get_option(key: str) -> int:
    """The value the LIBRARY holds for an option right now (`ifx_get_option`): whoever set it — this module, another binding of the
    same process, or the environment default."""
    v = C.c_int32(0)
    _hip.check(_hip.load().ifx_get_option(key.encode(), C.byref(v)), "ifx_get_option")
    _OPTIONS[key] = int(v.value)
    if key == "gemm_small_split":
        _TLS.small_split = int(v.value)
    return int(v.value)


_TLS = threading.local()     # gemm_small_split is per host thread in the library (ifx_core.hip); so is its mirror here


def _small_split() -> int:
    v = getattr(_TLS, "small_split", None)
    return get_option("gemm_small_split") if v is None else v


def device_error(clear: bool = True) -> int:
    """0, or the code of the first device-side wait a kernel gave up (`ifx_device_error`; meaningful once the stream has been
    synchronised).  `check_device()` raises on it."""
    return int(_hip.load().ifx_device_error(1 if clear else 0))


def check_device(what: str = "", sync: bool = False) -> None:
    """Raise if a kernel gave up a device-side wait since the last check.  The word is meaningful for work that has completed:
    `sync=True` synchronises the current stream first (the pipelines do this once per clip, before the clip is handed out — a split-K
    consumer that gave up leaves garbage in its tile, and without this check the clip would ship silently corrupt: ADVICE r5).
    On an error the split-K workspaces are dropped before raising: the consumer that gave up zeroed the tile's flags, a late
    producer may have raised one afterwards, and the next launch on that workspace relies on 'all flags zero on entry'."""
    if sync and torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()
    if device_error(clear=False):
        _GEMM_WS.clear()
        msg = _hip.load().ifx_last_error().decode("utf-8", "replace")       # reports and clears the word
        raise _hip.HipKernelError(f"{what + ': ' if what else ''}{msg}")


def set_option(key: str, value: int) -> None:
    """Kernel-selection override (`ifx_set_option`): 'gemm_variant' 0..29, 'attn_variant' 0..7, 'conv_variant' 0/1, 'gemm_small_split' 0/1,
    'spin_timeout_ms', 'spin_fault', 'attn_debug_counters' (include/inferix_hip.h); 0 = choose by shape."""
    _hip.check(_hip.load().ifx_set_option(key.encode(), int(value)), "ifx_set_option")
    _OPTIONS[key] = int(value)
    if key == "gemm_small_split":
        _TLS.small_split = int(value)
    if key == "attn_variant":
        _SPLIT_PLAN.clear()      # the split plan depends on the attention schedule
    if key == "gemm_variant":
        _GEMM_WS_NEED.clear()    # the workspace a GEMM shape asks for depends on the tile selection (small_split is part of the key)


class option_scope:
    """`with option_scope("gemm_small_split", 1): ...` — set a kernel-selection option for the launches enqueued inside and put the
    previous value back afterwards (the selection happens on the host at enqueue time, so the scope covers exactly those launches)."""

    def __init__(self, key: str, value: int):
        self.key, self.value, self.prev = key, int(value), 0

    def __enter__(self):
        self.prev = get_option(self.key)         # the library's value, not a mirror: options set through lib.ifx_set_option count
        if self.prev != self.value:
            set_option(self.key, self.value)
        return self

    def __exit__(self, *exc):
        if self.prev != self.value:
            set_option(self.key, self.prev)
        return False


_ATTN_WS: dict = {}      # (device index, stream) -> fp32 split-KV workspace, grown on demand
_SPLIT_PLAN: dict = {}   # (rows, heads, nkeys) -> (splits, workspace bytes)


def _attn_workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    key = (dev.index, _stream())
    ws = _ATTN_WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
        _ATTN_WS[key] = ws
    return ws


def attention(q: torch.Tensor, kv: KvCacheView, kv_len: int, scale: float = 0.0,
              out: Optional[torch.Tensor] = None, return_lse: bool = False, tag: str = "attn", kv_start: int = 0,
              splits: Optional[int] = None):
    """softmax(q K^T * scale) V over logical cache tokens [kv_start, kv_len) read in place. q `[rows, heads, 128]`.
    `splits`: key chunks run as independent workgroups (None = ifx_attn_split_plan decides; 1 = never split)."""
    lib = _hip.load()
    assert q.dim() == 3 and q.is_contiguous()
    rows, heads, hd = q.shape
    out = torch.empty_like(q) if out is None else out
    lse = torch.empty(heads, rows, dtype=torch.float32, device=q.device) if return_lse else None
    ks = kv.struct()
    d = heads * hd
    nk = kv_len - kv_start
    if splits is None:
        plan = _SPLIT_PLAN.get((rows, heads, nk))
        if plan is None:
            need = C.c_int64(0)
            plan = (int(lib.ifx_attn_split_plan(rows, heads, int(kv_start), int(kv_len), C.byref(need))), int(need.value))
            _SPLIT_PLAN[(rows, heads, nk)] = plan
        splits, ws_bytes = plan
    else:
        ws_bytes = splits * rows * heads * (hd + 1) * 4 if splits > 1 else 0
    lse_p = _dev(lse, "lse", torch.float32) if lse is not None else None
    with _timed(tag, 4.0 * rows * nk * d, 2.0 * (2 * rows * d + 2 * nk * d)):
        if splits > 1:
            ws = _attn_workspace(q.device, ws_bytes)
            _hip.check(lib.ifx_attn_fwd_paged_split(_dev(q, "q"), _dev(out, "out"), lse_p, C.byref(ks), rows, heads,
                                                    int(kv_start), int(kv_len), float(scale), int(splits),
                                                    ws.data_ptr(), ws.numel() * 4, _stream()),
                       "ifx_attn_fwd_paged_split")
        else:
            _hip.check(lib.ifx_attn_fwd_paged(_dev(q, "q"), _dev(out, "out"), lse_p, C.byref(ks), rows, heads,
                                              int(kv_start), int(kv_len), float(scale), _stream()),
                       "ifx_attn_fwd_paged")
    return (out, lse) if return_lse else out


def attention_split_plan(rows: int, heads: int, nkeys: int) -> int:
    """Key-chunk count `ifx_attn_split_plan` recommends for this launch shape (1 = do not split)."""
    plan = _SPLIT_PLAN.get((rows, heads, nkeys))
    if plan is None:
        need = C.c_int64(0)
        plan = (int(_hip.load().ifx_attn_split_plan(rows, heads, 0, int(nkeys), C.byref(need))), int(need.value))
        _SPLIT_PLAN[(rows, heads, nkeys)] = plan
    return plan[0]


def attention_partial(q: torch.Tensor, kv: KvCacheView, kv_len: int, kv_start: int, splits: int, workspace: torch.Tensor,
                      slot_base: int, slot_cap: int, scale: float = 0.0, tag: str = "attn") -> int:
    """fp32 partials of the key chunks of [kv_start, kv_len) into `workspace` slots from `slot_base`; returns the number
    of slots written.  Finish with `attention_merge`."""
    rows, heads, hd = q.shape
    ks = kv.struct()
    used = C.c_int32(0)
    nk = kv_len - kv_start
    with _timed(tag, 4.0 * rows * nk * heads * hd, 2.0 * (2 * rows * heads * hd + 2 * nk * heads * hd)):
        _hip.check(_hip.load().ifx_attn_fwd_partial(_dev(q, "q"), C.byref(ks), rows, heads, int(kv_start), int(kv_len),
                                                    float(scale), int(splits), workspace.data_ptr(), workspace.numel() * 4,
                                                    int(slot_base), int(slot_cap), C.byref(used), _stream()),
                   "ifx_attn_fwd_partial")
    return int(used.value)


def attention_merge(workspace: torch.Tensor, slot_cap: int, slots_used: int, out: torch.Tensor,
                    lse: Optional[torch.Tensor] = None) -> None:
    rows, heads, _ = out.shape
    _hip.check(_hip.load().ifx_attn_merge_partials(workspace.data_ptr(), int(slot_cap), int(slots_used), _dev(out, "out"),
                                                   _dev(lse, "lse", torch.float32) if lse is not None else None, rows, heads,
                                                   _stream()), "ifx_attn_merge_partials")


def attention_workspace(q: torch.Tensor, slot_cap: int) -> torch.Tensor:
    rows, heads, hd = q.shape
    return _attn_workspace(q.device, slot_cap * rows * heads * (hd + 1) * 4)


def kv_scatter_shards(gathered: torch.Tensor, world: int, frames: int, hw_local: int, frame_tokens: int,
                      local_start: int, kv: KvCacheView) -> None:
    """All-gathered `[world, 2, frames*hw_local, heads, 128]` K/V rows -> cache slots in the (frame, hw) token order."""
    assert gathered.is_contiguous() and gathered.numel() == world * 2 * frames * hw_local * kv.k.shape[-2] * kv.k.shape[-1]
    ks = kv.struct()
    with _timed("kv_scatter", 0.0, 2.0 * gathered.numel() * 2):
        _hip.check(_hip.load().ifx_kv_scatter_shards(_dev(gathered, "gathered"), world, frames, hw_local, frame_tokens,
                                                     int(local_start), C.byref(ks), _stream()), "ifx_kv_scatter_shards")


# ---- peer-to-peer K/V exchange (ifx_peer.hip): device memory other processes map, stream-ordered flags, the push kernel ----
class PeerBuffer:
    """A device allocation of libinferix_hip.so that other processes can map (`ifx_peer_alloc`); `fine_grained` for flag blocks."""

    def __init__(self, nbytes: int, fine_grained: bool = False):
        p = C.c_void_p()
        _hip.check(_hip.load().ifx_peer_alloc(int(nbytes), 1 if fine_grained else 0, C.byref(p)), "ifx_peer_alloc")
        self.ptr, self.nbytes = int(p.value), int(nbytes)

    def free(self) -> None:
        if self.ptr:
            _hip.check(_hip.load().ifx_peer_free(self.ptr), "ifx_peer_free")
            self.ptr = 0

    def tensor(self, dtype: torch.dtype, shape) -> torch.Tensor:
        """Zero-copy torch view of the allocation (the buffer must outlive it)."""
        typestr = {torch.int32: "<i4", torch.bfloat16: "<u2", torch.uint8: "|u1", torch.float32: "<f4"}[dtype]
        holder = type("_Cai", (), {"__cuda_array_interface__": {"shape": tuple(shape), "typestr": typestr, "data": (self.ptr, False),
                                                                 "version": 2}, "_keep": self})()
        t = torch.as_tensor(holder, device="cuda")
        return t.view(torch.bfloat16) if dtype == torch.bfloat16 else t


def peer_export(ptr: int) -> Tuple[bytes, int]:
    """(64-byte IPC handle of the allocation `ptr` lies in, offset of `ptr` in it)."""
    h = C.create_string_buffer(_hip.IFX_PEER_HANDLE_BYTES)
    off = C.c_int64()
    _hip.check(_hip.load().ifx_peer_export(int(ptr), h, C.byref(off)), "ifx_peer_export")
    return h.raw, int(off.value)


def peer_open(handle: bytes) -> int:
    p = C.c_void_p()
    _hip.check(_hip.load().ifx_peer_open(C.create_string_buffer(handle, _hip.IFX_PEER_HANDLE_BYTES), C.byref(p)), "ifx_peer_open")
    return int(p.value)


def peer_close(ptr: int) -> None:
    _hip.check(_hip.load().ifx_peer_close(int(ptr)), "ifx_peer_close")


def rmsnorm_rope_kv_push(kv_rows: torch.Tensor, wk: torch.Tensor, eps: float, rope: Optional[RopeGridSpec], dest_k: Sequence[int],
                         dest_v: Sequence[int], geometry: KvCacheView, local_start: int, frame_tokens: int, slot_hw_local: int,
                         slot_hw_offset: int, dim: int) -> None:
    """K <- RoPE(RMSNorm(k) * wk), V raw, from rows `[rows, (k | v)]`, stored at the cache slots of the rows' logical tokens in every
    destination (device addresses of caches with `geometry`'s layout: the peers' caches, or one staging buffer)."""
    rows, cols, ld = _rows2d(kv_rows, "kv_rows")
    assert cols >= 2 * dim and len(dest_k) == len(dest_v) and 1 <= len(dest_k) <= _hip.IFX_MAX_PEERS
    pc = _hip.PeerCaches()
    pc.count = len(dest_k)
    for i, (a, b) in enumerate(zip(dest_k, dest_v)):
        pc.k[i], pc.v[i] = int(a), int(b)
    rs = rope.struct() if rope is not None else None
    gs = geometry.struct()
    with _timed("kv_push", 0.0, 2.0 * rows * dim * 2 * (1 + len(dest_k))):
        _hip.check(_hip.load().ifx_rmsnorm_rope_kv_push(
            _dev(kv_rows, "kv_rows"), ld, _dev(wk, "wk"), C.byref(rs) if rs is not None else None, C.byref(pc), C.byref(gs),
            int(local_start), int(frame_tokens), int(slot_hw_local), int(slot_hw_offset), rows, dim, eps, _stream()),
            "ifx_rmsnorm_rope_kv_push")


def peer_signal(flag_blocks: Sequence[int], index: int, value: int) -> None:
    """flag_blocks[p][index] <- value for every peer p, ordered after the work already on the current stream."""
    pf = _hip.PeerFlags()
    pf.count = len(flag_blocks)
    for i, a in enumerate(flag_blocks):
        pf.flags[i] = int(a)
    _hip.check(_hip.load().ifx_peer_signal(C.byref(pf), int(index), int(value), _stream()), "ifx_peer_signal")


def peer_wait(flags_ptr: int, count: int, value: int, timeout_ms: int, status: Optional[torch.Tensor]) -> None:
    """The current stream waits until flags[i] >= value for all i < count (a timeout stores 1 + i into `status`)."""
    _hip.check(_hip.load().ifx_peer_wait(int(flags_ptr), int(count), int(value), int(timeout_ms),
                                         _dev(status, "status", torch.int32) if status is not None else None, _stream()), "ifx_peer_wait")


def lse_merge(out_a: torch.Tensor, lse_a: torch.Tensor, out_b: torch.Tensor, lse_b: torch.Tensor) -> None:
    lib = _hip.load()
    rows, heads, _ = out_a.shape
    _hip.check(lib.ifx_lse_merge(_dev(out_a, "out_a"), _dev(lse_a, "lse_a", torch.float32), _dev(out_b, "out_b"),
                                 _dev(lse_b, "lse_b", torch.float32), rows, heads, _stream()), "ifx_lse_merge")


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], *, epilogue: int = _hip.IFX_EPI_BIAS,
           residual: Optional[torch.Tensor] = None, mod: Optional[torch.Tensor] = None, gate_slot: int = 0,
           rows_per_group: int = 1, out: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None,
           split_col: int = 0) -> torch.Tensor:
    """y = epilogue(x @ w.T + bias) on MFMA (ifx_gemm_bf16). x `[..., K]`, w `[N, K]`.  `out2` `[M, N - split_col]` (ifx_epilogue.y2):
    the output columns from `split_col` on go THERE instead of `out` — the q|k|v projection storing V straight into the cache rows."""
    lib = _hip.load()
    M, K, ldx = _rows2d(x, "x")
    N = w.shape[0]
    assert w.shape[1] == K and w.is_contiguous()
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=BF16, device=x.device)
    _, _, ldy = _rows2d(out, "out")
    epi = _hip.Epilogue(epilogue, None, 0, None, 1, 0, 1)
    if residual is not None:
        _, _, ldr = _rows2d(residual, "residual")
        epi.residual, epi.ld_res = _dev(residual, "residual"), ldr
    if mod is not None:
        assert mod.dim() == 3 and mod.shape[-1] == N and mod.is_contiguous()
        assert rows_per_group > 0 and M <= mod.shape[0] * rows_per_group, "mod table does not cover every output row"
        epi.mod, epi.mod_slots, epi.gate_slot, epi.rows_per_group = _dev(mod, "mod"), mod.shape[1], gate_slot, rows_per_group
    if out2 is not None:
        r2, c2, ld2 = _rows2d(out2, "out2")
        assert r2 == M and c2 == N - split_col and 0 < split_col < N, (out2.shape, M, N, split_col)
        epi.y2, epi.ldy2, epi.split_col = _dev(out2, "out2"), ld2, int(split_col)
    wkey = (M, N, K, _small_split())
    need = _GEMM_WS_NEED.get(wkey) if out2 is None else 0          # the two-destination launch takes no workspace (bias epilogue, unsplit)
    if need is None:
        need = _GEMM_WS_NEED[wkey] = int(lib.ifx_gemm_workspace_bytes(M, N, K))
    with _timed("gemm", 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N * (2 if residual is not None else 1))):   # x + W + y (+ the residual rows)
        if need:
            ws = _gemm_workspace(x.device, need)
            _hip.check(lib.ifx_gemm_bf16_ws(_dev(x, "x"), ldx, _dev(w, "w"), _dev(bias, "bias") if bias is not None else None,
                                            _dev(out, "out"), ldy, M, N, K, C.byref(epi), ws.data_ptr(), ws.numel(), _stream()),
                       "ifx_gemm_bf16_ws")
        else:
            _hip.check(lib.ifx_gemm_bf16(_dev(x, "x"), ldx, _dev(w, "w"), _dev(bias, "bias") if bias is not None else None,
                                         _dev(out, "out"), ldy, M, N, K, C.byref(epi), _stream()), "ifx_gemm_bf16")
    return out


_GEMM_WS_NEED: dict = {}     # (M, N, K, small_split) -> bytes ifx_gemm_workspace_bytes asks for (0 = none)
_GEMM_WS: dict = {}          # (device index, stream) -> zero-initialised scratch for the split-K tiles, grown on demand


def _gemm_workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    key = (dev.index, _stream())
    ws = _GEMM_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)      # the arrival counters in front must start at zero
        _GEMM_WS[key] = ws
    return ws


def quant_per_token(x: torch.Tensor, fmt: int, q: Optional[torch.Tensor] = None,
                    scale: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-token dynamic quantisation of `[rows, K]` bf16 -> (bytes `[rows, K]` uint8, scale `[rows]` fp32)."""
    lib = _hip.load()
    rows, K, ldx = _rows2d(x, "x")
    q = torch.empty(rows, K, dtype=torch.uint8, device=x.device) if q is None else q
    scale = torch.empty(rows, dtype=torch.float32, device=x.device) if scale is None else scale
    with _timed("quant_per_token", 0.0, 3.0 * rows * K):
        _hip.check(lib.ifx_quant_per_token(_dev(x, "x"), ldx, _dev(q, "q", torch.uint8), q.stride(0),
                                           _dev(scale, "scale", torch.float32), rows, K, fmt, _stream()),
                   "ifx_quant_per_token")
    return q, scale


def linear_q8(xq: torch.Tensor, x_scale: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor,
              bias: Optional[torch.Tensor], fmt: int, *, epilogue: int = _hip.IFX_EPI_BIAS,
              residual: Optional[torch.Tensor] = None, mod: Optional[torch.Tensor] = None, gate_slot: int = 0,
              rows_per_group: int = 1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = epilogue(bf16((xq @ wq.T) * (x_scale ⊗ w_scale) + bias)) on fp8 / int8 MFMA (ifx_gemm_q8)."""
    lib = _hip.load()
    M, K = xq.shape
    N = wq.shape[0]
    assert wq.shape[1] == K and wq.is_contiguous() and xq.stride(1) == 1
    out = torch.empty(M, N, dtype=BF16, device=xq.device) if out is None else out
    _, _, ldy = _rows2d(out, "out")
    epi = _hip.Epilogue(epilogue, None, 0, None, 1, 0, 1)
    if residual is not None:
        _, _, ldr = _rows2d(residual, "residual")
        epi.residual, epi.ld_res = _dev(residual, "residual"), ldr
    if mod is not None:
        assert mod.dim() == 3 and mod.shape[-1] == N and mod.is_contiguous()
        assert rows_per_group > 0 and M <= mod.shape[0] * rows_per_group, "mod table does not cover every output row"
        epi.mod, epi.mod_slots, epi.gate_slot, epi.rows_per_group = _dev(mod, "mod"), mod.shape[1], gate_slot, rows_per_group
    wkey = ("q8", M, N, K, _small_split())
    need = _GEMM_WS_NEED.get(wkey)
    if need is None:
        need = _GEMM_WS_NEED[wkey] = int(lib.ifx_gemm_q8_workspace_bytes(M, N, K))
    with _timed("gemm_q8", 2.0 * M * N * K, 1.0 * (M * K + N * K) + 2.0 * M * N):
        if need:       # long-K, narrow-N launches: the 256-token tile with K split between two workgroups (ifx_gemm_q8_ws)
            ws = _gemm_workspace(xq.device, need)
            _hip.check(lib.ifx_gemm_q8_ws(_dev(xq, "xq", torch.uint8), xq.stride(0), _dev(x_scale, "x_scale", torch.float32),
                                          _dev(wq, "wq", torch.uint8), _dev(w_scale, "w_scale", torch.float32),
                                          _dev(bias, "bias") if bias is not None else None, _dev(out, "out"), ldy, M, N, K,
                                          fmt, C.byref(epi), ws.data_ptr(), ws.numel(), _stream()), "ifx_gemm_q8_ws")
        else:
            _hip.check(lib.ifx_gemm_q8(_dev(xq, "xq", torch.uint8), xq.stride(0), _dev(x_scale, "x_scale", torch.float32),
                                       _dev(wq, "wq", torch.uint8), _dev(w_scale, "w_scale", torch.float32),
                                       _dev(bias, "bias") if bias is not None else None, _dev(out, "out"), ldy, M, N, K,
                                       fmt, C.byref(epi), _stream()), "ifx_gemm_q8")
    return out


def linear_q8_quant_out(xq: torch.Tensor, x_scale: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor, fmt: int, out_divisor: torch.Tensor,
                        *, epilogue: int = _hip.IFX_EPI_GELU_ERF, via_bf16: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """e4m3 bytes `[M, N]` = quant_static(gelu(bf16((xq @ wq.T) * scales)), out_divisor) in the GEMM's epilogue (ifx_gemm_q8_quant_out)."""
    M, K = xq.shape
    N = wq.shape[0]
    assert wq.shape[1] == K and wq.is_contiguous() and xq.stride(1) == 1 and out_divisor.numel() == N
    out = torch.empty(M, N, dtype=torch.uint8, device=xq.device) if out is None else out
    epi = _hip.Epilogue(epilogue, None, 0, None, 1, 0, 1)
    with _timed("gemm_q8", 2.0 * M * N * K, 1.0 * (M * K + N * K) + 1.0 * M * N):
        _hip.check(_hip.load().ifx_gemm_q8_quant_out(_dev(xq, "xq", torch.uint8), xq.stride(0), _dev(x_scale, "x_scale", F32),
                                                     _dev(wq, "wq", torch.uint8), _dev(w_scale, "w_scale", F32), None,
                                                     _dev(out, "out", torch.uint8), out.stride(0), M, N, K, int(fmt), C.byref(epi),
                                                     _dev(out_divisor, "out_divisor", F32), 1 if via_bf16 else 0, _stream()),
                   "ifx_gemm_q8_quant_out")
    return out


def layernorm_quant_static(x: torch.Tensor, eps: float, divisors: torch.Tensor, gamma: Optional[torch.Tensor] = None,
                           beta: Optional[torch.Tensor] = None, via_bf16: bool = True) -> torch.Tensor:
    """LayerNorm (plain / affine) of `[rows, dim]` bf16 + `n_out` static e4m3 quantisations of its bf16 result, one per divisor
    vector of `divisors` `[n_out, dim]` fp32 -> bytes `[rows, n_out, dim]` (ifx_layernorm_quant_static)."""
    rows, dim, ldx = _rows2d(x, "x")
    assert ldx == dim, "layernorm_quant_static: dense rows"
    n_out = divisors.shape[0]
    assert divisors.shape == (n_out, dim) and divisors.is_contiguous()
    q = torch.empty(rows, n_out, dim, dtype=torch.uint8, device=x.device)
    mode = _hip.IFX_LN_AFFINE if gamma is not None else _hip.IFX_LN_PLAIN
    with _timed("layernorm", 0.0, (2.0 + n_out) * rows * dim):
        _hip.check(_hip.load().ifx_layernorm_quant_static(_dev(x, "x"), _dev(q, "q", torch.uint8), n_out * dim, _dev(divisors, "divisors", F32),
                                                          n_out, rows, dim, float(eps), mode, _dev(gamma, "gamma") if gamma is not None else None,
                                                          _dev(beta, "beta") if beta is not None else None, 1 if via_bf16 else 0, _stream()),
                   "ifx_layernorm_quant_static")
    return q


def kv_roll(kv: KvCacheView, sink_tokens: int, evicted: int, rolled: int, scratch: torch.Tensor) -> None:
    lib = _hip.load()
    ks = kv.struct()
    assert scratch.numel() >= rolled * kv.k.shape[1] * kv.k.shape[2]
    _hip.check(lib.ifx_kv_roll(C.byref(ks), sink_tokens, evicted, rolled, _dev(scratch, "scratch"), _stream()),
               "ifx_kv_roll")


# ---- VAE decoder ops (channels-last frames) ----------------------------------------------------------------------
_ZERO_PAGE = {}


def _zero_page(dev: torch.device) -> torch.Tensor:
    z = _ZERO_PAGE.get(dev)
    if z is None:
        z = _ZERO_PAGE[dev] = torch.zeros(256, dtype=torch.uint8, device=dev)
    return z


def _slots(v) -> "C.Array":
    return (C.c_int32 * len(v))(*[int(i) for i in v])


def to_planar(x: torch.Tensor) -> torch.Tensor:
    """channels-last frames `[t, h, w, c]` (c % 32 == 0) -> the 32-channel-plane layout `[t, c/32, h, w, 32]` of
    `ifx_conv3d_desc.in_planar` (a copy)."""
    t, h, w, c = x.shape
    return x.view(t, h, w, c // 32, 32).permute(0, 3, 1, 2, 4).contiguous()


def conv3d_cl(x: torch.Tensor, in_slots, w: torch.Tensor, bias: Optional[torch.Tensor], *, kt: int, ks: int,
              y: torch.Tensor, out_slots, upsample: bool = False, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Causal conv on channels-last frames (ifx_conv3d_cl).  `x` `[slots, hs, ws, cin]` — or, 5-D, the planar layout
    `[slots, cin/32, hs, ws, 32]` (`to_planar`, `rmsnorm_cl(out_planar=True)`) — logical input frame f at
    `x[in_slots[f]]` (negative slot = zero frame; `t_out + kt - 1` entries); `w` `[kt*ks*ks, cin/32, cout, 32]`; output frame t
    goes to `y[out_slots[t]]` `[ho, wo, cout]`; `residual` `[t_out, ho, wo, cout]`."""
    t_out = len(out_slots)
    assert len(in_slots) == t_out + kt - 1, (len(in_slots), t_out, kt)
    in_planar = x.dim() == 5
    if in_planar:
        _, planes, hs, ws, c32 = x.shape
        assert c32 == 32
        cin = planes * 32
    else:
        _, hs, ws, cin = x.shape
    taps, cc, cout, c32 = w.shape
    assert taps == kt * ks * ks and c32 == 32 and cc * 32 == cin and x.is_contiguous() and y.is_contiguous() and w.is_contiguous()
    ho, wo = (hs * 2, ws * 2) if upsample else (hs, ws)
    assert tuple(y.shape[1:]) == (ho, wo, cout), (tuple(y.shape), ho, wo, cout)
    assert max(out_slots) < y.shape[0] and max(in_slots) < x.shape[0]
    if residual is not None:
        assert residual.is_contiguous() and tuple(residual.shape) == (t_out, ho, wo, cout)
    ins, outs = _slots(in_slots), _slots(out_slots)
    d = _hip.Conv3dDesc(_dev(x, "x"), hs * ws * cin, ins, hs, ws, cin, 1 if upsample else 0, _dev(w, "w"),
                        _dev(bias, "bias") if bias is not None else None, kt, ks, _dev(y, "y"), ho * wo * cout, outs, cout,
                        t_out, _dev(residual, "residual") if residual is not None else None,
                        _zero_page(x.device).data_ptr(), 1 if in_planar else 0)
    flops = 2.0 * t_out * ho * wo * cout * cin * taps
    with _timed("conv3d", flops, 0.0):
        _hip.check(_hip.load().ifx_conv3d_cl(C.byref(d), _stream()), "ifx_conv3d_cl")
    return y


def rmsnorm_cl(x: torch.Tensor, gamma: torch.Tensor, y: torch.Tensor, out_slots, silu: bool) -> torch.Tensor:
    """Per-pixel channel RMS norm (+ SiLU) of `x` `[frames, h, w, c]` into `y[out_slots[f]]` (ifx_rmsnorm_cl); a 5-D `y`
    `[slots, c/32, h, w, 32]` takes the frames in the planar layout (IFX_NORM_OUT_PLANAR)."""
    frames, h, w_, c = x.shape
    out_planar = y.dim() == 5
    assert x.is_contiguous() and y.is_contiguous() and len(out_slots) == frames
    assert tuple(y.shape[1:]) == ((c // 32, h, w_, 32) if out_planar else (h, w_, c)), (tuple(y.shape), tuple(x.shape))
    assert gamma.numel() == c and max(out_slots) < y.shape[0]
    with _timed("rmsnorm_cl", 0.0, 4.0 * x.numel()):
        _hip.check(_hip.load().ifx_rmsnorm_cl(_dev(x, "x"), _dev(gamma, "gamma"), _dev(y, "y"), h * w_ * c, _slots(out_slots),
                                              frames, h * w_, c, (1 if silu else 0) | (2 if out_planar else 0), _stream()), "ifx_rmsnorm_cl")
    return y


def softmax_rows(scores: torch.Tensor, scale: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(scores * scale) over the last dim of a 2-D bf16 tensor whose row pitch is a multiple of 8."""
    rows, cols = scores.shape
    ld = scores.stride(0)
    out = torch.empty_like(scores) if out is None else out
    assert scores.stride(1) == 1 and out.stride() == scores.stride()
    _hip.check(_hip.load().ifx_softmax_rows(_dev(scores, "scores"), _dev(out, "probs"), rows, cols, ld, float(scale),
                                            _stream()), "ifx_softmax_rows")
    return out


# ---- umT5 text encoder ops -----------------------------------------------------------------------------------------
def t5_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, rel_bias: torch.Tensor, seq_lens: torch.Tensor,
                 batch: int, heads: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """T5 self-attention (ifx_t5_attention): q/k/v `[batch*L, heads*64]` (row-strided views of a fused qkv output are fine),
    `rel_bias` `[heads, 2L-1]`, `seq_lens` int32 `[batch]` on the device."""
    rows = q.shape[0]
    L = rows // batch
    assert rows == batch * L and q.shape[1] == heads * 64 and k.shape == q.shape and v.shape == q.shape
    assert tuple(rel_bias.shape) == (heads, 2 * L - 1) and rel_bias.is_contiguous()
    assert seq_lens.dtype == torch.int32 and seq_lens.is_cuda and seq_lens.numel() == batch
    out = torch.empty(rows, heads * 64, dtype=BF16, device=q.device) if out is None else out
    with _timed("attn_t5", 4.0 * batch * heads * L * L * 64, 0.0):
        _hip.check(_hip.load().ifx_t5_attention(_dev(q, "q"), q.stride(0), _dev(k, "k"), k.stride(0), _dev(v, "v"), v.stride(0),
                                                _dev(out, "out"), out.stride(0), _dev(rel_bias, "rel_bias"),
                                                seq_lens.data_ptr(), batch, L, heads, _stream()), "ifx_t5_attention")
    return out


def t5_gated_gelu(gate_fc1: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`fc1 * GELU(gate)` from the fused `[rows, 2*ffn]` GEMM output (ifx_t5_gated_gelu)."""
    rows, two_ffn = gate_fc1.shape
    ffn = two_ffn // 2
    assert gate_fc1.is_contiguous()
    out = torch.empty(rows, ffn, dtype=BF16, device=gate_fc1.device) if out is None else out
    _hip.check(_hip.load().ifx_t5_gated_gelu(_dev(gate_fc1, "gate_fc1"), _dev(out, "h"), rows, ffn, _stream()),
               "ifx_t5_gated_gelu")
    return out


# ---- strided attention, MAGI layer ops, static / per-tensor quantisers -------------------------------------------------
def attention_ld(q: torch.Tensor, kv: KvCacheView, kv_len: int, out: torch.Tensor, heads: int, kv_start: int = 0,
                 scale: float = 0.0, tag: str = "attn") -> torch.Tensor:
    """`attention` for query / output rows that are column blocks of wider matrices (ifx_attn_fwd_paged_ld): `q` and `out` are
    2-D `[rows, >= heads*128]` views whose row strides are passed on; the result lands in `out[:, :heads*128]`."""
    lib = _hip.load()
    rows = q.shape[0]
    assert q.dim() == 2 and out.dim() == 2 and out.shape[0] == rows and q.stride(1) == 1 and out.stride(1) == 1
    assert q.shape[1] >= heads * 128 and out.shape[1] >= heads * 128
    ks = kv.struct()
    nk = kv_len - kv_start
    key = (rows, heads, nk)
    plan = _SPLIT_PLAN.get(key)
    if plan is None:
        need = C.c_int64(0)
        plan = (int(lib.ifx_attn_split_plan(rows, heads, int(kv_start), int(kv_len), C.byref(need))), int(need.value))
        _SPLIT_PLAN[key] = plan
    splits, ws_bytes = plan
    ws = _attn_workspace(q.device, ws_bytes) if splits > 1 else None
    d = heads * 128
    with _timed(tag, 4.0 * rows * nk * d, 2.0 * (2 * rows * d + 2 * nk * kv.k.shape[1] * 128)):
        _hip.check(lib.ifx_attn_fwd_paged_ld(_dev(q, "q"), q.stride(0), _dev(out, "out"), out.stride(0), None, C.byref(ks), rows,
                                             heads, int(kv_start), int(kv_len), float(scale), int(splits),
                                             ws.data_ptr() if ws is not None else None,
                                             ws.numel() * 4 if ws is not None else 0, _stream()), "ifx_attn_fwd_paged_ld")
    return out


F32 = torch.float32


def magi_head_prep(mixed: torch.Tensor, *, layout: int, q_heads: int, kv_heads: int, eps: float, layernorm_1p: bool,
                   k_out: torch.Tensor, v_out: torch.Tensor, kv_head_stride: int, ld_kv: int, row0: int = 0,
                   split: Optional[int] = None, row1: int = 0, rope: Optional[torch.Tensor] = None, qn=None, kn=None, xn=None,
                   q_out: Optional[torch.Tensor] = None, qx_out: Optional[torch.Tensor] = None, q_scale: float = 0.0,
                   q_group: int = 0) -> None:
    """ifx_magi_head_prep: per-head LayerNorm (+ rotary) of the fused projection row and the k / v scatter (include/inferix_hip.h).
    `q_group` > 0: `q_out` is `[q_heads / q_group, rows, q_group * 128]` — the head -> rank all-to-all's send order — instead of `[rows, q_heads * 128]`.
    `qn` / `kn` = (weight, bias) fp32 `[128]`; `xn` = (weight, bias) bf16 `[128]`; `k_out` / `v_out` are base tensors of the
    destination (cache planes or a staging buffer), addressed dest(r) * ld_kv + head * kv_head_stride."""
    rows = mixed.shape[0]
    assert mixed.dim() == 2 and mixed.stride(1) == 1
    d = _hip.MagiHeadPrepDesc()
    d.inp, d.ld_in, d.rows, d.layout = _dev(mixed, "mixed"), mixed.stride(0), rows, layout
    d.q_heads, d.kv_heads, d.head_dim = q_heads, kv_heads, 128
    d.eps, d.layernorm_1p = float(eps), 1 if layernorm_1p else 0
    if layout == 0:
        assert rope is not None and rope.is_contiguous() and rope.dim() == 2 and rope.shape[0] == rows
        assert rope.shape[1] % 16 == 0 and rope.shape[1] <= 128, "rope rows are (sin | cos) of at most 64 pairs, a multiple of 8 each"
        d.rope = _dev(rope, "rope", F32)
        d.rope_half = rope.shape[1] // 2            # partial rotary: MAGI's table covers 96 of the 128 head channels
        d.qn_w, d.qn_b = _dev(qn[0], "q_layernorm.weight", F32), _dev(qn[1], "q_layernorm.bias", F32)
        d.kn_w, d.kn_b = _dev(kn[0], "k_layernorm.weight", F32), _dev(kn[1], "k_layernorm.bias", F32)
        if q_group > 0:
            assert q_out.dim() == 3 and q_out.is_contiguous() and q_out.shape == (q_heads // q_group, rows, q_group * 128), q_out.shape
            d.q_out, d.ld_q, d.q_group, d.q_group_stride = _dev(q_out, "q_out"), q_group * 128, int(q_group), rows * q_group * 128
        else:
            d.q_out, d.ld_q = _dev(q_out, "q_out"), q_out.stride(0)
            assert q_out.shape[0] >= rows
        d.qx_out, d.ld_qx = _dev(qx_out, "qx_out"), qx_out.stride(0)
        assert qx_out.shape[0] >= rows
    d.xn_w, d.xn_b = _dev(xn[0], "layernorm_xattn.weight"), _dev(xn[1], "layernorm_xattn.bias")
    d.k_out, d.v_out, d.ld_kv, d.kv_head_stride = _dev(k_out, "k_out"), _dev(v_out, "v_out"), int(ld_kv), int(kv_head_stride)
    d.row0, d.split, d.row1 = int(row0), int(rows if split is None else split), int(row1)
    d.q_scale = float(q_scale)                  # self-attention q pre-multiplied (attn_q_prescale); 0 = 1
    n_heads = (2 * q_heads + 2 * kv_heads) if layout == 0 else 2 * kv_heads
    with _timed("magi_head_prep", 0.0, 4.0 * rows * n_heads * 128):
        _hip.check(_hip.load().ifx_magi_head_prep(C.byref(d), _stream()), "ifx_magi_head_prep")


def kv_split_rows(kv: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, row0: int, split: int, row1: int) -> None:
    """`kv` `[n, heads, 256]` (K | V per head) -> rows of the cache planes `[slots, heads, 128]`: row r to `row0 + r` for r < split, to
    `row1 + r - split` behind it (ifx_kv_split_rows: MagiKVCacheManager's store rule in one launch)."""
    n, heads, hd2 = kv.shape
    assert hd2 == 256 and kv.is_contiguous() and k_cache.is_contiguous() and v_cache.is_contiguous()
    assert k_cache.shape[1:] == (heads, 128) and v_cache.shape == k_cache.shape
    assert 0 <= split <= n and row0 + split <= k_cache.shape[0] and row1 + (n - split) <= k_cache.shape[0]
    _hip.check(_hip.load().ifx_kv_split_rows(_dev(kv, "kv"), _dev(k_cache, "k_cache"), _dev(v_cache, "v_cache"), n, heads, int(row0),
                                             int(split), int(row1), _stream()), "ifx_kv_split_rows")


def magi_gate_norm_residual(x: torch.Tensor, residual: torch.Tensor, condition_map: torch.Tensor, gate: torch.Tensor,
                            norm_w: torch.Tensor, norm_b: torch.Tensor, eps: float, layernorm_1p: bool,
                            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bias_modulate_add in one pass (ifx_magi_gate_norm_residual).  `gate` `[groups, dim]` view (row-strided is fine)."""
    rows, dim, ldx = _rows2d(x, "x")
    _, _, ldr = _rows2d(residual, "residual")
    out = torch.empty(rows, dim, dtype=BF16, device=x.device) if out is None else out
    _, _, ldy = _rows2d(out, "out")
    assert gate.dim() == 2 and gate.shape[1] == dim and gate.stride(1) == 1
    assert condition_map.dtype == torch.int32 and condition_map.is_cuda and condition_map.numel() == rows and condition_map.is_contiguous()
    with _timed("magi_gate_norm", 0.0, 6.0 * rows * dim):
        _hip.check(_hip.load().ifx_magi_gate_norm_residual(
            _dev(x, "x"), ldx, _dev(residual, "residual"), ldr, condition_map.data_ptr(), _dev(gate, "gate"), gate.stride(0),
            _dev(norm_w, "norm_w", F32), _dev(norm_b, "norm_b", F32), 1 if layernorm_1p else 0, _dev(out, "out"), ldy, rows, dim,
            float(eps), _stream()), "ifx_magi_gate_norm_residual")
    return out


def act_rows(x: torch.Tensor, mode: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """SiLU (mode IFX_ACT_SILU) / tanh (IFX_ACT_TANH) of a contiguous bf16 tensor (ifx_act_rows)."""
    assert x.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    _hip.check(_hip.load().ifx_act_rows(_dev(x, "x"), _dev(out, "out"), x.numel(), int(mode), _stream()), "ifx_act_rows")
    return out


def quant_static(x: torch.Tensor, divisor: torch.Tensor, fmt: int, via_bf16: bool, q: Optional[torch.Tensor] = None,
                 row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q = cast(clamp(x / divisor)) with a per-input-channel (`[K]`) or single (`[1]`) fp32 divisor (ifx_quant_static);
    `via_bf16` = the bf16 intermediate of the reference's div_clamp_to."""
    rows, K, ldx = _rows2d(x, "x")
    q = torch.empty(rows, K, dtype=torch.uint8, device=x.device) if q is None else q
    divisor = divisor.reshape(-1)
    with _timed("quant_static", 0.0, 3.0 * rows * K):
        _hip.check(_hip.load().ifx_quant_static(_dev(x, "x"), ldx, _dev(q, "q", torch.uint8), q.stride(0), _dev(divisor, "divisor", F32),
                                                divisor.numel(), _dev(row_scale, "row_scale", F32) if row_scale is not None else None,
                                                rows, K, int(fmt), 1 if via_bf16 else 0, _stream()), "ifx_quant_static")
    return q


_AMAX_WS: dict = {}


def quant_per_tensor(x: torch.Tensor, fmt: int, q: Optional[torch.Tensor] = None,
                     scale: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Dynamic per-tensor quantisation (ifx_quant_per_tensor): bytes `[rows, K]` + the tensor's scale repeated per row `[rows]`."""
    rows, K, ldx = _rows2d(x, "x")
    q = torch.empty(rows, K, dtype=torch.uint8, device=x.device) if q is None else q
    scale = torch.empty(rows, dtype=F32, device=x.device) if scale is None else scale
    key = (x.device.index, _stream())
    ws = _AMAX_WS.get(key)
    if ws is None:
        ws = _AMAX_WS[key] = torch.zeros(1, dtype=torch.int32, device=x.device)
    _hip.check(_hip.load().ifx_quant_per_tensor(_dev(x, "x"), ldx, _dev(q, "q", torch.uint8), q.stride(0), _dev(scale, "scale", F32),
                                                ws.data_ptr(), rows, K, int(fmt), _stream()), "ifx_quant_per_tensor")
    return q, scale


def attention_ranges(q: torch.Tensor, kv: KvCacheView, q_ranges, k_ranges, out: torch.Tensor, heads: int, scale: float = 0.0,
                     tag: str = "attn") -> torch.Tensor:
    """Several (query range, key range) pairs in ONE launch (ifx_attn_fwd_ranges; MAGI core_attention).  `q` / `out` 2-D views
    `[rows, >= heads*128]` (row strides are passed on); ranges are host sequences of (start, end)."""
    n = len(q_ranges)
    assert 1 <= n <= 8 and len(k_ranges) == n and q.dim() == 2 and out.dim() == 2 and q.stride(1) == 1 and out.stride(1) == 1
    rows = q.shape[0]
    qa = (C.c_int32 * (2 * n))(*[int(v) for r in q_ranges for v in r])
    ka = (C.c_int32 * (2 * n))(*[int(v) for r in k_ranges for v in r])
    ks = kv.struct()
    d = heads * 128
    flops = sum(4.0 * (qe - qs) * (ke - kb) * d for (qs, qe), (kb, ke) in zip(q_ranges, k_ranges))
    with _timed(tag, flops, 0.0):
        _hip.check(_hip.load().ifx_attn_fwd_ranges(_dev(q, "q"), q.stride(0), _dev(out, "out"), out.stride(0), C.byref(ks), rows, heads,
                                                   n, qa, ka, float(scale), _stream()), "ifx_attn_fwd_ranges")
    return out


def attention_dedup(q: torch.Tensor, kv: KvCacheView, kv_len: int, last_key_multiplicity: int, out: Optional[torch.Tensor] = None,
                    scale: float = 0.0, tag: str = "attn") -> torch.Tensor:
    """Attention over keys [0, kv_len) whose LAST key stands for `last_key_multiplicity` identical rows (ifx_attn_fwd_dedup)."""
    assert q.dim() == 3 and q.is_contiguous()
    rows, heads, hd = q.shape
    out = torch.empty_like(q) if out is None else out
    ks = kv.struct()
    nk = kv_len - 1 + last_key_multiplicity
    with _timed(tag, 4.0 * rows * nk * heads * hd, 0.0):
        _hip.check(_hip.load().ifx_attn_fwd_dedup(_dev(q, "q"), _dev(out, "out"), C.byref(ks), rows, heads, int(kv_len),
                                                  int(last_key_multiplicity), float(scale), _stream()), "ifx_attn_fwd_dedup")
    return out
