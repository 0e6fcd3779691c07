"""Attention operator with the reference's call contract, backed by the HIP paged flash-attention kernel.

* `attention(q, k, v, ...)`  — inferix/models/attention/flash_attention.py:153-200: q `[B, Lq, H, D]`,
  k/v `[B, Lk, H, D]` -> `[B, Lq, H, D]` bf16, scale 1/sqrt(D) by default, no mask.
* `collect_supported_attn()` — inferix/models/attention/backends.py:154-166: `{name: fn}` where
  `fn(q, k, v, dropout_p, softmax_scale, causal, window_size, ...) -> (out [B, L, H, D], lse [B, H, L])`.
  The backend registered here is `"HipPagedFA"`.

Dense k/v tensors are wrapped as single-page caches (zero copy) so the same kernel serves both the
registry contract and the in-place paged path used by the model.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import hip_ops as ops


def _check_plain(dropout_p, causal, window_size, q_lens, k_lens):
    if dropout_p:
        raise NotImplementedError("HipPagedFA: dropout is not supported (inference kernel)")
    if causal:
        raise NotImplementedError("HipPagedFA: block causality is realised by the cache contents; "
                                  "token-level causal masks are not built")
    if tuple(window_size) != (-1, -1):
        raise NotImplementedError("HipPagedFA: sliding windows are realised by cache eviction")
    if q_lens is not None or k_lens is not None:
        raise NotImplementedError("HipPagedFA: varlen batches are not built (pass dense tensors)")


def hip_paged_fa_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, dropout_p: float = 0.0,
                         softmax_scale: Optional[float] = None, causal: bool = False,
                         window_size: Tuple[int, int] = (-1, -1), **_unused) -> Tuple[torch.Tensor, torch.Tensor]:
    """Registry backend: `(out [B, Lq, H, D], lse [B, H, Lq])` (fp32 natural-log LSE)."""
    _check_plain(dropout_p, causal, window_size, None, None)
    B = q.shape[0]
    outs, lses = [], []
    for b in range(B):
        kv = ops.KvCacheView(k[b].to(torch.bfloat16).contiguous(), v[b].to(torch.bfloat16).contiguous())
        o, l = ops.attention(q[b].to(torch.bfloat16).contiguous(), kv, k.shape[1], scale=softmax_scale or 0.0,
                             return_lse=True)
        outs.append(o)
        lses.append(l)
    return torch.stack(outs), torch.stack(lses)


def attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0.0, softmax_scale=None, q_scale=None, causal=False,
              window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, fa_version=None) -> torch.Tensor:
    _check_plain(dropout_p, causal, window_size, q_lens, k_lens)
    if q_scale is not None:
        q = q * q_scale
    out, _ = hip_paged_fa_forward(q, k, v, softmax_scale=softmax_scale)
    return out


flash_attention = attention


def collect_supported_attn() -> Dict[str, callable]:
    return {"HipPagedFA": hip_paged_fa_forward}
