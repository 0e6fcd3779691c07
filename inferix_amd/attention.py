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


def _lens(lens, B: int, full: int, what: str):
    if lens is None:
        return [full] * B
    vals = [int(t) for t in (lens.tolist() if isinstance(lens, torch.Tensor) else lens)]
    if len(vals) != B or any(t < 0 or t > full for t in vals):
        raise ValueError(f"{what} must hold one length in [0, {full}] per batch sample, got {vals}")
    return vals


def hip_paged_fa_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, dropout_p: float = 0.0,
                         softmax_scale: Optional[float] = None, causal: bool = False,
                         window_size: Tuple[int, int] = (-1, -1), k_lens=None, **_unused) -> Tuple[torch.Tensor, torch.Tensor]:
    """Registry backend: `(out [B, Lq, H, D], lse [B, H, Lq])` (fp32 natural-log LSE).  `k_lens` `[B]`: sample b attends its first
    `k_lens[b]` keys (the kernel's per-launch key count; nothing is packed or copied)."""
    _check_plain(dropout_p, causal, window_size, None, None)
    B = q.shape[0]
    klens = _lens(k_lens, B, k.shape[1], "k_lens")
    outs, lses = [], []
    for b in range(B):
        qb = q[b].to(torch.bfloat16).contiguous()
        if klens[b] == 0:            # flash-attn's varlen kernels return zeros (and lse = -inf) for a sample without keys
            outs.append(torch.zeros_like(qb))
            lses.append(torch.full((qb.shape[1], qb.shape[0]), float("-inf"), dtype=torch.float32, device=qb.device))
            continue
        kv = ops.KvCacheView(k[b].to(torch.bfloat16).contiguous(), v[b].to(torch.bfloat16).contiguous())
        o, l = ops.attention(qb, kv, klens[b], scale=softmax_scale or 0.0, return_lse=True)
        outs.append(o)
        lses.append(l)
    return torch.stack(outs), torch.stack(lses)


def attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0.0, softmax_scale=None, q_scale=None, causal=False,
              window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, fa_version=None) -> torch.Tensor:
    """flash_attention.py:42-150 with its varlen arguments: `k_lens[b]` keys per sample (the padded text context of
    WanT2VCrossAttention, wan_base/model.py:90-95).  `q_lens`: the reference packs `q[b, :q_lens[b]]` and unflattens the result to
    `[B, Lq]`, which only works when every length IS Lq — anything else raises there too (RuntimeError from unflatten); here it is a
    ValueError that says so."""
    _check_plain(dropout_p, causal, window_size, q_lens, k_lens)
    B, Lq = q.shape[0], q.shape[1]
    if any(t != Lq for t in _lens(q_lens, B, Lq, "q_lens")):
        raise ValueError(f"q_lens must all equal Lq = {Lq}: the reference unflattens the packed result to [B, Lq] (flash_attention.py:141)")
    if q_scale is not None:
        q = q * q_scale
    out, _ = hip_paged_fa_forward(q, k, v, softmax_scale=softmax_scale, k_lens=k_lens)
    return out.to(q.dtype)


flash_attention = attention


def collect_supported_attn() -> Dict[str, callable]:
    return {"HipPagedFA": hip_paged_fa_forward}
