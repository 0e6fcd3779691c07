"""umT5-XXL text encoder on MI355X: what stands between a prompt switch and the first block.

Drop-in for `WanTextEncoder` (inferix/models/self_forcing/wrapper.py:15-59) and the `T5Encoder` it wraps
(inferix/models/wan_base/text_encoder/t5.py:269-314): the reference's state-dict keys, `forward(text_prompts) ->
{"prompt_embeds": [B, 512, 4096]}` with the rows past each prompt's length zeroed, bf16 like the pipelines' `model.to(bfloat16)`.
The tokenizer (HuggingFace sentencepiece files, `tokenizer.py:38-72`) is host-side string work and is injected.

How it runs here: per layer 4 GEMM launches (fused q|k|v, o + residual, fused gate|fc1, fc2 + residual: `ifx_gemm_bf16`), 2 RMS
norms (`ifx_rmsnorm`, the same fp32-statistics / bf16-scale chain as `T5LayerNorm`), one `ifx_t5_attention` (K and V^T of a head
staged in LDS, relative-position bias from a 2L-1 table instead of the reference's [1, 64, L, L] tensor per layer) and one
`ifx_t5_gated_gelu`.  All prompts of a batch go through the linears as one [B*L, dim] matrix; the 9.3 GB of weights stay in HBM.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _hip
from . import hip_ops as ops

BF16 = torch.bfloat16


def relative_position_table(emb: torch.Tensor, L: int, num_buckets: int, max_dist: int = 128) -> torch.Tensor:
    """`T5RelativeEmbedding` (t5.py:235-266, bidirectional) as a table over relative offsets: out[h, d + L - 1] = bias of
    key - query = d.  The bucket arithmetic is the reference's, op for op (fp32 log), evaluated once per layer on 2L-1 offsets."""
    rel = torch.arange(-(L - 1), L)
    nb = num_buckets // 2
    buckets = (rel > 0).long() * nb
    a = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(a.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    buckets = buckets + torch.where(a < max_exact, a, large)
    return emb[buckets.to(emb.device)].t().contiguous()              # [heads, 2L-1]


def synthetic_t5_state_dict(vocab_size: int = 256384, dim: int = 4096, dim_attn: int = 4096, dim_ffn: int = 10240,
                            num_heads: int = 64, num_layers: int = 24, num_buckets: int = 32, seed: int = 0,
                            device="cpu") -> Dict[str, torch.Tensor]:
    """Random encoder weights with the reference's keys and init scales (t5.py:29-45) — benchmarks / smoke (no checkpoint here)."""
    g = torch.Generator(device=device).manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, device=device)
    hd = dim_attn // num_heads
    sd = {"token_embedding.weight": r(vocab_size, dim).to(BF16), "norm.weight": (1 + 0.1 * r(dim)).to(BF16)}
    for i in range(num_layers):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = (1 + 0.1 * r(dim)).to(BF16)
        sd[p + "norm2.weight"] = (1 + 0.1 * r(dim)).to(BF16)
        sd[p + "attn.q.weight"] = (r(dim_attn, dim) * (dim * hd) ** -0.25).to(BF16)
        sd[p + "attn.k.weight"] = (r(dim_attn, dim) * dim ** -0.5).to(BF16)
        sd[p + "attn.v.weight"] = (r(dim_attn, dim) * dim ** -0.5).to(BF16)
        sd[p + "attn.o.weight"] = (r(dim, dim_attn) * dim_attn ** -0.5).to(BF16)
        sd[p + "ffn.gate.0.weight"] = (r(dim_ffn, dim) * dim ** -0.5).to(BF16)
        sd[p + "ffn.fc1.weight"] = (r(dim_ffn, dim) * dim ** -0.5).to(BF16)
        sd[p + "ffn.fc2.weight"] = (r(dim, dim_ffn) * dim_ffn ** -0.5).to(BF16)
        sd[p + "pos_embedding.embedding.weight"] = (0.5 * r(num_buckets, num_heads)).to(BF16)
    return sd


class HipT5Encoder:
    """`T5Encoder` (shared_pos=False) on the HIP kernels.  `forward(ids, mask) -> [B, L, dim]` bf16."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], *, dim: int = 4096, dim_attn: int = 4096, dim_ffn: int = 10240,
                 num_heads: int = 64, num_layers: int = 24, num_buckets: int = 32, eps: float = 1e-6, device="cuda"):
        _hip.load()                                         # fail loudly without the HIP library
        if dim_attn // num_heads != 64:
            raise NotImplementedError(f"ifx_t5_attention is built for head_dim 64 (umT5), got {dim_attn // num_heads}")
        self.device = torch.device(device)
        self.dim, self.dim_attn, self.dim_ffn = dim, dim_attn, dim_ffn
        self.heads, self.layers, self.buckets, self.eps = num_heads, num_layers, num_buckets, eps
        g = lambda k: state_dict[k].to(device=self.device, dtype=BF16)
        self.emb = g("token_embedding.weight").contiguous()
        self.norm = g("norm.weight").contiguous()
        self.blocks = []
        for i in range(num_layers):
            p = f"blocks.{i}."
            self.blocks.append(dict(
                norm1=g(p + "norm1.weight").contiguous(), norm2=g(p + "norm2.weight").contiguous(),
                wqkv=torch.cat([g(p + "attn.q.weight"), g(p + "attn.k.weight"), g(p + "attn.v.weight")], 0).contiguous(),
                wo=g(p + "attn.o.weight").contiguous(),
                wgf=torch.cat([g(p + "ffn.gate.0.weight"), g(p + "ffn.fc1.weight")], 0).contiguous(),
                wfc2=g(p + "ffn.fc2.weight").contiguous(),
                pos=g(p + "pos_embedding.embedding.weight").contiguous()))
        self._tables: Dict[int, List[torch.Tensor]] = {}

    def _bias_tables(self, L: int) -> List[torch.Tensor]:
        t = self._tables.get(L)
        if t is None:
            t = self._tables[L] = [relative_position_table(b["pos"], L, self.buckets) for b in self.blocks]
        return t

    @torch.no_grad()
    def forward(self, ids: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, L = ids.shape
        if L % 32 or not 32 <= L <= 512:
            raise ValueError(f"sequence length {L}: ifx_t5_attention needs a multiple of 32 in [32, 512] (the tokenizer pads to 512)")
        ids = ids.to(self.device)
        seq = (mask.to(self.device).gt(0).sum(dim=1) if mask is not None else torch.full((B,), L, device=self.device)).to(torch.int32)
        x = self.emb[ids.reshape(-1)].contiguous()                      # [B*L, dim]  (t5.py:306)
        da = self.dim_attn
        tables = self._bias_tables(L)
        for blk, table in zip(self.blocks, tables):
            h = ops.rmsnorm(x, blk["norm1"], self.eps)
            qkv = ops.linear(h, blk["wqkv"], None)
            o = ops.t5_attention(qkv[:, :da], qkv[:, da:2 * da], qkv[:, 2 * da:], table, seq, B, self.heads)
            x = ops.linear(o, blk["wo"], None, epilogue=_hip.IFX_EPI_RESIDUAL, residual=x, out=x)       # x + attn(...)
            h = ops.rmsnorm(x, blk["norm2"], self.eps)
            hh = ops.t5_gated_gelu(ops.linear(h, blk["wgf"], None))
            x = ops.linear(hh, blk["wfc2"], None, epilogue=_hip.IFX_EPI_RESIDUAL, residual=x, out=x)    # x + ffn(...)
        return ops.rmsnorm(x, self.norm, self.eps).view(B, L, self.dim)

    __call__ = forward


class HipWanTextEncoder:
    """`WanTextEncoder` (wrapper.py:15-59).  `tokenizer(texts, return_mask=True, add_special_tokens=True) -> (ids, mask)` is
    injected (the reference's `HuggingfaceTokenizer`, or anything with that call)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], tokenizer: Optional[Callable] = None, *, device="cuda", **cfg):
        self.text_encoder = HipT5Encoder(state_dict, device=device, **cfg)
        self.tokenizer = tokenizer

    @property
    def device(self):
        return self.text_encoder.device

    @torch.no_grad()
    def encode_ids(self, ids: torch.Tensor, mask: torch.Tensor) -> Dict[str, torch.Tensor]:
        ctx = self.text_encoder(ids, mask)
        seq_lens = mask.to(ctx.device).gt(0).sum(dim=1).long()
        keep = torch.arange(ctx.shape[1], device=ctx.device).unsqueeze(0) < seq_lens.unsqueeze(1)
        ctx = torch.where(keep.unsqueeze(-1), ctx, torch.zeros((), dtype=ctx.dtype, device=ctx.device))   # `u[v:] = 0.0` (wrapper.py:54-55)
        return {"prompt_embeds": ctx}

    def forward(self, text_prompts: Sequence[str]) -> Dict[str, torch.Tensor]:
        if self.tokenizer is None:
            raise RuntimeError("HipWanTextEncoder: no tokenizer was given (pass the reference's HuggingfaceTokenizer, "
                               "text_encoder/tokenizer.py) — or call encode_ids(ids, mask)")
        ids, mask = self.tokenizer(list(text_prompts), return_mask=True, add_special_tokens=True)
        return self.encode_ids(ids, mask)

    __call__ = forward
