"""TEST INFRASTRUCTURE — CPU oracle of the umT5 text encoder as Inferix runs it.

Functional, state-dict-driven restatement; only tests, `smoke()` and bench.py's CPU leg may import it.  Follows op for op
(every op's result in the model dtype, bf16 in the pipelines: inferix/pipeline/self_forcing/pipeline.py:172):

  * `WanTextEncoder.forward`        inferix/models/self_forcing/wrapper.py:46-59   (ids, mask -> context, padding rows zeroed)
  * `T5Encoder.forward`             inferix/models/wan_base/text_encoder/t5.py:305-314
  * `T5SelfAttention.forward`       t5.py:172-177   (pre-norm residual block, per-layer relative position bias: shared_pos=False)
  * `T5Attention.forward`           t5.py:88-122    (no 1/sqrt(d) scaling, additive bias, finfo.min on masked keys, fp32 softmax)
  * `T5FeedForward.forward`         t5.py:138-143   (fc1(x) * GELU(gate(x)), fc2)
  * `GELU.forward`                  t5.py:50-52     (tanh formula evaluated op by op in the model dtype)
  * `T5LayerNorm.forward`           t5.py:63-68
  * `T5RelativeEmbedding`           t5.py:235-266   (bidirectional log-bucketed relative positions)

Pinned by `oracle/gen_golden_t5.py` against the reference's own `T5Encoder` on CPU (tests/golden/t5_encoder.npz).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

BF = torch.bfloat16


@dataclass(frozen=True)
class T5Config:
    vocab_size: int = 256384                         # t5.py:458-470 (`umt5_xxl`)
    dim: int = 4096
    dim_attn: int = 4096
    dim_ffn: int = 10240
    num_heads: int = 64
    num_layers: int = 24
    num_buckets: int = 32
    max_dist: int = 128

    @property
    def head_dim(self) -> int:
        return self.dim_attn // self.num_heads


def param_shapes(cfg: T5Config) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {"token_embedding.weight": (cfg.vocab_size, cfg.dim), "norm.weight": (cfg.dim,)}
    for i in range(cfg.num_layers):
        p = f"blocks.{i}."
        s[p + "norm1.weight"] = (cfg.dim,)
        s[p + "attn.q.weight"] = (cfg.dim_attn, cfg.dim)
        s[p + "attn.k.weight"] = (cfg.dim_attn, cfg.dim)
        s[p + "attn.v.weight"] = (cfg.dim_attn, cfg.dim)
        s[p + "attn.o.weight"] = (cfg.dim, cfg.dim_attn)
        s[p + "norm2.weight"] = (cfg.dim,)
        s[p + "ffn.gate.0.weight"] = (cfg.dim_ffn, cfg.dim)
        s[p + "ffn.fc1.weight"] = (cfg.dim_ffn, cfg.dim)
        s[p + "ffn.fc2.weight"] = (cfg.dim, cfg.dim_ffn)
        s[p + "pos_embedding.embedding.weight"] = (cfg.num_buckets, cfg.num_heads)
    return s


def make_params(cfg: T5Config, seed: int) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights (bf16) with the reference's init scales (t5.py:29-45), norm weights ~ 1 and a relative
    position table large enough to matter."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "norm.weight":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name == "token_embedding.weight":
            t = torch.randn(shape, generator=g)
        elif "pos_embedding" in name:
            t = 0.5 * torch.randn(shape, generator=g)
        elif name.endswith("attn.q.weight"):
            t = torch.randn(shape, generator=g) * (cfg.dim * cfg.head_dim) ** -0.25     # keeps q.k ~ O(1) without 1/sqrt(d)
        elif name.endswith("attn.o.weight") or name.endswith("ffn.fc2.weight"):
            t = torch.randn(shape, generator=g) * shape[1] ** -0.5
        else:
            t = torch.randn(shape, generator=g) * shape[1] ** -0.5
        W[name] = t.to(BF)
    return W


def relative_position_bucket(rel_pos: torch.Tensor, num_buckets: int, max_dist: int) -> torch.Tensor:
    """t5.py:247-266, bidirectional."""
    nb = num_buckets // 2
    buckets = (rel_pos > 0).long() * nb
    rel = rel_pos.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return buckets + torch.where(rel < max_exact, rel, large)


def position_bias(emb: torch.Tensor, lq: int, lk: int, num_buckets: int, max_dist: int) -> torch.Tensor:
    """t5.py:235-245 -> [1, heads, lq, lk]."""
    rel = torch.arange(lk).unsqueeze(0) - torch.arange(lq).unsqueeze(1)
    return emb[relative_position_bucket(rel, num_buckets, max_dist)].permute(2, 0, 1).unsqueeze(0).contiguous()


def t5_layer_norm(x: torch.Tensor, w: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    y = x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps)
    if w.dtype in (torch.float16, torch.bfloat16):
        y = y.type_as(w)
    return w * y


def gelu(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def attention(x: torch.Tensor, W: Dict[str, torch.Tensor], p: str, heads: int, mask, pos_bias) -> torch.Tensor:
    b, L, _ = x.shape
    q = F.linear(x, W[p + "q.weight"]).view(b, L, heads, -1)
    k = F.linear(x, W[p + "k.weight"]).view(b, L, heads, -1)
    v = F.linear(x, W[p + "v.weight"]).view(b, L, heads, -1)
    bias = x.new_zeros(b, heads, L, L)
    bias += pos_bias
    if mask is not None:
        bias.masked_fill_(mask.view(b, 1, 1, -1) == 0, torch.finfo(x.dtype).min)
    attn = torch.einsum("binc,bjnc->bnij", q, k) + bias
    attn = F.softmax(attn.float(), dim=-1).type_as(attn)
    o = torch.einsum("bnij,bjnc->binc", attn, v).reshape(b, L, -1)
    return F.linear(o, W[p + "o.weight"])


def encoder_forward(cfg: T5Config, W: Dict[str, torch.Tensor], ids: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """`T5Encoder.forward` (dropout is identity in eval)."""
    x = W["token_embedding.weight"][ids]
    L = x.shape[1]
    for i in range(cfg.num_layers):
        p = f"blocks.{i}."
        e = position_bias(W[p + "pos_embedding.embedding.weight"], L, L, cfg.num_buckets, cfg.max_dist)
        x = x + attention(t5_layer_norm(x, W[p + "norm1.weight"]), W, p + "attn.", cfg.num_heads, mask, e)
        h = t5_layer_norm(x, W[p + "norm2.weight"])
        h = F.linear(h, W[p + "ffn.fc1.weight"]) * gelu(F.linear(h, W[p + "ffn.gate.0.weight"]))
        x = x + F.linear(h, W[p + "ffn.fc2.weight"])
    return t5_layer_norm(x, W["norm.weight"])


def text_encoder_forward(cfg: T5Config, W: Dict[str, torch.Tensor], ids: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """`WanTextEncoder.forward` after the tokenizer (wrapper.py:49-59): rows past each prompt's length are zeroed."""
    ctx = encoder_forward(cfg, W, ids, mask)
    seq_lens = mask.gt(0).sum(dim=1).long()
    for u, v in zip(ctx, seq_lens):
        u[v:] = 0.0
    return ctx
