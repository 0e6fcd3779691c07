"""Seeded synthetic weights/inputs for benchmarks and smoke tests (no checkpoints exist offline).
Init style follows the reference's `init_weights()` (causal_model.py:1221-1243): xavier-uniform linears,
N(0, .02) text/time embeddings, zero-mean small biases; `head.head.weight` ~ N(0, .02) (zero upstream,
which would make every output pure bias).  Generated directly on the target device."""
from __future__ import annotations

import math
from typing import Dict

import torch


def synthetic_state_dict(model, seed: int = 0) -> Dict[str, torch.Tensor]:
    dev = model.device_
    g = torch.Generator(device=dev).manual_seed(seed)
    d, f = model.dim, model.ffn_dim

    def xavier(o, i):
        a = math.sqrt(6.0 / (i + o))
        return ((torch.rand(o, i, generator=g, device=dev) * 2 - 1) * a).to(torch.bfloat16)

    def normal(*shape, std=0.02):
        return (torch.randn(*shape, generator=g, device=dev) * std).to(torch.bfloat16)

    sd: Dict[str, torch.Tensor] = {}

    def lin(name, o, i, w=None):
        sd[name + ".weight"] = xavier(o, i) if w is None else w
        sd[name + ".bias"] = normal(o)

    pk = model.in_dim * math.prod(model.patch_size)
    sd["patch_embedding.weight"] = xavier(d, pk).view(d, model.in_dim, *model.patch_size)
    sd["patch_embedding.bias"] = normal(d)
    lin("text_embedding.0", d, model.text_dim, normal(d, model.text_dim))
    lin("text_embedding.2", d, d, normal(d, d))
    lin("time_embedding.0", d, model.freq_dim, normal(d, model.freq_dim))
    lin("time_embedding.2", d, d, normal(d, d))
    lin("time_projection.1", 6 * d, d)
    for i in range(model.num_layers):
        p = f"blocks.{i}."
        sd[p + "modulation"] = (torch.randn(1, 6, d, generator=g, device=dev) / d ** 0.5).to(torch.bfloat16)
        sd[p + "norm3.weight"] = (1 + normal(d, std=0.1).float()).to(torch.bfloat16)
        sd[p + "norm3.bias"] = normal(d, std=0.1)
        for a in ("self_attn", "cross_attn"):
            for n in "qkvo":
                lin(p + f"{a}.{n}", d, d)
            sd[p + f"{a}.norm_q.weight"] = (1 + normal(d, std=0.1).float()).to(torch.bfloat16)
            sd[p + f"{a}.norm_k.weight"] = (1 + normal(d, std=0.1).float()).to(torch.bfloat16)
        lin(p + "ffn.0", f, d)
        lin(p + "ffn.2", d, f)
    sd["head.modulation"] = (torch.randn(1, 2, d, generator=g, device=dev) / d ** 0.5).to(torch.bfloat16)
    od = model.out_dim * math.prod(model.patch_size)
    lin("head.head", od, d, normal(od, d))
    return sd
