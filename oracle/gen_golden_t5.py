"""TEST INFRASTRUCTURE — golden vectors for the umT5 encoder oracle (runs ONLY in the build container).

Imports the reference's own `T5Encoder` (inferix/models/wan_base/text_encoder/t5.py) on CPU at a small geometry with the real
head size (64) and bucket count (32), loads the seeded weights of `oracle/t5_oracle.make_params`, runs it in bf16 on two padded
prompts, checks the oracle against it and writes ids / mask / expected context to tests/golden/t5_encoder.npz.

    python oracle/gen_golden_t5.py
"""
from __future__ import annotations

import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import _refstub  # noqa: E402
import t5_oracle as T  # noqa: E402
from fixture_io import GOLDEN_DIR, save_npz, weights_checksum  # noqa: E402

torch.set_grad_enabled(False)
BF = torch.bfloat16


def main():
    _refstub.install()
    t5 = importlib.import_module("inferix.models.wan_base.text_encoder.t5")
    cfg = T.T5Config(vocab_size=200, dim=256, dim_attn=256, dim_ffn=512, num_heads=4, num_layers=3)
    seed = 77
    W = T.make_params(cfg, seed)
    model = t5.T5Encoder(vocab=cfg.vocab_size, dim=cfg.dim, dim_attn=cfg.dim_attn, dim_ffn=cfg.dim_ffn, num_heads=cfg.num_heads,
                         num_layers=cfg.num_layers, num_buckets=cfg.num_buckets, shared_pos=False, dropout=0.1).eval()
    model.load_state_dict({k: v.float() for k, v in W.items()}, strict=True)
    model = model.to(BF)

    g = torch.Generator().manual_seed(3)
    L = 192                                                         # > 128: exercises the clamped far buckets
    ids = torch.randint(1, cfg.vocab_size, (2, L), generator=g)
    lens = [150, 37]
    mask = torch.zeros(2, L, dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
        ids[b, n:] = 0
    ref = model(ids, mask)
    for u, n in zip(ref, lens):                                     # wrapper.py:54-55
        u[n:] = 0.0
    mine = T.text_encoder_forward(cfg, W, ids, mask)
    d = (mine.float() - ref.float()).abs().max().item()
    print(f"oracle vs reference T5Encoder: shape {tuple(ref.shape)}, max diff {d:.3e}, ref std {ref.float().std():.3f}")
    if d != 0.0:
        raise SystemExit("oracle disagrees with the reference")
    # bucket table
    rel = torch.arange(-300, 301).view(1, -1)
    emb = model.blocks[0].pos_embedding
    assert torch.equal(emb._relative_position_bucket(rel), T.relative_position_bucket(rel, cfg.num_buckets, cfg.max_dist))
    save_npz(os.path.join(GOLDEN_DIR, "t5_encoder.npz"), {
        "cfg": torch.tensor([cfg.vocab_size, cfg.dim, cfg.dim_attn, cfg.dim_ffn, cfg.num_heads, cfg.num_layers]), "seed": seed,
        "weights_checksum": weights_checksum(W), "ids": ids, "mask": mask, "context": ref,
        "buckets_m300_300": T.relative_position_bucket(rel, cfg.num_buckets, cfg.max_dist)})
    print("wrote tests/golden/t5_encoder.npz")


if __name__ == "__main__":
    main()
