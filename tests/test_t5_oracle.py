"""CPU: the umT5 encoder oracle against the golden context the reference's own `T5Encoder` produced
(oracle/gen_golden_t5.py), and the host-side relative-position table against the oracle's bias tensor."""
import torch

import t5_oracle as T
from fixture_io import golden, weights_checksum


def _setup():
    g = golden("t5_encoder.npz")
    v, d, da, f, h, n = [int(i) for i in g["cfg"]]
    cfg = T.T5Config(vocab_size=v, dim=d, dim_attn=da, dim_ffn=f, num_heads=h, num_layers=n)
    W = T.make_params(cfg, int(g["seed"]))
    assert weights_checksum(W) == int(g["weights_checksum"])
    return g, cfg, W


def test_oracle_matches_reference_golden():
    g, cfg, W = _setup()
    got = T.text_encoder_forward(cfg, W, g["ids"], g["mask"])
    assert got.dtype == torch.bfloat16 and torch.equal(got, g["context"])
    lens = g["mask"].sum(1).tolist()
    assert all(float(got[b, n:].abs().max()) == 0.0 for b, n in enumerate(lens))


def test_relative_position_buckets_and_host_table():
    g, cfg, W = _setup()
    rel = torch.arange(-300, 301).view(1, -1)
    assert torch.equal(T.relative_position_bucket(rel, 32, 128), g["buckets_m300_300"])
    # bidirectional: 16 buckets each side, exact up to 8, log-spaced to 128, clamped beyond
    b = T.relative_position_bucket(torch.tensor([0, -1, -7, -8, -127, -128, -400, 1, 7, 8, 127, 128, 400]), 32, 128)
    assert b.tolist() == [0, 1, 7, 8, 15, 15, 15, 17, 23, 24, 31, 31, 31]
    from inferix_amd.t5 import relative_position_table
    emb = W["blocks.0.pos_embedding.embedding.weight"]
    for L in (32, 192, 512):
        table = relative_position_table(emb, L, 32)                    # [heads, 2L-1]
        full = T.position_bias(emb, L, L, 32, 128)[0]                  # [heads, L, L]
        i, j = torch.meshgrid(torch.arange(L), torch.arange(L), indexing="ij")
        assert torch.equal(table[:, (j - i + L - 1)], full)


def test_host_synthetic_state_dict_has_the_reference_keys():
    from inferix_amd.t5 import synthetic_t5_state_dict
    cfg = T.T5Config(vocab_size=64, dim=128, dim_attn=128, dim_ffn=256, num_heads=2, num_layers=2)
    sd = synthetic_t5_state_dict(64, 128, 128, 256, 2, 2)
    shapes = T.param_shapes(cfg)
    assert set(sd) == set(shapes) and all(tuple(sd[k].shape) == shapes[k] for k in shapes)
