"""CPU: the VAE decoder oracle against the golden pixels the reference's own `WanVAE_` produced (oracle/gen_golden_vae.py),
and the stream properties the HIP path relies on."""
import torch

import vae_oracle as V
from fixture_io import golden, weights_checksum

BF = torch.bfloat16


def _setup():
    g = golden("vae_decode.npz")
    cfg = V.VaeConfig(dim=int(g["cfg_dim"]))
    W = V.make_decoder_params(cfg, int(g["seed"]))
    assert weights_checksum(W) == int(g["weights_checksum"]), "seeded weights drifted from the ones the golden was made with"
    return g, cfg, W


def test_param_shapes_cover_decoder():
    cfg = V.VaeConfig()
    shapes = V.decoder_param_shapes(cfg)
    # 96-wide Wan2.1 decoder: conv1 + 15 residual blocks (2 middle + 4 x 3... = 2 + 12) + head, 2 temporal upsamplers
    res = [p for p in V.decoder_plan(cfg) if p[0] == "res"]
    assert len(res) == 14 and [p[0] for p in V.decoder_plan(cfg)].count("up3d") == 2
    assert shapes["decoder.conv1.weight"] == (384, 16, 3, 3, 3) and shapes["decoder.head.2.weight"] == (3, 96, 3, 3, 3)
    assert shapes["decoder.upsamples.4.shortcut.weight"] == (384, 192, 1, 1, 1)
    assert shapes["decoder.upsamples.3.time_conv.weight"] == (768, 384, 3, 1, 1)
    n_causal = sum(1 for k, s in shapes.items() if k.endswith("weight") and len(s) == 5 and k.startswith("decoder."))
    assert n_causal == 33          # what count_conv3d(decoder) gives upstream (vae.py:468-474): the length of the feature cache


def test_oracle_matches_reference_golden_all_flows():
    g, cfg, W = _setup()
    orc = V.VaeDecoderOracle(cfg, W)
    for kw in (dict(use_cache=False), dict(use_cache=True, chunk_size=1), dict(use_cache=True, chunk_size=2)):
        got = orc.decode_to_pixel(g["latent"], **kw)
        assert got.dtype == torch.float32 and got.shape == g["pixels"].shape == (1, 9, 3, 64, 96)
        assert torch.equal(got, g["pixels"]), kw


def test_first_chunk_emits_one_frame_then_four():
    g, cfg, W = _setup()
    orc = V.VaeDecoderOracle(cfg, W)
    z = g["latent"].permute(0, 2, 1, 3, 4)
    a = orc.cached_decode(z[:, :, :1])
    b = orc.cached_decode(z[:, :, 1:2])
    assert a.shape[2] == 1 and b.shape[2] == 4                       # the 'Rep' rule of the temporal upsamplers
    pix = torch.cat([a, b], 2).float().clamp_(-1, 1)
    assert torch.equal(pix[0].permute(1, 0, 2, 3), g["pixels"][0, :5])


def test_batching_frames_is_the_same_function_of_the_stream():
    g, cfg, W = _setup()
    orc = V.VaeDecoderOracle(cfg, W)
    z = g["latent"].permute(0, 2, 1, 3, 4)
    multi = orc.cached_decode(z, frames_per_call=2).float().clamp_(-1, 1).permute(0, 2, 1, 3, 4)
    assert torch.equal(multi, g["pixels"])


def test_host_synthetic_state_dict_has_the_reference_keys():
    """The product-side random-init helper (bench / smoke) produces exactly the decoder's state-dict keys and shapes."""
    from inferix_amd.vae import synthetic_decoder_state_dict
    for dim in (32, 96):
        sd = synthetic_decoder_state_dict(dim=dim)
        shapes = V.decoder_param_shapes(V.VaeConfig(dim=dim))
        assert set(sd) == set(shapes) and all(tuple(sd[k].shape) == shapes[k] for k in shapes)


def test_encoder_oracle_matches_reference_golden():
    g = golden("vae_encode.npz")
    cfg = V.VaeConfig(dim=int(g["cfg_dim"]))
    EW = V.make_encoder_params(cfg, int(g["seed"]))
    assert weights_checksum(EW) == int(g["weights_checksum"])
    orc = V.VaeEncoderOracle(cfg, EW)
    assert torch.equal(orc.encode_to_latent(g["video"]), g["latent"])                       # 5 frames -> 2 latent frames
    assert torch.equal(orc.encode_to_latent(g["video"][:, :, :1]), g["latent_first_frame"])  # the image-to-video case
    assert tuple(g["latent"].shape) == (1, 2, 16, 8, 12)
    from inferix_amd.vae import synthetic_encoder_state_dict
    for dim in (32, 96):
        sd = synthetic_encoder_state_dict(dim=dim)
        shapes = V.encoder_param_shapes(V.VaeConfig(dim=dim))
        assert set(sd) == set(shapes) and all(tuple(sd[k].shape) == shapes[k] for k in shapes)
