"""TEST INFRASTRUCTURE — golden vectors for the VAE decoder oracle (runs ONLY in the build container).

Imports the reference's own `WanVAE_` (inferix/models/wan_base/vae.py) on CPU, loads the seeded synthetic decoder weights
of `oracle/vae_oracle.make_decoder_params`, runs the three decode flows `WanVAEWrapper.decode_to_pixel` offers
(wrapper.py:103-168: all-at-once, cached with chunk_size 1 — the per-block streaming call — and chunk_size 2) in bf16, checks
the oracle against each, and writes inputs + expected pixels to tests/golden/vae_decode.npz.

    python oracle/gen_golden_vae.py

Nothing of the reference travels: the fixture holds the latent, the seed / config of the weights and the expected output.
"""
from __future__ import annotations

import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import _refstub  # noqa: E402
import vae_oracle as V  # noqa: E402
from fixture_io import GOLDEN_DIR, save_npz, weights_checksum  # noqa: E402

torch.set_grad_enabled(False)
BF = torch.bfloat16


def ref_decode_to_pixel(model, latent, use_cache, chunk_size):
    """The wrapper's body (it cannot be constructed without a checkpoint file): same calls, same order."""
    zs = latent.permute(0, 2, 1, 3, 4)
    mean = torch.tensor(V.VAE_MEAN, dtype=torch.float32)
    std = torch.tensor(V.VAE_STD, dtype=torch.float32)
    scale = [mean.to(dtype=latent.dtype), 1.0 / std.to(dtype=latent.dtype)]
    out = []
    for u in zs:
        if not use_cache:
            out.append(model.decode(u.unsqueeze(0), scale).float().clamp_(-1, 1).squeeze(0))
        else:
            model.clear_cache()
            parts = [model.cached_decode(u[:, s:s + chunk_size].unsqueeze(0), scale).float().clamp_(-1, 1).squeeze(0)
                     for s in range(0, u.shape[1], chunk_size)]
            out.append(torch.cat(parts, dim=1))
            model.clear_cache()
    return torch.stack(out, dim=0).permute(0, 2, 1, 3, 4)


def main():
    _refstub.install()
    vae = importlib.import_module("inferix.models.wan_base.vae")
    cfg = V.VaeConfig(dim=32)
    seed = 4242
    W = V.make_decoder_params(cfg, seed)
    model = vae.WanVAE_(dim=cfg.dim, z_dim=cfg.z_dim, dim_mult=list(cfg.dim_mult), num_res_blocks=cfg.num_res_blocks,
                        attn_scales=[], temperal_downsample=list(cfg.temperal_downsample), dropout=0.0).eval()
    missing, unexpected = model.load_state_dict({k: v.float() for k, v in W.items()}, strict=False)
    assert not unexpected and all(k.startswith("encoder.") or k.startswith("conv1.") for k in missing), (missing, unexpected)
    model = model.to(BF)
    model.clear_cache()

    g = torch.Generator().manual_seed(7)
    latent = torch.randn(1, 3, cfg.z_dim, 8, 12, generator=g).to(BF)          # [B, T, C, h, w]

    ref_all = ref_decode_to_pixel(model, latent, False, 2)
    ref_c1 = ref_decode_to_pixel(model, latent, True, 1)
    ref_c2 = ref_decode_to_pixel(model, latent, True, 2)
    print("reference: all-at-once vs chunk 1 / chunk 2 max diff",
          (ref_all - ref_c1).abs().max().item(), (ref_all - ref_c2).abs().max().item())

    orc = V.VaeDecoderOracle(cfg, W)
    for name, ref, kw in (("all", ref_all, dict(use_cache=False)), ("chunk1", ref_c1, dict(use_cache=True, chunk_size=1)),
                          ("chunk2", ref_c2, dict(use_cache=True, chunk_size=2))):
        mine = orc.decode_to_pixel(latent, **kw)
        d = (mine - ref).abs().max().item()
        print(f"  oracle vs reference [{name}]: shape {tuple(ref.shape)} max diff {d:.3e}")
        if d != 0.0:
            raise SystemExit("oracle disagrees with the reference")
    # batching several latent frames per call (the HIP path) is the same function of the stream
    orc.clear_cache()
    z = latent.permute(0, 2, 1, 3, 4)
    multi = orc.cached_decode(z, frames_per_call=2).float().clamp_(-1, 1)
    print("  oracle, 2 latent frames per call vs 1:", (multi.permute(0, 2, 1, 3, 4) - ref_all).abs().max().item())

    # ---- encoder: the reference's encode() on a 5-frame and a 1-frame (image-to-video) clip
    enc_seed = 4343
    EW = V.make_encoder_params(cfg, enc_seed)
    model2 = vae.WanVAE_(dim=cfg.dim, z_dim=cfg.z_dim, dim_mult=list(cfg.dim_mult), num_res_blocks=cfg.num_res_blocks,
                         attn_scales=[], temperal_downsample=list(cfg.temperal_downsample), dropout=0.0).eval()
    model2.load_state_dict({k: v.float() for k, v in {**W, **EW}.items()}, strict=True)
    model2 = model2.to(BF)
    video = (torch.rand(1, 3, 5, 64, 96, generator=g) * 2 - 1).to(BF)
    mean = torch.tensor(V.VAE_MEAN, dtype=torch.float32)
    std = torch.tensor(V.VAE_STD, dtype=torch.float32)
    scale = [mean.to(BF), 1.0 / std.to(BF)]
    ref_lat = torch.stack([model2.encode(u.unsqueeze(0), scale).float().squeeze(0) for u in video]).permute(0, 2, 1, 3, 4)
    ref_img = torch.stack([model2.encode(u[:, :1].unsqueeze(0), scale).float().squeeze(0) for u in video]).permute(0, 2, 1, 3, 4)
    eorc = V.VaeEncoderOracle(cfg, EW)
    for name, ref, x in (("5 frames", ref_lat, video), ("1 frame", ref_img, video[:, :, :1])):
        mine = eorc.encode_to_latent(x)
        d = (mine - ref).abs().max().item()
        print(f"  encoder oracle vs reference [{name}]: shape {tuple(ref.shape)} max diff {d:.3e}, std {ref.std():.3f}")
        if d != 0.0:
            raise SystemExit("encoder oracle disagrees with the reference")
    save_npz(os.path.join(GOLDEN_DIR, "vae_encode.npz"), {
        "cfg_dim": cfg.dim, "seed": enc_seed, "weights_checksum": weights_checksum(EW), "video": video, "latent": ref_lat,
        "latent_first_frame": ref_img})
    print("wrote tests/golden/vae_encode.npz")

    save_npz(os.path.join(GOLDEN_DIR, "vae_decode.npz"), {
        "cfg_dim": cfg.dim, "seed": seed, "weights_checksum": weights_checksum(W), "latent": latent, "pixels": ref_all,
        "pixels_std": float(ref_all.std())})
    print("wrote tests/golden/vae_decode.npz; pixel std", float(ref_all.std()), "abs max", float(ref_all.abs().max()))


if __name__ == "__main__":
    main()
