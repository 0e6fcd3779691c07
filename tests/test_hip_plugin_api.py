"""GPU: the outer plugin API — `SelfForcingPipeline` mirrors the reference's class: config-driven
construction, checkpoint loading, T2V, per-block streaming with uint8 frame hand-off, segment chaining, error behaviour.
The text encoder and the VAE are outside this build and injected as small deterministic stand-ins."""
import os
import sys

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
import wan_oracle as O  # noqa: E402

BF = torch.bfloat16


class ToyVAE:
    """decode: 3 of the 16 latent channels, 2x nearest upsample, tanh -> [-1, 1]; encode: the inverse geometry."""
    def __init__(self):
        self.model = self
        self.cache_clears = 0
        self.decodes = []

    def clear_cache(self):
        self.cache_clears += 1

    def decode_to_pixel(self, latents, use_cache=True, chunk_size=1):
        self.decodes.append(tuple(latents.shape))
        x = torch.tanh(latents[:, :, :3].float())
        return x.repeat_interleave(2, dim=-1).repeat_interleave(2, dim=-2)

    def encode_to_latent(self, image):
        b, c, t, h, w = image.shape
        return torch.zeros(b, t, 16, h // 2, w // 2) + image.mean()


def _make(tmp_path, **over):
    cfg = O.tiny_config()
    conf = dict(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, num_frame_per_block=3,
                independent_first_frame=False, context_noise=0, timestep_shift=5.0, kv_cache_tokens=21 * cfg.frame_seqlen,
                latent_shape=[cfg.in_dim, cfg.latent_h, cfg.latent_w],
                model_kwargs=dict(patch_size=list(cfg.patch_size), text_len=cfg.text_len, in_dim=cfg.in_dim, dim=cfg.dim,
                                  ffn_dim=cfg.ffn_dim, freq_dim=cfg.freq_dim, text_dim=cfg.text_dim, out_dim=cfg.out_dim,
                                  num_heads=cfg.num_heads, num_layers=cfg.num_layers, eps=cfg.eps))
    conf.update(over)
    path = tmp_path / "self_forcing_tiny.yaml"
    path.write_text(yaml.safe_dump(conf))
    g = torch.Generator().manual_seed(5)
    pe = torch.randn(1, cfg.text_len, cfg.text_dim, generator=g).to(BF).cuda()
    from inferix_amd.pipeline import SelfForcingPipeline
    vae = ToyVAE()
    pipe = SelfForcingPipeline(str(path), text_encoder=lambda text_prompts: {"prompt_embeds": pe.expand(len(text_prompts), -1, -1)},
                               vae=vae)
    W = O.init_weights(cfg, seed=0)
    ck = tmp_path / "ckpt.pt"
    torch.save({"generator": {"model." + k: v for k, v in W.items()}, "generator_ema": {"model." + k: v * 0 for k, v in W.items()}}, ck)
    pipe.load_checkpoint(str(ck), use_ema=False)
    pipe.setup_devices(low_memory=False, verbose=False)
    return pipe, vae, cfg, W


def test_t2v_equals_inner_pipeline_and_checkpoint_loading(tmp_path):
    from inferix_amd.core import DecodeMode
    pipe, vae, cfg, W = _make(tmp_path)
    torch.manual_seed(123)
    video = pipe.run_text_to_video(["a prompt", "another"], num_output_frames=6, num_samples=1)
    assert video.shape == (2, 6, 3, 2 * cfg.latent_h, 2 * cfg.latent_w) and 0.0 <= float(video.min()) and float(video.max()) <= 1.0
    assert vae.decodes == [(1, 6, cfg.in_dim, cfg.latent_h, cfg.latent_w)] * 2 and vae.cache_clears == 2
    # the same seeds through the inner pipeline (whose parity with the reference is pinned by the rollout goldens)
    torch.manual_seed(123)
    _, lat = pipe._run_inference(["a prompt", "another"], 6, 1, decode_mode=DecodeMode.NO_DECODE, return_latents=True)
    ref = (vae.decode_to_pixel(lat) * 0.5 + 0.5).clamp(0, 1)
    assert torch.equal(video.float(), ref.float())
    # weights really came from the 'generator' entry of the checkpoint (the EMA entry is all zeros)
    from test_hip_model import build
    m = build(cfg, W)
    x = torch.randn(1, 3, cfg.in_dim, cfg.latent_h, cfg.latent_w, generator=torch.Generator().manual_seed(1)).to(BF).cuda()
    assert torch.isfinite(lat.float()).all() and lat.float().abs().sum() > 0
    sd_keys = set(W)
    assert sd_keys and pipe._checkpoint_state_dict is None
    with pytest.raises(ValueError):
        bad = tmp_path / "bad.pt"
        torch.save({"something": {}}, bad)
        pipe.load_checkpoint(str(bad))


def test_streaming_blocks_segments_and_errors(tmp_path, monkeypatch):
    from inferix_amd.core import StreamingMode
    pipe, vae, cfg, _ = _make(tmp_path)
    got = []
    torch.manual_seed(7)
    video = pipe.run_streaming_generation(["p0", "p1"], stream_callback=got.append, num_segments=2, segment_length=9,
                                          overlap_frames=3, num_samples=1)
    H, Wd = 2 * cfg.latent_h, 2 * cfg.latent_w
    # segment 0: 3 blocks; segment 1: 3 overlap frames prefilled + 2 generated blocks -> 5 callbacks of uint8 [3, H, W, C]
    assert len(got) == 5 and all(f.dtype == torch.uint8 and tuple(f.shape) == (3, H, Wd, 3) for f in got)
    assert video.shape == (1, 15, H, Wd, 3) and video.device.type == "cpu" and video.dtype == torch.float32
    assert torch.equal((video[0, :3] * 255.0).clamp(0, 255).to(torch.uint8), got[0].cpu())
    # deferred decode gives the same frames in the same order
    got2 = []
    torch.manual_seed(7)
    video2 = pipe.run_streaming_generation(["p0", "p1"], stream_callback=got2.append, num_segments=2, segment_length=9,
                                           overlap_frames=3, num_samples=1, streaming_mode=StreamingMode.DEFERRED_DECODE)
    assert torch.equal(video, video2) and all(torch.equal(a, b) for a, b in zip(got, got2))
    # segment chaining: the second segment starts from the last `overlap_frames` latents of the first
    torch.manual_seed(7)
    v0, lat0 = pipe._generate_segment_with_streaming("p0", None, None, segment_length=9)
    v1, lat1 = pipe._generate_segment_with_streaming("p1", lat0[:, -3:], None, segment_length=9)
    assert lat1.shape[1] == 9 and torch.equal(lat1[:, :3], lat0[:, -3:]) and v1.shape[1] == 6
    with pytest.raises(ValueError, match="multiple of 3"):
        pipe._generate_segment_with_streaming("p", None, None, segment_length=10)
    import torch.distributed as dist
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    with pytest.raises(NotImplementedError, match="distributed"):
        pipe.run_image_to_video(["p"], "x.png")


def test_missing_components_fail_loudly(tmp_path):
    from inferix_amd.pipeline import SelfForcingPipeline
    pipe, vae, cfg, _ = _make(tmp_path)
    pipe.pipeline.vae = None
    with pytest.raises(RuntimeError, match="vae"):
        pipe.run_text_to_video(["p"], num_output_frames=3)
    conf = tmp_path / "bidir.yaml"
    conf.write_text(yaml.safe_dump({"model_kwargs": {}}))
    with pytest.raises(NotImplementedError):
        SelfForcingPipeline(str(conf))
