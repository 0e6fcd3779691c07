"""GPU: the call sequences of the reference's example scripts — example/self_forcing/run_self_forcing.py:79-150,
example/causvid/run_causvid.py, example/quantization/run_self_forcing_quantized.py:47-64 and the interactive example
(example/streaming/run_interactive_streaming.py) — restated here and driven through the REFERENCE'S import paths (the `inferix` /
`dax` shims), i.e. what a user of the reference who swaps the package in gets.  Tiny model dimensions; text encoder and VAE are
deterministic stand-ins with the real components' interface (the HIP ones are exercised in test_hip_vae.py / test_hip_t5.py)."""
import os

import pytest
import torch
import yaml

import wan_oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


class ToyVAE4:
    """The Wan VAE's geometry without its arithmetic: decode T latents -> 1 + 4 (T - 1) frames of 3 channels at 2x the latent
    resolution in [-1, 1]; encode one pixel frame -> one latent frame."""
    def __init__(self):
        self.model = self
        self.encoded = []

    def clear_cache(self):
        pass

    def decode_to_pixel(self, latents, use_cache=True, chunk_size=1):
        x = torch.tanh(latents[:, :, :3].float()).repeat_interleave(2, dim=-1).repeat_interleave(2, dim=-2)
        return x.repeat_interleave(4, dim=1)[:, 3:]

    def encode_to_latent(self, pixel):
        b, c, t, h, w = pixel.shape
        assert c == 3 and t == 1
        self.encoded.append(pixel.float().mean().item())
        lat = torch.zeros(b, t, 16, h // 2, w // 2, device=pixel.device)
        lat[:, :, :3] = torch.atanh(pixel.float().clamp(-0.999, 0.999))[:, :, :, ::2, ::2].transpose(1, 2)
        return lat


def _conf(cfg, **over):
    conf = dict(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, num_frame_per_block=3,
                independent_first_frame=False, context_noise=0, timestep_shift=5.0, kv_cache_tokens=21 * cfg.frame_seqlen,
                latent_shape=[cfg.in_dim, cfg.latent_h, cfg.latent_w],
                model_kwargs=dict(patch_size=list(cfg.patch_size), text_len=cfg.text_len, in_dim=cfg.in_dim, dim=cfg.dim,
                                  ffn_dim=cfg.ffn_dim, freq_dim=cfg.freq_dim, text_dim=cfg.text_dim, out_dim=cfg.out_dim,
                                  num_heads=cfg.num_heads, num_layers=cfg.num_layers, eps=cfg.eps))
    conf.update(over)
    return conf


class PromptTable:
    """text_encoder stand-in: a deterministic embedding per prompt string; records what it was asked for."""
    def __init__(self, cfg):
        self.cfg, self.seen = cfg, []

    def __call__(self, text_prompts):
        self.seen.append(list(text_prompts))
        out = []
        for p in text_prompts:
            g = torch.Generator().manual_seed(sum(p.encode()) % 1000)
            out.append(torch.randn(self.cfg.text_len, self.cfg.text_dim, generator=g))
        return {"prompt_embeds": torch.stack(out).to(BF).cuda()}


def test_self_forcing_example_sequence_through_reference_import_paths(tmp_path):
    from inferix.core.memory.utils import get_cuda_free_memory_gb, gpu as get_gpu
    from inferix.core.utils import set_random_seed
    from inferix.models.wan_base.utils.parallel_config import ParallelConfig
    from inferix.pipeline.self_forcing.pipeline import SelfForcingPipeline
    cfg = O.tiny_config()
    default, specific = tmp_path / "default_config.yaml", tmp_path / "self_forcing_dmd.yaml"
    conf = _conf(cfg)
    default.write_text(yaml.safe_dump({**conf, "denoising_step_list": [1000], "model_kwargs": {**conf["model_kwargs"], "sink_size": 0}}))
    specific.write_text(yaml.safe_dump({"denoising_step_list": [1000, 750, 500, 250], "model_kwargs": {"local_attn_size": -1}}))
    W = O.init_weights(cfg, seed=0)
    ckpt = tmp_path / "checkpoint.pt"
    torch.save({"generator": {"model." + k: v for k, v in W.items()}}, ckpt)
    # ---- run_self_forcing.py: setup_distributed_environment (single process branch) + main
    set_random_seed(0)
    parallel_config = ParallelConfig()
    torch.cuda.set_device(0)
    gpu = get_gpu()
    low_memory = get_cuda_free_memory_gb(gpu) < 40
    assert not low_memory, "an MI355X reports hundreds of GB free"
    enc = PromptTable(cfg)
    pipeline = SelfForcingPipeline(config_path=str(specific), default_config_path=str(default), parallel_config=parallel_config,
                                   text_encoder=enc, vae=ToyVAE4())
    # nested keys merge as OmegaConf.merge does: the specific file's model_kwargs did not wipe the default's
    assert pipeline.config.model_kwargs["dim"] == cfg.dim and pipeline.config.model_kwargs["local_attn_size"] == -1
    assert list(pipeline.config.denoising_step_list) == [1000, 750, 500, 250]
    pipeline._memory_mode = "relaxed"                     # "--memory_mode" override of the script
    pipeline._vae_chunk_size = 3
    pipeline.load_checkpoint(str(ckpt), use_ema=False)
    pipeline.setup_devices(low_memory=low_memory, verbose=True, use_memory_manager=False)
    prompts = [p.strip() for p in "a cat; a dog".split(";") if p.strip()]
    out_dir = tmp_path / "out"
    video = pipeline.run_text_to_video(prompts=prompts, num_output_frames=6, num_samples=1, output_folder=str(out_dir),
                                       save_with_index=True, use_ema=False, low_memory=low_memory)
    assert video.shape == (2, 21, 3, 2 * cfg.latent_h, 2 * cfg.latent_w) and torch.isfinite(video).all()
    assert sorted(os.listdir(out_dir)) == ["0-0.pt", "1-0.pt"] and enc.seen == [["a cat"], ["a dog"]]
    # the template entry points of the base class
    # (`__call__` hands its keyword arguments to `run` as the INPUTS dict, base_pipeline.py:437-458: only the input keys count,
    #  so both calls below generate the default 21 latent frames)
    set_random_seed(1)
    v1 = pipeline(prompt="a cat")
    set_random_seed(1)
    v2 = pipeline.run({"prompts": ["a cat"]})
    assert torch.equal(v1, v2) and v1.shape[1] == 81


def test_quantized_example_sequence(tmp_path):
    """run_self_forcing_quantized.py: quantize_transformer(pipeline.pipeline.generator.model, ...) after setup_devices."""
    from dax.quant.quantization import quantize_dynamic
    from dax.quant.quantization.qconfig import (get_dynamic_fp8_per_token_act_per_channel_weight_qconfig,
                                                get_dynamic_int8_per_token_act_per_channel_weight_qconfig)
    from inferix.core.utils import set_random_seed
    from inferix.pipeline.self_forcing.pipeline import SelfForcingPipeline
    cfg = O.tiny_config()
    path = tmp_path / "c.yaml"
    path.write_text(yaml.safe_dump(_conf(cfg)))
    pipeline = SelfForcingPipeline(config_path=str(path), text_encoder=PromptTable(cfg), vae=ToyVAE4())
    pipeline.pipeline.generator.model.load_state_dict(O.init_weights(cfg, seed=0))
    pipeline.setup_devices(low_memory=False)
    set_random_seed(2)
    ref = pipeline.run_text_to_video(["p"], num_output_frames=3)
    for qconfig in (get_dynamic_fp8_per_token_act_per_channel_weight_qconfig(), get_dynamic_int8_per_token_act_per_channel_weight_qconfig()):
        quantize_dynamic(pipeline.pipeline.generator.model, {"": qconfig, "text_embedding": None, "proj_out": None, "head": None})
        assert pipeline.pipeline.generator.model.quantized_linears > 0
        set_random_seed(2)
        vq = pipeline.run_text_to_video(["p"], num_output_frames=3)
        rel = float((vq - ref).norm() / ref.norm())
        assert torch.isfinite(vq).all() and 0 < rel < 0.2, rel            # 8-bit linears: close to, not equal to, the bf16 video


def test_causvid_example_sequence_same_and_different_prompts(tmp_path):
    """run_causvid.py -> CausVidPipeline.run_text_to_video: `num_rollout` chained segments of one prompt in ONE request, and the
    different-prompt mode = a new KVCacheRequest per segment (continuous-prompt rollover), boundary frame re-encoded from pixels."""
    from inferix.core.utils import set_random_seed
    from inferix.models.wan_base.utils.parallel_config import ParallelConfig
    from inferix.pipeline.causvid.pipeline import CausVidPipeline
    cfg = O.tiny_config()
    conf = _conf(cfg, denoising_step_list=[1000, 757, 522, 0], warp_denoising_step=False, timestep_shift=8.0,
                 image_or_video_shape=[1, 9, cfg.in_dim, cfg.latent_h, cfg.latent_w])
    del conf["kv_cache_tokens"]
    path = tmp_path / "causvid.yaml"
    path.write_text(yaml.safe_dump(conf))
    W = O.init_weights(cfg, seed=0)
    os.makedirs(tmp_path / "ckpt")
    torch.save({"generator": {"model." + k: v for k, v in W.items()}}, tmp_path / "ckpt" / "model.pt")
    enc, vae = PromptTable(cfg), ToyVAE4()
    pipeline = CausVidPipeline(config_path=str(path), default_config_path=None, wan_base_model_path=None, enable_kv_offload=False,
                               parallel_config=ParallelConfig(), text_encoder=enc, vae=vae)
    assert pipeline.frames_per_segment == 9 and pipeline.pipeline.kv_cache_tokens == 9 * cfg.frame_seqlen
    pipeline.load_checkpoint(str(tmp_path / "ckpt"))
    pipeline.setup_devices(low_memory=False)
    seen = []
    inner = pipeline.pipeline.inference

    def spy(**kw):
        video, lat = inner(**kw)
        seen.append((kw["text_prompts"][0], kw["kv_cache_requests"][0].request_id,
                     None if kw["start_latents"] is None else kw["start_latents"].clone(), lat.clone(), video.shape))
        return video, lat
    pipeline.pipeline.inference = spy
    px = 1 + 4 * (9 - 1)                       # pixel frames of a 9-latent segment
    # ---- same prompt, 2 rollouts
    set_random_seed(3)
    res = pipeline.run_text_to_video(["one prompt"], output_folder=str(tmp_path / "o1"), num_rollout=2, num_overlap_frames=3)
    assert len(res) == 1 and res[0].dtype == torch.uint8 and tuple(res[0].shape) == (2 * (px - 9), 2 * cfg.latent_h, 2 * cfg.latent_w, 3)
    assert [s[1] for s in seen] == ["one prompt"] * 2 and seen[0][2] is None
    start = seen[1][2]
    assert start.shape[1] == 3 and torch.equal(start[:, 1:], seen[0][3][:, -2:]), "last overlap-1 latents are carried over as they are"
    assert torch.equal(seen[1][3][:, :3], start), "the next segment is prefilled with the start latents"
    assert len(vae.encoded) == 2 and os.path.exists(tmp_path / "o1" / "prompt_0.pt")
    # ---- different prompts: rollover
    seen.clear()
    set_random_seed(3)
    res = pipeline.run_text_to_video(["first", "second", "third"], output_folder=str(tmp_path / "o2"), num_overlap_frames=3,
                                     is_diff_prompt=True)
    assert len(res) == 1 and res[0].shape[0] == 3 * (px - 9)
    assert [s[0] for s in seen] == ["first", "second", "third"] and len({s[1] for s in seen}) == 3, "one request per segment"
    assert enc.seen[-3:] == [["first"], ["second"], ["third"]]
    assert torch.equal(seen[2][3][:, :3], seen[2][2])
    # interactive mode: prompts come from the source until it says Quit
    seen.clear()
    feed = iter(["alpha", "beta", "Quit"])
    res = pipeline.run_text_to_video([], output_folder=str(tmp_path / "o3"), num_overlap_frames=3, is_diff_prompt=True,
                                     is_interactive=True, prompt_source=lambda i: next(feed))
    assert [s[0] for s in seen] == ["alpha", "beta"] and res[0].shape[0] == 2 * (px - 9)
    with pytest.raises(NotImplementedError):
        pipeline.run_image_to_video("p", "x.png")
    with pytest.raises(AssertionError):
        pipeline.run_text_to_video(["p"], num_overlap_frames=2)


def test_interactive_streaming_example_sequence(tmp_path):
    """run_interactive_streaming.py: an InteractiveSession feeds run_interactive_generation; a prompt submitted during segment 1
    is encoded for segment 2, whose first frames are the overlap of segment 1."""
    from inferix.core.interactive import InteractiveSession
    from inferix.core.types import InputApplyPolicy
    from inferix.core.utils import set_random_seed
    from inferix.pipeline.self_forcing.pipeline import SelfForcingPipeline
    cfg = O.tiny_config()
    path = tmp_path / "c.yaml"
    path.write_text(yaml.safe_dump(_conf(cfg)))
    enc = PromptTable(cfg)
    pipeline = SelfForcingPipeline(config_path=str(path), text_encoder=enc, vae=ToyVAE4())
    pipeline.pipeline.generator.model.load_state_dict(O.init_weights(cfg, seed=0))
    pipeline.setup_devices()
    session = InteractiveSession(apply_policy=InputApplyPolicy.NEXT_SEGMENT)
    status = []
    session.set_status_callback(status.append)
    frames = []

    def stream(f):
        frames.append(f)
        if len(frames) == 1:
            session.submit_input(prompt="a dog running")
    set_random_seed(4)
    video = pipeline.run_interactive_generation(session=session, initial_prompt="a cat walking", num_segments=2, segment_length=6,
                                                overlap_frames=3, stream_callback=stream, block_size=3)
    assert enc.seen == [["a cat walking"], ["a dog running"]]
    assert len(frames) == 3 and all(f.dtype == torch.uint8 for f in frames)             # 2 blocks + 1 new block
    assert video.shape[0] == 1 and video.shape[1] == sum(f.shape[0] for f in frames) and len(status) == 2
    with pytest.raises(ValueError):
        pipeline.run_interactive_generation(session=InteractiveSession(), initial_prompt="x", num_segments=1, segment_length=7)
