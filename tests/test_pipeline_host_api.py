"""CPU: the host-side plugin surface that sits either side of the denoising path — `AbstractInferencePipeline` template methods,
segment / interactive loops with their boundary checks, `InteractiveSession`, and the `inferix.*` / `dax.*` import paths of
the reference's example scripts (re-exports of inferix_amd).  No kernels run here: the generator is a stub."""
import importlib
import pkgutil
import threading

import pytest
import torch

from inferix_amd.core import (ControlCommand, DecodeMode, InputApplyPolicy, InteractiveSession, MemoryMode, SessionState,
                              calculate_total_frames, validate_overlap_config)
from inferix_amd.pipeline import AbstractInferencePipeline


class StubPipeline(AbstractInferencePipeline):
    """Generates 'latents' whose value encodes (segment number, prompt id) so the loops can be checked exactly."""

    def __init__(self):
        super().__init__({})
        self.calls = []
        self.setups = 0

    def load_checkpoint(self, checkpoint_path, **kw):
        self.ckpt = checkpoint_path

    def setup_devices(self, low_memory=False, verbose=True, use_memory_manager=False):
        self.setups += 1

    def run_text_to_video(self, prompts, **kw):
        return ("t2v", list(prompts), kw)

    def run_image_to_video(self, prompts, image_path, **kw):
        return ("i2v", list(prompts), image_path, kw)

    def _generate_segment_with_streaming(self, prompt, initial_latent, stream_callback, segment_length=21, **kw):
        n_ctx = 0 if initial_latent is None else initial_latent.shape[1]
        self.calls.append((prompt, n_ctx, kw.get("guidance_scale")))
        lat = torch.full((1, segment_length, 2, 2, 2), float(len(self.calls)))
        if initial_latent is not None:
            lat[:, :n_ctx] = initial_latent
        video = lat[:, n_ctx:, 0, :, :, None].expand(-1, -1, -1, -1, 3).clone()        # [B, T_new, H, W, C]
        if stream_callback is not None:
            stream_callback(video[0])
        return video, lat


def test_run_and_call_dispatch():
    p = StubPipeline()
    assert p(prompt="a")[:2] == ("t2v", ["a"]) and p.setups == 1
    assert p(prompts=["a", "b"], image_path="x.png")[:3] == ("i2v", ["a", "b"], "x.png") and p.setups == 1      # setup once
    assert p.run({"prompt": "a", "image_path": "y"})[0] == "i2v"
    with pytest.raises(ValueError):
        p.run({"image_path": "y"})
    with pytest.raises(TypeError):
        AbstractInferencePipeline({})                                               # abstract


def test_memory_mode_and_decode_latent():
    p = StubPipeline()
    p._apply_memory_mode(MemoryMode.AGGRESSIVE)
    assert (p._free_cache_before_vae, p._vae_chunk_size) == (True, 2)
    p._apply_memory_mode("relaxed")
    assert (p._free_cache_before_vae, p._vae_chunk_size) == (False, 7)
    p._apply_memory_mode(MemoryMode.BALANCED, vae_chunk_size=3)
    assert (p._free_cache_before_vae, p._vae_chunk_size) == (True, 3)

    class V:
        def __init__(self):
            self.calls = []

        def decode_to_pixel(self, lat, use_cache=True, chunk_size=2):
            self.calls.append((lat.shape[1], chunk_size))
            return lat[:, :, :3] * 4 - 2                                            # outside [-1, 1] on purpose
    lat = torch.rand(1, 7, 4, 2, 2)
    v = V()
    assert p._decode_latent(lat, v, DecodeMode.NO_DECODE) is None and not v.calls
    full = p._decode_latent(lat, v, DecodeMode.AFTER_ALL, chunk_size=5)
    assert v.calls == [(7, 5)] and 0 <= float(full.min()) and float(full.max()) <= 1
    got = []
    per = p._decode_latent(lat, v, DecodeMode.PER_BLOCK, chunk_size=1, stream_callback=got.append, block_size=3)
    assert [c[0] for c in v.calls[1:]] == [3, 3, 1] and len(got) == 3 and torch.equal(per, full)


def test_streaming_segment_loop_cycles_prompts_and_chains_overlap():
    p = StubPipeline()
    video = p.run_streaming_generation(["p0", "p1"], num_segments=3, segment_length=6, overlap_frames=3)
    assert [c[:2] for c in p.calls] == [("p0", 0), ("p1", 3), ("p0", 3)]
    assert video.shape[1] == 6 + 3 + 3 == calculate_total_frames(3, 6, 3)
    assert p.run_streaming_generation(["p"], num_segments=0) is None


def test_interactive_generation_checkpoints():
    p = StubPipeline()
    s = InteractiveSession(apply_policy=InputApplyPolicy.NEXT_SEGMENT)
    statuses = []
    s.set_status_callback(statuses.append)
    # a prompt queued before the start is picked up at the FIRST checkpoint; of two queued inputs only the latest survives
    s.submit_input(prompt="discarded")
    q = s.submit_input(prompt="second", guidance_scale=3.0)
    assert q.will_apply_at == "Segment 1"
    streamed = []

    def cb(frames):                                   # the UI thread: change the prompt while segment 2 of 4 is being generated
        streamed.append(frames)
        if len(streamed) == 2:
            s.submit_input(prompt="third")
    video = p.run_interactive_generation(s, "first", num_segments=4, segment_length=6, overlap_frames=3, stream_callback=cb)
    assert [c[0] for c in p.calls] == ["second", "second", "third", "third"]
    assert [c[2] for c in p.calls] == [3.0] * 4 and [c[1] for c in p.calls] == [0, 3, 3, 3]
    assert video.shape[1] == 15 and s.state == SessionState.COMPLETED and s.current_prompt == "third"
    assert len(statuses) == 4 and statuses[-1].current_segment == 3 and statuses[-1].frames_generated == 15
    assert statuses[-1].progress_percent > statuses[0].progress_percent
    # STOP at a boundary ends the loop and returns what exists; a stop before the first segment returns None
    p2, s2 = StubPipeline(), InteractiveSession()
    out = p2.run_interactive_generation(s2, "x", num_segments=5, segment_length=6, overlap_frames=3,
                                        stream_callback=lambda f: s2.submit_input(control=ControlCommand.STOP) if len(p2.calls) == 2 else None)
    assert len(p2.calls) == 2 and out.shape[1] == 9
    # several segments without overlap: the reference's loop raises at the first boundary (base_pipeline.py:1068-1069) — same here
    with pytest.raises(ValueError, match="overlap_frames must be positive"):
        StubPipeline().run_interactive_generation(InteractiveSession(), "x", num_segments=2, segment_length=3, overlap_frames=0)
    s3 = InteractiveSession()
    s3.stop()
    assert StubPipeline().run_interactive_generation(s3, "x", num_segments=2, segment_length=3, overlap_frames=0) is None
    # pause blocks the generation thread at the checkpoint until resume
    p4, s4 = StubPipeline(), InteractiveSession()
    s4.pause()
    assert s4.should_pause()
    t = threading.Thread(target=lambda: p4.run_interactive_generation(s4, "x", num_segments=1, segment_length=3, overlap_frames=0))
    t.start()
    t.join(timeout=0.5)
    assert t.is_alive() and not p4.calls
    s4.resume()
    t.join(timeout=10)
    assert not t.is_alive() and len(p4.calls) == 1
    # an exception inside a segment marks the session as failed and propagates
    class Boom(StubPipeline):
        def _generate_segment_with_streaming(self, *a, **k):
            raise RuntimeError("kernel failed")
    s5 = InteractiveSession()
    with pytest.raises(RuntimeError):
        Boom().run_interactive_generation(s5, "x", num_segments=1, segment_length=3, overlap_frames=0)
    assert s5.state == SessionState.ERROR


def test_next_block_policy_ignores_segment_checkpoints():
    s = InteractiveSession(apply_policy=InputApplyPolicy.NEXT_BLOCK)
    s.submit_input(prompt="b")
    assert s.evaluate_checkpoint("segment", 0, "a").new_prompt is None
    assert s.evaluate_checkpoint("block", 1, "a").new_prompt == "b"
    assert s.evaluate_checkpoint("block", 2, "b").new_prompt is None           # applied once
    s2 = InteractiveSession.from_prompts(["p0", "p1", "p2"])
    assert s2.initial_prompt == "p0" and s2.evaluate_checkpoint("segment", 0, "p0").new_prompt == "p2"


def test_boundary_validation_errors():
    p = StubPipeline()
    for kw in (dict(segment_length=0, overlap_frames=0, block_size=3, num_segments=1),
               dict(segment_length=7, overlap_frames=0, block_size=3, num_segments=1),
               dict(segment_length=6, overlap_frames=2, block_size=3, num_segments=1),
               dict(segment_length=6, overlap_frames=6, block_size=3, num_segments=1),
               dict(segment_length=6, overlap_frames=-3, block_size=3, num_segments=1),
               dict(segment_length=6, overlap_frames=3, block_size=3, num_segments=0)):
        with pytest.raises(ValueError):
            p._validate_boundary_config(**kw)
    assert validate_overlap_config(0, 3) and validate_overlap_config(6, 3)
    lat = torch.zeros(1, 3, 2, 2, 2)
    b = p._validate_segment_boundary(2, 4, 3, 9, 3, lat)
    assert (b.start_frame, b.end_frame, b.unique_frames, b.overlap_with_previous, b.is_first, b.is_last) == (12, 20, 6, 3, False, False)
    assert p._validate_segment_boundary(0, 1, 3, 9, 3, None).unique_frames == 9
    for args in ((0, 2, 3, 9, 3, lat), (1, 2, 3, 9, 3, None), (1, 2, 3, 9, 3, torch.zeros(1, 2, 2, 2, 2))):
        with pytest.raises(ValueError):
            p._validate_segment_boundary(*args)
    assert p._extract_overlap_latent(torch.arange(10.).view(1, 10, 1, 1, 1), 3, 0).flatten().tolist() == [7., 8., 9.]
    for args in ((lat, 0, 0), (None, 3, 0), (lat, 4, 0)):
        with pytest.raises(ValueError):
            p._extract_overlap_latent(*args)


def test_reference_import_paths_resolve_to_the_hip_implementation():
    """Every module of the `inferix` / `dax` shims imports, and the names the reference's example scripts use
    (example/self_forcing/run_self_forcing.py:7-10, example/causvid/run_causvid.py:6-9,
    example/quantization/run_self_forcing_quantized.py:12-23, example/streaming/run_interactive_streaming.py:36-39) are the
    inferix_amd objects."""
    import dax
    import inferix
    for pkg in (inferix, dax):
        for m in pkgutil.walk_packages(pkg.__path__, pkg.__name__ + "."):
            importlib.import_module(m.name)
    import inferix_amd.pipeline as P
    from inferix.core.interactive import InteractiveSession as S
    from inferix.core.memory.utils import get_cuda_free_memory_gb, gpu  # noqa: F401
    from inferix.core.types import InputApplyPolicy as IAP, StreamingMode  # noqa: F401
    from inferix.core.types.inference import DecodeMode as DM, InferenceParams, ModelMetaArgs  # noqa: F401
    from inferix.core.utils import set_random_seed
    from inferix.kvcache_manager import KVCacheManager, KVCacheRequest  # noqa: F401
    from inferix.models.attention.backends import collect_supported_attn
    from inferix.models.wan_base.utils.parallel_config import ParallelConfig
    from inferix.pipeline.base_pipeline import AbstractInferencePipeline as A
    from inferix.pipeline.causvid.pipeline import CausVidPipeline
    from inferix.pipeline.self_forcing.CausalInferencePipeline import CausalInferencePipeline
    from inferix.pipeline.self_forcing.pipeline import SelfForcingPipeline
    from dax.quant.quantization import quantize_dynamic
    from dax.quant.quantization.qconfig import get_dynamic_fp8_per_token_act_per_channel_weight_qconfig as f8
    import inferix_amd.quant as Q
    assert SelfForcingPipeline is P.SelfForcingPipeline and CausVidPipeline is P.CausVidPipeline and A is AbstractInferencePipeline
    assert issubclass(SelfForcingPipeline, A) and issubclass(CausVidPipeline, A) and CausalInferencePipeline is P.CausalInferencePipeline
    assert S is InteractiveSession and IAP is InputApplyPolicy and DM is DecodeMode
    assert quantize_dynamic is Q.quantize_dynamic and f8().fmt == 0 and "HipPagedFA" in collect_supported_attn()
    assert set_random_seed(7) == 7 and float(torch.rand(1)) == float(torch.manual_seed(7) and torch.rand(1))
    assert ParallelConfig().world_size == 1
    # MAGI (config 5): the model class, the layer classes and the scheduler half of SampleTransport at the reference's paths
    from inferix.models.magi.dit.dit_model import VideoDiTModel
    from inferix.models.magi.dit.dit_module import TransformerBlock, TransformerLayer  # noqa: F401
    from inferix.pipeline.magi.video_generate import ChunkSchedule, find_dit_model, generate_sequences, init_intervel, init_t  # noqa: F401
    import inferix_amd.magi.model as MM_
    assert VideoDiTModel is MM_.HipVideoDiTModel and generate_sequences(4, 4, 0)[0] == [0, 0, 0, 0, 1, 2, 3]
    wrapped = type("DDP", (), {})()
    wrapped.module = type("M", (), {"forward_dispatcher": lambda self: None})()
    assert find_dit_model(wrapped) is wrapped.module


def test_load_image_decode_matches_the_reference_transform(tmp_path):
    """`SelfForcingPipeline.load_image` (reference pipeline.py:212-221: Resize((480, 832)) -> ToTensor -> Normalize(.5, .5) on a PIL
    image).  torchvision is not in this image, so the three steps are checked one by one: at the target size the resize is the
    identity and the result is exactly (px / 255 - 0.5) / 0.5; a constant image stays constant under the bilinear resize; the file
    route works for PNG and JPEG containers and palette / RGBA inputs are converted to RGB."""
    import numpy as np
    import pytest
    Image = pytest.importorskip("PIL.Image")
    from inferix_amd.pipeline.self_forcing import decode_image_file
    rng = np.random.default_rng(0)
    px = rng.integers(0, 256, size=(48, 80, 3), dtype=np.uint8)
    p = tmp_path / "a.png"
    Image.fromarray(px).save(p)
    x = decode_image_file(str(p), 48, 80)
    assert x.shape == (3, 48, 80) and x.dtype == torch.float32
    want = (torch.from_numpy(px).permute(2, 0, 1).float() / 255 - 0.5) / 0.5
    assert torch.equal(x, want)
    y = decode_image_file(str(p), 96, 160)                      # upscale: stays inside the source range
    assert y.shape == (3, 96, 160) and float(y.min()) >= float(want.min()) and float(y.max()) <= float(want.max())
    flat = np.full((30, 50, 4), 200, dtype=np.uint8)            # RGBA, constant colour
    q = tmp_path / "b.png"
    Image.fromarray(flat, "RGBA").save(q)
    z = decode_image_file(str(q), 60, 104)
    assert z.shape == (3, 60, 104) and torch.equal(z, torch.full_like(z, (200 / 255 - 0.5) / 0.5))
    j = tmp_path / "c.jpg"
    Image.fromarray(flat[..., :3]).save(j)
    assert decode_image_file(str(j), 60, 104).shape == (3, 60, 104)
