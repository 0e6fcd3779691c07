"""SelfForcingPipeline — the outer plugin API of the path (first row).

Mirror of `inferix.pipeline.self_forcing.pipeline.SelfForcingPipeline` (pipeline/self_forcing/pipeline.py:26-810) on top of
`AbstractInferencePipeline`'s streaming loop (pipeline/base_pipeline.py:468-615): same constructor, `load_checkpoint`,
`setup_devices`, `run_text_to_video`, `run_image_to_video`, `run_streaming_generation`,
`_generate_segment_with_streaming`, same return conventions and error behaviour.  What is behind it differs:

  * the generator is the HIP model (`HipCausalWanModel` behind `HipWanDiffusionWrapper`), built from the config's
    `model_kwargs`; everything is resident in HBM (288 GB), so the reference's meta-device construction, layered
    materialisation, `DynamicSwapInstaller` / memory-manager contexts and the generator off-loading around the deferred VAE
    decode have no counterpart — `low_memory`, `use_memory_manager`, `use_mmap` are accepted and have nothing to do;
  * the umT5 text encoder and the Wan VAE are OUT of this build's scope: they are injected
    (`text_encoder(text_prompts=[...]) -> {"prompt_embeds": ...}`, `vae.decode_to_pixel(latents, use_cache, chunk_size)`,
    `vae.encode_to_latent(image)`, `vae.model.clear_cache()`), e.g. the reference's own modules; a run that needs one that
    was not given fails with a clear error;
  * the latent geometry comes from the config (`latent_shape`, default `[16, 60, 104]` = 480p), not from literals.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from ..core.types import DecodeMode, StreamingMode
from ..kvcache_manager import KVCacheManager, KVCacheRequest
from .base_pipeline import AbstractInferencePipeline
from .causal_inference import CausalInferencePipeline


def _load_config(path_or_dict, default=None) -> SimpleNamespace:
    def read(p):
        if p is None:
            return {}
        if isinstance(p, dict):
            return dict(p)
        import yaml
        with open(p) as f:
            return yaml.safe_load(f) or {}
    def merge(a, b):                         # OmegaConf.merge(default, config): nested mappings merge key by key
        out = dict(a)
        for k, v in b.items():
            out[k] = merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
        return out
    return SimpleNamespace(**merge(read(default), read(path_or_dict)))


def decode_image_file(image_path: str, height: int, width: int) -> torch.Tensor:
    """`[3, height, width]` fp32 in [-1, 1] from an image file: the reference's `Resize((480, 832))` -> `ToTensor` -> `Normalize(.5, .5)`
    on a PIL image (pipeline/self_forcing/pipeline.py:212-221).  torchvision's `Resize` of a PIL image IS `Image.resize(size[::-1],
    BILINEAR)`, `ToTensor` is uint8 / 255 and the normalisation `(x - 0.5) / 0.5` in fp32 — the same three steps on Pillow + torch give
    the same bits without torchvision (which this image does not have)."""
    try:
        from PIL import Image
    except ImportError as exc:               # host-side decoder only; the tensor route of run_image_to_video needs none
        raise RuntimeError("load_image needs Pillow to decode an image file; pass image=tensor [1, 3, 1, H, W] in [-1, 1] to "
                           "run_image_to_video instead") from exc
    import numpy as np
    with Image.open(image_path) as im:
        im = im.convert("RGB").resize((width, height), Image.BILINEAR)
        px = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy())              # [H, W, 3]
    x = px.permute(2, 0, 1).to(torch.float32).div(255)
    return (x - 0.5) / 0.5


class SelfForcingPipeline(AbstractInferencePipeline):
    def __init__(self, config_path, default_config_path: Optional[str] = None, parallel_config=None,
                 profiling_config=None, *, text_encoder=None, vae=None, generator=None, device=None):
        from ..wan import ParallelConfig
        self.config = _load_config(config_path, default_config_path)
        if not hasattr(self.config, "denoising_step_list"):
            raise NotImplementedError("configs without denoising_step_list select the reference's bidirectional "
                                      "CausalDiffusionInferencePipeline, which is not part of this path")
        if not torch.cuda.is_available():
            raise RuntimeError("SelfForcingPipeline needs an MI355X: the HIP path has no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.parallel_config = parallel_config or ParallelConfig()
        AbstractInferencePipeline.__init__(self, self.config, profiling_config)
        self._memory_mode = getattr(self.config, "memory_mode", "balanced")
        self._vae_chunk_size = getattr(self.config, "vae_chunk_size", None)        # None = the inner pipeline's preset
        self._checkpoint_state_dict: Optional[Dict[str, torch.Tensor]] = None
        self.latent_shape = list(getattr(self.config, "latent_shape", [16, 60, 104]))
        torch.set_grad_enabled(False)
        self._initialize_pipeline(text_encoder=text_encoder, vae=vae, generator=generator)

    # ------------------------------------------------------------------------------------------------------
    def _initialize_pipeline(self, text_encoder=None, vae=None, generator=None):
        if generator is None:
            from ..wan import HipCausalWanModel, HipWanDiffusionWrapper
            mk = dict(getattr(self.config, "model_kwargs", {}) or {})
            shift = float(mk.pop("timestep_shift", getattr(self.config, "timestep_shift", 5.0)))
            mk.pop("is_causal", None)
            model = HipCausalWanModel(parallel_config=self.parallel_config, device=self.device, **mk)
            generator = HipWanDiffusionWrapper(model=model, timestep_shift=shift, parallel_config=self.parallel_config)
        args = self.config
        if not hasattr(args, "frame_seq_length"):
            ps = getattr(generator.model, "patch_size", (1, 2, 2))
            args.frame_seq_length = (self.latent_shape[1] // ps[1]) * (self.latent_shape[2] // ps[2])
        self.pipeline = CausalInferencePipeline(args, self.device, generator=generator, text_encoder=text_encoder, vae=vae,
                                                parallel_config=self.parallel_config, profiler=self._profiler)
        self._pipeline_type = "causal"
        if self.parallel_config.world_size > 1 and getattr(generator.model, "cp", None) is None and dist.is_initialized():
            from ..sequence_parallel import attach_sequence_parallel
            attach_sequence_parallel(generator.model, dist.group.WORLD)

    def _init_model(self) -> Any:
        return self.pipeline

    def load_checkpoint(self, checkpoint_path: str, **kwargs) -> None:
        """`{'generator': sd, 'generator_ema': sd}` checkpoints (pipeline.py:87-126); the state dict is applied in
        `setup_devices`.  `.safetensors` files are read as a bare generator state dict."""
        if not checkpoint_path:
            return
        use_ema = kwargs.get("use_ema", False)
        if checkpoint_path.endswith(".safetensors"):
            from safetensors.torch import load_file
            self._checkpoint_state_dict = load_file(checkpoint_path)
            return
        try:
            state = torch.load(checkpoint_path, map_location="cpu", mmap=bool(kwargs.get("use_mmap", True)))
        except Exception:
            state = torch.load(checkpoint_path, map_location="cpu")
        key = "generator_ema" if use_ema else "generator"
        if key not in state:
            other = "generator" if use_ema else "generator_ema"
            key = other if other in state else None
        if key is None:
            raise ValueError(f"No valid checkpoint key found. Available: {list(state.keys())}")
        self._checkpoint_state_dict = state[key]

    def setup_devices(self, low_memory: bool = False, verbose: bool = True, use_memory_manager: bool = False) -> None:
        """Weights to HBM.  `low_memory` / `use_memory_manager` exist for 16-24 GB GPUs upstream (base_pipeline.py:134-201)."""
        if self._checkpoint_state_dict is not None:
            sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in self._checkpoint_state_dict.items()}
            self.pipeline.generator.model.load_state_dict(sd)
            self._checkpoint_state_dict = None
        for name in ("text_encoder", "vae"):
            comp = getattr(self.pipeline, name)
            if isinstance(comp, torch.nn.Module):
                comp.to(self.device)

    # ------------------------------------------------------------------------------------------------------
    def _need(self, what: str):
        comp = getattr(self.pipeline, what)
        if comp is None:
            raise RuntimeError(f"this call needs a {what}: pass {what}=... to SelfForcingPipeline (the umT5 encoder and the "
                               f"Wan VAE are outside this build, the reference's modules plug in unchanged)")
        return comp

    def _noise(self, num_samples: int, frames: int) -> torch.Tensor:
        return torch.randn([num_samples, frames, *self.latent_shape], device=self.device, dtype=torch.bfloat16)

    def run_text_to_video(self, prompts: List[str], num_output_frames: int = 21, num_samples: int = 1,
                          output_folder: Optional[str] = None, save_with_index: bool = False, use_ema: bool = False,
                          low_memory: bool = False, **kwargs) -> torch.Tensor:
        return self._run_inference(prompts=prompts, num_output_frames=num_output_frames, num_samples=num_samples,
                                   output_folder=output_folder, save_with_index=save_with_index, use_ema=use_ema,
                                   low_memory=low_memory, **kwargs)

    def run_image_to_video(self, prompts: List[str], image_path: str, num_output_frames: int = 21, num_samples: int = 1,
                           output_folder: Optional[str] = None, save_with_index: bool = False, use_ema: bool = False,
                           low_memory: bool = False, **kwargs) -> torch.Tensor:
        if dist.is_initialized():
            raise NotImplementedError("I2V does not support distributed inference yet.")
        vae = self._need("vae")
        image = kwargs.pop("image", None)
        if image is None:
            image = self.load_image(image_path)
        initial_latent = vae.encode_to_latent(image).to(device=self.device, dtype=torch.bfloat16)
        initial_latent = initial_latent.repeat(num_samples, 1, 1, 1, 1)
        return self._run_inference(prompts=prompts, num_output_frames=num_output_frames - 1, num_samples=num_samples,
                                   initial_latent=initial_latent, output_folder=output_folder,
                                   save_with_index=save_with_index, use_ema=use_ema, low_memory=low_memory, **kwargs)

    def load_image(self, image_path: str) -> torch.Tensor:
        """Image file -> `[1, 3, 1, H, W]` bf16 in [-1, 1] on the device (pipeline.py:212-221).  The target size follows `latent_shape`
        (8x the latent grid: 480 x 832 for the stock config, which upstream hard-codes)."""
        x = decode_image_file(image_path, int(self.latent_shape[1]) * 8, int(self.latent_shape[2]) * 8)
        return x.unsqueeze(0).unsqueeze(2).to(device=self.device, dtype=torch.bfloat16)

    def _run_inference(self, prompts: List[str], num_output_frames: int, num_samples: int,
                       initial_latent: Optional[torch.Tensor] = None, output_folder: Optional[str] = None,
                       save_with_index: bool = False, use_ema: bool = False, low_memory: bool = False,
                       decode_mode: DecodeMode = DecodeMode.AFTER_ALL, return_latents: bool = False):
        """One `pipeline.inference` per prompt, `num_samples` requests each (pipeline.py:325-449).  Returns the videos
        `[len(prompts)*B, T, C, H, W]` in [0, 1] (latents with `decode_mode=NO_DECODE`)."""
        self._need("text_encoder")
        if decode_mode != DecodeMode.NO_DECODE:
            self._need("vae")
        if output_folder and self.parallel_config.local_rank == 0:
            os.makedirs(output_folder, exist_ok=True)
        if dist.is_initialized():
            dist.barrier()
        kvm = KVCacheManager(device=self.device)
        reqs = [KVCacheRequest(f"req_{i}") for i in range(num_samples)]
        videos, latents = [], []
        for prompt_idx, prompt in enumerate(prompts):
            video, lat = self.pipeline.inference(noise=self._noise(num_samples, num_output_frames),
                                                 text_prompts=[prompt] * num_samples, return_latents=True,
                                                 initial_latent=initial_latent, kv_cache_manager=kvm,
                                                 kv_cache_requests=reqs, low_memory=low_memory,
                                                 vae_chunk_size=self._vae_chunk_size, profile=self._profiling_enabled,
                                                 decode_mode=decode_mode)
            videos.append(video)
            latents.append(lat)
            if output_folder and self.parallel_config.rank == 0 and decode_mode != DecodeMode.NO_DECODE:
                self._save_video(video, prompt_idx, prompt, num_samples, output_folder, save_with_index, use_ema)
            self._clear_vae_cache()
            if dist.is_initialized():
                dist.barrier()
        if not videos:
            return None
        out = torch.cat(videos, dim=0)
        return (out, torch.cat(latents, dim=0)) if return_latents else out

    def _save_video(self, video, prompt_idx, prompt, num_samples, output_folder, save_with_index, use_ema):
        """Raw frames as a tensor file (container muxing — torchvision.io.write_video upstream — is outside this build)."""
        for s in range(num_samples):
            name = f"{prompt_idx}-{s}" if save_with_index else f"{prompt[:100]}-{s}"
            torch.save((video[s].float().clamp(0, 1) * 255).to(torch.uint8).cpu(),
                       os.path.join(output_folder, f"{name}{'_ema' if use_ema else ''}.pt"))

    def _clear_vae_cache(self):
        vae = self.pipeline.vae
        model = getattr(vae, "model", None)
        if model is not None and hasattr(model, "clear_cache"):
            model.clear_cache()

    # ------------------------------------------------------------------------------------------------------
    def _cleanup_segment_memory(self):
        self._clear_vae_cache()

    def _select_streaming_mode(self, streaming_mode: StreamingMode, low_memory: bool) -> StreamingMode:
        """AUTO resolves to TRUE_STREAMING: generator and VAE are co-resident in 288 GB (pipeline.py:502-547 falls back to
        DEFERRED_DECODE only below 24 GB)."""
        return StreamingMode.TRUE_STREAMING if streaming_mode == StreamingMode.AUTO else streaming_mode

    def _to_frames(self, block_latent: torch.Tensor) -> torch.Tensor:
        vae = self._need("vae")
        v = vae.decode_to_pixel(block_latent, use_cache=True, chunk_size=1)
        v = (v * 0.5 + 0.5).clamp(0, 1)
        return v.permute(0, 1, 3, 4, 2).contiguous()            # b t c h w -> b t h w c

    def _generate_segment_with_streaming(self, prompt: str, initial_latent: Optional[torch.Tensor],
                                         stream_callback: Optional[Callable[[torch.Tensor], None]],
                                         segment_length: int = 21, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        """One segment with per-block decode + streaming (pipeline.py:549-810).
        -> (video `[B, segment_length, H, W, C]` float in [0, 1] on the host, final latents `[B, T, C, H, W]`);
        `stream_callback` gets uint8 `[T_block, H, W, C]` per sample per block."""
        num_samples = kwargs.get("num_samples", 1)
        low_memory = kwargs.get("low_memory", False)
        mode = self._select_streaming_mode(kwargs.get("streaming_mode", StreamingMode.AUTO), low_memory)
        rank = self.parallel_config.rank
        nfb = getattr(self.pipeline.args, "num_frame_per_block", 3)
        if initial_latent is None and getattr(self.pipeline.args, "independent_first_frame", False):
            if (segment_length - 1) % nfb != 0:
                raise ValueError(f"For independent_first_frame mode, segment_length must be 1 + N*{nfb}. Got "
                                 f"{segment_length}. Valid values: {[1 + nfb * i for i in range(1, 10)]}")
        elif segment_length % nfb != 0:
            raise ValueError(f"segment_length must be a multiple of {nfb}. Got {segment_length}. "
                             f"Valid values: {[nfb * i for i in range(1, 10)]}")
        self._need("text_encoder")
        self._need("vae")
        n_ctx = initial_latent.shape[1] if initial_latent is not None else 0
        noise = self._noise(num_samples, segment_length - n_ctx)
        kvm = KVCacheManager(device=self.device)
        reqs = [KVCacheRequest(f"stream_req_{i}") for i in range(num_samples)]
        decoded: List[torch.Tensor] = []
        saved: List[torch.Tensor] = []

        def emit(block_latent: torch.Tensor):
            frames = self._to_frames(block_latent)
            if stream_callback is not None:
                for s in range(frames.shape[0]):
                    stream_callback(torch.clamp(frames[s] * 255.0, 0, 255).to(torch.uint8))
            decoded.append(frames.cpu())

        def block_callback(block_latent: torch.Tensor, block_index: int):
            if rank != 0:
                return
            if mode == StreamingMode.TRUE_STREAMING:
                emit(block_latent)
            else:
                saved.append(block_latent.clone())
        _, final_latents = self.pipeline.inference(noise=noise, text_prompts=[prompt] * num_samples, return_latents=True,
                                                   initial_latent=initial_latent, kv_cache_manager=kvm,
                                                   kv_cache_requests=reqs, low_memory=low_memory,
                                                   profile=self._profiling_enabled, block_callback=block_callback,
                                                   decode_mode=DecodeMode.NO_DECODE, free_cache_before_vae=True)
        for req in reqs:
            if req.request_id in kvm.request_to_kv_caches:
                kvm.free(req)
        for block_latent in saved:                       # DEFERRED_DECODE: all diffusion first, then the blocks in order
            emit(block_latent)
        if decoded:
            video = torch.cat(decoded, dim=1)
        else:                                            # ranks > 0, or nothing was generated
            video = self._to_frames(final_latents).cpu()
        self._clear_vae_cache()
        return video, final_latents
