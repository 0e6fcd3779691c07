"""`CausVidPipeline`: the CausVid plugin class (inferix/pipeline/causvid/pipeline.py:16-387) over the HIP generator.

Same constructor and methods as upstream — `load_checkpoint`, `setup_devices`, `run_text_to_video(prompts, output_folder,
num_rollout, num_overlap_frames, is_diff_prompt, is_interactive)` with its two drivers `_run_inference_same_prompt` (one
KVCacheRequest per prompt, `num_rollout` chained segments, :188-222) and `_run_inference_diff_prompt` (a NEW request per
segment = per prompt, the previous request's caches cleared — the continuous-prompt KV rollover of BASELINE config 3,
:224-266), `_generate_one_segment` (:269-309) and `_encode_start_frame` (:315-332): the boundary frame of a segment is taken
from its decoded PIXELS, re-encoded by the VAE encoder into one latent frame and joined with the last `overlap - 1` latents.

Components the build does not ship (tokenizer-fed text encoder weights, VAE weights) are injected through `text_encoder=` /
`vae=`; `inferix_amd.t5.HipWanTextEncoder` and `inferix_amd.vae.HipWanVAEWrapper` are the HIP implementations of both.
Video files: frames are returned (uint8 `[T, H, W, C]` arrays per prompt) and saved as tensors; container muxing (diffusers'
`export_to_video` upstream) is outside this build.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

from ..kvcache_manager import KVCacheManager, KVCacheRequest
from .base_pipeline import AbstractInferencePipeline
from .causvid_inference import CausVidInferencePipeline
from .self_forcing import _load_config


class CausVidPipeline(AbstractInferencePipeline):
    def __init__(self, config_path, default_config_path: Optional[str] = None, wan_base_model_path: Optional[str] = None,
                 enable_kv_offload: bool = False, parallel_config=None, *, text_encoder=None, vae=None, generator=None,
                 device=None, latent_shape=(16, 60, 104)):
        from ..wan import ParallelConfig
        config = _load_config(config_path, default_config_path)
        super().__init__(config)
        if enable_kv_offload:
            raise NotImplementedError("kv_offload parks the cache in host memory for 24 GB GPUs; the HIP kernels read it in "
                                      "place from HBM (288 GB)")
        if not torch.cuda.is_available():
            raise RuntimeError("CausVidPipeline needs an MI355X: the HIP path has no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.parallel_config = parallel_config or ParallelConfig()
        self.wan_base_model_path = wan_base_model_path
        self.enable_kv_offload = enable_kv_offload
        self._memory_mode = getattr(self.config, "memory_mode", "balanced")
        self._vae_chunk_size = getattr(self.config, "vae_chunk_size", None)
        self.latent_shape = list(getattr(self.config, "latent_shape", latent_shape))
        self.frames_per_segment = int(getattr(self.config, "image_or_video_shape", [1, 21])[1])
        self._initialize_pipeline(text_encoder=text_encoder, vae=vae, generator=generator)

    def _initialize_pipeline(self, text_encoder=None, vae=None, generator=None):
        torch.set_grad_enabled(False)
        args = self.config
        if generator is None:
            from ..wan import HipCausalWanModel, HipCausVidDiffusionWrapper
            mk = dict(getattr(args, "model_kwargs", {}) or {})
            shift = float(mk.pop("timestep_shift", getattr(args, "timestep_shift", 8.0)))
            model = HipCausalWanModel(parallel_config=self.parallel_config, device=self.device, **mk)
            generator = HipCausVidDiffusionWrapper(model=model, timestep_shift=shift, parallel_config=self.parallel_config)
        ps = getattr(generator.model, "patch_size", (1, 2, 2))
        if not hasattr(args, "frame_seq_length"):
            args.frame_seq_length = (self.latent_shape[1] // ps[1]) * (self.latent_shape[2] // ps[2])
        if not hasattr(args, "kv_cache_tokens"):
            args.kv_cache_tokens = self.frames_per_segment * args.frame_seq_length
        self.pipeline = CausVidInferencePipeline(args, wan_base_model_path=self.wan_base_model_path, device=self.device,
                                                 parallel_config=self.parallel_config, generator=generator,
                                                 text_encoder=text_encoder, vae=vae)
        if self.parallel_config.world_size > 1 and getattr(generator.model, "cp", None) is None and dist.is_initialized():
            from ..sequence_parallel import attach_sequence_parallel
            attach_sequence_parallel(generator.model, dist.group.WORLD)

    # ---- plugin contract --------------------------------------------------------------------------------------------------
    def load_checkpoint(self, checkpoint_path: str, **kwargs) -> None:
        """`<checkpoint_path>/model.pt` holding {'generator': sd} / {'generator_ema': sd} / a bare state dict (:76-97)."""
        state = torch.load(os.path.join(checkpoint_path, "model.pt"), map_location="cpu")
        key = "generator" if "generator" in state else ("generator_ema" if "generator_ema" in state else None)
        sd = state[key] if key is not None else state
        sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}
        self.pipeline.generator.model.load_state_dict(sd)

    def setup_devices(self, low_memory: bool = False, verbose: bool = True, use_memory_manager: bool = False) -> None:
        for name in ("text_encoder", "vae"):
            comp = getattr(self.pipeline, name)
            if isinstance(comp, torch.nn.Module):
                comp.to(self.device)

    def run_image_to_video(self, prompt, image_path, **kwargs):
        raise NotImplementedError("CausVid does not support image to video generation")

    def _generate_segment_with_streaming(self, prompt, initial_latent, stream_callback, segment_length: int = 21, **kwargs):
        raise NotImplementedError("Streaming generation is not yet implemented for CausVidPipeline. "
                                  "Please use the standard run_text_to_video method with num_rollout parameter.")

    # ---- text to video ------------------------------------------------------------------------------------------------------
    def run_text_to_video(self, prompts: List, output_folder: Optional[str] = None, num_rollout: int = 3,
                          num_overlap_frames: int = 3, is_diff_prompt: bool = False, is_interactive: bool = False,
                          prompt_source: Optional[Callable[[int], str]] = None, **kwargs):
        """-> list of uint8 frame arrays `[T, H, W, C]` (one per prompt, or one for the whole diff-prompt video); also saved
        under `output_folder` on rank 0.  `prompt_source(segment_id) -> str` replaces the reference's stdin prompt of the
        interactive mode (`get_prompt_from_shell`, :362-387; "Quit" ends the video) — under torch.distributed it is called on
        rank 0 and the string is broadcast exactly as upstream."""
        if self.pipeline.text_encoder is None or self.pipeline.vae is None:
            raise RuntimeError("CausVidPipeline.run_text_to_video needs text_encoder= and vae= (the boundary frame of every "
                               "segment goes through the VAE decoder and encoder)")
        assert num_overlap_frames % self.pipeline.num_frame_per_block == 0, \
            "num_overlap_frames must be divisible by num_frame_per_block"
        output_folder = output_folder or "./output"
        if self.parallel_config.rank == 0:
            os.makedirs(output_folder, exist_ok=True)
        kvm = KVCacheManager(device=self.device)
        if is_diff_prompt:
            return self._run_inference_diff_prompt(prompts=prompts, output_folder=output_folder,
                                                   num_overlap_frames=num_overlap_frames, kv_cache_manager=kvm,
                                                   is_interactive=is_interactive, prompt_source=prompt_source)
        return self._run_inference_same_prompt(prompts=prompts, output_folder=output_folder, num_rollout=num_rollout,
                                               num_overlap_frames=num_overlap_frames, kv_cache_manager=kvm)

    def _run_inference_same_prompt(self, prompts: List, output_folder: str, num_rollout: int = 3, num_overlap_frames: int = 3,
                                   kv_cache_manager: Optional[KVCacheManager] = None, **kwargs):
        results = []
        for prompt_idx, prompt in enumerate(prompts):
            reqs = [KVCacheRequest(prompt)]
            start_latents, all_video = None, []
            for _ in range(num_rollout):
                start_latents = self._generate_one_segment(prompt, start_latents, all_video, num_overlap_frames, reqs, kv_cache_manager)
            self.pipeline.clear_cache(kv_cache_manager, reqs)
            results.append(self._save_video(all_video, output_folder, f"prompt_{prompt_idx}"))
        return results

    def _run_inference_diff_prompt(self, prompts: List, output_folder: str, num_overlap_frames: int = 3,
                                   kv_cache_manager: Optional[KVCacheManager] = None, is_interactive: bool = False,
                                   prompt_source: Optional[Callable[[int], str]] = None, **kwargs):
        segment_id, start_latents, all_video = 0, None, []
        while True:
            if is_interactive:
                prompt = get_prompt(segment_id, prompt_source, self.device)
            else:
                prompt = prompts[segment_id] if segment_id < len(prompts) else "Quit"
            if prompt == "Quit":
                break
            reqs = [KVCacheRequest(f"segment_{segment_id}:{prompt}")]         # a NEW request per segment: the rollover
            segment_id += 1
            start_latents = self._generate_one_segment(prompt, start_latents, all_video, num_overlap_frames, reqs, kv_cache_manager)
            self.pipeline.clear_cache(kv_cache_manager, reqs)
        if not all_video:
            return []
        return [self._save_video(all_video, output_folder, f"segments_{segment_id}")]

    def _generate_one_segment(self, prompt: str, start_latents: Optional[torch.Tensor], all_video: List,
                              num_overlap_frames: int, kv_cache_requests: List, kv_cache_manager: KVCacheManager) -> torch.Tensor:
        """One segment of `frames_per_segment` latent frames (21 = 7 blocks), :269-309."""
        noise = torch.randn([1, self.frames_per_segment, *self.latent_shape], device=self.device, dtype=torch.bfloat16)
        video, latents = self.pipeline.inference(noise=noise, text_prompts=[prompt], return_latents=True,
                                                 start_latents=start_latents, kv_cache_manager=kv_cache_manager,
                                                 kv_cache_requests=kv_cache_requests, vae_chunk_size=self._vae_chunk_size)
        frames = video[0].permute(0, 2, 3, 1)                                                   # [T, H, W, C] in [0, 1]
        start_frame = self._encode_start_frame(video, num_overlap_frames)
        start_latents = torch.cat([start_frame, latents[:, -(num_overlap_frames - 1):]], dim=1) if num_overlap_frames > 1 \
            else start_frame
        keep = frames[:-(4 * (num_overlap_frames - 1) + 1)]
        all_video.append((keep.float().clamp(0, 1) * 255).to(torch.uint8).cpu())
        return start_latents

    def _encode_start_frame(self, video: torch.Tensor, num_overlap_frames: int) -> torch.Tensor:
        """The pixel frame `4 * (overlap - 1) + 1` from the end -> one latent frame `[B, 1, 16, h, w]` bf16 (:315-332)."""
        i = video.shape[1] - 4 * (num_overlap_frames - 1) - 1
        frame = (video[:, i:i + 1] * 2.0 - 1.0).transpose(2, 1).to(torch.bfloat16)               # [B, 3, 1, H, W]
        return self.pipeline.vae.encode_to_latent(frame).to(torch.bfloat16)

    def _save_video(self, all_video: List[torch.Tensor], output_folder: str, video_name: str):
        video = torch.cat(all_video, dim=0)
        if self.parallel_config.rank == 0:
            torch.save(video, os.path.join(output_folder, f"{video_name}.pt"))
        return video


def get_prompt(segment_id: int, prompt_source: Optional[Callable[[int], str]], device) -> str:
    """The interactive prompt of segment `segment_id`: asked on rank 0 (from `prompt_source`, default stdin) and broadcast to the
    other ranks as a byte tensor — `get_prompt_from_shell` (:362-387)."""
    ask = prompt_source or (lambda i: input(f"> Please give prompt for segment {i}. [You can input Quit to abort]:"))
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return ask(segment_id)
    rank = dist.get_rank()
    if rank == 0:
        data = torch.tensor(list(ask(segment_id).encode("utf-8")), dtype=torch.uint8, device=device)
        n = torch.tensor(data.numel(), dtype=torch.long, device=device)
    else:
        n = torch.tensor(0, dtype=torch.long, device=device)
    dist.broadcast(n, src=0)
    if rank != 0:
        data = torch.zeros(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(data, src=0)
    return bytes(data.tolist()).decode("utf-8")
