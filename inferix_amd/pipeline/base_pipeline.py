"""`AbstractInferencePipeline`: the plugin base class of the reference (inferix/pipeline/base_pipeline.py:16-1270), restricted
to what sits either side of the denoising path — the template methods (`run`, `__call__`, `setup`), the segment loops
(`run_streaming_generation` :468-615, `run_interactive_generation` :747-934 with its boundary checks :936-1090), the decode
strategies (`_decode_latent` :1217-1270) and the memory-mode presets (`_apply_memory_mode` :1188-1215).

Not reproduced (out of scope): the profiling reporter (the `_get_profiler_context` hooks are no-ops unless a profiler
object is injected), the asynchronous memory manager / component offload of :203-357 (everything is resident in 288 GB of HBM).
"""
from __future__ import annotations

import contextlib
from abc import ABC, abstractmethod
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch

from ..core.interactive import (ControlCommand, SegmentBoundary, calculate_total_frames, validate_overlap_config)
from ..core.types import DecodeMode, MemoryMode


class AbstractInferencePipeline(ABC):
    def __init__(self, config: Dict[str, Any], profiling_config=None):
        self.config = config
        self.profiling_config = profiling_config
        self._profiling_enabled = bool(getattr(profiling_config, "enabled", False))
        self._profiler = getattr(profiling_config, "profiler", None)
        self._is_setup = False
        self._free_cache_before_vae = True
        self._vae_chunk_size: Optional[int] = None

    # ---- profiling hooks (the reporter itself is outside this build) ---------------------------------------------------
    def _get_profiler_context(self, stage_name: str, metadata: Optional[Dict[str, Any]] = None):
        prof = self._profiler
        if self._profiling_enabled and prof is not None and hasattr(prof, "stage"):
            return prof.stage(stage_name, metadata)
        return contextlib.nullcontext()

    def cleanup_profiling(self):
        if self._profiling_enabled and self._profiler is not None and hasattr(self._profiler, "cleanup"):
            self._profiler.cleanup()

    # ---- plugin contract --------------------------------------------------------------------------------------------------
    @abstractmethod
    def load_checkpoint(self, checkpoint_path: str, **kwargs) -> None:
        ...

    @abstractmethod
    def run_text_to_video(self, prompts: List[str], **kwargs) -> Any:
        ...

    @abstractmethod
    def run_image_to_video(self, prompts: List[str], image_path: str, **kwargs) -> Any:
        ...

    def setup_devices(self, low_memory: bool = False, verbose: bool = True, use_memory_manager: bool = False) -> None:
        """Subclasses move their weights to HBM here; the 16-24 GB offload modes of the reference do not apply."""

    def setup(self):
        self.setup_devices()
        self._is_setup = True

    def run(self, inputs: Dict[str, Any], **kwargs) -> Any:
        """Dispatch on the input keys (base_pipeline.py:408-435): `prompts` (+ `image_path`), or the single-prompt spellings."""
        with self._get_profiler_context("pipeline_run"):
            if "prompts" in inputs and "image_path" in inputs:
                return self.run_image_to_video(inputs["prompts"], inputs["image_path"], **kwargs)
            if "prompts" in inputs:
                return self.run_text_to_video(inputs["prompts"], **kwargs)
            if "prompt" in inputs and "image_path" in inputs:
                return self.run_image_to_video([inputs["prompt"]], inputs["image_path"], **kwargs)
            if "prompt" in inputs:
                return self.run_text_to_video([inputs["prompt"]], **kwargs)
            raise ValueError("Invalid inputs for pipeline execution")

    def __call__(self, **kwargs) -> Any:
        with self._get_profiler_context("pipeline_call"):
            if not self._is_setup:
                self.setup()
            return self.run(inputs=kwargs)

    # ---- memory / decode presets -------------------------------------------------------------------------------------------
    def _apply_memory_mode(self, mode: MemoryMode, vae_chunk_size: Optional[int] = None):
        """AGGRESSIVE: free the KV cache before the VAE, chunks of 2; BALANCED: free, 4; RELAXED: keep, 7.  An explicit
        `vae_chunk_size` wins (base_pipeline.py:1188-1215)."""
        if isinstance(mode, str):
            mode = MemoryMode(mode)
        self._free_cache_before_vae = mode != MemoryMode.RELAXED
        preset = {MemoryMode.AGGRESSIVE: 2, MemoryMode.RELAXED: 7}.get(mode, 4)
        self._vae_chunk_size = vae_chunk_size if vae_chunk_size is not None else preset

    def _decode_latent(self, latent: torch.Tensor, vae, decode_mode: DecodeMode = DecodeMode.AFTER_ALL, chunk_size: int = 2,
                       stream_callback: Optional[Callable[[torch.Tensor], None]] = None, block_size: int = 3) -> Optional[torch.Tensor]:
        """`[B, T, C, H, W]` latents -> pixels in [0, 1] (None for NO_DECODE); PER_BLOCK decodes `block_size` frames at a time
        and hands every decoded block to `stream_callback` (base_pipeline.py:1217-1270)."""
        if decode_mode == DecodeMode.NO_DECODE:
            return None
        if decode_mode == DecodeMode.AFTER_ALL:
            return (vae.decode_to_pixel(latent, use_cache=True, chunk_size=chunk_size) * 0.5 + 0.5).clamp(0, 1)
        videos = []
        for start in range(0, latent.shape[1], block_size):
            block = vae.decode_to_pixel(latent[:, start:start + block_size], use_cache=True, chunk_size=chunk_size)
            block = (block * 0.5 + 0.5).clamp(0, 1)
            if stream_callback:
                stream_callback(block)
            videos.append(block)
        return torch.cat(videos, dim=1)

    # ---- segment loops ---------------------------------------------------------------------------------------------------
    @abstractmethod
    def _generate_segment_with_streaming(self, prompt: str, initial_latent: Optional[torch.Tensor],
                                         stream_callback: Optional[Callable[[torch.Tensor], None]], segment_length: int = 21,
                                         **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (video `[B, T, H, W, C]` float in [0, 1] on the host, final latents `[B, T, C, H, W]`)."""

    def _cleanup_segment_memory(self):
        pass

    def run_streaming_generation(self, prompts: List[str], stream_callback: Optional[Callable[[torch.Tensor], None]] = None,
                                 num_segments: int = 1, segment_length: int = 21, overlap_frames: int = 3,
                                 **kwargs) -> Optional[torch.Tensor]:
        """Prompts cycle over segments; the last `overlap_frames` latent frames of a segment condition the next one; videos are
        concatenated along time -> `[B, T_total, H, W, C]` on the host (base_pipeline.py:468-615)."""
        videos = []
        initial_latent = None
        for seg in range(num_segments):
            video, final_latent = self._generate_segment_with_streaming(
                prompt=prompts[seg % len(prompts)], initial_latent=initial_latent, stream_callback=stream_callback,
                segment_length=segment_length, **kwargs)
            videos.append(video)
            if seg < num_segments - 1:
                initial_latent = final_latent[:, -overlap_frames:]
            self._cleanup_segment_memory()
        if not videos:
            return None
        return torch.cat(videos, dim=1) if len(videos) > 1 else videos[0]

    def run_interactive_generation(self, session, initial_prompt: str, num_segments: int = 1, segment_length: int = 21,
                                   overlap_frames: int = 3, stream_callback: Optional[Callable[[torch.Tensor], None]] = None,
                                   block_size: int = 3, **kwargs) -> Optional[torch.Tensor]:
        """`run_streaming_generation` with the prompts coming from an `InteractiveSession` (base_pipeline.py:747-934): at every
        segment boundary the session's checkpoint is evaluated — STOP ends the loop, a queued prompt / guidance change takes
        effect for the segment about to start, a pause blocks until resume — then the segment is generated from the previous
        one's overlap latents.  Every segment gets fresh KV-cache requests, so a prompt change never reads the old prompt's cache."""
        self._validate_boundary_config(segment_length=segment_length, overlap_frames=overlap_frames, block_size=block_size,
                                       num_segments=num_segments)
        session.set_initial_prompt(initial_prompt)
        session.set_generation_params(total_segments=num_segments, blocks_per_segment=segment_length // block_size)
        session.start_session()
        videos: List[torch.Tensor] = []
        initial_latent = None
        prompt = initial_prompt
        guidance = kwargs.pop("guidance_scale", 7.5)
        failed = False
        try:
            for seg in range(num_segments):
                self._validate_segment_boundary(segment_idx=seg, num_segments=num_segments, overlap_frames=overlap_frames,
                                                segment_length=segment_length, block_size=block_size, initial_latent=initial_latent)
                cp = session.evaluate_checkpoint(checkpoint_type="segment", checkpoint_index=seg, current_prompt=prompt,
                                                 current_guidance=guidance)
                if cp.command == ControlCommand.STOP:
                    break
                if cp.new_prompt:
                    prompt = cp.new_prompt
                if cp.new_guidance:
                    guidance = cp.new_guidance
                while session.should_pause():
                    if session.should_stop():
                        break
                    session.wait_for_resume(timeout=0.1)
                if session.should_stop():
                    break
                video, final_latent = self._generate_segment_with_streaming(
                    prompt=prompt, initial_latent=initial_latent, stream_callback=stream_callback, segment_length=segment_length,
                    guidance_scale=guidance, **kwargs)
                videos.append(video)
                session.update_progress(segment_idx=seg, block_idx=segment_length // block_size - 1,
                                        frames_generated=sum(v.shape[1] for v in videos),
                                        gpu_memory_gb=torch.cuda.memory_allocated() / 1e9 if torch.cuda.is_available() else 0.0)
                if seg < num_segments - 1:
                    initial_latent = self._extract_overlap_latent(final_latent=final_latent, overlap_frames=overlap_frames,
                                                                  segment_idx=seg)
                self._cleanup_segment_memory()
        except Exception:
            failed = True
            raise
        finally:
            session.end_session(error=failed)
        if not videos:
            return None
        return torch.cat(videos, dim=1) if len(videos) > 1 else videos[0]

    # ---- boundary checks (base_pipeline.py:936-1090) ------------------------------------------------------------------------
    def _validate_boundary_config(self, segment_length: int, overlap_frames: int, block_size: int, num_segments: int):
        if segment_length <= 0:
            raise ValueError(f"segment_length must be positive, got {segment_length}")
        if segment_length % block_size != 0:
            raise ValueError(f"segment_length ({segment_length}) must be divisible by block_size ({block_size})")
        validate_overlap_config(overlap_frames, block_size)
        if overlap_frames >= segment_length:
            raise ValueError(f"overlap_frames ({overlap_frames}) must be less than segment_length ({segment_length})")
        if num_segments <= 0:
            raise ValueError(f"num_segments must be positive, got {num_segments}")

    def _validate_segment_boundary(self, segment_idx: int, num_segments: int, overlap_frames: int, segment_length: int,
                                   block_size: int, initial_latent: Optional[torch.Tensor]) -> SegmentBoundary:
        first, last = segment_idx == 0, segment_idx == num_segments - 1
        start = 0 if first else segment_idx * (segment_length - overlap_frames)
        if initial_latent is not None:
            if first:
                raise ValueError(f"First segment (idx=0) should have initial_latent=None, got tensor with shape {tuple(initial_latent.shape)}")
            if initial_latent.shape[1] != overlap_frames:
                raise ValueError(f"initial_latent has {initial_latent.shape[1]} frames, expected {overlap_frames} overlap frames "
                                 f"for segment {segment_idx}")
        elif not first:
            raise ValueError(f"Non-first segment (idx={segment_idx}) requires initial_latent with {overlap_frames} frames, got None")
        return SegmentBoundary(segment_idx=segment_idx, start_frame=start, end_frame=start + segment_length - 1,
                               unique_frames=segment_length if first else segment_length - overlap_frames,
                               overlap_with_previous=0 if first else overlap_frames, is_first=first, is_last=last)

    def _extract_overlap_latent(self, final_latent: torch.Tensor, overlap_frames: int, segment_idx: int) -> torch.Tensor:
        if overlap_frames <= 0:
            raise ValueError(f"overlap_frames must be positive, got {overlap_frames}")
        if final_latent is None:
            raise ValueError("final_latent is None, cannot extract overlap")
        if final_latent.shape[1] < overlap_frames:
            raise ValueError(f"final_latent has {final_latent.shape[1]} frames, cannot extract {overlap_frames} overlap frames")
        return final_latent[:, -overlap_frames:]

    @staticmethod
    def total_frames(num_segments: int, segment_length: int, overlap_frames: int) -> int:
        return calculate_total_frames(num_segments, segment_length, overlap_frames)
