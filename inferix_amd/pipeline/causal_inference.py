"""CausalInferencePipeline — the block loop x denoise-step loop of Self-Forcing.

Same constructor and `inference()` signature as the reference
(inferix/pipeline/self_forcing/CausalInferencePipeline.py:57-123): injected `generator`,
`text_encoder`, `vae`; per block: `len(denoising_step_list)` generator forwards with re-noising in
between, then a clean-context re-run at `context_noise` that overwrites the block's KV in place
(`:257-393`); `block_callback(block_latent, block_index)` per block; NO_DECODE returns latents.

MI355X-first differences behind the same API:
  * `kv_cache_meta` index entries are HOST tensors: the slot arithmetic runs on the host and the
    reference's >=5 `.item()` device syncs per layer per forward (causal_model.py:255,282-300) vanish;
  * timing uses events on the current stream and a single synchronize per step only when a profiler
    asks for it (`record_diffusion_step` / `record_block_computation` keep the reference's contract,
    inferix/profiling/profiler.py:387,418);
  * `renoise` (extension): optional list of pre-drawn Gaussian tensors consumed instead of
    `torch.randn_like` (teacher-forced parity runs).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Union

import torch

from ..schedulers import const_timestep

from ..core.types import DecodeMode
from ..kvcache_manager import KVCacheManager, KVCacheRequest


class CausalInferencePipeline(torch.nn.Module):
    def __init__(self, args, device, generator=None, text_encoder=None, vae=None, parallel_config=None,
                 profiler=None):
        super().__init__()
        if generator is None:
            from ..wan import HipWanDiffusionWrapper
            generator = HipWanDiffusionWrapper(model_path=getattr(args, "model_path", "weights/Wan2.1-T2V-1.3B"),
                                               **getattr(args, "model_kwargs", {}), is_causal=True,
                                               parallel_config=parallel_config)
        self.generator, self.text_encoder, self.vae = generator, text_encoder, vae
        self.parallel_config = parallel_config if parallel_config is not None else generator.parallel_config
        self._profiler = profiler
        self.device_ = torch.device(device)
        self.scheduler = self.generator.get_scheduler()
        steps = torch.tensor(list(args.denoising_step_list), dtype=torch.long)
        if getattr(args, "warp_denoising_step", False):
            ts = torch.cat((self.scheduler.timesteps.cpu(), torch.tensor([0], dtype=torch.float32)))
            steps = ts[1000 - steps]
        self.denoising_step_list = steps
        model = self.generator.model
        self.num_transformer_blocks = getattr(model, "num_layers", 30)
        self.frame_seq_length = getattr(args, "frame_seq_length", 1560)
        self.kv_cache_meta = None
        self.crossattn_cache_meta = None
        self._ts_cache: dict = {}
        self.args = args
        self.num_frame_per_block = getattr(args, "num_frame_per_block", 1)
        self.independent_first_frame = getattr(args, "independent_first_frame", False)
        self.local_attn_size = model.local_attn_size
        if self.num_frame_per_block > 1:
            model.num_frame_per_block = self.num_frame_per_block

    # ------------------------------------------------------------------
    def _gen_kw(self, x, cond, timestep, start_frame, kvm, reqs) -> dict:
        return dict(noisy_image_or_video=x, conditional_dict=cond, timestep=timestep,
                    kv_cache_meta=self.kv_cache_meta, crossattn_cache_meta=self.crossattn_cache_meta,
                    current_start=start_frame * self.frame_seq_length, kv_cache_manager=kvm, kv_cache_requests=reqs)

    def _timestep(self, value, shape, device, dtype=torch.int64) -> torch.Tensor:
        """`torch.ones(shape) * value` (CausalInferencePipeline.py:330,371), ONE tensor per (value, shape) reused across blocks: the
        model and the x0 / add_noise conversions memoise what they derive from a timestep tensor (the modulation tables, the sigma
        lookups) on the scalar it holds — `schedulers.const_timestep` tags the tensor with it — ~25 glue launches per forward less.
        A tagged tensor is a constant: never written, in place or through a raw pointer."""
        key = (float(value), tuple(shape), str(device), dtype)
        t = self._ts_cache.get(key)
        if t is None:
            if len(self._ts_cache) >= 32:
                self._ts_cache.clear()
            t = self._ts_cache[key] = const_timestep(value, shape, device, dtype)
        return t

    def _gen(self, x, cond, timestep, start_frame, kvm, reqs):
        return self.generator(**self._gen_kw(x, cond, timestep, start_frame, kvm, reqs))

    def _pairing(self) -> bool:
        """Whether the clean-context re-run of block b is enqueued TOGETHER with the first denoising step of block b + 1, layer by
        layer on two streams (`generator.forward_pair`): the two calls depend on each other only through layer l's cache rows, so the
        second chain's launches fill the tails and the dependent-launch gaps of the first (- 2 .. 3 % of a clip on one GPU,
        - 4 .. 5 % on an emulated sequence-parallel rank).  The results are bit-identical to the sequential calls
        (tests/test_hip_model.py).  `args.pair_forwards` / env IFX_PAIR_FORWARDS: 1 on, 0 off; unset = on."""
        import os
        if not hasattr(self.generator, "forward_pair"):
            return False
        want = getattr(self.args, "pair_forwards", None)
        if want is None and os.environ.get("IFX_PAIR_FORWARDS", "") != "":
            want = os.environ["IFX_PAIR_FORWARDS"] not in ("0", "false", "off")
        if want is None:
            want = True
        return bool(want)

    def inference(self, noise: torch.Tensor, text_prompts: List[str], kv_cache_manager: KVCacheManager,
                  kv_cache_requests: List[KVCacheRequest], initial_latent: Optional[torch.Tensor] = None,
                  return_latents: bool = False, profile: bool = False, low_memory: bool = False,
                  free_cache_before_vae: bool = True, decode_mode: DecodeMode = DecodeMode.AFTER_ALL,
                  vae_chunk_size: Optional[int] = None, block_callback: Optional[Callable] = None,
                  vae_decode_context=None, renoise: Optional[Sequence[torch.Tensor]] = None
                  ) -> Union[torch.Tensor, tuple]:
        B, T, Cc, Hh, Ww = noise.shape
        nfb = self.num_frame_per_block
        if not self.independent_first_frame or initial_latent is not None:
            assert T % nfb == 0
            num_blocks = T // nfb
        else:
            assert (T - 1) % nfb == 0
            num_blocks = (T - 1) // nfb
        n_in = initial_latent.shape[1] if initial_latent is not None else 0
        cond = self.text_encoder(text_prompts=text_prompts)
        dev = noise.device
        output = torch.zeros([B, T + n_in, Cc, Hh, Ww], device=dev, dtype=noise.dtype)
        renoise = list(renoise) if renoise is not None else None

        # ---- caches -----------------------------------------------------------------------
        if self.kv_cache_meta is None:
            self._initialize_kv_cache(kv_cache_manager, kv_cache_requests, dtype=noise.dtype)
        else:
            for m in self.kv_cache_meta:
                m["global_end_index"] = torch.tensor([0], dtype=torch.long)
                m["local_end_index"] = torch.tensor([0], dtype=torch.long)
        if self.crossattn_cache_meta is None:
            self._initialize_crossattn_cache(kv_cache_manager, kv_cache_requests, dtype=noise.dtype)
        else:
            for m in self.crossattn_cache_meta:
                m["is_init"] = False

        # ---- prefill of given context frames at t = 0 ---------------------------------------
        cur = 0
        if initial_latent is not None:
            t0 = torch.zeros([B, 1], device=dev, dtype=torch.int64)
            if self.independent_first_frame:
                assert (n_in - 1) % nfb == 0
                n_pref = (n_in - 1) // nfb
                output[:, :1] = initial_latent[:, :1]
                self._gen(initial_latent[:, :1], cond, t0, cur, kv_cache_manager, kv_cache_requests)
                cur += 1
            else:
                assert n_in % nfb == 0
                n_pref = n_in // nfb
            for _ in range(n_pref):
                ref = initial_latent[:, cur:cur + nfb]
                output[:, cur:cur + nfb] = ref
                self._gen(ref, cond, t0, cur, kv_cache_manager, kv_cache_requests)
                cur += nfb

        # ---- temporal (blocks) x spatial (denoise steps) loops -------------------------------
        frames = [nfb] * num_blocks
        if self.independent_first_frame and initial_latent is None:
            frames = [1] + frames
        want_steps = self._profiler is not None and hasattr(self._profiler, "record_diffusion_step")
        want_blocks = profile and self._profiler is not None and hasattr(self._profiler, "record_block_computation")
        self.block_times_ms: List[float] = []
        pair = self._pairing() and not want_steps and not profile      # the per-step / per-block timers want one call per interval
        pending = None                   # keyword arguments of the previous block's clean-context re-run, deferred into the next step
        for block_index, nf in enumerate(frames):
            if profile:
                b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                b0.record()
            x = noise[:, cur - n_in:cur + nf - n_in]
            x0 = timestep = None
            nsteps = len(self.denoising_step_list)
            for index, tcur in enumerate(self.denoising_step_list):
                if want_steps:
                    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s0.record()
                timestep = self._timestep(tcur, (B, nf), dev)
                if pending is not None:
                    # the previous block's clean-context re-run and this first step, layer-interleaved (same results as back to back)
                    _, (_, x0) = self.generator.forward_pair(pending, self._gen_kw(x, cond, timestep, cur, kv_cache_manager,
                                                                                     kv_cache_requests))
                    pending = None
                else:
                    _, x0 = self._gen(x, cond, timestep, cur, kv_cache_manager, kv_cache_requests)
                if index < nsteps - 1:
                    flat = x0.flatten(0, 1)
                    eps = renoise.pop(0).to(flat.device, flat.dtype) if renoise is not None else torch.randn_like(flat)
                    tn = self._timestep(self.denoising_step_list[index + 1], (B * nf,), dev, torch.long)
                    x = self.scheduler.add_noise(flat, eps, tn).unflatten(0, x0.shape[:2])
                if want_steps:
                    s1.record()
                    torch.cuda.synchronize()
                    try:
                        self._profiler.record_diffusion_step(step=index, timestep=float(tcur) / 1000.0, block_size=nf,
                                                             computation_time_ms=s0.elapsed_time(s1),
                                                             guidance_scale=getattr(self.args, "guidance_scale", None))
                    except Exception:   # profiler errors never break generation (reference behaviour)
                        pass
            if x0 is not None:
                output[:, cur:cur + nf] = x0
                ctx_t = self._timestep(getattr(self.args, "context_noise", 0), tuple(timestep.shape), dev, timestep.dtype)
                if pair and block_index + 1 < len(frames):
                    pending = self._gen_kw(x0, cond, ctx_t, cur, kv_cache_manager, kv_cache_requests)
                else:
                    self._gen(x0, cond, ctx_t, cur, kv_cache_manager, kv_cache_requests)
            if profile:
                b1.record()
                torch.cuda.synchronize()
                self.block_times_ms.append(b0.elapsed_time(b1))
                if want_blocks:
                    try:
                        self._profiler.record_block_computation(
                            block_index=block_index, block_size=nf, computation_time_ms=self.block_times_ms[-1],
                            memory_usage_mb=torch.cuda.max_memory_allocated() / (1024 * 1024))
                    except Exception:
                        pass
            cur += nf
            if block_callback is not None and x0 is not None:
                block_callback(output[:, cur - nf:cur], block_index)

        cp = getattr(getattr(self.generator, "model", None), "cp", None)
        if cp is not None and hasattr(cp, "check_now"):
            cp.check_now()             # a sequence-parallel rank: a peer-store wait that gave up is reported before the clip is handed out
        if output.is_cuda:
            from .. import hip_ops
            hip_ops.check_device("CausalInferencePipeline.inference", sync=True)    # a split-K wait that gave up: the clip is garbage — raise, do not ship it
        if free_cache_before_vae:
            self.clear_cache(kv_cache_manager, kv_cache_requests)
        if decode_mode == DecodeMode.NO_DECODE:
            return (output, output) if return_latents else output
        if self.vae is None or not hasattr(self.vae, "decode_to_pixel"):
            raise RuntimeError("decode requested but no VAE with decode_to_pixel() was injected")
        chunk = vae_chunk_size if vae_chunk_size is not None else 2
        if vae_decode_context is not None:
            with vae_decode_context:
                video = self.vae.decode_to_pixel(output, use_cache=True, chunk_size=chunk)
        else:
            video = self.vae.decode_to_pixel(output, use_cache=True, chunk_size=chunk)
        video = (video * 0.5 + 0.5).clamp(0, 1)
        return (video, output) if return_latents else video

    # ------------------------------------------------------------------
    def _initialize_kv_cache(self, kv_cache_manager, kv_cache_requests, dtype):
        size = self.local_attn_size * self.frame_seq_length if self.local_attn_size != -1 \
            else getattr(self.args, "kv_cache_tokens", 32760)
        # The reference shards each rank's cache `size / ring_size` tokens x `heads / ulysses_size` heads
        # (CausalInferencePipeline.py:444-470 -> self_forcing_kv_cache_manager.py:45-57) for its all-to-all + ring attention.
        # Here every degree maps onto the sequence-parallel exchange with a REPLICATED cache in the single-GPU token order
        # (inferix_amd/sequence_parallel.py), so ParallelConfig(ulysses_size, ring_size) does not resize anything.
        u = r = 1
        blocks = self.generator.model.blocks
        for l in range(self.num_transformer_blocks):
            for req in kv_cache_requests:
                blocks[l].kv_cache_manager.allocate_kv_cache(kv_cache_manager=kv_cache_manager, kv_cache_request=req,
                                                             sequence_length=size, dtype=dtype, ulysses_size=u,
                                                             ring_size=r)
        self.kv_cache_meta = [{"global_end_index": torch.tensor([0], dtype=torch.long),
                               "local_end_index": torch.tensor([0], dtype=torch.long)}
                              for _ in range(self.num_transformer_blocks)]

    def _initialize_crossattn_cache(self, kv_cache_manager, kv_cache_requests, dtype):
        blocks = self.generator.model.blocks
        text_len = getattr(self.generator.model, "text_len", 512)
        for l in range(self.num_transformer_blocks):
            for req in kv_cache_requests:
                blocks[l].kv_cache_manager.allocate_crossattn_cache(kv_cache_manager=kv_cache_manager,
                                                                    kv_cache_request=req, crossattn_length=text_len,
                                                                    dtype=dtype)
        self.crossattn_cache_meta = [{"is_init": False} for _ in range(self.num_transformer_blocks)]

    def clear_cache(self, kv_cache_manager, kv_cache_requests):
        blocks = self.generator.model.blocks
        for l in range(self.num_transformer_blocks):
            for req in kv_cache_requests:
                blocks[l].kv_cache_manager.clear_cache(kv_cache_manager=kv_cache_manager, kv_cache_request=req)
        self.kv_cache_meta = None
        self.crossattn_cache_meta = None
