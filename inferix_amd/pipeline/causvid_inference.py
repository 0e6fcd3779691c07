"""CausVid block loop + continuous-prompt KV rollover.

`CausVidInferencePipeline.inference` mirrors the reference's CausVid inner pipeline
(inferix/pipeline/causvid/CausalInferencePipeline.py:94-257): the LAST entry of `denoising_step_list` is
dropped (`:37`), blocks covered by `start_latents` are prefilled at t = 0, every block is addressed by explicit
cache slots `kv_start/kv_end = block * frames_per_block * frame_seq_length`, re-noising uses a `[B]` timestep, and
the clean block is re-run at t = 0 to overwrite its KV.  `rollover` mirrors the per-segment request swap of
`CausVidPipeline` (inferix/pipeline/causvid/pipeline.py:224-309): every segment gets a NEW `KVCacheRequest` (its
own prompt), is prefilled with the previous segment's last `overlap` latents, and the old request is freed.
The pixel-space re-encode of the boundary frame (`_encode_start_frame`, VAE) is outside the path: callers pass a
`reencode(latents)` hook (identity by default).  Frame geometry is a parameter (`frame_seq_length`: 1560 for
480p, 3600 for 720p latents) instead of the reference's hard-wired 480x832.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

from ..schedulers import const_timestep

from ..kvcache_manager import KVCacheManager, KVCacheRequest


class CausVidInferencePipeline(torch.nn.Module):
    def __init__(self, args, wan_base_model_path=None, device="cuda", enable_kv_offload=False, parallel_config=None,
                 generator=None, text_encoder=None, vae=None):
        super().__init__()
        if generator is None:
            from ..wan import HipCausVidDiffusionWrapper
            generator = HipCausVidDiffusionWrapper(model_path=wan_base_model_path or "weights/Wan2.1-T2V-1.3B",
                                                   **getattr(args, "model_kwargs", {}),
                                                   enable_kv_offload=enable_kv_offload, parallel_config=parallel_config)
        self.generator, self.text_encoder, self.vae = generator, text_encoder, vae
        self.parallel_config = parallel_config if parallel_config is not None else generator.parallel_config
        self.scheduler = generator.get_scheduler()
        steps = torch.tensor(list(args.denoising_step_list), dtype=torch.long)[:-1]
        if getattr(args, "warp_denoising_step", False):
            ts = torch.cat((self.scheduler.timesteps.cpu(), torch.tensor([0], dtype=torch.float32)))
            steps = ts[1000 - steps]
        self.denoising_step_list = steps
        self.num_transformer_blocks = getattr(generator.model, "num_layers", 30)
        self.frame_seq_length = getattr(args, "frame_seq_length", 1560)
        self.kv_cache_tokens = getattr(args, "kv_cache_tokens", 32760)
        self.is_kv_cache_initialized = False
        self.args = args
        self.num_frame_per_block = getattr(args, "num_frame_per_block", 1)
        if self.num_frame_per_block > 1:
            generator.model.num_frame_per_block = self.num_frame_per_block

    # ---- caches ---------------------------------------------------------------------------------
    def _initialize_kv_cache(self, kv_cache_manager, kv_cache_requests, dtype):
        for blk in self.generator.model.blocks:
            for req in kv_cache_requests:
                blk.kv_cache_manager.allocate_kv_cache(kv_cache_manager=kv_cache_manager, kv_cache_request=req,
                                                       sequence_length=self.kv_cache_tokens, dtype=dtype)

    def _initialize_crossattn_cache(self, kv_cache_manager, kv_cache_requests, dtype):
        tl = getattr(self.generator.model, "text_len", 512)
        for blk in self.generator.model.blocks:
            for req in kv_cache_requests:
                blk.kv_cache_manager.allocate_crossattn_cache(kv_cache_manager=kv_cache_manager, kv_cache_request=req,
                                                              crossattn_length=tl, dtype=dtype)

    def _reset_crossattn_cache(self):
        for blk in self.generator.model.blocks:
            blk.is_cross_attn_init = False

    def clear_cache(self, kv_cache_manager, kv_cache_requests):
        for blk in self.generator.model.blocks:
            for req in kv_cache_requests:
                blk.kv_cache_manager.clear_cache(kv_cache_manager=kv_cache_manager, kv_cache_request=req)
        self.is_kv_cache_initialized = False

    # ---- one segment ------------------------------------------------------------------------------
    def _gen_kw(self, x, cond, timestep, block_index, kvm, reqs) -> dict:
        n = self.num_frame_per_block * self.frame_seq_length
        return dict(noisy_image_or_video=x, conditional_dict=cond, timestep=timestep,
                    current_start=block_index * n, current_end=(block_index + 1) * n,
                    kv_start=block_index * n, kv_end=(block_index + 1) * n, kv_cache_manager=kvm, kv_cache_requests=reqs)

    def _gen(self, x, cond, timestep, block_index, kvm, reqs):
        return self.generator(**self._gen_kw(x, cond, timestep, block_index, kvm, reqs))

    def _pairing(self) -> bool:
        """As CausalInferencePipeline._pairing: the clean-context re-run of a block is enqueued layer-interleaved with the next block's
        first denoising step (`generator.forward_pair`; bit-identical results).  `args.pair_forwards` / env IFX_PAIR_FORWARDS; on by default."""
        import os
        if not hasattr(self.generator, "forward_pair"):
            return False
        want = getattr(self.args, "pair_forwards", None) if getattr(self, "args", None) is not None else None
        if want is None and os.environ.get("IFX_PAIR_FORWARDS", "") != "":
            want = os.environ["IFX_PAIR_FORWARDS"] not in ("0", "false", "off")
        return True if want is None else bool(want)

    def _timestep(self, value, shape, device, dtype=torch.int64) -> torch.Tensor:
        """`torch.ones(shape) * value`, one tensor per (value, shape) reused across blocks (the model memoises its modulation tables
        and the sigma lookups on the scalar a `schedulers.const_timestep` tensor holds; see CausalInferencePipeline._timestep).
        Never written, in place or through a raw pointer."""
        cache = self.__dict__.setdefault("_ts_cache", {})
        key = (float(value), tuple(shape), str(device), dtype)
        t = cache.get(key)
        if t is None:
            if len(cache) >= 32:
                cache.clear()
            t = cache[key] = const_timestep(value, shape, device, dtype)
        return t

    def inference(self, noise: torch.Tensor, text_prompts: List[str], start_latents: Optional[torch.Tensor],
                  return_latents: bool = True, kv_cache_manager: Optional[KVCacheManager] = None,
                  kv_cache_requests: Optional[List] = None, vae_chunk_size: Optional[int] = None,
                  decode: bool = True, renoise: Optional[Sequence[torch.Tensor]] = None):
        B, T, C, H, W = noise.shape
        nfb = self.num_frame_per_block
        cond = self.text_encoder(text_prompts=text_prompts)
        dev = noise.device
        output = torch.zeros([B, T, C, H, W], device=dev, dtype=noise.dtype)
        renoise = list(renoise) if renoise is not None else None
        if not self.is_kv_cache_initialized:
            self._initialize_kv_cache(kv_cache_manager, kv_cache_requests, dtype=noise.dtype)
            self._initialize_crossattn_cache(kv_cache_manager, kv_cache_requests, dtype=noise.dtype)
            self.is_kv_cache_initialized = True
        # NOTE (deliberate deviation): the reference resets the per-block `is_cross_attn_init` flag only when the
        # caches already exist (`:123-136`), so after a request swap (`clear_cache` + new KVCacheRequest) the flag
        # stays True and the new prompt's text K/V are never computed — the blocks read the freshly allocated,
        # uninitialised cross cache.  Here a new segment always recomputes them.
        self._reset_crossattn_cache()
        n_in = start_latents.shape[1] // nfb if start_latents is not None else 0
        pair = self._pairing()
        pending = None                   # the previous block's clean-context re-run, deferred into this block's first step
        for blk in range(T // nfb):
            sl = slice(blk * nfb, (blk + 1) * nfb)
            if blk < n_in:
                ref = start_latents[:, sl]
                output[:, sl] = ref
                self._gen(ref, cond, torch.zeros([B, nfb], device=dev, dtype=torch.int64), blk, kv_cache_manager,
                          kv_cache_requests)
                continue
            x = noise[:, sl]
            x0 = timestep = None
            nsteps = len(self.denoising_step_list)
            for index, tcur in enumerate(self.denoising_step_list):
                timestep = self._timestep(tcur, (B, nfb), dev)
                if pending is not None:
                    _, x0 = self.generator.forward_pair(pending, self._gen_kw(x, cond, timestep, blk, kv_cache_manager, kv_cache_requests))
                    pending = None
                else:
                    x0 = self._gen(x, cond, timestep, blk, kv_cache_manager, kv_cache_requests)
                if index < nsteps - 1:
                    flat = x0.flatten(0, 1)
                    eps = renoise.pop(0).to(flat.device, flat.dtype) if renoise is not None else torch.randn_like(flat)
                    tn = self._timestep(self.denoising_step_list[index + 1], (B,), dev, torch.long)
                    x = self.scheduler.add_noise(flat, eps, tn).view(x0.shape)
            if x0 is None:
                raise RuntimeError(f"no denoising step ran for block {blk}")
            output[:, sl] = x0
            t0 = self._timestep(0, tuple(timestep.shape), dev, timestep.dtype)          # `timestep * 0`
            if pair and blk + 1 < T // nfb:
                pending = self._gen_kw(x0, cond, t0, blk, kv_cache_manager, kv_cache_requests)
            else:
                self._gen(x0, cond, t0, blk, kv_cache_manager, kv_cache_requests)
        cp = getattr(getattr(self.generator, "model", None), "cp", None)
        if cp is not None and hasattr(cp, "check_now"):
            cp.check_now()             # a sequence-parallel rank: a peer-store wait that gave up is reported before the clip is handed out
        if output.is_cuda:
            from .. import hip_ops
            hip_ops.check_device("CausVidInferencePipeline.inference", sync=True)    # a split-K wait that gave up: the clip is garbage — raise, do not ship it
        if not decode or self.vae is None:
            return (output, output) if return_latents else output
        chunk = vae_chunk_size if vae_chunk_size is not None else 2
        video = (self.vae.decode_to_pixel(output, use_cache=True, chunk_size=chunk) * 0.5 + 0.5).clamp(0, 1)
        return (video, output) if return_latents else video

    # ---- continuous-prompt rollover ---------------------------------------------------------------------
    def rollover(self, prompts: Sequence[str], noises: Sequence[torch.Tensor], kv_cache_manager: KVCacheManager,
                 overlap_frames: int, reencode: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                 renoise: Optional[Sequence[Sequence[torch.Tensor]]] = None) -> List[torch.Tensor]:
        """One segment per prompt; segment i > 0 starts from the last `overlap_frames` latents of segment i-1 (after
        `reencode`), in a fresh KVCacheRequest; the previous request's caches are freed (pipeline.py:247-260)."""
        outs: List[torch.Tensor] = []
        start = None
        prev_req = None
        for i, (prompt, noise) in enumerate(zip(prompts, noises)):
            req = [KVCacheRequest(f"segment_{i}")]
            if prev_req is not None:
                self.clear_cache(kv_cache_manager, prev_req)
                kv_cache_manager.free(prev_req[0])
            lat = self.inference(noise, [prompt], start, return_latents=False, kv_cache_manager=kv_cache_manager,
                                 kv_cache_requests=req, decode=False,
                                 renoise=renoise[i] if renoise is not None else None)
            outs.append(lat)
            tail = lat[:, -overlap_frames:]
            start = reencode(tail) if reencode is not None else tail
            prev_req = req
        return outs
