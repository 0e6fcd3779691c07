"""HipWanDiffusionWrapper — one denoise step = model forward + flow->x0.

Mirror of `WanDiffusionWrapper` (inferix/models/self_forcing/wrapper.py:171-398): the generator object that
`CausalInferencePipeline(..., generator=...)` accepts.  Same `forward` keyword signature and return
`(flow_pred, pred_x0)` `[B, F, C, H, W]`; `.model`, `.get_scheduler()`, `.parameters()` as the pipeline uses."""
from __future__ import annotations

import json
import os
from typing import List, Optional

import torch

from ..kvcache_manager import KVCacheManager, KVCacheRequest
from ..schedulers import FlowMatchScheduler
from .causal_model import HipCausalWanModel, ParallelConfig


class HipWanDiffusionWrapper(torch.nn.Module):
    def __init__(self, model: Optional[HipCausalWanModel] = None, model_path: Optional[str] = None,
                 model_name: str = "Wan2.1-T2V-1.3B", timestep_shift: float = 8.0, is_causal: bool = True,
                 local_attn_size: int = -1, sink_size: int = 0, enable_kv_offload: bool = False,
                 parallel_config: Optional[ParallelConfig] = None, device="cuda"):
        super().__init__()
        if not is_causal:
            raise NotImplementedError("only the causal (KV-cached) generator is built")
        self.parallel_config = parallel_config if parallel_config is not None else ParallelConfig()
        self.enable_kv_offload = enable_kv_offload
        if model is None:
            if model_path is None or not os.path.exists(model_path):
                raise FileNotFoundError(f"Model path not found: {model_path}")
            model = self.load_pretrained(model_path, local_attn_size=local_attn_size, sink_size=sink_size,
                                         parallel_config=self.parallel_config, device=device)
        self.model = model
        self.uniform_timestep = False
        self.scheduler = FlowMatchScheduler(shift=timestep_shift, sigma_min=0.0, extra_one_step=True)
        self.scheduler.set_timesteps(1000, training=True)
        self.seq_len = 32760
        self._sig64 = None
        from ..schedulers import TensorMemo
        self._sigma_memo = TensorMemo()

    @staticmethod
    def load_pretrained(model_path: str, **kw) -> HipCausalWanModel:
        """diffusers-style directory: config.json + *.safetensors with the reference's key names."""
        from safetensors.torch import load_file
        with open(os.path.join(model_path, "config.json")) as f:
            cfg = json.load(f)
        keys = ("model_type", "patch_size", "text_len", "in_dim", "dim", "ffn_dim", "freq_dim", "text_dim",
                "out_dim", "num_heads", "num_layers", "eps")
        model = HipCausalWanModel(**{k: cfg[k] for k in keys if k in cfg}, **kw)
        sd = {}
        for fn in sorted(os.listdir(model_path)):
            if fn.endswith(".safetensors"):
                sd.update(load_file(os.path.join(model_path, fn)))
        return model.load_state_dict(sd)

    def parameters(self, recurse: bool = True):
        return self.model.parameters()

    def get_scheduler(self) -> FlowMatchScheduler:
        return self.scheduler

    def _convert_flow_pred_to_x0(self, flow_pred: torch.Tensor, xt: torch.Tensor, timestep: torch.Tensor) -> torch.Tensor:
        """x0 = x_t - sigma_t * flow in fp64, sigma picked by nearest timestep (wrapper.py:259-283)."""
        dev = flow_pred.device
        epoch = getattr(self.scheduler, "table_epoch", 0)
        if self._sig64 is None or self._sig64[0].device != dev or self._sig64[2] != epoch:
            self._sig64 = (self.scheduler.sigmas.double().to(dev), self.scheduler.timesteps.double().to(dev), epoch)
        sig, ts, _ = self._sig64
        # the sigma of a pipeline-made constant timestep tensor is looked up once per value (schedulers.TensorMemo: keyed on the
        # host-known scalar, never on a storage address; any other tensor is looked up every call)
        sigma = self._sigma_memo.get(timestep, (str(dev), epoch), lambda: sig[torch.argmin(
            (ts.unsqueeze(0) - timestep.to(dev).double().unsqueeze(1)).abs(), dim=1)].reshape(-1, 1, 1, 1))
        return (xt.double() - sigma * flow_pred.double()).to(flow_pred.dtype)

    @torch.no_grad()
    def forward(self, noisy_image_or_video: torch.Tensor, conditional_dict: dict, timestep: torch.Tensor,
                kv_cache_meta: Optional[List[dict]] = None, crossattn_cache_meta: Optional[List[dict]] = None,
                current_start: Optional[int] = None, classify_mode: Optional[bool] = False,
                concat_time_embeddings: Optional[bool] = False, clean_x: Optional[torch.Tensor] = None,
                aug_t: Optional[torch.Tensor] = None, cache_start: Optional[int] = None,
                kv_cache_manager: Optional[KVCacheManager] = None,
                kv_cache_requests: Optional[List[KVCacheRequest]] = None):
        if kv_cache_meta is None or classify_mode or clean_x is not None:
            raise NotImplementedError("HipWanDiffusionWrapper implements the KV-cached inference call only")
        flow = self.model(noisy_image_or_video.permute(0, 2, 1, 3, 4), t=timestep,
                          context=conditional_dict["prompt_embeds"], seq_len=self.seq_len,
                          kv_cache_meta=kv_cache_meta, crossattn_cache_meta=crossattn_cache_meta,
                          current_start=current_start, cache_start=cache_start,
                          kv_cache_manager=kv_cache_manager, kv_cache_requests=kv_cache_requests
                          ).permute(0, 2, 1, 3, 4)
        from ..schedulers import carry_tag
        x0 = self._convert_flow_pred_to_x0(flow.flatten(0, 1), noisy_image_or_video.flatten(0, 1).to(flow.device),
                                           carry_tag(timestep, timestep.flatten(0, 1))).unflatten(0, flow.shape[:2])
        return flow, x0


    @torch.no_grad()
    def forward_pair(self, first: dict, second: dict):
        """Two `forward` calls (their keyword arguments) enqueued layer-interleaved on two streams (HipCausalWanModel.forward_pair) — for
        the clean-context re-run of one block and the first denoising step of the next, which depend on each other only through the
        cache, layer by layer.  Returns `((flow, x0), (flow, x0))`, bit-identical to the two calls made one after the other."""
        def model_kw(kw):
            explicit = kw.get("kv_start") is not None and kw.get("kv_end") is not None        # the CausVid call names its slots
            if (kw.get("kv_cache_meta") is None and not explicit) or kw.get("classify_mode") or kw.get("clean_x") is not None:
                raise NotImplementedError("HipWanDiffusionWrapper implements the KV-cached inference call only")
            return dict(x=kw["noisy_image_or_video"].permute(0, 2, 1, 3, 4), t=kw["timestep"],
                        context=kw["conditional_dict"]["prompt_embeds"], kv_cache_meta=kw.get("kv_cache_meta"),
                        crossattn_cache_meta=kw.get("crossattn_cache_meta"), current_start=kw["current_start"],
                        kv_start=kw.get("kv_start"), kv_end=kw.get("kv_end"),
                        kv_cache_manager=kw["kv_cache_manager"], kv_cache_requests=kw["kv_cache_requests"])
        flows = self.model.forward_pair(model_kw(first), model_kw(second))
        out = []
        for kw, flow in zip((first, second), flows):
            flow = flow.permute(0, 2, 1, 3, 4)
            xt, ts = kw["noisy_image_or_video"], kw["timestep"]
            x0 = self._convert_flow_pred_to_x0(flow.flatten(0, 1), xt.flatten(0, 1).to(flow.device), ts.flatten(0, 1)).unflatten(0, flow.shape[:2])
            out.append((flow, x0))
        return out[0], out[1]


class HipCausVidDiffusionWrapper(HipWanDiffusionWrapper):
    """CausVid generator (inferix/models/causvid/wrapper.py:269-304): explicit `kv_start/kv_end` cache slots per
    call, returns `pred_x0` only.  Same kernels as Self-Forcing; the default timestep shift is 8.0."""

    @torch.no_grad()
    def forward(self, noisy_image_or_video: torch.Tensor, conditional_dict: dict, timestep: torch.Tensor,
                kv_start: Optional[int] = None, kv_end: Optional[int] = None, current_start: Optional[int] = None,
                current_end: Optional[int] = None, kv_cache_manager: Optional[KVCacheManager] = None,
                kv_cache_requests: Optional[List[KVCacheRequest]] = None) -> torch.Tensor:
        if kv_start is None or kv_end is None:
            raise ValueError("CausVid generator needs explicit kv_start / kv_end")
        flow = self.model(noisy_image_or_video.permute(0, 2, 1, 3, 4), t=timestep,
                          context=conditional_dict["prompt_embeds"], seq_len=self.seq_len, kv_start=kv_start,
                          kv_end=kv_end, current_start=current_start, current_end=current_end,
                          kv_cache_manager=kv_cache_manager, kv_cache_requests=kv_cache_requests
                          ).permute(0, 2, 1, 3, 4)
        return self._convert_flow_pred_to_x0(flow.flatten(0, 1), noisy_image_or_video.flatten(0, 1).to(flow.device),
                                             timestep.flatten(0, 1)).unflatten(0, flow.shape[:2])

    @torch.no_grad()
    def forward_pair(self, first: dict, second: dict):
        """The CausVid form of HipWanDiffusionWrapper.forward_pair: two calls' keyword arguments (each names its own cache slots),
        `(pred_x0, pred_x0)` back — bit-identical to the two calls one after the other."""
        (_, xa), (_, xb) = super().forward_pair(first, second)
        return xa, xb
