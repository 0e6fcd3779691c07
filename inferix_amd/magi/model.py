"""MAGI-1 `VideoDiTModel` on MI355X: one denoise forward of the whole model (BASELINE config 5 as a MODEL step, not a layer stack).

Mirrors `inferix/models/magi/dit/dit_model.py` of the reference:
  forward_pre_process / get_embedding_and_meta :111-330   patch embedding, rope table, timestep / caption embedders, condition and
                                                          condition map, packed cross- / core-attention ranges, cp_pre_process
  forward                                      :353-398   the layer stack (HipMagiTransformerBlock: every heavy op a HIP kernel)
  TransformerBlock.forward tail                dit_module.py:1386-1388   final LayerNorm on the fp32 hidden states
  forward_post_process / unpatchify            :97-107,332-351   FinalLinear, cp_post_process, (T H W) N (pT pH pW C) -> N C T H W

What runs where.  The 34 layers are >99.9 % of a forward's arithmetic and go through inferix_amd.magi.dit (HIP GEMMs, attention,
norms).  The embedders, the rope table, the final LayerNorm and the final linear are fp32 modules upstream (`_high_precision_promoter`,
:620-637; evaluated under `torch.autocast("cuda", dtype=torch.float32)`): here they are a handful of fp32 torch calls on the device —
glue by the contract's definition (a 64-wide patchify GEMM, two-layer MLPs on one row per denoising range, an 800-row caption
projection, a 3072 -> 64 head), with the same rounding points as the reference: the sinusoid is rounded to bf16 before the fp32
timestep MLP (`t_freq.to(params_dtype)`), x / condition / y_xattn_flat are rounded to bf16 behind the embedders.

Parity: tests/test_hip_magi_model.py against tests/golden/magi_model_tiny.npz (the reference's own VideoDiTModel.forward run by
oracle/gen_golden_magi_model.py; the layers follow the reference's CPU path as the layer goldens do).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from . import context_parallel as cpl
from .dit import HipMagiTransformerBlock
from .types import InferenceParams, ModelMetaArgs, PackedCoreAttnParams, PackedCrossAttnParams

BF16 = torch.bfloat16


def _cfg(c, name, default):
    return getattr(c, name, default)


class HipVideoDiTModel:
    """`VideoDiTModel(config)` with `.forward(x, t, y, caption_dropout_mask, xattn_mask, kv_range, inference_params, **kwargs)`.
    `config` carries `model_config` / `engine_config` (/ `runtime_config`) as the reference's MagiConfig does."""

    def __init__(self, config, device="cuda"):
        self.model_config, self.engine_config = config.model_config, config.engine_config
        self.runtime_config = getattr(config, "runtime_config", None)
        mc = self.model_config
        self.device = torch.device(device)
        self.patch_size, self.t_patch_size = _cfg(mc, "patch_size", 2), _cfg(mc, "t_patch_size", 1)
        self.in_channels, self.out_channels = _cfg(mc, "in_channels", 16), _cfg(mc, "out_channels", 16)
        self.caption_max_length = _cfg(mc, "caption_max_length", 800)
        self.x_rescale_factor, self.half_channel_vae = float(_cfg(mc, "x_rescale_factor", 1.0)), bool(_cfg(mc, "half_channel_vae", False))
        self.frequency_embedding_size = 256
        self.videodit_blocks = HipMagiTransformerBlock(mc, self.engine_config, device)
        self.w: Dict[str, torch.Tensor] = {}

    # ---- weights --------------------------------------------------------------------------------------------------------------
    def load_state_dict(self, W: Dict[str, torch.Tensor]) -> None:
        """State dict with the reference's names: `x_embedder.weight`, `t_embedder.mlp.{0,2}.*`, `y_embedder.*`, `rope.bands`,
        `videodit_blocks.layers.{i}.*`, `videodit_blocks.final_layernorm.*`, `final_linear.linear.weight`."""
        dev = self.device
        for k in ("x_embedder.weight", "t_embedder.mlp.0.weight", "t_embedder.mlp.0.bias", "t_embedder.mlp.2.weight", "t_embedder.mlp.2.bias",
                  "y_embedder.null_caption_embedding", "y_embedder.y_proj_xattn.0.weight", "y_embedder.y_proj_xattn.0.bias",
                  "y_embedder.y_proj_adaln.0.weight", "y_embedder.y_proj_adaln.0.bias", "rope.bands",
                  "videodit_blocks.final_layernorm.weight", "videodit_blocks.final_layernorm.bias", "final_linear.linear.weight"):
            if k not in W:
                raise KeyError(f"HipVideoDiTModel.load_state_dict: missing {k}")
            self.w[k] = W[k].to(dev, torch.float32).contiguous()          # fp32 modules (_high_precision_promoter)
        self.videodit_blocks.load_state_dict(W, "videodit_blocks.layers.")

    def load_synthetic(self, seed: int = 0, fp8: Optional[bool] = None) -> None:
        """Random weights of the whole model with the reference's shapes and dtypes, generated on the device layer by layer (no
        checkpoint exists offline): benchmarks and smoke runs.  `fp8` (default `engine_config.fp8_quant`): layers 1 .. n - 2 carry the
        static-scale e4m3 linears of an fp8_quant checkpoint (dit_module.py:434-490)."""
        from .dit import synthetic_layer_state_dict
        mc, dev = self.model_config, self.device
        fp8 = bool(_cfg(self.engine_config, "fp8_quant", False)) if fp8 is None else fp8
        g = torch.Generator(device=dev).manual_seed(seed)
        h, heads = mc.hidden_size, mc.num_attention_heads
        cond, xat = int(h * _cfg(mc, "cond_hidden_ratio", 0.25)), int(h * _cfg(mc, "xattn_cond_hidden_ratio", 1.0))
        cin = self.in_channels * (2 if self.half_channel_vae else 1)
        cap = _cfg(mc, "caption_channels", 4096)
        rnd = lambda *shape: torch.randn(*shape, generator=g, device=dev)
        mat = lambda *shape: rnd(*shape) * math.prod(shape[1:]) ** -0.5
        n = h // heads // 8
        bands = 1.0 / (10000.0 ** (torch.arange(0, n, dtype=torch.int64, device=dev).to(torch.float32) / n))
        self.w = {"x_embedder.weight": mat(h, cin, self.t_patch_size, self.patch_size, self.patch_size),
                  "t_embedder.mlp.0.weight": mat(cond, self.frequency_embedding_size), "t_embedder.mlp.0.bias": 0.1 * rnd(cond),
                  "t_embedder.mlp.2.weight": mat(cond, cond), "t_embedder.mlp.2.bias": 0.1 * rnd(cond),
                  "y_embedder.null_caption_embedding": rnd(self.caption_max_length, cap),
                  "y_embedder.y_proj_xattn.0.weight": mat(xat, cap), "y_embedder.y_proj_xattn.0.bias": 0.1 * rnd(xat),
                  "y_embedder.y_proj_adaln.0.weight": mat(cond, cap), "y_embedder.y_proj_adaln.0.bias": 0.1 * rnd(cond),
                  "rope.bands": bands, "videodit_blocks.final_layernorm.weight": 0.1 * rnd(h),
                  "videodit_blocks.final_layernorm.bias": 0.1 * rnd(h),
                  "final_linear.linear.weight": mat(self.patch_size * self.patch_size * self.t_patch_size * self.out_channels, h)}
        layers = self.videodit_blocks.layers
        for i, layer in enumerate(layers):
            layer.load_state_dict(synthetic_layer_state_dict(mc, seed=seed * 1000 + i, device=dev, fp8=fp8 and 0 < i < len(layers) - 1))

    # ---- pieces of get_embedding_and_meta ---------------------------------------------------------------------------------------
    def _timestep_embedding(self, t: torch.Tensor) -> torch.Tensor:
        """TimestepEmbedder.forward (dit_module.py:76-106): cos | sin of t * 1000 * 10000^(-i/half), rounded to the parameter dtype,
        then Linear -> SiLU -> Linear in fp32."""
        half = self.frequency_embedding_size // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(0, half, dtype=torch.float32, device=t.device) / half)
        args = t[:, None].float() * freqs[None] * 1000.0
        tf = torch.cat([torch.cos(args), torch.sin(args)], dim=-1).to(BF16).float()
        w = self.w
        return F.linear(F.silu(F.linear(tf, w["t_embedder.mlp.0.weight"], w["t_embedder.mlp.0.bias"])),
                        w["t_embedder.mlp.2.weight"], w["t_embedder.mlp.2.bias"])

    def _rope(self, t_total: int, H: int, W: int, rows: int) -> torch.Tensor:
        """LearnableRotaryEmbeddingCat.get_embed (dit_module.py:602-776, in_pixels=False) for [t_total, H, W] with the reference
        shape [t_total, H / r, W / r], r = sqrt(H W / 256) (dit_model.py:157-166); the last `rows` rows (sin | cos)."""
        dev, bands = self.device, self.w["rope.bands"]
        r = math.sqrt((H * W) / (16 * 16))
        shape, ref = [t_total, H, W], [t_total, H / r, W / r]
        axes = [torch.arange(s, dtype=torch.int64, device=dev).to(torch.float32) for s in shape]
        axes[1] = axes[1] - (H - 1) / 2
        axes[2] = axes[2] - (W - 1) / 2
        scaled = [a if f == 1 else a / (f - 1) * (rf - 1) for a, f, rf in zip(axes, shape, ref)]
        grid = torch.stack(torch.meshgrid(*scaled, indexing="ij"), dim=-1).unsqueeze(-1)
        pos = grid * bands
        n = t_total * H * W
        return torch.cat([pos.sin().reshape(n, -1), pos.cos().reshape(n, -1)], -1)[-rows:].contiguous()

    def forward_pre_process(self, x, t, y, caption_dropout_mask=None, xattn_mask=None, kv_range=None, **kwargs
                            ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, ModelMetaArgs]:
        assert kv_range is not None, "Please ensure kv_range is provided"
        if caption_dropout_mask is None:
            raise ValueError("caption_dropout_mask is required at inference: the AdaLN branch of the caption embedder reads the learned "
                             "null caption it selects (dit_module.py:141-160)")
        assert xattn_mask is not None
        dev, w = self.device, self.w
        x = x.to(dev) * self.x_rescale_factor
        if self.half_channel_vae:
            assert x.shape[1] == 16
            x = torch.cat([x, x], dim=1)
        x, t, y = x.float(), t.to(dev).float(), y.to(dev).float()
        # Part 1: patch embedding (a Conv3d whose stride equals its kernel)
        # stride == kernel: the convolution is one [tokens, C pT pH pW] x [C pT pH pW, hidden] product (MIOpen's fp32 Conv3d for this
        # shape is a naive direct kernel: 6 ms per call at 720 x 720, as long as four of the 34 layers)
        pt, pp = self.t_patch_size, self.patch_size
        N, C, Tf, Hf, Wf = x.shape
        T, H, Wd = Tf // pt, Hf // pp, Wf // pp
        cols = x[:, :, :T * pt, :H * pp, :Wd * pp].reshape(N, C, T, pt, H, pp, Wd, pp).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(N, T * H * Wd, -1)
        xe = F.linear(cols, w["x_embedder.weight"].reshape(w["x_embedder.weight"].shape[0], -1))      # [N, (T H W), hidden]
        range_num, dn = kwargs["range_num"], kwargs["denoising_range_num"]
        slice_point = kwargs.get("slice_point", 0)
        frame_in_range = T // dn
        # Part 2: rope
        rope = self._rope(T + frame_in_range * slice_point, H, Wd, T * H * Wd)
        # Part 3: timestep embedding
        assert t.shape[0] == N and t.shape[1] == dn, f"Invalid t shape {tuple(t.shape)}"
        te = self._timestep_embedding(t.flatten())
        if _cfg(self.engine_config, "distill", False):
            if kwargs["num_steps"] == 12:
                factor = 4 / kwargs["distill_interval"] * 2
            else:
                factor = kwargs["num_steps"] / 4 * 2
            te = te + self._timestep_embedding(torch.ones_like(t.flatten()) * factor)
        te = te.reshape(N, dn, -1)
        # Part 4: caption: cross-attention rows from the caption itself, the AdaLN row from the learned null caption
        y_xattn = F.silu(F.linear(y, w["y_embedder.y_proj_xattn.0.weight"], w["y_embedder.y_proj_xattn.0.bias"]))
        null = w["y_embedder.null_caption_embedding"]
        drop = caption_dropout_mask.to(dev).bool()
        cap = torch.where(drop[:, None, None], null[None, -1, :], null[None, -2, :])
        y_adaln = F.linear(cap, w["y_embedder.y_proj_adaln.0.weight"], w["y_embedder.y_proj_adaln.0.bias"]).squeeze(1)
        mask = xattn_mask.to(dev).squeeze(1).squeeze(1)
        condition = te + y_adaln.unsqueeze(1)
        assert condition.shape[0] == N and condition.shape[1] == dn
        per = (T * H * Wd) // dn
        condition_map = torch.repeat_interleave(torch.arange(N * dn, device=dev), per).reshape(N, -1).transpose(0, 1).contiguous()
        y_flat = torch.masked_select(y_xattn.squeeze(1), mask.unsqueeze(-1).bool()).reshape(-1, y_xattn.shape[-1])
        # Part 5 / 6: packed ranges
        y_index = mask.reshape(mask.shape[0], -1).sum(-1)
        clip = H * Wd * frame_in_range
        cu_q = torch.tensor([0] + [clip] * dn * N, device=dev).cumsum(-1).to(torch.int32)
        cu_k = torch.cat([y_index.new_zeros(1), y_index]).to(torch.int64).cumsum(-1).to(torch.int32)
        q_ranges = torch.stack([cu_q[:-1], cu_q[1:]], 1)
        k_ranges_x = torch.stack([cu_k[:-1], cu_k[1:]], 1)
        cross = PackedCrossAttnParams(q_ranges=q_ranges, kv_ranges=k_ranges_x, cu_seqlens_q=cu_q, cu_seqlens_kv=cu_k,
                                      max_seqlen_q=clip, max_seqlen_kv=self.caption_max_length)
        kv_range = kv_range.to(dev)
        flat_kv = torch.unique(kv_range, sorted=True)
        ardf = dict(clip_token_nums=clip, slice_point=slice_point, range_num=range_num, denoising_range_num=dn, q_range=q_ranges,
                    k_range=kv_range, max_seqlen_q=clip, max_seqlen_k=int((flat_kv[-1] - flat_kv[0]).item()))
        xs = xe.to(BF16).transpose(0, 1).contiguous()                                              # "N C T H W -> (T H W) N C"
        core = PackedCoreAttnParams(q_range=ardf["q_range"], k_range=ardf["k_range"], np_q_range=ardf["q_range"].cpu().numpy(),
                                    np_k_range=ardf["k_range"].cpu().numpy(), max_seqlen_q=clip, max_seqlen_k=ardf["max_seqlen_k"])
        ec = self.engine_config
        xs, condition_map, rope, pad, sizes, core, cross = cpl.cp_pre_process(
            _cfg(ec, "cp_size", 1), _cfg(ec, "cp_strategy", "none"), xs, condition_map, rope, None, ardf, core, cross)
        meta = ModelMetaArgs(H=H, W=Wd, cp_pad_size=pad, cp_split_sizes=sizes, slice_point=slice_point, denoising_range_num=dn,
                             range_num=range_num, extract_prefix_video_feature=kwargs.get("extract_prefix_video_feature", False),
                             fwd_extra_1st_chunk=kwargs["fwd_extra_1st_chunk"],
                             distill_nearly_clean_chunk=kwargs.get("distill_nearly_clean_chunk", False), clip_token_nums=clip,
                             enable_cuda_graph=False, core_attn_params=core, cross_attn_params=cross)
        return xs, condition.to(BF16), condition_map, y_flat.to(BF16), rope, meta

    def forward_post_process(self, hidden: torch.Tensor, meta: ModelMetaArgs) -> torch.Tensor:
        mc, w = self.model_config, self.w
        gamma = w["videodit_blocks.final_layernorm.weight"]
        gamma = gamma + 1 if bool(_cfg(mc, "apply_layernorm_1p", False)) else gamma
        hn = F.layer_norm(hidden.float(), (hidden.shape[-1],), gamma, w["videodit_blocks.final_layernorm.bias"], mc.layernorm_epsilon)
        o = F.linear(hn, w["final_linear.linear.weight"])                                        # (thw / cp, N, pT pH pW C)
        ec = self.engine_config
        o = cpl.cp_post_process(_cfg(ec, "cp_size", 1), _cfg(ec, "cp_strategy", "none"), o, meta)
        S, N, _ = o.shape
        pt, p, C = self.t_patch_size, self.patch_size, self.out_channels
        H, Wd = meta.H, meta.W
        T = S // (H * Wd)
        o = o.reshape(T, H, Wd, N, pt, p, p, C).permute(3, 7, 0, 4, 1, 5, 2, 6).reshape(N, C, T * pt, H * p, Wd * p).contiguous()
        if self.half_channel_vae:
            assert o.shape[1] == 32
            o = o[:, :16]
        return o / self.x_rescale_factor

    @torch.no_grad()
    def forward(self, x, t, y, caption_dropout_mask=None, xattn_mask=None, kv_range=None,
                inference_params: Optional[InferenceParams] = None, **kwargs) -> torch.Tensor:
        if x.shape[0] > 1:
            # a batch: only the unconditional pass of forward_3cfg has one (chunks turned into batch rows, no cache, dit_model.py:436-489).
            # The layers run the cached path with batch 1; rows of a batch do not interact, so they go through one at a time — row b with
            # its own timesteps, its captions (range_num = denoising ranges per row) and its key ranges rebased to the row.
            if inference_params is not None:
                raise NotImplementedError("HipVideoDiTModel.forward: a batch with a KV cache (the reference batches only forward_3cfg's "
                                          "unconditional pass, inference_params=None)")
            N, dn = x.shape[0], kwargs["denoising_range_num"]
            rows = kv_range.shape[0] // N
            drop = caption_dropout_mask.expand(N) if caption_dropout_mask.numel() == 1 else caption_dropout_mask
            outs = []
            for b in range(N):
                kv = kv_range[b * rows:(b + 1) * rows]
                outs.append(self.forward(x[b:b + 1], t[b:b + 1], y[b * dn:(b + 1) * dn], drop[b:b + 1], xattn_mask[b * dn:(b + 1) * dn],
                                         kv - kv.min(), None, **kwargs))
            return torch.cat(outs, dim=0)
        xs, condition, condition_map, y_flat, rope, meta = self.forward_pre_process(x, t, y, caption_dropout_mask, xattn_mask, kv_range, **kwargs)
        hs = self.videodit_blocks(xs.clone(), condition, condition_map, y_flat, rope, inference_params, meta)
        return self.forward_post_process(hs, meta)

    __call__ = forward

    @torch.no_grad()
    def forward_3cfg(self, x, timestep, y, mask, kv_range, inference_params, **kwargs):
        """dit_model.py:399-492: the three forwards of classifier-free guidance — (previous chunks + text) WITHOUT touching the cache,
        (previous chunks, null caption) which writes it, and the unconditional one: the denoising chunks as batch rows that see only
        themselves, no cache.  -> (out_cond_pre_and_text, out_cond_pre, out_uncond, denoise_width)."""
        assert x.shape[0] == 2 and mask.shape[0] % 2 == 0
        x = torch.cat([x[0:1], x[0:1]], dim=0)
        half = y.shape[0] // 2
        drop = torch.tensor([False, True], dtype=torch.bool, device=x.device)
        kwargs = dict(kwargs)
        inference_params.update_kv_cache = False
        out_text = self.forward(x[0:1], timestep[0:1], y[0:half], caption_dropout_mask=drop[0:1], xattn_mask=mask[0:half],
                                kv_range=kv_range, inference_params=inference_params, **kwargs)
        inference_params.update_kv_cache = True
        out_pre = self.forward(x[1:2], timestep[1:2], y[half:], caption_dropout_mask=drop[1:2], xattn_mask=mask[half:],
                               kv_range=kv_range, inference_params=inference_params, **kwargs)
        cw = kwargs["chunk_width"]
        dn = kwargs["denoising_range_num"] - (1 if kwargs.get("fwd_extra_1st_chunk", False) else 0)     # UnconditionGuard (:448-469)
        denoise_width = cw * dn
        u = x[0:1, :, -denoise_width:].squeeze(0)
        uncond_x = u.reshape(-1, dn, cw, *u.shape[2:]).transpose(0, 1)                                   # chunk_to_batch: [dn, C, cw, H, W]
        ukw = dict(kwargs, range_num=1, denoising_range_num=1, slice_point=0, fwd_extra_1st_chunk=False)
        out_uncond = self.forward(uncond_x, timestep[0:1, -dn:].transpose(0, 1), y[half:][-dn:],
                                  caption_dropout_mask=torch.tensor([True], dtype=torch.bool, device=x.device), xattn_mask=mask[half:][-dn:],
                                  kv_range=self.generate_kv_range_for_uncondition(uncond_x), inference_params=None, **ukw)
        o = out_uncond.transpose(0, 1)
        out_uncond = o.reshape(1, -1, dn * cw, *o.shape[3:])                                             # batch_to_chunk
        return out_text, out_pre, out_uncond, denoise_width

    def _dispatch_3cfg(self, x, timestep, y, mask, kv_range, inference_params, **kwargs) -> torch.Tensor:
        """dit_model.py:500-535: per denoising chunk, scales looked up by its timestep in `cfg_t_range`:
        (1 - s_prev) uncond + (s_prev - s_text) cond_pre + s_text cond_pre_and_text."""
        rc = self.runtime_config
        out_text, out_pre, out_uncond, denoise_width = self.forward_3cfg(x, timestep, y, mask, kv_range, inference_params, **kwargs)
        dev = out_text.device
        prev_s = torch.tensor(rc.prev_chunk_scales, device=dev)
        text_s = torch.tensor(rc.text_scales, device=dev)
        t_range = torch.tensor(rc.cfg_t_range, device=dev)
        assert len(prev_s) == len(t_range) and len(text_s) == len(t_range), "prev_chunk_scales / text_scales and cfg_t_range differ in length"
        n, cw = kwargs["denoising_range_num"], kwargs["chunk_width"]
        if kwargs["fwd_extra_1st_chunk"]:
            n -= 1
        cfg_t = timestep[0, -n:].to(dev)
        pieces = []
        for ci in range(n):
            idx = torch.searchsorted(t_range - 1e-7, cfg_t[ci]) - 1                                       # get_cfg_scale (:494-497)
            assert 0 <= int(idx) < len(prev_s)
            sp, st = prev_s[idx], text_s[idx]
            sl = slice(ci * cw, (ci + 1) * cw)
            pieces.append((1 - sp) * out_uncond[:, :, sl] + (sp - st) * out_pre[:, :, -denoise_width:][:, :, sl]
                          + st * out_text[:, :, -denoise_width:][:, :, sl])
        x = torch.cat([x[0:1, :, :-denoise_width].to(dev), torch.cat(pieces, dim=2)], dim=2)
        return torch.cat([x, x], dim=0)

    def generate_kv_range_for_uncondition(self, uncond_x: torch.Tensor) -> torch.Tensor:
        """dit_model.py:91-100: one self-contained key range per batch row of `uncond_x`."""
        B, _, T, H, W = uncond_x.shape
        n = (T // self.t_patch_size) * (H // self.patch_size) * (W // self.patch_size)
        return torch.tensor([[b * n, (b + 1) * n] for b in range(B)], dtype=torch.int32, device=self.device)

    @torch.no_grad()
    def forward_dispatcher(self, x, timestep, y, mask, kv_range, inference_params, **kwargs) -> torch.Tensor:
        """dit_model.py:499-596 for `runtime_config.cfg_number == 1` (the distilled checkpoints: one conditional forward, no guidance
        batch).  x `[2, C, T, H, W]` with equal halves, y / mask hold the conditional rows first.  When the newest finished chunk is
        nearly clean (`distill_nearly_clean_chunk`) it is forwarded a second time as an extra range that sees only itself, and the two
        results are blended `prev_chunks_scale : 1 - prev_chunks_scale` (env `prev_chunks_scale`, 0.7)."""
        import os
        rc = self.runtime_config
        if rc is not None and getattr(rc, "cfg_number", None) == 3:
            return self._dispatch_3cfg(x, timestep, y, mask, kv_range, inference_params, **kwargs)
        if rc is None or getattr(rc, "cfg_number", None) != 1:
            raise NotImplementedError("HipVideoDiTModel.forward_dispatcher: runtime_config.cfg_number must be 1 (distilled checkpoints) or 3 "
                                      "(the reference's own NotImplementedError for anything else, dit_model.py:596)")
        assert x.shape[0] == 2
        x = torch.cat([x[0:1], x[0:1]], dim=0)
        kwargs = dict(kwargs)
        kwargs["caption_dropout_mask"] = torch.tensor([False], dtype=torch.bool, device=x.device)
        inference_params.update_kv_cache = True
        half = y.shape[0] // 2
        cw = kwargs["chunk_width"]
        if kwargs.get("distill_nearly_clean_chunk", False):
            scale = float(os.getenv("prev_chunks_scale", 0.7))
            s0 = 1 if kwargs["fwd_extra_1st_chunk"] else 0
            width = x.shape[2]
            new_x = x[0:1, :, s0 * cw:(s0 + 1) * cw]
            new_kv = self.generate_kv_range_for_uncondition(new_x) + kv_range.max()
            kwargs["denoising_range_num"] += 1
            out = self.forward(torch.cat([x[0:1], new_x], dim=2), torch.cat([timestep[0:1], timestep[0:1, s0:s0 + 1]], dim=1),
                               torch.cat([y[0:half], y[s0:s0 + 1]], dim=0), xattn_mask=torch.cat([mask[0:half], mask[s0:s0 + 1]], dim=0),
                               kv_range=torch.cat([kv_range.to(new_kv.device), new_kv], dim=0), inference_params=inference_params, **kwargs)
            out[:, :, s0 * cw:(s0 + 1) * cw] = out[:, :, s0 * cw:(s0 + 1) * cw] * scale + out[:, :, width:] * (1 - scale)
            out = out[:, :, :width]
        else:
            out = self.forward(x[0:1], timestep[0:1], y[0:half], xattn_mask=mask[0:half], kv_range=kv_range,
                               inference_params=inference_params, **kwargs)
        denoise_width = cw * kwargs["denoising_range_num"]
        if kwargs["fwd_extra_1st_chunk"]:
            denoise_width -= cw
        x = torch.cat([x[0:1, :, :-denoise_width], out[:, :, -denoise_width:]], dim=2)
        return torch.cat([x[0:1], x[0:1]], dim=0)
