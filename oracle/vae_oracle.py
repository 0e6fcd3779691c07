"""TEST INFRASTRUCTURE — CPU oracle of the Wan2.1 VAE *decoder* as Inferix streams it.

A functional, state-dict-driven restatement of the reference's decode path; only tests, `smoke()` and bench.py's CPU leg may
import it.  It follows, op for op and rounding for rounding (every op's result is a bf16 tensor, as in the reference which
runs the VAE under `model.to(dtype=torch.bfloat16)`, inferix/pipeline/self_forcing/pipeline.py:172):

  * `WanVAEWrapper.decode_to_pixel`          inferix/models/self_forcing/wrapper.py:103-168 (latent [B,T,C,H,W] -> pixels)
  * `WanVAE_.decode` / `cached_decode`       inferix/models/wan_base/vae.py:543-566, 573-594   (one latent frame at a time)
  * `Decoder3d.forward`                      vae.py:415-466   (conv1, middle, upsamples, head with the feature cache)
  * `ResidualBlock.forward`                  vae.py:201-219   (norm, SiLU, causal conv x2, 1x1x1 shortcut)
  * `AttentionBlock.forward`                 vae.py:238-262   (single-head spatial attention per frame)
  * `Resample.forward` upsample2d/3d         vae.py:101-141   (time conv + frame interleave, nearest 2x + conv2d, 'Rep' rule)
  * `CausalConv3d.forward`                   vae.py:26-34     (two frames of causal padding, shortened by the cache)
  * `RMS_norm.forward`                       vae.py:52-55

  * encoder (image-to-video start frames): `WanVAEWrapper.encode_to_latent` wrapper.py:88-101, `WanVAE_.encode` vae.py:512-541
    (chunks of 1, 4, 4, ... frames), `Encoder3d.forward` vae.py:318-377, `Resample` downsample2d/3d vae.py:91-100, 143-156.

Pinned by `oracle/gen_golden_vae.py` against the reference's own `WanVAE_` run on CPU (tests/golden/vae_decode.npz, vae_encode.npz).

The feature cache is restated as "the last two input frames of every causal conv" keyed by layer name instead of the
reference's positional list; `None` = nothing cached yet, `REP` = the first-chunk marker of the temporal upsamplers.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

BF = torch.bfloat16
CACHE_T = 2                      # vae.py:12
REP = "Rep"

VAE_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
            0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]      # wrapper.py:65-72 (published
VAE_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,            # latent statistics of Wan2.1)
           3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


@dataclass(frozen=True)
class VaeConfig:
    dim: int = 96                                    # vae.py:619-626 (`_video_vae` defaults)
    z_dim: int = 16
    dim_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    temperal_downsample: Tuple[bool, ...] = (False, True, True)

    @property
    def temperal_upsample(self) -> Tuple[bool, ...]:
        return tuple(self.temperal_downsample[::-1])             # vae.py:498


def decoder_plan(cfg: VaeConfig) -> List[tuple]:
    """Execution order of `Decoder3d` (vae.py:381-413): ('res', prefix, cin, cout) | ('attn', prefix, c) |
    ('up3d' | 'up2d', prefix, c)."""
    dims = [cfg.dim * u for u in [cfg.dim_mult[-1]] + list(cfg.dim_mult[::-1])]
    plan = [("res", "decoder.middle.0", dims[0], dims[0]), ("attn", "decoder.middle.1", dims[0]),
            ("res", "decoder.middle.2", dims[0], dims[0])]
    n = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(cfg.num_res_blocks + 1):
            plan.append(("res", f"decoder.upsamples.{n}", cin, cout))
            cin = cout
            n += 1
        if i != len(cfg.dim_mult) - 1:
            plan.append(("up3d" if cfg.temperal_upsample[i] else "up2d", f"decoder.upsamples.{n}", cout))
            n += 1
    return plan


def decoder_param_shapes(cfg: VaeConfig) -> Dict[str, Tuple[int, ...]]:
    dims0 = cfg.dim * cfg.dim_mult[-1]
    s: Dict[str, Tuple[int, ...]] = {"conv2.weight": (cfg.z_dim, cfg.z_dim, 1, 1, 1), "conv2.bias": (cfg.z_dim,),
                                     "decoder.conv1.weight": (dims0, cfg.z_dim, 3, 3, 3), "decoder.conv1.bias": (dims0,)}
    last = dims0
    for item in decoder_plan(cfg):
        kind, p = item[0], item[1]
        if kind == "res":
            cin, cout = item[2], item[3]
            s[f"{p}.residual.0.gamma"] = (cin, 1, 1, 1)
            s[f"{p}.residual.2.weight"] = (cout, cin, 3, 3, 3)
            s[f"{p}.residual.2.bias"] = (cout,)
            s[f"{p}.residual.3.gamma"] = (cout, 1, 1, 1)
            s[f"{p}.residual.6.weight"] = (cout, cout, 3, 3, 3)
            s[f"{p}.residual.6.bias"] = (cout,)
            if cin != cout:
                s[f"{p}.shortcut.weight"] = (cout, cin, 1, 1, 1)
                s[f"{p}.shortcut.bias"] = (cout,)
            last = cout
        elif kind == "attn":
            c = item[2]
            s[f"{p}.norm.gamma"] = (c, 1, 1)
            s[f"{p}.to_qkv.weight"] = (3 * c, c, 1, 1)
            s[f"{p}.to_qkv.bias"] = (3 * c,)
            s[f"{p}.proj.weight"] = (c, c, 1, 1)
            s[f"{p}.proj.bias"] = (c,)
        else:
            c = item[2]
            s[f"{p}.resample.1.weight"] = (c // 2, c, 3, 3)
            s[f"{p}.resample.1.bias"] = (c // 2,)
            if kind == "up3d":
                s[f"{p}.time_conv.weight"] = (2 * c, c, 3, 1, 1)
                s[f"{p}.time_conv.bias"] = (2 * c,)
            last = c // 2
    s["decoder.head.0.gamma"] = (last, 1, 1, 1)
    s["decoder.head.2.weight"] = (3, last, 3, 3, 3)
    s["decoder.head.2.bias"] = (3,)
    return s


def make_decoder_params(cfg: VaeConfig, seed: int) -> Dict[str, torch.Tensor]:
    """Seeded synthetic decoder weights (bf16).  Convolutions get variance-preserving gaussians (so activations neither die
    nor blow up through 19 residual blocks), gammas ~ 1, biases small; the attention `proj` is NOT zero as upstream
    initialises it (vae.py:236) so that the attention block contributes to the goldens."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape in decoder_param_shapes(cfg).items():
        if name.endswith("gamma"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5
        W[name] = t.to(BF)
    return W


# ------------------------------------------------------------------------------------------------------------------
def causal_conv3d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, cache_x: Optional[torch.Tensor]) -> torch.Tensor:
    """vae.py:26-34: pad (k_t - 1) * 1 ... the layer pads 2*padding_t frames in front (none behind); cached frames
    replace that many zero frames.  Spatial padding symmetric."""
    kt, kh, kw = w.shape[2:]
    pad_t = kt - 1 if kt > 1 else 0
    if cache_x is not None and pad_t > 0:
        x = torch.cat([cache_x.to(x.dtype), x], dim=2)
        pad_t -= cache_x.shape[2]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, pad_t, 0))
    return F.conv3d(x, w, b)


def rms_norm(x: torch.Tensor, gamma: torch.Tensor, channel_dim: int = 1) -> torch.Tensor:
    """vae.py:52-55 with bias = 0 (every decoder norm): F.normalize(x, dim=C) * sqrt(C) * gamma (+ 0.)."""
    return F.normalize(x, dim=channel_dim) * (x.shape[channel_dim] ** 0.5) * gamma + 0.0


class VaeDecoderOracle:
    """State-dict-driven decoder with the reference's streaming cache semantics."""

    def __init__(self, cfg: VaeConfig, W: Dict[str, torch.Tensor], dtype=BF):
        self.cfg, self.dtype = cfg, dtype
        self.W = {k: v.to(dtype) for k, v in W.items()}
        self.plan = decoder_plan(cfg)
        self.mean = torch.tensor(VAE_MEAN[:cfg.z_dim], dtype=torch.float32)
        self.std = torch.tensor(VAE_STD[:cfg.z_dim], dtype=torch.float32)
        self.clear_cache()

    # ---- cache -------------------------------------------------------------------------------------------------
    def clear_cache(self) -> None:                                    # vae.py:603-611
        self.cache: Dict[str, object] = {}

    def _cached_conv(self, name: str, x: torch.Tensor) -> torch.Tensor:
        """The recurring pattern of vae.py:207-216 / 417-427 / 455-464: remember the last two input frames (topped up
        with the previous chunk's last frame when the chunk is a single frame), convolve with the old cache."""
        old = self.cache.get(name)
        cache_x = x[:, :, -CACHE_T:].clone()
        if cache_x.shape[2] < 2 and old is not None:
            cache_x = torch.cat([old[:, :, -1:], cache_x], dim=2)
        y = causal_conv3d(x, self.W[name + ".weight"], self.W[name + ".bias"], old)
        self.cache[name] = cache_x
        return y

    # ---- blocks ------------------------------------------------------------------------------------------------
    def _res(self, p: str, x: torch.Tensor, cin: int, cout: int) -> torch.Tensor:
        h = x if cin == cout else causal_conv3d(x, self.W[p + ".shortcut.weight"], self.W[p + ".shortcut.bias"], None)
        x = F.silu(rms_norm(x, self.W[p + ".residual.0.gamma"]))
        x = self._cached_conv(p + ".residual.2", x)
        x = F.silu(rms_norm(x, self.W[p + ".residual.3.gamma"]))
        x = self._cached_conv(p + ".residual.6", x)
        return x + h

    def _attn(self, p: str, x: torch.Tensor) -> torch.Tensor:
        b, c, t, h, w = x.shape
        y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        y = rms_norm(y, self.W[p + ".norm.gamma"])
        qkv = F.conv2d(y, self.W[p + ".to_qkv.weight"], self.W[p + ".to_qkv.bias"])
        q, k, v = qkv.reshape(b * t, 1, 3 * c, h * w).permute(0, 1, 3, 2).contiguous().chunk(3, dim=-1)
        o = F.scaled_dot_product_attention(q, k, v)
        o = o.squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
        o = F.conv2d(o, self.W[p + ".proj.weight"], self.W[p + ".proj.bias"])
        return o.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4) + x

    def _upsample(self, kind: str, p: str, x: torch.Tensor) -> torch.Tensor:
        b, c, t, h, w = x.shape
        if kind == "up3d":
            name = p + ".time_conv"
            old = self.cache.get(name)
            if old is None:                               # first chunk: no temporal upsampling at all (vae.py:107-109)
                self.cache[name] = REP
            else:
                cache_x = x[:, :, -CACHE_T:].clone()
                if cache_x.shape[2] < 2:
                    head = torch.zeros_like(cache_x) if isinstance(old, str) else old[:, :, -1:]
                    cache_x = torch.cat([head, cache_x], dim=2)
                y = causal_conv3d(x, self.W[name + ".weight"], self.W[name + ".bias"],
                                  None if isinstance(old, str) else old)
                self.cache[name] = cache_x
                y = y.reshape(b, 2, c, t, h, w)           # channel halves -> even / odd output frames (vae.py:130-132)
                x = torch.stack((y[:, 0], y[:, 1]), 3).reshape(b, c, 2 * t, h, w)
                t = 2 * t
        y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        y = F.interpolate(y.float(), scale_factor=(2.0, 2.0), mode="nearest-exact").to(x.dtype)      # vae.py:58-64
        y = F.conv2d(y, self.W[p + ".resample.1.weight"], self.W[p + ".resample.1.bias"], padding=1)
        return y.reshape(b, t, c // 2, 2 * h, 2 * w).permute(0, 2, 1, 3, 4)

    def decoder_frames(self, x: torch.Tensor) -> torch.Tensor:
        """`Decoder3d.forward` on a chunk `[1, z, t, h, w]` (the reference always passes t = 1)."""
        x = self._cached_conv("decoder.conv1", x)
        for item in self.plan:
            if item[0] == "res":
                x = self._res(item[1], x, item[2], item[3])
            elif item[0] == "attn":
                x = self._attn(item[1], x)
            else:
                x = self._upsample(item[0], item[1], x)
        x = F.silu(rms_norm(x, self.W["decoder.head.0.gamma"]))
        return self._cached_conv("decoder.head.2", x)

    # ---- entry points ------------------------------------------------------------------------------------------
    def cached_decode(self, z: torch.Tensor, frames_per_call: int = 1) -> torch.Tensor:
        """vae.py:573-594.  `z` `[1, z_dim, T, h, w]` already in model dtype.  `frames_per_call` > 1 is NOT what the
        reference does (it always feeds one latent frame); it exists to show that feeding several frames at once is the
        same computation once the first chunk has passed (the MI355X path batches a block's frames)."""
        scale0 = self.mean.to(z.dtype).view(1, -1, 1, 1, 1)
        scale1 = (1.0 / self.std.to(z.dtype)).view(1, -1, 1, 1, 1)      # wrapper.py:117-118: scale computed in latent dtype
        z = z / scale1 + scale0
        x = causal_conv3d(z, self.W["conv2.weight"], self.W["conv2.bias"], None)
        outs, i = [], 0
        while i < x.shape[2]:
            n = 1 if not self.cache else frames_per_call         # the first chunk is always a single frame ('Rep' rule)
            outs.append(self.decoder_frames(x[:, :, i:i + n]))
            i += n
        return torch.cat(outs, dim=2)

    def decode_to_pixel(self, latent: torch.Tensor, use_cache: bool = False, chunk_size: int = 2) -> torch.Tensor:
        """wrapper.py:103-168: latent `[B, T, C, H, W]` -> pixels `[B, T_out, 3, 8H, 8W]` float32 clamped to [-1, 1]."""
        zs = latent.permute(0, 2, 1, 3, 4)
        if use_cache:
            assert latent.shape[0] == 1, "Batch size must be 1 when using cache"
        out = []
        for u in zs:
            if not use_cache:
                self.clear_cache()
                dec = self.cached_decode(u.unsqueeze(0)).float().clamp_(-1, 1).squeeze(0)
                self.clear_cache()
            else:
                self.clear_cache()
                parts = [self.cached_decode(u[:, s:s + chunk_size].unsqueeze(0)).float().clamp_(-1, 1).squeeze(0)
                         for s in range(0, u.shape[1], chunk_size)]
                dec = torch.cat(parts, dim=1)
                self.clear_cache()
            out.append(dec)
        return torch.stack(out, dim=0).permute(0, 2, 1, 3, 4)


# ==================================================================================================================
# Encoder (vae.py:264-377, 512-541)
# ==================================================================================================================
def encoder_plan(cfg: VaeConfig) -> List[tuple]:
    """Execution order of `Encoder3d` (vae.py:283-316): ('res', prefix, cin, cout) | ('down2d' | 'down3d', prefix, c) | ('attn', prefix, c)."""
    dims = [cfg.dim * u for u in [1] + list(cfg.dim_mult)]
    plan, n = [], 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(cfg.num_res_blocks):
            plan.append(("res", f"encoder.downsamples.{n}", cin, cout))
            cin = cout
            n += 1
        if i != len(cfg.dim_mult) - 1:
            plan.append(("down3d" if cfg.temperal_downsample[i] else "down2d", f"encoder.downsamples.{n}", cout))
            n += 1
    top = dims[-1]
    plan += [("res", "encoder.middle.0", top, top), ("attn", "encoder.middle.1", top), ("res", "encoder.middle.2", top, top)]
    return plan


def encoder_param_shapes(cfg: VaeConfig) -> Dict[str, Tuple[int, ...]]:
    z2 = 2 * cfg.z_dim
    s: Dict[str, Tuple[int, ...]] = {"conv1.weight": (z2, z2, 1, 1, 1), "conv1.bias": (z2,),
                                     "encoder.conv1.weight": (cfg.dim, 3, 3, 3, 3), "encoder.conv1.bias": (cfg.dim,)}
    last = cfg.dim
    for item in encoder_plan(cfg):
        kind, p = item[0], item[1]
        if kind == "res":
            cin, cout = item[2], item[3]
            s[f"{p}.residual.0.gamma"] = (cin, 1, 1, 1)
            s[f"{p}.residual.2.weight"] = (cout, cin, 3, 3, 3)
            s[f"{p}.residual.2.bias"] = (cout,)
            s[f"{p}.residual.3.gamma"] = (cout, 1, 1, 1)
            s[f"{p}.residual.6.weight"] = (cout, cout, 3, 3, 3)
            s[f"{p}.residual.6.bias"] = (cout,)
            if cin != cout:
                s[f"{p}.shortcut.weight"] = (cout, cin, 1, 1, 1)
                s[f"{p}.shortcut.bias"] = (cout,)
            last = cout
        elif kind == "attn":
            c = item[2]
            s[f"{p}.norm.gamma"] = (c, 1, 1)
            s[f"{p}.to_qkv.weight"] = (3 * c, c, 1, 1)
            s[f"{p}.to_qkv.bias"] = (3 * c,)
            s[f"{p}.proj.weight"] = (c, c, 1, 1)
            s[f"{p}.proj.bias"] = (c,)
        else:
            c = item[2]
            s[f"{p}.resample.1.weight"] = (c, c, 3, 3)
            s[f"{p}.resample.1.bias"] = (c,)
            if kind == "down3d":
                s[f"{p}.time_conv.weight"] = (c, c, 3, 1, 1)
                s[f"{p}.time_conv.bias"] = (c,)
    s["encoder.head.0.gamma"] = (last, 1, 1, 1)
    s["encoder.head.2.weight"] = (z2, last, 3, 3, 3)
    s["encoder.head.2.bias"] = (z2,)
    return s


def make_encoder_params(cfg: VaeConfig, seed: int) -> Dict[str, torch.Tensor]:
    """Seeded synthetic encoder weights (bf16), same recipe as `make_decoder_params`."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape in encoder_param_shapes(cfg).items():
        if name.endswith("gamma"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5
        W[name] = t.to(BF)
    return W


class VaeEncoderOracle(VaeDecoderOracle):
    """`Encoder3d` + `WanVAE_.encode` with the streaming cache; reuses the residual / attention / cached-conv restatements."""

    def __init__(self, cfg: VaeConfig, W: Dict[str, torch.Tensor], dtype=BF):
        self.cfg, self.dtype = cfg, dtype
        self.W = {k: v.to(dtype) for k, v in W.items()}
        self.plan = encoder_plan(cfg)
        self.mean = torch.tensor(VAE_MEAN[:cfg.z_dim], dtype=torch.float32)
        self.std = torch.tensor(VAE_STD[:cfg.z_dim], dtype=torch.float32)
        self.clear_cache()

    def _downsample(self, kind: str, p: str, x: torch.Tensor) -> torch.Tensor:
        b, c, t, h, w = x.shape
        y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        y = F.conv2d(F.pad(y, (0, 1, 0, 1)), self.W[p + ".resample.1.weight"], self.W[p + ".resample.1.bias"], stride=2)   # vae.py:91-94
        x = y.reshape(b, t, c, y.shape[-2], y.shape[-1]).permute(0, 2, 1, 3, 4)
        if kind == "down3d":                                                      # vae.py:143-156
            name = p + ".time_conv"
            old = self.cache.get(name)
            if old is None:
                self.cache[name] = x.clone()
            else:
                cache_x = x[:, :, -1:].clone()
                x = F.conv3d(torch.cat([old[:, :, -1:], x], 2), self.W[name + ".weight"], self.W[name + ".bias"], stride=(2, 1, 1))
                self.cache[name] = cache_x
        return x

    def encoder_frames(self, x: torch.Tensor) -> torch.Tensor:
        """`Encoder3d.forward` on a chunk `[1, 3, t, H, W]`."""
        x = self._cached_conv("encoder.conv1", x)
        for item in self.plan:
            if item[0] == "res":
                x = self._res(item[1], x, item[2], item[3])
            elif item[0] == "attn":
                x = self._attn(item[1], x)
            else:
                x = self._downsample(item[0], item[1], x)
        x = F.silu(rms_norm(x, self.W["encoder.head.0.gamma"]))
        return self._cached_conv("encoder.head.2", x)

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """vae.py:512-541: `[1, 3, T, H, W]` (T = 1 + 4k) -> normalised mu `[1, z, 1 + k, H/8, W/8]`."""
        self.clear_cache()
        T = x.shape[2]
        outs = [self.encoder_frames(x[:, :, :1])]
        for i in range(1, 1 + (T - 1) // 4):
            outs.append(self.encoder_frames(x[:, :, 1 + 4 * (i - 1):1 + 4 * i]))
        out = torch.cat(outs, 2)
        mu, _ = causal_conv3d(out, self.W["conv1.weight"], self.W["conv1.bias"], None).chunk(2, dim=1)
        scale0 = self.mean.to(x.dtype).view(1, -1, 1, 1, 1)
        scale1 = (1.0 / self.std.to(x.dtype)).view(1, -1, 1, 1, 1)
        mu = (mu - scale0) * scale1
        self.clear_cache()
        return mu

    def encode_to_latent(self, pixel: torch.Tensor) -> torch.Tensor:
        """wrapper.py:88-101: pixel `[B, 3, T, H, W]` -> latent `[B, T', z, H/8, W/8]` float32."""
        out = [self.encode(u.unsqueeze(0)).float().squeeze(0) for u in pixel]
        return torch.stack(out, dim=0).permute(0, 2, 1, 3, 4)
