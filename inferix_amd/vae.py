"""Wan2.1 VAE decoder on MI355X — the per-block decode of the streaming pipelines.

Drop-in for the decode half of `WanVAEWrapper` (inferix/models/self_forcing/wrapper.py:62-168) and the decoder of `WanVAE_`
(inferix/models/wan_base/vae.py:380-466, 543-611): same state-dict keys, same `decode_to_pixel(latent, use_cache,
chunk_size)` / `model.clear_cache()` surface, same streaming semantics (two-frame causal feature cache per conv, the 'Rep'
first-chunk rule of the temporal upsamplers), bf16 like the reference's `model.to(dtype=torch.bfloat16)`.

MI355X-first differences in HOW (not in what is computed):
  * activations are channels-last frames `[t, h, w, c]`; every conv is `ifx_conv3d_cl` (LDS-tiled direct convolution on MFMA).
  * the feature cache is not a list of cloned tensors that get concatenated in front of the next chunk (vae.py:207-216):
    each causal conv owns a frame ring `[2 + T, h, w, c]`; its producer (`ifx_rmsnorm_cl`, or the previous conv's epilogue)
    writes new frames straight into free slots, the conv reads [two history slots | new slots] through a slot table, and
    "updating the cache" is relabelling the last two slots.  Nothing is copied, concatenated or padded.
  * nearest-2x upsample + conv2d is one launch (the upsampled frame is never materialised); the temporal upsampler's channel
    halves are written directly to the even / odd output frames.
  * the reference feeds one latent frame per decoder call (vae.py:581-592, sized for 24 GB GPUs); with 288 GB of HBM the
    frames of a block go through together after the first chunk — the same function of the stream (causal convs commute
    with chunking; pinned in oracle/gen_golden_vae.py), 3x fewer launches and 3x larger grids.
  * the single-head 384-wide spatial attention of the middle block is two `ifx_gemm_bf16` launches around `ifx_softmax_rows`.

`HipWanVAEEncoder` (image-to-video start frames, once per request) runs on the same kernels.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _hip
from . import hip_ops as ops

BF16 = torch.bfloat16

VAE_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
            0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]     # wrapper.py:65-72
VAE_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
           3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


def _repack_conv(w: torch.Tensor, cin_pad: int = 0) -> torch.Tensor:
    """torch conv weight `[cout, cin, (kt,) kh, kw]` -> `[taps, cin/32, cout, 32]` bf16 (ifx_conv3d_cl layout: tap-major, then
    32-channel chunks, so that the 16-row x 64-byte pieces the kernel DMAs are contiguous)."""
    cout, cin = w.shape[:2]
    t = w.reshape(cout, cin, -1).permute(2, 0, 1)                       # [taps, cout, cin]
    cin_p = max(cin_pad, (cin + 31) // 32 * 32)
    if cin_p > cin:
        t = torch.nn.functional.pad(t, (0, cin_p - cin))
    taps = t.shape[0]
    return t.reshape(taps, cout, cin_p // 32, 32).permute(0, 2, 1, 3).contiguous().to(BF16)


def synthetic_decoder_state_dict(dim: int = 96, z_dim: int = 16, dim_mult: Sequence[int] = (1, 2, 4, 4),
                                 num_res_blocks: int = 2, temperal_downsample: Sequence[bool] = (False, True, True),
                                 seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random decoder weights with the reference's state-dict keys and shapes (vae.py:380-413), variance-preserving so that
    activations stay O(1) through the 14 residual blocks — for benchmarks and smoke runs (there is no checkpoint here)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, cout, cin, *k):
        fan = cin
        for d in k:
            fan *= d
        sd[name + ".weight"] = (torch.randn(cout, cin, *k, generator=g) * fan ** -0.5).to(BF16)
        sd[name + ".bias"] = (0.02 * torch.randn(cout, generator=g)).to(BF16)

    def gamma(name, c, *ones):
        sd[name] = (1.0 + 0.1 * torch.randn(c, *ones, generator=g)).to(BF16)

    dims = [dim * u for u in [dim_mult[-1]] + list(dim_mult[::-1])]
    up = tuple(temperal_downsample[::-1])
    conv("conv2", z_dim, z_dim, 1, 1, 1)
    conv("decoder.conv1", dims[0], z_dim, 3, 3, 3)

    def res(p, cin, cout):
        gamma(p + ".residual.0.gamma", cin, 1, 1, 1)
        conv(p + ".residual.2", cout, cin, 3, 3, 3)
        gamma(p + ".residual.3.gamma", cout, 1, 1, 1)
        conv(p + ".residual.6", cout, cout, 3, 3, 3)
        if cin != cout:
            conv(p + ".shortcut", cout, cin, 1, 1, 1)

    res("decoder.middle.0", dims[0], dims[0])
    gamma("decoder.middle.1.norm.gamma", dims[0], 1, 1)
    conv("decoder.middle.1.to_qkv", 3 * dims[0], dims[0], 1, 1)
    conv("decoder.middle.1.proj", dims[0], dims[0], 1, 1)
    res("decoder.middle.2", dims[0], dims[0])
    n, last = 0, dims[0]
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(num_res_blocks + 1):
            res(f"decoder.upsamples.{n}", cin, cout)
            cin = cout
            n += 1
        last = cout
        if i != len(dim_mult) - 1:
            conv(f"decoder.upsamples.{n}.resample.1", cout // 2, cout, 3, 3)
            if up[i]:
                conv(f"decoder.upsamples.{n}.time_conv", 2 * cout, cout, 3, 1, 1)
            n += 1
    gamma("decoder.head.0.gamma", last, 1, 1, 1)
    conv("decoder.head.2", 3, last, 3, 3, 3)
    return sd


def synthetic_encoder_state_dict(dim: int = 96, z_dim: int = 16, dim_mult: Sequence[int] = (1, 2, 4, 4), num_res_blocks: int = 2,
                                 temperal_downsample: Sequence[bool] = (False, True, True), seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random encoder weights with the reference's keys and shapes (vae.py:264-316) — the counterpart of
    `synthetic_decoder_state_dict` for smoke runs of `encode_to_latent`."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, cout, cin, *k):
        fan = cin
        for d in k:
            fan *= d
        sd[name + ".weight"] = (torch.randn(cout, cin, *k, generator=g) * fan ** -0.5).to(BF16)
        sd[name + ".bias"] = (0.02 * torch.randn(cout, generator=g)).to(BF16)

    def gamma(name, c, *ones):
        sd[name] = (1.0 + 0.1 * torch.randn(c, *ones, generator=g)).to(BF16)

    def res(p, cin, cout):
        gamma(p + ".residual.0.gamma", cin, 1, 1, 1)
        conv(p + ".residual.2", cout, cin, 3, 3, 3)
        gamma(p + ".residual.3.gamma", cout, 1, 1, 1)
        conv(p + ".residual.6", cout, cout, 3, 3, 3)
        if cin != cout:
            conv(p + ".shortcut", cout, cin, 1, 1, 1)

    dims = [dim * u for u in [1] + list(dim_mult)]
    conv("conv1", 2 * z_dim, 2 * z_dim, 1, 1, 1)
    conv("encoder.conv1", dims[0], 3, 3, 3, 3)
    n = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(num_res_blocks):
            res(f"encoder.downsamples.{n}", cin, cout)
            cin = cout
            n += 1
        if i != len(dim_mult) - 1:
            conv(f"encoder.downsamples.{n}.resample.1", cout, cout, 3, 3)
            if temperal_downsample[i]:
                conv(f"encoder.downsamples.{n}.time_conv", cout, cout, 3, 1, 1)
            n += 1
    top = dims[-1]
    res("encoder.middle.0", top, top)
    gamma("encoder.middle.1.norm.gamma", top, 1, 1)
    conv("encoder.middle.1.to_qkv", 3 * top, top, 1, 1)
    conv("encoder.middle.1.proj", top, top, 1, 1)
    res("encoder.middle.2", top, top)
    gamma("encoder.head.0.gamma", top, 1, 1, 1)
    conv("encoder.head.2", 2 * z_dim, top, 3, 3, 3)
    return sd


class FrameRing:
    """Input frames of one causal conv: `2 + cap` physical slots; `hist` = slots of the last two frames of the stream
    (-1 = a zero frame in front of the stream)."""

    def __init__(self, cap: int, h: int, w: int, c: int, device, planar: bool = False):
        # planar: frames as 32-channel planes `[c/32, h, w, 32]` (ifx_conv3d_desc.in_planar) — the layout the 3x3x3 convs read their
        # halo patches from at the full LDS-DMA rate (contiguous 64-byte pixels per channel chunk)
        self.planar = planar
        self.buf = torch.empty((cap + 2, c // 32, h, w, 32) if planar else (cap + 2, h, w, c), dtype=BF16, device=device)
        self.cap = cap
        self.hist = [-1, -1]

    def reset(self) -> None:
        self.hist = [-1, -1]

    def new_slots(self, t: int) -> List[int]:
        assert t <= self.cap, f"{t} frames per call, ring built for {self.cap}"
        free = [s for s in range(self.cap + 2) if s not in self.hist]
        return free[:t]

    def commit(self, new: Sequence[int]) -> List[int]:
        """-> in_slots for the conv over `new`; afterwards the history is the last two frames of the stream."""
        ins = self.hist + list(new)
        self.hist = ins[-2:]
        return ins


class _VaeLayers:
    """What encoder and decoder share: scratch / frame-ring management and the residual, attention and causal-conv layers."""

    def _init_layers(self, device, max_frames_per_call: int) -> None:
        _hip.load()                                         # fail loudly without the HIP library
        self.device = torch.device(device)
        self.max_frames = max_frames_per_call
        self._rings: Dict[Tuple[str, int, int], FrameRing] = {}
        self._scratch: Dict[Tuple, torch.Tensor] = {}
        self._started = False
        self._rep: Dict[str, bool] = {}
        self.mean = torch.tensor(VAE_MEAN[:self.z_dim], dtype=torch.float32, device=self.device)
        self.std = torch.tensor(VAE_STD[:self.z_dim], dtype=torch.float32, device=self.device)

    def _load_res(self, W: Dict[str, torch.Tensor], g, p: str, cin: int, cout: int) -> None:
        for n in ("residual.2", "residual.6"):
            W[f"{p}.{n}.w"] = _repack_conv(g(f"{p}.{n}.weight"))
            W[f"{p}.{n}.b"] = g(f"{p}.{n}.bias").to(BF16).contiguous()
        W[f"{p}.residual.0.gamma"] = g(f"{p}.residual.0.gamma").reshape(-1).to(BF16).contiguous()
        W[f"{p}.residual.3.gamma"] = g(f"{p}.residual.3.gamma").reshape(-1).to(BF16).contiguous()
        if cin != cout:
            W[f"{p}.shortcut.w"] = _repack_conv(g(f"{p}.shortcut.weight"))
            W[f"{p}.shortcut.b"] = g(f"{p}.shortcut.bias").to(BF16).contiguous()

    def _load_attn(self, W: Dict[str, torch.Tensor], g, p: str, c: int) -> None:
        W[f"{p}.norm.gamma"] = g(f"{p}.norm.gamma").reshape(-1).to(BF16).contiguous()
        W[f"{p}.to_qkv.w"] = g(f"{p}.to_qkv.weight").reshape(3 * c, c).to(BF16).contiguous()
        W[f"{p}.to_qkv.b"] = g(f"{p}.to_qkv.bias").to(BF16).contiguous()
        W[f"{p}.proj.w"] = g(f"{p}.proj.weight").reshape(c, c).to(BF16).contiguous()
        W[f"{p}.proj.b"] = g(f"{p}.proj.bias").to(BF16).contiguous()

    # ---- buffers ----------------------------------------------------------------------------------------------------
    def _ring(self, name: str, cap: int, h: int, w: int, c: int, planar: bool = False) -> FrameRing:
        key = (name, h, w)
        r = self._rings.get(key)
        if r is None or r.cap < cap:
            hist = r.hist if r is not None else None
            assert hist is None or hist == [-1, -1], "frame ring would have to grow mid-stream: raise max_frames_per_call"
            r = self._rings[key] = FrameRing(cap, h, w, c, self.device, planar)
        return r

    def _tmp(self, tag: str, *shape) -> torch.Tensor:
        key = (tag,) + tuple(shape)
        t = self._scratch.get(key)
        if t is None:
            t = self._scratch[key] = torch.empty(*shape, dtype=BF16, device=self.device)
        return t

    def clear_cache(self) -> None:                                    # vae.py:603-611
        for r in self._rings.values():
            r.reset()
        self._rep.clear()
        self._started = False

    # ---- layers -----------------------------------------------------------------------------------------------------
    def _causal_conv(self, name: str, src: torch.Tensor, gamma: Optional[str], out: torch.Tensor,
                     residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[norm + SiLU ->] ring -> 3x3x3 causal conv.  `src` plain `[t, h, w, cin]`; `out` plain `[t, h, w, cout]`."""
        t, h, w, c = src.shape
        ring = self._ring(name, self._cap(h), h, w, c, planar=c % 32 == 0)
        new = ring.new_slots(t)
        if gamma is not None:
            ops.rmsnorm_cl(src, self.W[gamma], ring.buf, new, silu=True)          # (planar ring: 5-D buffer -> IFX_NORM_OUT_PLANAR)
        elif ring.planar:
            ring.buf[new] = src.view(t, h, w, c // 32, 32).permute(0, 3, 1, 2, 4)
        else:
            ring.buf[new] = src
        ins = ring.commit(new)
        return ops.conv3d_cl(ring.buf, ins, self.W[name + ".w"], self.W[name + ".b"], kt=3, ks=3, y=out,
                             out_slots=list(range(t)), residual=residual)

    def _cap(self, h: int) -> int:
        """Frames per call a ring at this resolution must hold: temporal upsampling doubles them twice."""
        return self.max_frames * 4

    def _res(self, p: str, x: torch.Tensor, cin: int, cout: int) -> torch.Tensor:
        """`ResidualBlock.forward` (vae.py:201-219).  The block's output overwrites its shortcut operand in place (every
        output element is produced by the thread that read that element of the residual)."""
        t, h, w, _ = x.shape
        if cin != cout:
            hres = self._tmp("res.short", t, h, w, cout)
            ops.conv3d_cl(x, list(range(t)), self.W[p + ".shortcut.w"], self.W[p + ".shortcut.b"], kt=1, ks=1, y=hres,
                          out_slots=list(range(t)))
        else:
            hres = x
        y1 = self._tmp("res.y1", t, h, w, cout)
        self._causal_conv(p + ".residual.2", x, p + ".residual.0.gamma", y1)
        return self._causal_conv(p + ".residual.6", y1, p + ".residual.3.gamma", hres, residual=hres)

    def _attn(self, p: str, x: torch.Tensor) -> torch.Tensor:
        t, h, w, c = x.shape
        hw = h * w
        kpad = (hw + 63) // 64 * 64
        xn = self._tmp("attn.xn", t, h, w, c)
        ops.rmsnorm_cl(x, self.W[p + ".norm.gamma"], xn, list(range(t)), silu=False)
        qkv = ops.linear(xn.view(t * hw, c), self.W[p + ".to_qkv.w"], self.W[p + ".to_qkv.b"],
                         out=self._tmp("attn.qkv", t * hw, 3 * c)).view(t, hw, 3 * c)
        key = ("attn.s", hw, kpad)
        if key not in self._scratch:                              # pad columns stay zero for the K % 64 rule of the GEMM
            self._scratch[key] = torch.zeros(2, hw, kpad, dtype=BF16, device=self.device)
            self._scratch[("attn.vt", c, kpad)] = torch.zeros(c, kpad, dtype=BF16, device=self.device)
        s_buf, p_buf = self._scratch[key][0], self._scratch[key][1]
        vt = self._scratch[("attn.vt", c, kpad)]
        o = self._tmp("attn.o", t, hw, c)
        out = self._tmp("attn.out", t, h, w, c)
        for f in range(t):
            q = qkv[f, :, :c]
            k = self._tmp("attn.k", hw, c)
            k.copy_(qkv[f, :, c:2 * c])
            vt[:, :hw].copy_(qkv[f, :, 2 * c:].t())
            ops.linear(q, k, None, out=s_buf[:, :hw])
            ops.softmax_rows(s_buf[:, :hw], c ** -0.5, out=p_buf[:, :hw])
            ops.linear(p_buf, vt, None, out=o[f])
        ops.linear(o.view(t * hw, c), self.W[p + ".proj.w"], self.W[p + ".proj.b"], epilogue=_hip.IFX_EPI_RESIDUAL,
                   residual=x.view(t * hw, c), out=out.view(t * hw, c))
        return out


class HipWanVAEDecoder(_VaeLayers):
    """`Decoder3d` + `conv2` of `WanVAE_` with the streaming cache (`cached_decode` / `decode` / `clear_cache`)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], *, dim: int = 96, z_dim: int = 16,
                 dim_mult: Sequence[int] = (1, 2, 4, 4), num_res_blocks: int = 2,
                 temperal_downsample: Sequence[bool] = (False, True, True), device="cuda", max_frames_per_call: int = 3):
        self.dim, self.z_dim, self.dim_mult = dim, z_dim, tuple(dim_mult)
        self._init_layers(device, max_frames_per_call)
        self.num_res_blocks = num_res_blocks
        self.temperal_upsample = tuple(temperal_downsample[::-1])
        self.plan = self._plan()
        self._load(state_dict)

    # ---- structure (vae.py:381-413) ---------------------------------------------------------------------------------
    def _plan(self) -> List[tuple]:
        dims = [self.dim * u for u in [self.dim_mult[-1]] + list(self.dim_mult[::-1])]
        plan = [("res", "decoder.middle.0", dims[0], dims[0]), ("attn", "decoder.middle.1", dims[0]),
                ("res", "decoder.middle.2", dims[0], dims[0])]
        n = 0
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            if i in (1, 2, 3):
                cin = cin // 2
            for _ in range(self.num_res_blocks + 1):
                plan.append(("res", f"decoder.upsamples.{n}", cin, cout))
                cin = cout
                n += 1
            if i != len(self.dim_mult) - 1:
                plan.append(("up3d" if self.temperal_upsample[i] else "up2d", f"decoder.upsamples.{n}", cout))
                n += 1
        self.head_dim = dims[-1]
        self.dims0 = dims[0]
        return plan

    def _load(self, sd: Dict[str, torch.Tensor]) -> None:
        dev = self.device
        g = lambda k: sd[k].to(dev)
        W: Dict[str, torch.Tensor] = {}
        self.zpad = max(32, (self.z_dim + 31) // 32 * 32)
        W["conv2.w"] = _repack_conv(g("conv2.weight"), self.zpad)
        W["conv2.b"] = g("conv2.bias").to(BF16).contiguous()
        W["decoder.conv1.w"] = _repack_conv(g("decoder.conv1.weight"), self.zpad)
        W["decoder.conv1.b"] = g("decoder.conv1.bias").to(BF16).contiguous()
        for item in self.plan:
            kind, p = item[0], item[1]
            if kind == "res":
                self._load_res(W, g, p, item[2], item[3])
            elif kind == "attn":
                self._load_attn(W, g, p, item[2])
            else:
                c = item[2]
                W[f"{p}.resample.w"] = _repack_conv(g(f"{p}.resample.1.weight"))
                W[f"{p}.resample.b"] = g(f"{p}.resample.1.bias").to(BF16).contiguous()
                if kind == "up3d":
                    tw = g(f"{p}.time_conv.weight")                  # [2c, c, 3, 1, 1]: halves -> even / odd frames
                    tb = g(f"{p}.time_conv.bias").to(BF16)
                    for half in (0, 1):
                        W[f"{p}.time_conv.w{half}"] = _repack_conv(tw[half * c:(half + 1) * c])
                        W[f"{p}.time_conv.b{half}"] = tb[half * c:(half + 1) * c].contiguous()
        W["decoder.head.0.gamma"] = g("decoder.head.0.gamma").reshape(-1).to(BF16).contiguous()
        W["decoder.head.2.w"] = _repack_conv(g("decoder.head.2.weight"))
        W["decoder.head.2.b"] = g("decoder.head.2.bias").to(BF16).contiguous()
        self.W = W

    def _upsample(self, kind: str, p: str, x: torch.Tensor) -> torch.Tensor:
        t, h, w, c = x.shape
        if kind == "up3d":
            if not self._rep.get(p):                            # first chunk: no temporal upsampling (vae.py:107-109)
                self._rep[p] = True
            else:
                ring = self._ring(p + ".time_conv", self._cap(h), h, w, c)
                new = ring.new_slots(t)
                ring.buf[new] = x
                ins = ring.commit(new)
                y = self._tmp(p + ".tc", 2 * t, h, w, c)
                for half in (0, 1):
                    ops.conv3d_cl(ring.buf, ins, self.W[f"{p}.time_conv.w{half}"], self.W[f"{p}.time_conv.b{half}"], kt=3, ks=1,
                                  y=y, out_slots=[2 * i + half for i in range(t)])
                x, t = y, 2 * t
        out = self._tmp(p + ".up", t, 2 * h, 2 * w, c // 2)
        return ops.conv3d_cl(x, list(range(t)), self.W[p + ".resample.w"], self.W[p + ".resample.b"], kt=1, ks=3, y=out,
                             out_slots=list(range(t)), upsample=True)

    def _decoder_frames(self, x: torch.Tensor) -> torch.Tensor:
        """`Decoder3d.forward` (vae.py:415-466) on `t` latent frames `[t, h, w, zpad]` (already through `conv2`)."""
        t, h, w, _ = x.shape
        y = self._tmp("conv1.out", t, h, w, self.dims0)
        x = self._causal_conv("decoder.conv1", x, None, y)
        for item in self.plan:
            if item[0] == "res":
                x = self._res(item[1], x, item[2], item[3])
            elif item[0] == "attn":
                x = self._attn(item[1], x)
            else:
                x = self._upsample(item[0], item[1], x)
        t, h, w, _ = x.shape
        out = self._tmp("head.out", t, h, w, 3)
        return self._causal_conv("decoder.head.2", x, "decoder.head.0.gamma", out)

    # ---- entry points -----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def cached_decode(self, z: torch.Tensor) -> torch.Tensor:
        """vae.py:573-594: `z` `[1, z_dim, T, h, w]` (latent dtype) -> `[1, 3, T_out, 8h, 8w]` bf16, continuing the stream."""
        assert z.dim() == 5 and z.shape[0] == 1 and z.shape[1] == self.z_dim, tuple(z.shape)
        z = z.to(self.device)
        scale0 = self.mean.to(z.dtype).view(1, -1, 1, 1, 1)
        scale1 = (1.0 / self.std.to(z.dtype)).view(1, -1, 1, 1, 1)
        z = (z / scale1 + scale0).to(BF16)                                   # wrapper.py:117-118, vae.py:575-579
        T, h, w = z.shape[2:]
        zc = torch.zeros(T, h, w, self.zpad, dtype=BF16, device=self.device)
        zc[..., :self.z_dim] = z[0].permute(1, 2, 3, 0)
        x = torch.empty(T, h, w, self.zpad, dtype=BF16, device=self.device)
        x[..., self.z_dim:] = 0
        x16 = self._tmp("conv2.out", T, h, w, self.z_dim)
        ops.conv3d_cl(zc, list(range(T)), self.W["conv2.w"], self.W["conv2.b"], kt=1, ks=1, y=x16, out_slots=list(range(T)))
        x[..., :self.z_dim] = x16
        outs, i = [], 0
        while i < T:
            n = 1 if not self._started else min(self.max_frames, T - i)      # first chunk alone ('Rep' rule)
            self._started = True
            outs.append(self._decoder_frames(x[i:i + n]).clone())
            i += n
        y = torch.cat(outs, dim=0)                                           # [T_out, H, W, 3]
        return y.permute(3, 0, 1, 2).unsqueeze(0)

    def decode(self, z: torch.Tensor) -> torch.Tensor:                       # vae.py:543-566
        self.clear_cache()
        out = self.cached_decode(z)
        self.clear_cache()
        return out


class HipWanVAEEncoder(_VaeLayers):
    """`Encoder3d` + `conv1` of `WanVAE_` (vae.py:264-377, 512-541): the start frames of image-to-video, once per request.
    Not a throughput path, so it reuses the decoder's kernels as they are: the stride-2 `Conv2d` behind `ZeroPad2d((0,1,0,1))`
    (vae.py:91-94) is the pad-1 stride-1 convolution sampled at the odd positions (`y[1::2, 1::2]`, computed at full
    resolution and sliced — 4x the arithmetic of a strided kernel, on a handful of frames), and the stride-2 temporal conv of
    `downsample3d` (vae.py:98-99, 143-156) is one launch per output frame with its three input frames named in the slot table."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], *, dim: int = 96, z_dim: int = 16,
                 dim_mult: Sequence[int] = (1, 2, 4, 4), num_res_blocks: int = 2,
                 temperal_downsample: Sequence[bool] = (False, True, True), device="cuda"):
        self.dim, self.z_dim, self.dim_mult = dim, z_dim, tuple(dim_mult)
        self.num_res_blocks = num_res_blocks
        self.temperal_downsample = tuple(temperal_downsample)
        self._init_layers(device, 1)                       # rings hold the 4-frame chunks of `encode` (cap = 4 x max_frames)
        self.plan = self._plan()
        self._load(state_dict)
        self._last: Dict[str, torch.Tensor] = {}           # last frame in front of each temporal downsampler

    def _plan(self) -> List[tuple]:                         # vae.py:283-316
        dims = [self.dim * u for u in [1] + list(self.dim_mult)]
        plan, n = [], 0
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            for _ in range(self.num_res_blocks):
                plan.append(("res", f"encoder.downsamples.{n}", cin, cout))
                cin = cout
                n += 1
            if i != len(self.dim_mult) - 1:
                plan.append(("down3d" if self.temperal_downsample[i] else "down2d", f"encoder.downsamples.{n}", cout))
                n += 1
        top = dims[-1]
        self.top = top
        return plan + [("res", "encoder.middle.0", top, top), ("attn", "encoder.middle.1", top), ("res", "encoder.middle.2", top, top)]

    def _load(self, sd: Dict[str, torch.Tensor]) -> None:
        dev = self.device
        g = lambda k: sd[k].to(dev)
        W: Dict[str, torch.Tensor] = {}
        W["conv1.w"] = _repack_conv(g("conv1.weight"))
        W["conv1.b"] = g("conv1.bias").to(BF16).contiguous()
        W["encoder.conv1.w"] = _repack_conv(g("encoder.conv1.weight"), 32)       # RGB padded to one 32-channel chunk
        W["encoder.conv1.b"] = g("encoder.conv1.bias").to(BF16).contiguous()
        for item in self.plan:
            kind, p = item[0], item[1]
            if kind == "res":
                self._load_res(W, g, p, item[2], item[3])
            elif kind == "attn":
                self._load_attn(W, g, p, item[2])
            else:
                W[f"{p}.resample.w"] = _repack_conv(g(f"{p}.resample.1.weight"))
                W[f"{p}.resample.b"] = g(f"{p}.resample.1.bias").to(BF16).contiguous()
                if kind == "down3d":
                    W[f"{p}.time_conv.w"] = _repack_conv(g(f"{p}.time_conv.weight"))
                    W[f"{p}.time_conv.b"] = g(f"{p}.time_conv.bias").to(BF16).contiguous()
        W["encoder.head.0.gamma"] = g("encoder.head.0.gamma").reshape(-1).to(BF16).contiguous()
        W["encoder.head.2.w"] = _repack_conv(g("encoder.head.2.weight"))
        W["encoder.head.2.b"] = g("encoder.head.2.bias").to(BF16).contiguous()
        self.W = W

    def clear_cache(self) -> None:
        super().clear_cache()
        self._last.clear()

    def _downsample(self, kind: str, p: str, x: torch.Tensor) -> torch.Tensor:
        t, h, w, c = x.shape
        full = self._tmp(p + ".full", t, h, w, c)
        ops.conv3d_cl(x, list(range(t)), self.W[p + ".resample.w"], self.W[p + ".resample.b"], kt=1, ks=3, y=full,
                      out_slots=list(range(t)))
        x = full[:, 1::2, 1::2].contiguous()
        if kind == "down3d":
            last = self._last.get(p)
            if last is None:                                     # first chunk: no temporal conv (vae.py:146-148)
                self._last[p] = x[-1:].clone()
            else:
                seq = torch.cat([last, x], 0)                    # [1 + t, h/2, w/2, c]
                n_out = (t - 2) // 2 + 1
                y = self._tmp(p + ".tc", n_out, x.shape[1], x.shape[2], c)
                for j in range(n_out):
                    ops.conv3d_cl(seq, [2 * j, 2 * j + 1, 2 * j + 2], self.W[p + ".time_conv.w"], self.W[p + ".time_conv.b"],
                                  kt=3, ks=1, y=y, out_slots=[j])
                self._last[p] = x[-1:].clone()
                x = y
        return x

    def _encoder_frames(self, x: torch.Tensor) -> torch.Tensor:
        """`Encoder3d.forward` (vae.py:318-377) on a chunk `[t, H, W, 32]` (RGB in the first three channels)."""
        t, h, w, _ = x.shape
        x = self._causal_conv("encoder.conv1", x, None, self._tmp("conv1.out", t, h, w, self.dim))
        for item in self.plan:
            if item[0] == "res":
                x = self._res(item[1], x, item[2], item[3])
            elif item[0] == "attn":
                x = self._attn(item[1], x)
            else:
                x = self._downsample(item[0], item[1], x)
        t, h, w, _ = x.shape
        out = self._tmp("head.out", t, h, w, 2 * self.z_dim)
        return self._causal_conv("encoder.head.2", x, "encoder.head.0.gamma", out)

    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """vae.py:512-541: `[1, 3, T, H, W]` (T = 1 + 4k) -> normalised mu `[1, z, 1 + k, H/8, W/8]` in the input dtype."""
        assert x.dim() == 5 and x.shape[0] == 1 and x.shape[1] == 3, tuple(x.shape)
        self.clear_cache()
        x = x.to(self.device)
        T, H, Wd = x.shape[2:]
        xc = torch.zeros(T, H, Wd, 32, dtype=BF16, device=self.device)
        xc[..., :3] = x[0].permute(1, 2, 3, 0)
        outs = [self._encoder_frames(xc[:1]).clone()]
        for i in range(1, 1 + (T - 1) // 4):
            outs.append(self._encoder_frames(xc[1 + 4 * (i - 1):1 + 4 * i]).clone())
        out = torch.cat(outs, 0)                                          # [T', h, w, 2z]
        t2 = out.shape[0]
        mu32 = self._tmp("conv1x1.out", t2, out.shape[1], out.shape[2], 2 * self.z_dim)
        ops.conv3d_cl(out, list(range(t2)), self.W["conv1.w"], self.W["conv1.b"], kt=1, ks=1, y=mu32, out_slots=list(range(t2)))
        mu = mu32[..., :self.z_dim].permute(3, 0, 1, 2).unsqueeze(0).to(x.dtype)
        scale0 = self.mean.to(x.dtype).view(1, -1, 1, 1, 1)
        scale1 = (1.0 / self.std.to(x.dtype)).view(1, -1, 1, 1, 1)
        mu = (mu - scale0) * scale1                                       # vae.py:533-538, wrapper.py:91-92
        self.clear_cache()
        return mu


class HipWanVAEWrapper:
    """`WanVAEWrapper` (wrapper.py:62-168): `decode_to_pixel(latent, use_cache, chunk_size)`, `encode_to_latent(pixel)` and
    `.model.clear_cache()`, so that the pipelines' `vae=` seam takes it unchanged.  The encoder is built lazily from the same
    state dict on the first `encode_to_latent` (text-to-video never needs it)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], *, device="cuda", max_frames_per_call: int = 3, **cfg):
        self.model = HipWanVAEDecoder(state_dict, device=device, max_frames_per_call=max_frames_per_call, **cfg)
        self._encoder: Optional[HipWanVAEEncoder] = None
        self._enc_args = (state_dict, device, cfg)

    @torch.no_grad()
    def encode_to_latent(self, pixel: torch.Tensor) -> torch.Tensor:
        """pixel `[B, 3, T, H, W]` in [-1, 1] -> latent `[B, 1 + (T-1)/4, 16, H/8, W/8]` float32 (wrapper.py:88-101)."""
        if self._encoder is None:
            sd, device, cfg = self._enc_args
            if "encoder.conv1.weight" not in sd:
                raise RuntimeError("HipWanVAEWrapper.encode_to_latent: the state dict has no encoder.* weights")
            self._encoder = HipWanVAEEncoder(sd, device=device, **cfg)
        out = [self._encoder.encode(u.unsqueeze(0)).float().squeeze(0) for u in pixel]
        return torch.stack(out, dim=0).permute(0, 2, 1, 3, 4)

    @torch.no_grad()
    def decode_to_pixel(self, latent: torch.Tensor, use_cache: bool = False, chunk_size: int = 2) -> torch.Tensor:
        """latent `[B, T, C, H, W]` -> pixels `[B, T_out, 3, 8H, 8W]` float32 in [-1, 1] (wrapper.py:103-168).  All three
        flows of the reference produce the same pixels (all-at-once `decode` is frame-by-frame internally), so `chunk_size`
        only bounds how many latent frames go through the decoder together."""
        zs = latent.permute(0, 2, 1, 3, 4)
        if use_cache:
            assert latent.shape[0] == 1, "Batch size must be 1 when using cache"
        out = []
        for u in zs:
            self.model.clear_cache()
            if use_cache:
                parts = [self.model.cached_decode(u[:, s:s + chunk_size].unsqueeze(0)) for s in range(0, u.shape[1], chunk_size)]
                dec = torch.cat(parts, dim=2)
            else:
                dec = self.model.cached_decode(u.unsqueeze(0))
            self.model.clear_cache()
            out.append(dec.float().clamp_(-1, 1).squeeze(0))
        return torch.stack(out, dim=0).permute(0, 2, 1, 3, 4)
