"""MAGI's chunk schedule: which chunks a denoise forward carries, at which timesteps, over which key ranges.

Host mirror of the planning half of `SampleTransport` (inferix/pipeline/magi/video_generate.py):
  generate_sequences                         :166-182   per-stage chunk window and timestep window
  init_t / init_intervel                     :185-236   the time grid (sd3 / square / piecewise / identity) and distill intervals
  get_timestep / get_denoise_step_of_each_chunk :324-359
  generate_denoise_status_and_sequences      :554-572
  total_forward_step                         :574-585
  forward_velocity (steps 3-7)               :587-668   slice_point / range_num / denoising_range_num, the extra clean chunk, kv ranges
  integrate                                  :531-552   x += v * (t[i + 1] - t[i]) per chunk

Pure integer / small-tensor work on the host: `ChunkSchedule.plan(step)` is everything `forward_velocity` computes before it calls the
model, `ChunkSchedule.run(model, ...)` is the reference's forward_velocity + integrate_velocity loop for one clip, with or without a
prefix video (`extract_prefix_video_feature` / `try_pad_prefix_video`, :391-454).  tests/test_magi_schedule.py checks plans, timesteps and key ranges against a golden the reference's
own methods produced for every step of the 4.5B distill configuration.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .kv_ranges import chunk_token_nums, generate_kvrange_for_denoising_video


def generate_sequences(chunk_num: int, window_size: int, chunk_offset: int) -> Tuple[List[int], List[int], List[int], List[int]]:
    """Stage i (one per chunk entering or leaving the window) denoises chunks [clip_start[i], clip_end[i]) whose positions inside the
    window are [t_start[i], t_end[i])."""
    stages = range(chunk_offset, chunk_num + window_size - 1)
    clip_start = [max(chunk_offset, i - window_size + 1) for i in stages]
    clip_end = [min(chunk_num, i + 1) for i in stages]
    t_start = [max(0, i - chunk_num + 1) for i in stages]
    t_end = [min(window_size, i - chunk_offset + 1) for i in stages]
    return clip_start, clip_end, t_start, t_end


def init_t(t_schedule_config: Optional[Dict], num_steps: int, device="cpu", shortcut_mode: str = "") -> torch.Tensor:
    """The time grid t[0 .. num_steps] (0: noise, 1: clean)."""
    cfg = t_schedule_config or {}
    if num_steps == 12:
        base = torch.linspace(0, 1, 5, device=device) / 4
        accu = torch.linspace(0, 1, 5, device=device)
        base = base[:3] if shortcut_mode == "16,16,8" else torch.cat([base[:1], base[2:4]], dim=0)
        t = torch.cat([base + a for a in accu], dim=0)[: num_steps + 1]
    else:
        t = torch.linspace(0, 1, num_steps + 1, device=device)
    func = cfg.get("tSchedulerFunc", "sd3")
    if func == "sd3":
        shift = cfg.get("shift", 3.0)
        assert shift >= 1.0, "shift should >=1"
        inv = 1.0 / shift
        t = t ** 2
        t = inv * t / (1 + (inv - 1) * t)
    elif func == "square":
        t = t ** 2
    elif func == "piecewise":
        low = t < 0.875
        t = torch.where(low, t * (0.5 / 0.875), 0.5 + (t - 0.875) * (0.5 / (1 - 0.875)))
    return t


def init_interval(num_steps: int, device="cpu", shortcut_mode: str = "") -> torch.Tensor:
    if num_steps % 3 == 0:
        pat = [1, 1, 2] if shortcut_mode == "16,16,8" else [2, 1, 1]
        return torch.tensor(pat * (num_steps // 3), device=device)
    return torch.ones(num_steps, device=device)


@dataclass
class ForwardPlan:
    """What one `forward_velocity` call hands to the model."""
    step: int
    denoise_step_per_stage: int
    denoise_stage: int
    denoise_idx: int
    chunk_start: int
    chunk_end: int
    t_start: int
    t_end: int
    fwd_extra_1st_chunk: bool
    slice_point: int
    range_num: int
    denoising_range_num: int
    t_index: List[int]                       # rows of the time grid, one per denoising chunk (without the clean chunk's entry)
    denoise_step_of_each_chunk: List[int]    # with num_steps in front for the clean chunk


class ChunkSchedule:
    """The schedule of one clip: `chunk_num` chunks of `chunk_width` latent frames, `window_size` of them in flight."""

    def __init__(self, num_steps: int, window_size: int, chunk_num: int, chunk_width: int, chunk_offset: int = 0):
        assert num_steps % window_size == 0
        self.num_steps, self.window_size, self.chunk_num, self.chunk_width = num_steps, window_size, chunk_num, chunk_width
        self.chunk_offset = chunk_offset
        self.seq = generate_sequences(chunk_num, window_size, chunk_offset)

    def total_forward_step(self) -> int:
        return (self.num_steps // self.window_size) * (self.chunk_num + self.window_size - 1 - self.chunk_offset)

    def plan(self, step: int) -> ForwardPlan:
        per = self.num_steps // self.window_size
        stage, idx = step // per, step % per
        cs, ce, ts, te = (s[stage] for s in self.seq)
        extra = cs > self.chunk_offset and idx == 0
        t_index = [i * per + idx for i in range(ts, te)][::-1]
        steps = ([self.num_steps] if extra else []) + t_index
        return ForwardPlan(step=step, denoise_step_per_stage=per, denoise_stage=stage, denoise_idx=idx, chunk_start=cs, chunk_end=ce,
                           t_start=ts, t_end=te, fwd_extra_1st_chunk=extra, slice_point=cs - 1 if extra else cs, range_num=ce,
                           denoising_range_num=ce - cs + (1 if extra else 0), t_index=t_index, denoise_step_of_each_chunk=steps)

    def timestep(self, t_total: torch.Tensor, plan: ForwardPlan, clean_t: float, advance: int = 0) -> torch.Tensor:
        """get_timestep: the chunks' times, newest chunk last; the clean chunk (if any) in front at `clean_t`."""
        t = t_total[[i + advance for i in plan.t_index]]
        if plan.fwd_extra_1st_chunk and advance == 0:
            t = torch.cat([torch.ones(1, device=t.device) * clean_t, t], 0)
        return t

    def kv_range(self, plan: ForwardPlan, tokens: int, noise2clean_kvrange: Sequence[int], clean_chunk_kvrange: int,
                 device="cpu") -> torch.Tensor:
        return generate_kvrange_for_denoising_video(tokens, plan.slice_point, plan.denoising_range_num, plan.denoise_step_of_each_chunk,
                                                    self.num_steps, noise2clean_kvrange, clean_chunk_kvrange, 1, device)

    def extract_prefix_video_feature(self, model, prefix_video, y, emb_masks, inference_params, tokens: int, distill_interval) -> None:
        """One forward over the prefix video's whole chunks for the sake of its KV-cache writes (:391-435)."""
        from .kv_ranges import generate_kvrange_for_prefix_video
        rc, off, cw = model.runtime_config, self.chunk_offset, self.chunk_width
        xc = prefix_video[:, :, :off * cw]
        if xc.shape[0] == 1:
            xc = torch.cat([xc, xc], 0)
        null_y = torch.cat([y[1:2, :off], y[1:2, :off]], 0)                   # clean feature without y embedding
        null_m = torch.cat([emb_masks[1:2, :off], emb_masks[1:2, :off]], 0)
        t = (torch.ones(off, device=xc.device) * rc.clean_t).unsqueeze(0).repeat(xc.size(0), 1)
        kv = generate_kvrange_for_prefix_video(tokens, off, rc.noise2clean_kvrange, rc.clean_chunk_kvrange, 1, xc.device)
        model.forward_dispatcher(x=xc, timestep=t, y=null_y.flatten(0, 1).unsqueeze(1), mask=null_m.flatten(0, 1).unsqueeze(1), kv_range=kv,
                                 inference_params=inference_params, chunk_width=cw, num_steps=self.num_steps, slice_point=0, range_num=off,
                                 denoising_range_num=off, fwd_extra_1st_chunk=False, extract_prefix_video_feature=True,
                                 distill_interval=distill_interval)

    # ---- the loop -------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def run(self, model, x: torch.Tensor, y: torch.Tensor, emb_masks: torch.Tensor, inference_params, t_schedule_config=None,
            steps: Optional[Sequence[int]] = None, on_forward=None, prefix_video: Optional[torch.Tensor] = None) -> torch.Tensor:
        """forward_velocity + integrate_velocity for every step (or the listed `steps`) of one clip.
        x `[2 N, C, T, H, W]` noise (both halves equal, as upstream's `torch.cat([x, x])`), y `[2, chunk_num, L, C]` caption
        embeddings (row 1: the null caption), emb_masks `[2, chunk_num, L]`.  `prefix_video` `[2 N, C, Tp, H, W]` clean latents in front
        (image / video continuation; the schedule must have been built with `chunk_offset = Tp // chunk_width`): its whole chunks go
        through the model once, at clean_t with the null caption, to fill the KV cache (`extract_prefix_video_feature`, :391-435), and
        every forward has the prefix frames it overlaps pasted over its noise at t = 1 (`try_pad_prefix_video`, :437-454).
        Returns x denoised in place."""
        rc, ec, mc = model.runtime_config, model.engine_config, model.model_config
        dev = x.device
        if (prefix_video.size(2) // self.chunk_width if prefix_video is not None else 0) != self.chunk_offset:
            raise ValueError(f"the schedule was built for chunk_offset {self.chunk_offset}, the prefix video holds "
                             f"{0 if prefix_video is None else prefix_video.size(2) // self.chunk_width} whole chunks")
        shortcut = getattr(ec, "shortcut_mode", "")
        t_total = init_t(t_schedule_config or {}, self.num_steps, dev, shortcut)
        interval = init_interval(self.num_steps, dev, shortcut)
        tokens = chunk_token_nums(self.chunk_width, x.shape[3], x.shape[4], getattr(mc, "patch_size", 2))
        cw = self.chunk_width
        for step in (range(self.total_forward_step()) if steps is None else steps):
            p = self.plan(step)
            xc = x[:, :, p.chunk_start * cw: p.chunk_end * cw].clone()
            yc, mk = y[:, p.chunk_start:p.chunk_end], emb_masks[:, p.chunk_start:p.chunk_end]
            if p.fwd_extra_1st_chunk:
                xc = torch.cat([x[:, :, (p.chunk_start - 1) * cw: p.chunk_start * cw].clone(), xc], dim=2)
                yc = torch.cat([y[1:2, 0:1].expand(yc.size(0), -1, -1, -1), yc], dim=1)          # clean feature without y embedding
                mk = torch.cat([emb_masks[1:2, 1:2].expand(mk.size(0), -1, -1), mk], dim=1)
            if self.chunk_offset > 0 and step == 0:
                self.extract_prefix_video_feature(model, prefix_video, y, emb_masks, inference_params, tokens, interval[0])
            t = self.timestep(t_total, p, rc.clean_t).unsqueeze(0).repeat(xc.size(0), 1)
            kv = self.kv_range(p, tokens, rc.noise2clean_kvrange, rc.clean_chunk_kvrange, dev)
            if prefix_video is not None:                                      # try_pad_prefix_video
                start = p.slice_point * cw
                if prefix_video.size(2) > start:
                    n = min(prefix_video.size(2) - start, xc.size(2))
                    xc[:, :, :n] = prefix_video[:, :, start:start + n]
                    clean = (prefix_video.size(2) - start) // cw
                    if clean > 0:
                        t[:, :clean] = 1.0
            nearly_clean_t = t[0, int(p.fwd_extra_1st_chunk)].item()
            kwargs = dict(chunk_width=cw, fwd_extra_1st_chunk=p.fwd_extra_1st_chunk, num_steps=self.num_steps, slice_point=p.slice_point,
                          range_num=p.range_num, denoising_range_num=p.denoising_range_num,
                          distill_nearly_clean_chunk=nearly_clean_t > getattr(ec, "distill_nearly_clean_chunk_threshold", 0.3),
                          distill_interval=interval[p.denoise_idx])
            v = model.forward_dispatcher(x=xc, timestep=t, y=yc.flatten(0, 1).unsqueeze(1), mask=mk.flatten(0, 1).unsqueeze(1),
                                         kv_range=kv, inference_params=inference_params, **kwargs)
            if on_forward is not None:
                on_forward(p)
            if p.fwd_extra_1st_chunk:
                xc, v = xc[:, :, cw:], v[:, :, cw:]
            dt = self.timestep(t_total, p, rc.clean_t, advance=1) - t_total[p.t_index]
            N, C, T, H, W = xc.shape
            xc = (xc.reshape(N, C, -1, cw, H, W) + v.reshape(N, C, -1, cw, H, W) * dt.reshape(1, 1, -1, 1, 1, 1)).reshape(N, C, T, H, W)
            x[:, :, p.chunk_start * cw: p.chunk_end * cw] = xc
        return x
