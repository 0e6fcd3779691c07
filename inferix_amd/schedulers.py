"""Flow-matching scheduler of the Self-Forcing / CausVid wrappers
(inferix/models/schedulers/flow_match.py:106-176): shifted sigma table, timestep lookup by nearest
entry, `add_noise`.  Host-side tables + one axpby on the device."""
from __future__ import annotations

import torch


class TensorMemo:
    """value = f(tensor) memoised on the tensor's IDENTITY: storage address + version counter (any in-place write bumps it) + shape,
    with the tensor kept alive in the entry — no other tensor can be allocated at that address while the entry exists, so a hit is the
    same values.  For the handful of timestep tensors a clip reuses (sigma lookups: an argmin over the 1000-entry table per call)."""

    def __init__(self, capacity: int = 16):
        self.capacity, self.entries = capacity, {}

    def get(self, t: torch.Tensor, extra, make):
        # (a VIEW of the same storage — `timestep.flatten(0, 1)` makes a new tensor object per call — is the same values: the entry's
        #  tensor keeps that storage alive, views share its version counter, and shape + strides are part of the key)
        key = (t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()), t.dtype, extra)
        hit = self.entries.get(key)
        if hit is not None:
            return hit[1]
        v = make()
        if len(self.entries) >= self.capacity:
            self.entries.pop(next(iter(self.entries)))
        self.entries[key] = (t, v)
        return v


class FlowMatchScheduler:
    def __init__(self, num_inference_steps=100, num_train_timesteps=1000, shift=3.0, sigma_max=1.0,
                 sigma_min=0.003 / 1.002, inverse_timesteps=False, extra_one_step=False, reverse_sigmas=False):
        self.num_train_timesteps, self.shift = num_train_timesteps, shift
        self.sigma_max, self.sigma_min = sigma_max, sigma_min
        self.inverse_timesteps, self.extra_one_step, self.reverse_sigmas = inverse_timesteps, extra_one_step, reverse_sigmas
        self.set_timesteps(num_inference_steps)

    def set_timesteps(self, num_inference_steps=100, denoising_strength=1.0, training=False):
        start = self.sigma_min + (self.sigma_max - self.sigma_min) * denoising_strength
        n = num_inference_steps + 1 if self.extra_one_step else num_inference_steps
        s = torch.linspace(start, self.sigma_min, n)
        if self.extra_one_step:
            s = s[:-1]
        if self.inverse_timesteps:
            s = torch.flip(s, dims=[0])
        s = self.shift * s / (1 + (self.shift - 1) * s)
        if self.reverse_sigmas:
            s = 1 - s
        self.sigmas = s
        self.timesteps = s * self.num_train_timesteps
        if training:
            x = self.timesteps
            y = torch.exp(-2 * ((x - num_inference_steps / 2) / num_inference_steps) ** 2)
            y = y - y.min()
            self.linear_timesteps_weights = y * (num_inference_steps / y.sum())

    def _lookup(self, timestep: torch.Tensor, device) -> torch.Tensor:
        if timestep.ndim == 2:
            timestep = timestep.flatten(0, 1)
        self.sigmas = self.sigmas.to(device)
        self.timesteps = self.timesteps.to(device)
        return torch.argmin((self.timesteps.unsqueeze(0) - timestep.to(device).unsqueeze(1)).abs(), dim=1)

    def add_noise(self, original_samples, noise, timestep):
        """(1 - sigma) * x0 + sigma * noise, sigma fp32, cast to noise dtype. [B*T, C, H, W], [B*T]."""
        memo = self.__dict__.setdefault("_sigma_memo", TensorMemo())
        sigma = memo.get(timestep, (str(noise.device), self.sigmas.data_ptr()),
                         lambda: self.sigmas.to(noise.device)[self._lookup(timestep, noise.device)].reshape(-1, 1, 1, 1))
        return ((1 - sigma) * original_samples + sigma * noise).type_as(noise)

    def step(self, model_output, timestep, sample, to_final=False):
        idx = self._lookup(timestep, model_output.device)
        sigma = self.sigmas[idx].reshape(-1, 1, 1, 1)
        if to_final or bool((idx + 1 >= len(self.timesteps)).any()):
            nxt = 1 if (self.inverse_timesteps or self.reverse_sigmas) else 0
        else:
            nxt = self.sigmas[idx + 1].reshape(-1, 1, 1, 1)
        return sample + model_output * (nxt - sigma)

    def training_target(self, sample, noise, timestep):
        return noise - sample
