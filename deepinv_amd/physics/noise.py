"""Noise models applied by ``Physics.forward`` only (reference deepinv/physics/noise.py:11-330).

Gaussian noise on a HIP device is one fused pass (csrc/random.hip: Philox4x32-10 + Box-Muller, y = x + sigma n); on
other devices (and for the other models) it is the reference's torch expression.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class NoiseModel(nn.Module):
    def __init__(self, noise_model=None, rng: torch.Generator | None = None):
        super().__init__()
        self._fn = noise_model
        self.rng = rng

    def forward(self, x, seed: int | None = None, **kwargs):
        self.rng_manual_seed(seed)
        return x if self._fn is None else self._fn(x)

    def rng_manual_seed(self, seed: int | None = None):
        if seed is not None:
            if self.rng is None:
                raise ValueError("seed given but the noise model has no random generator (rng=None)")
            self.rng.manual_seed(seed)

    def randn_like(self, x, seed: int | None = None):
        self.rng_manual_seed(seed)
        return torch.empty_like(x).normal_(generator=self.rng)

    def update_parameters(self, **kwargs):
        pass


class ZeroNoise(NoiseModel):
    def forward(self, x, *args, **kwargs):
        return x


class GaussianNoise(NoiseModel):
    r""":math:`y = x + \sigma\epsilon`, :math:`\epsilon\sim\mathcal N(0,I)` (noise.py:197-330)."""

    def __init__(self, sigma=0.1, rng: torch.Generator | None = None):
        super().__init__(rng=rng)
        self.register_buffer("sigma", self._as_sigma(sigma), persistent=True)

    @staticmethod
    def _as_sigma(sigma):
        if isinstance(sigma, torch.Tensor):
            return sigma.detach().clone().float()
        return torch.tensor(float(sigma))

    def forward(self, x, sigma=None, seed=None, **kwargs):
        if sigma is not None:
            self.sigma = self._as_sigma(sigma).to(self.sigma.device)
        s = self.sigma.to(x.device)
        if x.is_cuda and x.dtype == torch.float32 and not (torch.is_grad_enabled() and x.requires_grad) \
                and (s.numel() == 1 or s.numel() == x.shape[0]):
            from ..hip import random as hrand

            self.rng_manual_seed(seed)
            return hrand.gaussian_noise(x, s, self.rng)
        if s.ndim > 0 and s.numel() > 1:
            s = s.reshape(-1, *([1] * (x.ndim - 1)))
        return x + self.randn_like(x, seed=seed) * s

    def update_parameters(self, sigma=None, **kwargs):
        if sigma is not None:
            self.sigma = self._as_sigma(sigma).to(self.sigma.device)
