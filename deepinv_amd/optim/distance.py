"""Distances (reference deepinv/optim/distance.py:13-115)."""
from __future__ import annotations

import torch

from .potential import Potential


class Distance(Potential):
    def __init__(self, d=None):
        super().__init__(fn=d)

    def fn(self, x, y, *args, **kwargs):
        return self._fn(x, y, *args, **kwargs)

    def forward(self, x, y, *args, **kwargs):
        return self.fn(x, y, *args, **kwargs)


class L2Distance(Distance):
    r""":math:`\frac{1}{2\sigma^2}\|x-y\|^2` (distance.py:47-115)."""

    def __init__(self, sigma=1.0):
        super().__init__()
        self.norm = 1 / (sigma ** 2)

    def fn(self, x, y, *args, **kwargs):
        z = x - y
        return 0.5 * torch.linalg.vector_norm(z, ord=2, dim=tuple(range(1, z.dim()))) ** 2 * self.norm

    def grad(self, x, y, *args, **kwargs):
        return (x - y) * self.norm

    def prox(self, x, y, *args, gamma=1.0, **kwargs):
        return (x + self.norm * gamma * y) / (1 + gamma * self.norm)
