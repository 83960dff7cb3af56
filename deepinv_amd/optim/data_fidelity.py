"""Data-fidelity terms (reference deepinv/optim/data_fidelity.py:26-338)."""
from __future__ import annotations

import torch

from ..physics.forward import LinearPhysics
from .distance import Distance, L2Distance
from .potential import Potential


class DataFidelity(Potential):
    r""":math:`f(x) = d(A(x), y)` (data_fidelity.py:26-160)."""

    def __init__(self, d=None):
        super().__init__()
        self.d = Distance(d=d)

    def fn(self, x, y, physics, *args, **kwargs):
        return self.d(physics.A(x), y, *args, **kwargs)

    def grad(self, x, y, physics, *args, **kwargs):
        return physics.A_vjp(x, self.d.grad(physics.A(x), y, *args, **kwargs))

    def grad_d(self, u, y, *args, **kwargs):
        return self.d.grad(u, y, *args, **kwargs)

    def prox_d(self, u, y, *args, **kwargs):
        return self.d.prox(u, y, *args, **kwargs)

    def prox_d_conjugate(self, u, y, *args, **kwargs):
        return self.d.prox_conjugate(u, y, *args, **kwargs)


class ZeroFidelity(DataFidelity):
    def fn(self, x, y, physics, *args, **kwargs):
        return torch.zeros(x.shape[0], device=x.device)

    def grad(self, x, y, physics, *args, **kwargs):
        return torch.zeros_like(x)

    def prox(self, x, y, physics, *args, gamma=1.0, **kwargs):
        return x


class L2(DataFidelity):
    r""":math:`\frac{1}{2\sigma^2}\|Ax-y\|^2` (data_fidelity.py:237-338)."""

    def __init__(self, sigma=1.0):
        super().__init__()
        self.d = L2Distance(sigma=sigma)
        self.norm = 1 / (sigma ** 2)

    def prox(self, x, y, physics, *args, gamma=1.0, **kwargs):
        return physics.prox_l2(x, y, self.norm * gamma)

    def grad(self, x, y, physics, *args, **kwargs):
        if isinstance(physics, LinearPhysics):
            return self.norm * (physics.A_adjoint_A(x) - physics.A_adjoint(y))
        return super().grad(x, y, physics, *args, **kwargs)
