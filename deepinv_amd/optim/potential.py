"""``Potential`` base class (reference deepinv/optim/potential.py:14-185)."""
from __future__ import annotations

from typing import Callable

import torch
import torch.nn as nn


def gradient_descent(grad_f: Callable, x: torch.Tensor, step_size=1.0, max_iter=100, tol=1e-5):
    """plain gradient descent used by the generic ``Potential.prox`` (deepinv/optim/utils.py)."""
    for _ in range(int(max_iter)):
        x_prev = x
        x = x - grad_f(x) * step_size
        if ((x - x_prev).norm() / (x.norm() + 1e-8)) < tol:
            break
    return x


class Potential(nn.Module):
    def __init__(self, fn: Callable = None):
        super().__init__()
        self._fn = fn

    def fn(self, x, *args, **kwargs):
        return self._fn(x, *args, **kwargs)

    def forward(self, x, *args, **kwargs):
        return self.fn(x, *args, **kwargs)

    def grad(self, x, *args, **kwargs):
        with torch.enable_grad():
            x = x.requires_grad_()
            h = self.forward(x, *args, **kwargs)
            return torch.autograd.grad(h, x, torch.ones_like(h), create_graph=True, only_inputs=True)[0]

    def prox(self, x, *args, gamma=1.0, stepsize_inter=1.0, max_iter_inter=50, tol_inter=1e-3, **kwargs):
        grad = lambda z: gamma * self.grad(z, *args, **kwargs) + (z - x)
        return gradient_descent(grad, x, step_size=stepsize_inter, max_iter=max_iter_inter, tol=tol_inter)

    def prox_conjugate(self, x, *args, gamma=1.0, lamb=1.0, **kwargs):
        return x - gamma * self.prox(x / gamma, *args, gamma=lamb / gamma, **kwargs)
