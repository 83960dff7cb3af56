"""Priors (reference deepinv/optim/prior.py:22-109)."""
from __future__ import annotations

import torch

from .potential import Potential


class Prior(Potential):
    def __init__(self, g=None, *args, **kwargs):
        super().__init__(*args, fn=g, **kwargs)
        self.explicit_prior = self._fn is not None


class ZeroPrior(Prior):
    def __init__(self):
        super().__init__()
        self.explicit_prior = True

    def fn(self, x, *args, **kwargs):
        return torch.zeros(x.shape[0], device=x.device)

    def grad(self, x, *args, **kwargs):
        return torch.zeros_like(x)

    def prox(self, x, ths=1.0, gamma=1.0, *args, **kwargs):
        return x


class PnP(Prior):
    r"""Plug-and-play prior :math:`\operatorname{prox}_{\gamma g}(x) = D_\sigma(x)` (prior.py:86-109)."""

    def __init__(self, denoiser, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.denoiser = denoiser
        self.explicit_prior = False

    def prox(self, x, sigma_denoiser, *args, **kwargs):
        return self.denoiser(x, sigma_denoiser)
