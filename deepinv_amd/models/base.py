"""``Denoiser`` / ``Reconstructor`` base classes (reference deepinv/models/base.py:10-150)."""
from __future__ import annotations

import torch


class Denoiser(torch.nn.Module):
    def __init__(self, device="cpu"):
        super().__init__()
        self.to(device)

    def forward(self, x, sigma, **kwargs):
        raise NotImplementedError()


class Reconstructor(torch.nn.Module):
    def __init__(self, device="cpu"):
        super().__init__()
        self.to(device)

    def forward(self, y, physics, **kwargs):
        raise NotImplementedError()
