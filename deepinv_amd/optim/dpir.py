"""DPIR = 8-iteration PnP-HQS with log-spaced noise levels (reference deepinv/optim/dpir.py:10-81)."""
from __future__ import annotations

import torch

from .data_fidelity import L2
from .optimizers import BaseOptim, create_iterator
from .prior import PnP


def get_DPIR_params(noise_level_img, device="cpu", max_iter=8):
    s1, s2 = 49.0 / 255.0, noise_level_img
    sigma = torch.logspace(torch.log10(torch.tensor(s1, dtype=torch.float32)),
                           torch.log10(torch.tensor(float(s2), dtype=torch.float32)), steps=max_iter,
                           dtype=torch.float32, device="cpu").to(device)
    stepsize = (sigma / max(0.01, noise_level_img)) ** 2
    return sigma, (1 / 0.23) * stepsize, max_iter


class DPIR(BaseOptim):
    def __init__(self, sigma=0.1, denoiser=None, device="cpu"):
        if denoiser is None:
            raise ValueError("pass a denoiser (no network access to download pretrained DRUNet weights)")
        prior = PnP(denoiser=denoiser)
        sig, step, max_iter = get_DPIR_params(sigma, device=device)
        super().__init__(create_iterator("HQS", prior=prior, cost_fn=None, g_first=False), max_iter=max_iter,
                         data_fidelity=L2(), prior=prior, early_stop=False,
                         params_algo={"stepsize": step, "g_param": sig})
