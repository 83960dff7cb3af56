"""Cartesian MRI acceleration-mask generators on the device (reference deepinv/physics/generator/mri.py:15-384,
generator/base.py:20-170).  Same constructor arguments, same `step()` contract and the same distributions as the
reference; the per-sample Python loops are one kernel launch for the whole batch (csrc/random.hip)."""
from __future__ import annotations

import warnings

import torch
import torch.nn as nn

from ..hip import random as hrand


def ceildiv(a, b):
    return -(a // -b)


class PhysicsGenerator(nn.Module):
    """base.py:20-170 (the part the mask generators use)"""

    def __init__(self, step=lambda **kwargs: {}, rng: torch.Generator | None = None, device="cpu", dtype=torch.float32,
                 **kwargs):
        super().__init__()
        self.step_func, self.kwargs = step, kwargs
        self.factory_kwargs = {"device": device, "dtype": dtype}
        self.device = torch.device(device)
        if rng is not None and torch.device(rng.device).type != self.device.type:
            raise ValueError(f"The random generator is not on the same device as the Physics Generator. Got random "
                             f"generator on {rng.device} and the Physics Generator named {self.__class__.__name__} on {device}.")
        self.rng = rng if rng is not None else torch.Generator(device=device)

    def rng_manual_seed(self, seed: int | None = None):
        if seed is not None:
            self.rng = self.rng.manual_seed(seed)

    def step(self, batch_size: int = 1, seed: int | None = None, **kwargs):
        self.rng_manual_seed(seed)
        return self.step_func(batch_size, seed, **{**self.kwargs, **kwargs})


class BaseMaskGenerator(PhysicsGenerator):
    """mri.py:15-131: vertical k-space lines, fixed low-frequency band + child-specific high-frequency sampling"""

    mode = 0

    def __init__(self, img_size, acceleration: int = 4, center_fraction: float | None = None, rng=None, device="cpu",
                 *args, **kwargs):
        super().__init__(*args, **kwargs, rng=rng, device=device)
        self.img_size, self.acc = img_size, acceleration
        self.center_fraction = center_fraction if center_fraction is not None else (0.08 if acceleration < 8 else 0.04)
        if len(img_size) == 2:
            (self.H, self.W), self.C, self.T = img_size, 1, 0
        elif len(img_size) == 3:
            (self.C, self.H, self.W), self.T = img_size, 0
        elif len(img_size) == 4:
            self.C, self.T, self.H, self.W = img_size
        else:
            raise ValueError("img_size must be (H, W) or (C, H, W) or (C, T, H, W)")
        self.calculate_lines(self.W)

    def calculate_lines(self, W: int):
        self.n_center = int(self.center_fraction * W)
        self.n_lines = int(W // self.acc - self.n_center)
        if self.n_lines < 0:
            raise ValueError("center_fraction is too high for this acceleration factor.")
        if self.n_lines == 0:
            warnings.warn("Number of high frequency lines to be sampled is 0. Reduce acceleration factor or reduce "
                          "center_fraction.")

    # ---- what the child classes define: centre band and the kernel's sampling parameters
    def _center(self, W):
        return W // 2 - self.n_center // 2, W // 2 + ceildiv(self.n_center, 2)

    def _sampling(self, W):
        raise NotImplementedError

    def step(self, batch_size=1, seed: int | None = None, img_size=None, **kwargs) -> dict:
        """mri.py:93-131: {'mask': [B, C, H, W] or [B, C, T, H, W] in {0, 1}}"""
        self.rng_manual_seed(seed)
        _B = 1 if batch_size == 0 else batch_size
        _T = self.T if self.T > 0 else 1
        _H, _W = (self.H, self.W) if img_size is None else img_size
        self.calculate_lines(_W)
        if self.n_lines + self.n_center >= _W:
            mask = torch.ones((_B, self.C, _T, _H, _W), **self.factory_kwargs)
        else:
            pdf, accel, n_off = self._sampling(_W)
            mask = hrand.mri_mask_lines(_B, self.C, _T, _H, _W, self.n_lines, self._center(_W), self.mode, pdf, accel, n_off,
                                        self.device, self.rng).to(self.factory_kwargs["dtype"])
        if self.T == 0:
            mask = mask[:, :, 0]
        if batch_size == 0:
            mask = mask[0]
        return {"mask": mask}


class RandomMaskGenerator(BaseMaskGenerator):
    """uniform random high-frequency lines (mri.py:134-196)"""

    def get_pdf(self, W: int) -> torch.Tensor:
        return torch.ones(W, device=self.device)

    def _sampling(self, W):
        pdf = self.get_pdf(W).float()
        lo, hi = self._center(W)
        pdf[lo:hi] = 0     # lines are never randomly sampled from the already sampled centre
        return (pdf / pdf.sum()).contiguous(), 1.0, 0


class GaussianMaskGenerator(RandomMaskGenerator):
    """tail-adjusted Gaussian density over the columns (mri.py:262-301)"""

    def get_pdf(self, W: int) -> torch.Tensor:
        x = torch.arange(W, device=self.device)
        pdf = torch.exp(-(0.5 / (W / 10.0) ** 2) * (x - W / 2) ** 2)
        return pdf + (W / (2.0 * self.acc) * 1.0 / W)


class EquispacedMaskGenerator(BaseMaskGenerator):
    """equispaced lines with a random offset per sample, sheared over time (mri.py:304-384)"""

    mode = 1

    def get_pdf(self):
        raise NotImplementedError("get_pdf is undefined for this mask generator.")

    def _center(self, W):
        pad = (W - self.n_center + 1) // 2
        return pad, pad + self.n_center

    def _sampling(self, W):
        adjusted = (self.acc * (self.n_center - W)) / (self.n_center * self.acc - W)
        return None, adjusted, round(adjusted)


class PolyOrderMaskGenerator(BaseMaskGenerator):
    """polynomial variable density (mri.py:199-281, Millard & Chiew): density (1 - r)^p over the normalised distance r from the
    centre column, shifted by a constant found by bisection so that the mean sampling rate is 1 / acceleration, centre band
    probability 1; every column of a mask row is then an independent Bernoulli draw (one launch for the whole batch, mode 2 of
    dinv_mri_mask_lines).  The number of sampled lines is random (only its mean is 1 / acceleration), as in the reference."""

    mode = 2

    def __init__(self, *args, poly_order: int = 8, **kwargs):
        super().__init__(*args, **kwargs)
        self.poly_order = poly_order
        self.pdf = self.get_pdf()

    def get_pdf(self, max_iter: int = 100, tol: float = 1e-3) -> torch.Tensor:
        """the Bernoulli probabilities per column (fp32 arithmetic in the reference's order: its bisection stops at the first
        midpoint whose rate is within `tol` of the target, so the same sequence of midpoints must be visited)"""
        lo_c, hi_c = BaseMaskGenerator._center(self, self.W)
        base = (1 - torch.linspace(-1, 1, self.W).abs()) ** self.poly_order
        base[lo_c:hi_c] = 1
        target = 1.0 / self.acc
        lo, hi = -1.0, 1.0
        for _ in range(max_iter):
            shift = (lo + hi) / 2
            cand = (base + shift).clamp_(0, 1)
            cand[lo_c:hi_c] = 1
            rate = cand.mean().item()
            if rate < target - tol:
                lo = shift
            elif rate > target + tol:
                hi = shift
            else:
                return cand.to(self.device)
        raise ValueError(f"get_pdf did not converge after {max_iter} iterations")

    def _sampling(self, W):
        if W != self.W:
            raise ValueError(f"PolyOrderMaskGenerator samples masks of the width it was built for ({self.W}), got {W}")
        return self.pdf.float().contiguous(), 1.0, 0
