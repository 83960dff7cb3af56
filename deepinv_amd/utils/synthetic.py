"""Synthetic problem inputs used by bench.py / tests."""
from __future__ import annotations

import math

import torch


def radial_mask(H: int, W: int, n_spokes: int) -> torch.Tensor:
    """{0,1} k-space mask made of ``n_spokes`` lines through the centre at angles k*pi/n_spokes,
    rasterised with integer index arithmetic (bit-reproducible).  80 spokes on 320x320 is ~4x
    undersampling.  The reference only ships Cartesian generators (generator/mri.py:136-327) but its
    MRI operators accept any (H,W) mask (mixins.py:127-146)."""
    mask = torch.zeros(H, W)
    cy, cx = H // 2, W // 2
    L = int(math.ceil(math.hypot(H, W)))
    t = torch.arange(-L, L + 1, dtype=torch.float64)
    for k in range(n_spokes):
        a = math.pi * k / n_spokes
        yy = torch.round(cy + t * math.sin(a)).long()
        xx = torch.round(cx + t * math.cos(a)).long()
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        mask[yy[ok], xx[ok]] = 1.0
    return mask
