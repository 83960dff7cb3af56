"""Measurement-synthesis kernels (deepinv_amd/csrc/random.hip) on the host emulation: Gaussian noise statistics and the
Cartesian mask generators' invariants (reference deepinv/physics/generator/mri.py:93-196, 262-384)."""
import ctypes

import numpy as np
import pytest
import torch

import emu_lib as E


def noise(n, per, x, sigma_t, sigma_f, seed, off):
    y = torch.empty(n)
    E.check(E.lib().dinv_gaussian_noise(ctypes.c_int64(n), ctypes.c_int64(per), E.p(x), E.p(sigma_t), ctypes.c_float(sigma_f),
                                        ctypes.c_uint64(seed), ctypes.c_uint64(off), E.p(y), None))
    return y


def mask_lines(B, C, T, H, W, n_lines, c_lo, c_hi, mode, pdf, accel, n_off, seed, off):
    m = torch.full((B, C, T, H, W), float("nan"))
    E.check(E.lib().dinv_mri_mask_lines(B, C, T, H, W, n_lines, c_lo, c_hi, mode, E.p(pdf), ctypes.c_double(accel), n_off,
                                        ctypes.c_uint64(seed), ctypes.c_uint64(off), E.p(m), None))
    return m


def test_gaussian_noise_statistics_and_reproducibility():
    n = 200_003
    x = torch.linspace(-1, 1, n)
    y = noise(n, n, x, None, 0.5, 1234, 0)
    z = (y - x) / 0.5
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1) < 0.01
    assert abs(float((z ** 3).mean())) < 0.03 and abs(float((z ** 4).mean()) - 3) < 0.06      # skewness, kurtosis
    assert abs(float((z.abs() < 1).float().mean()) - 0.6827) < 0.005
    assert torch.equal(y, noise(n, n, x, None, 0.5, 1234, 0))                   # same (seed, offset) -> same numbers
    assert not torch.equal(y, noise(n, n, x, None, 0.5, 1234, (n + 3) // 4))   # the next offset block is fresh
    # per-sample sigma: [B] table, `per` elements per sample
    ys = noise(4 * 1000, 1000, None, torch.tensor([0.0, 1.0, 2.0, 3.0]), 0.0, 7, 0).reshape(4, 1000)
    assert float(ys[0].abs().max()) == 0 and abs(float(ys[2].std()) - 2) < 0.15 and abs(float(ys[3].std()) - 3) < 0.2


@pytest.mark.parametrize("W,acc,cf", [(64, 4, 0.08), (320, 4, 0.08), (128, 8, 0.04)])
def test_random_line_masks(W, acc, cf):
    """n_center centre columns always sampled + exactly n_lines others, no duplicates, every image row and channel the
    same, rows differ across the batch; column frequencies follow the density"""
    n_center = int(cf * W)
    n_lines = int(W // acc - n_center)
    lo, hi = W // 2 - n_center // 2, W // 2 - n_center // 2 + n_center
    x = torch.arange(W)
    pdf = torch.exp(-(0.5 / (W / 10.0) ** 2) * (x - W / 2) ** 2) + (W / (2.0 * acc) / W)     # GaussianMaskGenerator.get_pdf
    pdf[lo:hi] = 0
    pdf = (pdf / pdf.sum()).float().contiguous()
    B, C, T, H = 64, 2, 3, 4
    m = mask_lines(B, C, T, H, W, n_lines, lo, hi, 0, pdf, 1.0, 0, 99, 0)
    assert set(m.unique().tolist()) == {0.0, 1.0}
    assert torch.equal(m, m[:, :1, :, :1].expand_as(m))                 # same lines for every channel and image row
    lines = m[:, 0, :, 0]                                               # [B, T, W]
    assert torch.all(lines[..., lo:hi] == 1)
    assert torch.all(lines.sum(-1) == n_center + n_lines)
    assert len({tuple(r.tolist()) for r in lines.reshape(-1, W)}) > B * T // 2      # they vary over batch and time
    # first-draw marginals are proportional to pdf; with n_lines draws without replacement the inclusion frequencies
    # are monotone in pdf: compare the empirical frequency of the most / least likely quartiles of columns
    freq = lines.reshape(-1, W).mean(0)
    outside = torch.ones(W, dtype=torch.bool)
    outside[lo:hi] = False
    order = torch.argsort(pdf[outside])
    f = freq[outside][order]
    q = len(f) // 4
    assert float(f[-q:].mean()) > float(f[:q].mean())


@pytest.mark.parametrize("W,acc,cf,T", [(64, 8, 0.04, 8), (320, 4, 0.08, 1)])
def test_equispaced_masks_match_the_reference_formula(W, acc, cf, T):
    """columns = round(arange((t + offset_b) % a, W - 1, a)) + the centre band (mri.py:352-384), for whatever offset
    the kernel drew in [0, round(a))"""
    n_center = int(cf * W)
    pad = (W - n_center + 1) // 2
    a = (acc * (n_center - W)) / (n_center * acc - W)
    B, H = 16, 2
    m = mask_lines(B, 1, T, H, W, 0, pad, pad + n_center, 1, None, a, round(a), 5, 0)[:, 0, :, 0]     # [B, T, W]
    offsets = set()
    for b in range(B):
        ok = False
        for off in range(round(a)):
            good = True
            for t in range(T):
                ref = torch.zeros(W)
                ref[pad:pad + n_center] = 1
                idx = torch.arange((t + off) % a, W - 1, a).round().type(torch.int64)
                ref[idx] = 1
                good &= bool(torch.equal(ref, m[b, t]))
            if good:
                ok = True
                offsets.add(off)
        assert ok, b
    assert len(offsets) > 1     # the offset really is random over the batch
