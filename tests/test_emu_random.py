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


# ---- distributional parity with the REFERENCE's generators (tests/golden/mask_generators.npz, made by
# tests/golden/make_golden_r6.py from deepinv.physics.generator.{Random,Gaussian,Equispaced,PolyOrder}MaskGenerator: per-column
# inclusion counts over 4096 masks).  Two-sample test per column: z = (a/Na - b/Nb) / sqrt(p (1 - p) (1/Na + 1/Nb)) with the pooled
# rate p; sum z^2 over the non-degenerate columns ~ chi-square(df) under equal distributions (the columns of a without-replacement
# draw are negatively correlated, which only lowers the sum).  Bound: df + 5 sqrt(2 df) (false-alarm rate ~ 1e-6).
def chi_square_vs_reference(counts, n, ref_counts, n_ref):
    a, b = np.asarray(counts, float), np.asarray(ref_counts, float)
    p = (a + b) / (n + n_ref)
    live = (p > 0) & (p < 1)
    assert np.array_equal(a[~live] / n, b[~live] / n_ref)            # always / never sampled columns agree exactly
    z2 = (a[live] / n - b[live] / n_ref) ** 2 / (p[live] * (1 - p[live]) * (1.0 / n + 1.0 / n_ref))
    df = int(live.sum())
    return float(z2.sum()), df, df + 5.0 * np.sqrt(2.0 * df)


def _golden_masks():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mask_generators.npz"))


@pytest.mark.parametrize("kind", ["random", "gaussian"])
def test_random_line_masks_have_the_reference_distribution(kind):
    """Gumbel top-n on the device against torch.multinomial(replacement=False) in the reference (mri.py:167-181, rand.py:75-77)"""
    d = _golden_masks()
    W, acc = 128, 8
    n_lines, n_center = (int(v) for v in d[f"{kind}_{W}_{acc}_lines"])
    lo, hi = W // 2 - n_center // 2, W // 2 - n_center // 2 + n_center
    x = torch.arange(W)
    pdf = torch.ones(W) if kind == "random" else torch.exp(-(0.5 / (W / 10.0) ** 2) * (x - W / 2) ** 2) + (W / (2.0 * acc) / W)
    pdf[lo:hi] = 0
    pdf = (pdf / pdf.sum()).float().contiguous()
    N = 1024
    m = mask_lines(N, 1, 1, 1, W, n_lines, lo, hi, 0, pdf, 1.0, 0, 4242, 0)[:, 0, 0, 0]
    assert torch.all(m.sum(-1) == n_lines + n_center)
    stat, df, bound = chi_square_vs_reference(m.sum(0).numpy(), N, d[f"{kind}_{W}_{acc}_counts"], int(d["n_masks"]))
    assert stat < bound, (stat, df, bound)


def test_bernoulli_column_masks_have_the_reference_distribution():
    """mode 2 (PolyOrderMaskGenerator.sample_mask, mri.py:273-281: torch.bernoulli(pdf)) with the reference's own pdf"""
    d = _golden_masks()
    W, acc, N = 128, 8, 1024
    for order in (4, 8):
        pdf = torch.from_numpy(d[f"poly{order}_{W}_{acc}_pdf"]).float().contiguous()
        m = mask_lines(N, 1, 1, 1, W, 0, 0, 0, 2, pdf, 1.0, 0, 77 + order, 0)[:, 0, 0, 0]
        stat, df, bound = chi_square_vs_reference(m.sum(0).numpy(), N, d[f"poly{order}_{W}_{acc}_counts"], int(d["n_masks"]))
        assert stat < bound, (order, stat, df, bound)
        assert torch.all(m[:, pdf == 1] == 1) and torch.all(m[:, pdf == 0] == 0)


def test_equispaced_patterns_are_the_reference_patterns():
    """every mask the kernel draws is one of the reference's column patterns (one per offset) and the offsets are uniform"""
    d = _golden_masks()
    W, acc, cf = 128, 8, 0.04
    pats = torch.from_numpy(d[f"equispaced_{W}_{acc}_patterns"]).float()
    n_center = int(cf * W)
    pad = (W - n_center + 1) // 2
    a = (acc * (n_center - W)) / (n_center * acc - W)
    N = 1024
    m = mask_lines(N, 1, 1, 1, W, 0, pad, pad + n_center, 1, None, a, round(a), 31, 0)[:, 0, 0, 0]
    which = (m[:, None, :] == pats[None]).all(-1)                     # [N, n_patterns]
    assert bool(which.any(1).all())
    assert pats.shape[0] == round(a)
    counts = which.float().sum(0).numpy()
    ref = d[f"equispaced_{W}_{acc}_pattern_counts"].astype(float)
    # both are uniform over the offsets: chi-square of the kernel's histogram against the uniform law, and of the reference's
    for c, n in ((counts, N), (ref, ref.sum())):
        e = n / len(c)
        assert float(((c - e) ** 2 / e).sum()) < (len(c) - 1) + 5 * np.sqrt(2.0 * (len(c) - 1))


def test_poly_order_pdf_equals_the_reference():
    """PolyOrderMaskGenerator.get_pdf (host arithmetic): the bisected Bernoulli probabilities of the reference, element for element"""
    import deepinv_amd as dinv

    d = _golden_masks()
    for W, acc in ((320, 4), (128, 8)):
        for order in (4, 8):
            gen = dinv.physics.generator.PolyOrderMaskGenerator((2, 8, W), acceleration=acc, poly_order=order)
            assert torch.equal(gen.pdf, torch.from_numpy(d[f"poly{order}_{W}_{acc}_pdf"]))
    with pytest.raises(ValueError):
        dinv.physics.generator.PolyOrderMaskGenerator((2, 8, 64), acceleration=4, center_fraction=0.5)
