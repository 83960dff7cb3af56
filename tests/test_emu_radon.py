"""The LDS-tiled Radon kernels (deepinv_amd/csrc/radon_tiled.hip) executed on the HOST by the fiber emulation of
tests/emu (the same source the GPU runs) against the CPU oracle, and the host-built window plan checked at the
BASELINE config-3 geometry (512x512, 720 angles).  CPU only; the GPU parity tests are in test_tomography_gpu.py."""
import ctypes

import numpy as np
import pytest
import torch

import emu_lib as E
from oracle import physics_cpu as O


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / b.norm())


def misses():
    L = E.lib()
    return (ctypes.c_int.in_dll(L, "dinv_emu_window_misses").value, ctypes.c_int.in_dll(L, "dinv_emu_segment_misses").value)


CASES = [
    # W, angles (deg), n_img, circle
    (16, torch.linspace(0, 180, 13)[:-1], 2, False),
    (24, torch.linspace(0, 180, 10)[:-1], 3, True),
    (64, torch.linspace(0, 180, 61)[:-1], 4, False),                                     # several chunks per class
    (97, torch.tensor([0., 44.9, 45., 45.1, 89.9, 90., 90.1, 134.9, 135., 135.1, 179.9]), 5, False),   # class borders
    (50, torch.tensor([-30., 200., 359., 720.5, 17., 93., -91.]), 1, True),               # any angle, any order
    (33, torch.tensor([10., 80., 20., 100., 170., 45., 46., 44.]), 9, False),             # ragged last image group
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_tiled_forward_adjoint_match_oracle(case):
    W, ang, B, circle = CASES[case]
    geo = E.RadonGeom(ang, W, circle)
    g = torch.Generator().manual_seed(W)
    x = torch.rand(B, 1, W, W, generator=g)
    y, plan = E.radon_forward_tiled(x, geo)
    assert not torch.isnan(y).any()
    y_ref = O.radon_forward(x, ang, circle)
    assert rel(y, y_ref) < 2e-6
    # same samples, same weights as the gather kernel of radon.hip; only the summation order may differ
    assert float((y - E.radon_forward_gather(x, geo)).abs().max() / y_ref.abs().max()) < 1e-5
    v = torch.randn(y.shape, generator=g)
    xa = E.radon_adjoint_tiled(v, geo)
    assert rel(xa, O.radon_adjoint(v, ang, W, circle)) < 1e-5
    dot = abs(float((y.double() * v.double()).sum()) - float((x.double() * xa.double()).sum()))
    assert dot / float(y.double().norm() * v.double().norm()) < 1e-7      # exact transpose up to summation order
    assert misses() == (0, 0)   # every tap inside the planned window, every contributing detector inside the staged segment


@pytest.mark.parametrize("kw", [1, 2, 4, 8])
def test_tiled_forward_every_chunk_width(kw):
    W, ang = 40, torch.linspace(0, 180, 24)[:-1]
    geo = E.RadonGeom(ang, W, False)
    x = torch.rand(2, 1, W, W, generator=torch.Generator().manual_seed(kw))
    y, plan = E.radon_forward_tiled(x, geo, kw=kw)
    assert plan.kw == kw and rel(y, O.radon_forward(x, ang)) < 2e-6 and misses()[0] == 0


def test_normalisation_by_device_scalar():
    """tomography.py:253-254: A(x) = radon(x) / ||A|| with the norm read by the kernel (no host read-back)"""
    W, ang = 20, torch.linspace(0, 180, 8)[:-1]
    geo = E.RadonGeom(ang, W, False)
    x = torch.rand(1, 1, W, W, generator=torch.Generator().manual_seed(3))
    nrm = torch.tensor([37.25])
    y, _ = E.radon_forward_tiled(x, geo, norm=nrm)
    assert rel(y, O.radon_forward(x, ang) / nrm) < 2e-6
    v = torch.randn(y.shape, generator=torch.Generator().manual_seed(4))
    assert rel(E.radon_adjoint_tiled(v, geo, norm=nrm), O.radon_adjoint(v, ang, W) / nrm) < 1e-5


@pytest.mark.parametrize("shape", [(1, 1, 23, 12), (2, 1, 91, 7), (1, 2, 182, 30), (1, 1, 47, 1)])
def test_ramp_filter_fft_matches_oracle(shape):
    """AbstractFilter.forward (radon.py:79-149): pad to 2^k, rfft * F, irfft, crop - even and odd angle counts"""
    y = torch.randn(shape, generator=torch.Generator().manual_seed(shape[2]))
    assert rel(E.ramp_fft(y), O.ramp_filter(y)) < 2e-6


def test_plan_covers_every_tap_at_config3_geometry():
    """512x512, 720 angles (BASELINE configs[2]): replay the kernel's band / window logic in numpy (fp64 positions)
    for a sample of (chunk, ray block) pairs and check that every valid sample's taps lie in the planned window."""
    ang = torch.linspace(0, 180, 721)[:-1]
    geo = E.RadonGeom(ang, 512, False)
    plan, blob = geo.plan(8)
    G, A, kw, BH = geo.G, geo.A, plan.kw, plan.band_h
    assert G == 725 and plan.fits == 1 and plan.widest_window <= 136 and kw == 8
    nch_max = (A + kw - 1) // kw + 4
    angles = blob[:nch_max * kw].reshape(nch_max, kw)
    dirs = blob[nch_max * kw: nch_max * kw + nch_max]
    off = nch_max * kw + nch_max
    wtab = blob[off: off + nch_max * plan.n_jblocks * plan.n_bands].reshape(nch_max, plan.n_jblocks, plan.n_bands)
    nch = plan.n_chunks_plain + plan.n_chunks_swap
    assert sorted(a for a in angles[:nch].ravel() if a >= 0) == list(range(A))   # every angle exactly once
    cs = geo.cs.double().numpy()
    ctr = 0.5 * (G - 1)
    rng = np.random.default_rng(0)
    ii = np.arange(G) - ctr
    for ch in rng.choice(nch, 24, replace=False):
        swap = ch >= plan.n_chunks_plain
        for jb in rng.choice(plan.n_jblocks, 4, replace=False):
            jj = np.arange(jb * 64, min(jb * 64 + 64, G)) - ctr
            for a in angles[ch]:
                if a < 0:
                    continue
                c, s = cs[a]
                assert (abs(s) > abs(c)) == swap and np.sign(s if swap else c) == dirs[ch]
                ix = ctr + c * jj[:, None] + s * ii[None, :]
                iy = ctr - s * jj[:, None] + c * ii[None, :]
                u, v = (iy, ix) if swap else (ix, iy)
                u0, v0 = np.floor(u).astype(int), np.floor(v).astype(int)
                ok = (u0 >= -1) & (u0 <= G - 1) & (v0 >= -1) & (v0 <= G - 1)
                band = (v0 + 1) // BH
                w = wtab[ch, jb][np.clip(band, 0, plan.n_bands - 1)]
                wx0, ww = (w & 0xffff) - 8, w >> 16
                assert np.all(band[ok] < plan.n_bands)
                assert np.all((u0 >= wx0)[ok] & (u0 + 1 <= wx0 + ww - 1)[ok])
                assert np.all(ww[ok] <= plan.win_w)


FAN_A = {"pixel_spacing": 0.1, "source_radius": 6.0, "detector_radius": 6.0, "n_detector_pixels": 20, "detector_spacing": 0.34}
FAN_CASES = [
    # W, angles, n_img, circle, fan parameters (None: the reference's defaults, 258 detector pixels that mostly miss)
    (16, torch.linspace(0, 360, 11)[:-1], 2, False, FAN_A),
    (16, torch.linspace(0, 360, 11)[:-1], 2, True, FAN_A),
    (16, torch.linspace(0, 180, 7)[:-1], 2, False, None),
    (31, torch.tensor([-20., 33., 91., 180., 271.5]), 3, False,
     {"pixel_spacing": 0.05, "source_radius": 3.0, "detector_radius": 5.0, "n_detector_pixels": 47, "detector_spacing": 0.11}),
    (24, torch.linspace(0, 360, 9)[:-1], 9, True,                     # detector finer than the pixels: many d per pixel
     {"pixel_spacing": 0.1, "source_radius": 8.0, "detector_radius": 2.0, "n_detector_pixels": 150, "detector_spacing": 0.03}),
    (20, torch.linspace(0, 360, 13)[:-1], 2, False,                   # source close to the image: the stretch varies 10x along a ray
     {"pixel_spacing": 0.1, "source_radius": 1.6, "detector_radius": 12.0, "n_detector_pixels": 40, "detector_spacing": 0.9}),
    (20, torch.linspace(0, 360, 13)[:-1], 2, True,                    # very coarse detector: a detector pixel covers many image pixels
     {"pixel_spacing": 0.02, "source_radius": 5.0, "detector_radius": 5.0, "n_detector_pixels": 6, "detector_spacing": 0.5}),
]


@pytest.mark.parametrize("case", range(len(FAN_CASES)))
def test_fan_beam_forward_adjoint_match_oracle(case):
    """fan-beam kernels of radon.hip (fan_beam_grid, radon.py:16-52) on the host emulation vs the CPU oracle; the adjoint
    is the exact transpose (dot test) although the detector-candidate window is only a bound"""
    W, ang, B, circle, fan = FAN_CASES[case]
    geo = E.FanGeom(ang, W, circle, fan)
    g = torch.Generator().manual_seed(100 + case)
    x = torch.rand(B, 1, W, W, generator=g)
    y = E.radon_fan_forward(x, geo)
    assert not torch.isnan(y).any() and y.shape == (B, 1, geo.n_det, len(ang))
    y_ref = O.radon_fan_forward(x, ang, fan, circle)
    assert rel(y, y_ref) < 2e-6
    v = torch.randn(y.shape, generator=g)
    xa = E.radon_fan_adjoint(v, geo)
    assert rel(xa, O.radon_fan_adjoint(v, ang, W, fan, circle)) < 1e-5
    dot = abs(float((y.double() * v.double()).sum()) - float((x.double() * xa.double()).sum()))
    assert dot / float(y.double().norm() * v.double().norm()) < 1e-7
