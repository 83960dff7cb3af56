"""Tomography (Radon fwd / exact adjoint / ramp / FBP) vs the CPU oracle.  fp32, tol 1e-4 relative."""
import pytest
import torch

from conftest import dot_test, rel_err
from oracle import physics_cpu as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("W,nang,circle,B,C", [(16, 16, False, 2, 1), (16, 16, True, 1, 2), (33, 20, False, 3, 1),
                                               (64, 45, False, 9, 1), (64, 45, True, 2, 3), (128, 60, False, 1, 1)])
def test_radon_forward_adjoint(dev, W, nang, circle, B, C):
    import deepinv_amd as dinv

    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, C, W, W, generator=g)
    phys = dinv.physics.Tomography(angles=nang, img_width=W, circle=circle, normalize=False, device=dev)
    ang = phys.angles.cpu()
    y = phys.A(x.to(dev))
    y_ref = O.radon_forward(x, ang, circle)
    assert y.shape == y_ref.shape
    assert rel_err(y, y_ref) < TOL
    v = torch.randn(y_ref.shape, generator=g)
    xa = phys.A_adjoint(v.to(dev))
    assert rel_err(xa, O.radon_adjoint(v, ang, W, circle)) < TOL
    assert dot_test(phys, x.to(dev), y) < 1e-5
    assert abs(float(phys.adjointness_test(x.to(dev)))) < 1e-3 * max(1.0, float(y_ref.abs().sum()) ** 0.5)


def test_tomography_docstring_known_answer(dev):
    """literal vectors of the reference docstring (deepinv/physics/tomography.py:91-114)"""
    import deepinv_amd as dinv

    seed = torch.manual_seed(0)
    x = torch.randn(1, 1, 4, 4)
    angles = torch.linspace(0, 45, steps=3)
    physics = dinv.physics.Tomography(angles=angles, img_width=4, circle=True, normalize=True, device=dev)
    physics.operator_norm.fill_(1.0)  # compare the un-normalised projector against the oracle below
    ref = O.radon_forward(x, angles, circle=True)
    assert torch.allclose(physics.A(x.to(dev)).cpu(), ref, atol=1e-5)


@pytest.mark.parametrize("N,A", [(23, 16), (91, 45), (725, 32)])
def test_ramp_filter(dev, N, A):
    from deepinv_amd.hip import radon as hr

    y = torch.randn(2, 1, N, A, generator=torch.Generator().manual_seed(1))
    out = hr.ramp_filter(y.to(dev))
    assert rel_err(out, O.ramp_filter(y)) < TOL


def test_fbp_and_normalisation(dev):
    import deepinv_amd as dinv

    W, nang = 64, 90
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 1, W, W, generator=g)
    torch.manual_seed(0)
    phys = dinv.physics.Tomography(angles=nang, img_width=W, normalize=True, device=dev)
    # ||A^T A|| ~ 1 after normalisation (reference test_physics.py:1460-1466)
    sq = phys.compute_sqnorm(torch.randn(1, 1, W, W, device=dev), tol=1e-4, verbose=False)
    assert abs(float(sq) - 1.0) < 1e-2
    nrm = phys.operator_norm.cpu()
    ang = phys.angles.cpu()
    y_ref = O.radon_forward(x, ang) / nrm
    y = phys.A(x.to(dev))
    assert rel_err(y, y_ref) < TOL
    rec = phys.A_dagger(y, fbp=True)
    assert rel_err(rec, O.tomography_fbp(y_ref, ang, W, operator_norm=nrm)) < TOL
    # FBP is a decent inverse (reference tolerance 65 %: test_physics.py:1468-1477)
    assert rel_err(rec, x) < 0.65
    # CG pseudo-inverse through the kernels (5 % in the reference)
    phys2 = dinv.physics.Tomography(angles=nang, img_width=W, normalize=True, device=dev, max_iter=100, tol=1e-5)
    r = phys2.A_dagger(phys2.A(x.to(dev)))
    assert rel_err(r, x) < 0.15


def test_radon_autograd(dev):
    import deepinv_amd as dinv

    phys = dinv.physics.Tomography(angles=12, img_width=16, normalize=False, device=dev)
    x = torch.rand(1, 1, 16, 16, device=dev, requires_grad=True)
    y = phys.A(x)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    assert rel_err(x.grad, phys.A_adjoint(w)) < 1e-6


@pytest.mark.parametrize("W,nang,circle", [(16, 12, False), (16, 12, True), (48, 30, False)])
def test_iradon_branch_adjoint_via_backprop_false(dev, W, nang, circle):
    """B3: interpolating (inexact) back-projection, ApplyRadon pair (radon.py:360-531, tomography.py:251,275-288,344-348)"""
    import deepinv_amd as dinv

    g = torch.Generator().manual_seed(5)
    phys = dinv.physics.Tomography(angles=nang, img_width=W, circle=circle, normalize=False, adjoint_via_backprop=False,
                                   device=dev)
    ang = phys.angles.cpu()
    x = torch.rand(2, 1, W, W, generator=g)
    y_ref = O.radon_forward(x, ang, circle)
    assert rel_err(phys.A(x.to(dev)), y_ref) < TOL
    v = torch.randn(y_ref.shape, generator=g)
    assert rel_err(phys.A_adjoint(v.to(dev)), O.iradon_backproject(v, ang, W, circle)) < TOL
    fbp_ref = O.iradon_backproject(O.ramp_filter(y_ref), ang, W, circle) * torch.pi / (2 * nang)
    assert rel_err(phys.A_dagger(y_ref.to(dev), fbp=True), fbp_ref) < TOL
    # autograd of A uses the interpolating back-projection (ApplyRadon.backward)
    xg = x.to(dev).requires_grad_(True)
    (phys.A(xg) * v.to(dev)).sum().backward()
    assert rel_err(xg.grad, O.iradon_backproject(v, ang, W, circle)) < TOL


@pytest.mark.parametrize("W,angles,B", [(97, [0., 44.9, 45., 45.1, 89.9, 90., 90.1, 134.9, 135., 135.1, 179.9], 5),
                                        (50, [-30., 200., 359., 720.5, 17., 93., -91.], 1), (192, 120, 8)])
def test_tiled_and_gather_kernels_agree(dev, W, angles, B, monkeypatch):
    """The LDS-tiled kernels (default) against the round-1 gather kernels (hip.radon.ENABLE_TILED = False): same samples, same
    weights - only the summation order may differ; class borders, arbitrary angle lists, a full 8-image group."""
    import deepinv_amd as dinv

    ang = angles if isinstance(angles, int) else torch.tensor(angles)
    g = torch.Generator().manual_seed(W)
    x = torch.rand(B, 1, W, W, generator=g).to(dev)
    phys = dinv.physics.Tomography(angles=ang, img_width=W, normalize=False, device=dev)
    y = phys.A(x)
    v = torch.randn(y.shape, generator=g).to(dev)
    xa = phys.A_adjoint(v)
    assert dot_test(phys, x, y) < 1e-5
    from deepinv_amd.hip import radon as HR
    monkeypatch.setattr(HR, "ENABLE_TILED", False)
    assert rel_err(y, phys.A(x)) < 1e-5
    assert rel_err(xa, phys.A_adjoint(v)) < 1e-5
    monkeypatch.setattr(HR, "ENABLE_RAMP_FFT", False)
    r_direct = phys.filter(y)
    monkeypatch.setattr(HR, "ENABLE_RAMP_FFT", True)
    assert rel_err(phys.filter(y), r_direct) < 1e-5


def test_angles_update_rebuilds_geometry(dev):
    """new angle VALUES with the same count must not reuse the old tables"""
    import deepinv_amd as dinv

    W = 32
    x = torch.rand(1, 1, W, W, generator=torch.Generator().manual_seed(1))
    a1, a2 = torch.linspace(0, 90, 10), torch.linspace(5, 170, 10)
    phys = dinv.physics.Tomography(angles=a1, img_width=W, normalize=False, device=dev)
    assert rel_err(phys.A(x.to(dev)), O.radon_forward(x, a1)) < TOL
    phys.update_parameters(angles=a2.to(dev))
    assert rel_err(phys.A(x.to(dev)), O.radon_forward(x, a2)) < TOL
    phys.angles.copy_(a1.to(dev))          # in-place edit: caught by the version counter
    assert rel_err(phys.A(x.to(dev)), O.radon_forward(x, a1)) < TOL


FAN_B = {"pixel_spacing": 0.02, "source_radius": 4.0, "detector_radius": 3.0, "n_detector_pixels": 96, "detector_spacing": 0.05}


@pytest.mark.parametrize("W,angles,circle,B,C,fan", [
    (32, 24, False, 3, 1, FAN_B), (48, 36, True, 2, 2, FAN_B), (64, 30, False, 9, 1, None),
    (40, 17, False, 1, 1, {"pixel_spacing": 0.05, "source_radius": 8.0, "detector_radius": 2.0, "n_detector_pixels": 200,
                           "detector_spacing": 0.02})])
def test_fan_beam_forward_adjoint_fbp(dev, W, angles, circle, B, C, fan):
    """Tomography(fan_beam=True) vs the CPU oracle (fan_beam_grid, radon.py:16-52): forward, exact adjoint (dot test),
    normalisation by the power method, FBP = RampFilter + adjoint (tomography.py:270-279), autograd"""
    import deepinv_amd as dinv

    g = torch.Generator().manual_seed(W)
    x = torch.rand(B, C, W, W, generator=g)
    phys = dinv.physics.Tomography(angles=angles, img_width=W, circle=circle, normalize=False, fan_beam=True,
                                   fan_parameters=fan, device=dev)
    ang = phys.angles.cpu()
    y = phys.A(x.to(dev))
    y_ref = O.radon_fan_forward(x, ang, fan, circle)
    assert y.shape == y_ref.shape and rel_err(y, y_ref) < TOL
    v = torch.randn(y_ref.shape, generator=g)
    assert rel_err(phys.A_adjoint(v.to(dev)), O.radon_fan_adjoint(v, ang, W, fan, circle)) < TOL
    assert dot_test(phys, x.to(dev), y) < 1e-5
    assert rel_err(phys.A_dagger(y, fbp=True), O.tomography_fan_fbp(y_ref, ang, W, fan, None, circle)) < TOL
    xg = x.to(dev).requires_grad_(True)
    (phys.A(xg) * v.to(dev)).sum().backward()
    assert rel_err(xg.grad, phys.A_adjoint(v.to(dev))) < 1e-6
    if fan is FAN_B and not circle:
        pn = dinv.physics.Tomography(angles=angles, img_width=W, circle=circle, normalize=True, fan_beam=True,
                                     fan_parameters=fan, device=dev)
        nrm = float(pn.operator_norm)
        assert rel_err(pn.A(x.to(dev)) * nrm, y) < 1e-5
        assert 0.9 < float(pn.compute_norm(torch.randn(1, 1, W, W, device=dev), squared=False, verbose=False)) < 1.1
        assert rel_err(pn.A_dagger(pn.A(x.to(dev)), fbp=True), phys.A_dagger(y, fbp=True)) < 1e-4
