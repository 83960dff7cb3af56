"""Loop-algebra kernels (csrc/elementwise.hip) and the CG fast path vs plain torch / the CPU oracle."""
import pytest
import torch

from conftest import rel_err
from oracle import optim_cpu as OO
from oracle import physics_cpu as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(3, 2, 17, 19), (2, 1, 64, 64), (1, 7), (4, 2, 320, 320)])
def test_lincomb_and_dot(dev, shape):
    from deepinv_amd.hip import elementwise as ew

    g = torch.Generator().manual_seed(0)
    x, y, z = (torch.randn(*shape, generator=g).to(dev) for _ in range(3))
    assert rel_err(ew.lincomb(0.5, x, -1.25, y, 2.0, z), 0.5 * x - 1.25 * y + 2.0 * z) < 1e-6
    assert rel_err(ew.lincomb(1.0, x, -1.0, y), x - y) < 1e-6
    n = x[0].numel()
    if n % 4 == 0:
        d = ew.batched_dot(x, y)
        ref = (x.double() * y.double()).flatten(1).sum(1)
        assert rel_err(d, ref) < 1e-5
        assert torch.equal(d, ew.batched_dot(x, y))  # deterministic


@pytest.mark.parametrize("shape", [(3, 2, 17, 19), (1, 7), (16, 3, 256, 256)])
def test_affine_clamp(dev, shape):
    """dinv_affine: DiffPIR's affine updates (diffusion.py:463-507) in one launch each"""
    from deepinv_amd.hip import elementwise as ew

    g = torch.Generator().manual_seed(1)
    x, y, z = (torch.randn(*shape, generator=g).to(dev) for _ in range(3))
    assert rel_err(ew.affine(0.5, x, -1.25, y, 2.0, z, d=0.75), 0.5 * x - 1.25 * y + 2.0 * z + 0.75) < 1e-6
    assert torch.equal(ew.affine(1.0, x, lo=0.0, hi=1.0), x.clamp(0, 1))
    assert torch.equal(ew.affine(1.0, x, lo=0.0, hi=1.0), ((2 * x - 1).clamp(-1, 1) / 2 + 0.5)) or \
        rel_err(ew.affine(1.0, x, lo=0.0, hi=1.0), (2 * x - 1).clamp(-1, 1) / 2 + 0.5) < 1e-6
    assert rel_err(ew.affine(0.3, x, d=0.5), x * 0.3 + 0.5) < 1e-6


def test_lincomb_and_affine_propagate_nan(dev):
    """NaN in -> NaN out, as torch.clamp and the reference's plain arithmetic do (fminf / fmaxf drop NaN)"""
    from deepinv_amd.hip import elementwise as ew

    x = torch.randn(4, 2, 33, 35, generator=torch.Generator().manual_seed(2))
    x.view(-1)[::97] = float("nan")
    x = x.to(dev)
    y = torch.ones_like(x)
    o = ew.lincomb(1.0, x, -1.0, y)
    assert torch.equal(torch.isnan(o), torch.isnan(x)) and torch.equal(o[~torch.isnan(o)], (x - y)[~torch.isnan(x)])
    o = ew.affine(1.0, x, lo=0.0, hi=1.0)
    ref = x.clamp(0, 1)
    assert torch.equal(torch.isnan(o), torch.isnan(ref)) and torch.equal(o[~torch.isnan(o)], ref[~torch.isnan(ref)])


def test_cg_fast_path_matches_oracle_cg(dev):
    import deepinv_amd as dinv

    g = torch.Generator().manual_seed(1)
    img, coils, B = (32, 32), 4, 3
    x = torch.rand(B, 2, *img, generator=g)
    maps = torch.randn(1, coils, *img, dtype=torch.complex64, generator=g) / 2
    mask = (torch.rand(*img, generator=g) > 0.5).float()
    phys = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, *img), device=dev, max_iter=40, tol=1e-6)
    A = lambda v: O.multicoil_A(v, maps, mask)
    AT = lambda v: O.multicoil_AT(v, maps, mask)
    y = A(x)
    z = torch.rand(B, 2, *img, generator=g)
    out = phys.prox_l2(z.to(dev), y.to(dev), 0.9)
    ref = OO.prox_l2_cg(z, y, 0.9, A, AT, max_iter=40, tol=1e-6)
    assert rel_err(out, ref) < 1e-4
    # pseudo-inverse through the same path
    xd = phys.A_dagger(y.to(dev))
    assert torch.isfinite(xd).all()


@pytest.mark.gpu
def test_cdiv_real_gpu(dev):
    """dinv_cdiv_real (the division of Downsampling.prox_l2: half spectrum over aliased symbol + 1/gamma) against torch's broadcast
    complex division, and its argument check"""
    from deepinv_amd.hip import elementwise as EW

    g = torch.Generator().manual_seed(5)
    s = torch.randn(16, 3, 64, 33, 2, generator=g).to(dev)
    d = (torch.rand(1, 1, 64, 33, generator=g) + 0.05).to(dev)
    sc = torch.view_as_complex(s)
    out = EW.cdiv_real(sc, d, 1.0 / 7e5)
    ref = sc / (d + 1.0 / 7e5)
    assert out.shape == ref.shape and float((out - ref).abs().max() / ref.abs().max()) < 3e-7
    with pytest.raises(ValueError):
        EW.cdiv_real(sc, d[..., :32].contiguous())


def test_mask_solve_gpu(dev):
    """dinv_mask_solve against the reference's tensor expressions of DecomposablePhysics.prox_l2 / A_dagger (forward.py:1223-1252):
    bit for bit against their CPU evaluation, mask shared by the batch and mask per sample"""
    from deepinv_amd.hip import elementwise as EW

    g = torch.Generator().manual_seed(6)
    x = torch.randn(4, 2, 64, 48, generator=g).to(dev)
    for mshape in ((1, 2, 64, 48), (4, 2, 64, 48)):
        m = ((torch.rand(*mshape, generator=g) > 0.4).float() * (0.5 + torch.rand(*mshape, generator=g))).to(dev)
        m.view(-1)[7] = 5e-6
        # (the host twin of this test, tests/test_emu_elementwise.py, is bit-exact against ATen's CPU kernels - the reference's
        # path; PyTorch-ROCm's device division differs from the correctly rounded one in the last bit)
        assert torch.allclose(EW.mask_solve(x, m, 1 / 0.37), x / (torch.conj(m) * m + 1 / 0.37), rtol=3e-7, atol=0)
        assert torch.allclose(EW.mask_solve(x, m, dagger=True), x * torch.where(m > 1e-5, m.reciprocal(), 0.0), rtol=3e-7, atol=0)
        assert torch.equal(EW.mask_solve(x, m, 1 / 0.37).cpu(), x.cpu() / (m.cpu() * m.cpu() + 1 / 0.37))
    with pytest.raises(ValueError):
        EW.mask_solve(x, torch.ones(1, 2, 64, 47, device=dev))
