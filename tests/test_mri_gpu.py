"""GPU parity of the HIP MRI path against the CPU oracle (tolerance 1e-4 relative fp32, the
north_star bound; measured errors are ~1e-6), exact zero pattern, dot test <= 1e-5."""
import pytest
import torch

from conftest import dot_test, rel_err
from oracle import physics_cpu as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _g(seed=0):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("shape,dims", [((3, 17, 11), (-2, -1)), ((2, 16, 8), (-2, -1)), ((2, 5, 17, 11), (-3, -2, -1)),
                                        ((1, 320, 320), (-2, -1)), ((2, 7, 19), (-1,)), ((2, 29, 6), (-2,))])
def test_centered_fft_matches_oracle(dev, shape, dims):
    import deepinv_amd as dinv

    x = torch.randn(*shape, dtype=torch.complex64, generator=_g())
    for fn_o, fn in ((O.cfft, dinv.physics.MRIMixin.fft), (O.cifft, dinv.physics.MRIMixin.ifft)):
        ref = fn_o(x, dims)
        out = fn(x.to(dev), dim=dims).cpu()
        assert rel_err(torch.view_as_real(out), torch.view_as_real(ref)) < TOL


@pytest.mark.parametrize("img,three_d", [((17, 11), False), ((16, 8), False), ((5, 17, 11), True), ((320, 320), False),
                                         ((4, 32, 20), True), ((64, 256), False), ((16, 128, 64), True)])
@pytest.mark.parametrize("batched_mask", [False, True])
def test_single_coil_mri(dev, img, three_d, batched_mask):
    import deepinv_amd as dinv

    B = 3
    g = _g(1)
    x = torch.randn(B, 2, *img, generator=g)
    mshape = (B, 1, *img) if batched_mask else img
    mask = (torch.rand(*mshape, generator=g) > 0.6).float()
    phys = dinv.physics.MRI(mask=mask, img_size=(2, *img), three_d=three_d, device=dev)
    y = phys.A(x.to(dev))
    y_ref = O.mri_A(x, mask, three_d)
    assert y.shape == y_ref.shape
    assert rel_err(y, y_ref) < TOL
    # bit-exact zero pattern (reference test_physics.py:1052-1077)
    full_mask = O.check_mask(mask, three_d).expand_as(y_ref)
    assert torch.equal(y.cpu() == 0, full_mask == 0)
    xa = phys.A_adjoint(y)
    assert rel_err(xa, O.mri_AT(y_ref, mask, three_d)) < TOL
    assert dot_test(phys, x.to(dev), y) < 1e-5
    # closed forms of DecomposablePhysics
    z = torch.randn(B, 2, *img, generator=g)
    p = phys.prox_l2(z.to(dev), y, 0.7)
    assert rel_err(p, O.mri_prox_l2(z, y_ref, 0.7, mask, three_d)) < TOL
    assert rel_err(phys.A_dagger(y), O.mri_dagger(y_ref, mask, three_d)) < TOL


@pytest.mark.parametrize("img,three_d,coils", [((17, 11), False, 7), ((5, 17, 11), True, 15), ((320, 320), False, 8),
                                               ((16, 64, 48), True, 12), ((64, 64), False, 1),
                                               # static-plan pipeline (every dim in {16,32,64,128,256,320,512})
                                               ((16, 64, 128), True, 3), ((32, 256), False, 2), ((128, 64), False, 5),
                                               ((512, 320), False, 1), ((16, 32, 64), True, 2)])
@pytest.mark.parametrize("batched", [False, True])
def test_multicoil_mri(dev, img, three_d, coils, batched):
    import deepinv_amd as dinv

    B = 2
    g = _g(2)
    x = torch.randn(B, 2, *img, generator=g)
    mb = B if batched else 1
    maps = torch.randn(mb, coils, *img, dtype=torch.complex64, generator=g) / coils ** 0.5
    mask = (torch.rand(mb, 1, *img, generator=g) > 0.5).float()
    phys = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, *img), three_d=three_d, device=dev)
    y = phys.A(x.to(dev))
    y_ref = O.multicoil_A(x, maps, mask, three_d)
    assert y.shape == y_ref.shape == (B, 2, coils, *img)
    assert rel_err(y, y_ref) < TOL
    assert torch.all(y.cpu()[O.check_mask(mask, three_d)[:, :, None].expand_as(y_ref) == 0] == 0)
    xa = phys.A_adjoint(y)
    assert rel_err(xa, O.multicoil_AT(y_ref, maps, mask, three_d)) < TOL
    assert dot_test(phys, x.to(dev), y) < 1e-5
    xr = phys.A_adjoint(y, rss=True)
    assert rel_err(xr, O.multicoil_AT_rss(y_ref, mask, three_d)) < TOL


@pytest.mark.parametrize("img,coils,B", [((320, 320), 8, 3), ((256, 256), 4, 2), ((320, 256), 2, 2), ((256, 512), 3, 1), ((512, 320), 2, 1)])
def test_wave_pipelines_against_oracle_and_cooperative_pipelines(dev, img, coils, B, monkeypatch):
    """The two families of 2-D pipelines behind dinv_mri_forward / _adjoint / _normal - wave-autonomous (csrc/mri_wave.hpp:
    rows pass with the radix-R column stage as its epilogue + 64-point column pass) and workgroup-cooperative (csrc/mri.hip) -
    on the same inputs: each against the CPU oracle (deepinv/physics/mri.py:254-324), the dot test, A^T A against A^T(A x),
    and against each other.  (The library picks the wave family for A always, for A^T / A^T A from 16 slices up; the test
    hook forces it.)"""
    import deepinv_amd as dinv
    from deepinv_amd.hip import mri as M

    g = _g(7)
    x = torch.randn(B, 2, *img, generator=g)
    maps = torch.randn(1, coils, *img, dtype=torch.complex64, generator=g) / coils ** 0.5
    mask = (torch.rand(1, 1, *img, generator=g) > 0.6).float()
    v = torch.randn(B, 2, coils, *img, generator=g)
    y_ref = O.multicoil_A(x, maps, mask)
    xa_ref = O.multicoil_AT(v, maps, mask)
    xn_ref = O.multicoil_AT(y_ref, maps, mask)
    outs = {}
    for force in (True, False):
        monkeypatch.setattr(M, "FORCE_WAVE_PIPELINES", force)
        phys = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, *img), device=dev)
        y = phys.A(x.to(dev))
        assert rel_err(y, y_ref) < TOL
        assert torch.all(y.cpu()[O.check_mask(mask)[:, :, None].expand_as(y_ref) == 0] == 0)      # exact zeros under the mask
        xa = phys.A_adjoint(v.to(dev))
        assert rel_err(xa, xa_ref) < TOL
        xn = phys.A_adjoint_A(x.to(dev))
        assert rel_err(xn, xn_ref) < TOL
        assert dot_test(phys, x.to(dev), y) < 1e-5
        outs[force] = (y.cpu(), xa.cpu(), xn.cpu())
    for a, b in zip(outs[True], outs[False]):
        assert rel_err(a, b) < 1e-5


def test_mask_update_persists_and_autograd(dev):
    import deepinv_amd as dinv

    g = _g(3)
    x = torch.randn(2, 2, 16, 12, generator=g).to(dev).requires_grad_(True)
    phys = dinv.physics.MRI(img_size=(2, 16, 12), device=dev)
    m2 = (torch.rand(16, 12, generator=g) > 0.5).float()
    y = phys.A(x, mask=m2.to(dev))
    assert torch.equal(phys.mask.cpu(), O.check_mask(m2))  # "the new mask is stored" (mri.py:28)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    assert rel_err(x.grad, O.mri_AT(w.cpu(), m2)) < TOL  # backward(A) = A_adjoint


def test_cpu_tensor_fails_loudly():
    import deepinv_amd as dinv

    phys = dinv.physics.MRI(img_size=(2, 8, 8))
    with pytest.raises(RuntimeError):
        phys.A(torch.randn(1, 2, 8, 8))


@pytest.mark.parametrize("mb,sb", [(1, 1), (3, 3), (1, 3)])
def test_parameter_gradients_match_autograd_oracle(dev, mb, sb):
    """dL/d(coil_maps), dL/d(mask), dL/dx of A and A_adjoint against autograd through the CPU oracle (the reference
    gets these from autograd; needed e.g. for learned sampling patterns).  Shared (batch 1) and per-sample params."""
    from deepinv_amd.hip.mri import mri_adjoint, mri_forward
    from oracle import physics_cpu as OP

    g = torch.Generator().manual_seed(5)
    B, N, H, W = 3, 4, 16, 12
    x = torch.randn(B, 2, H, W, generator=g)
    maps = torch.randn(sb, N, H, W, dtype=torch.complex64, generator=g) / 2
    mask = torch.rand(mb, 2, H, W, generator=g)
    wy = torch.randn(B, 2, N, H, W, generator=g)
    wx = torch.randn(B, 2, H, W, generator=g)
    yin = torch.randn(B, 2, N, H, W, generator=g)

    def grads(fwd, adj, to):
        xs, ms, ks, ys = (t.clone().to(to).requires_grad_() for t in (x, maps, mask, yin))
        l1 = (fwd(xs, ms, ks) * wy.to(to)).sum()
        g1 = torch.autograd.grad(l1, (xs, ms, ks))
        l2 = (adj(ys, ms, ks) * wx.to(to)).sum()
        g2 = torch.autograd.grad(l2, (ys, ms, ks))
        return [t.cpu() for t in (*g1, *g2)]

    ref = grads(lambda a, b, c: OP.multicoil_A(a, b, c), lambda a, b, c: OP.multicoil_AT(a, b, c), "cpu")
    got = grads(lambda a, b, c: mri_forward(a, b, c), lambda a, b, c: mri_adjoint(a, b, c), dev)
    for name, r, o in zip(("A/x", "A/maps", "A/mask", "AT/y", "AT/maps", "AT/mask"), ref, got):
        assert o.shape == r.shape, name
        real = lambda t: torch.view_as_real(t.resolve_conj()) if t.is_complex() else t
        assert rel_err(real(o), real(r)) < 1e-4, name


@pytest.mark.parametrize("img,three_d,coils", [((32, 48), False, 5), ((8, 16, 32), True, 4)])
def test_coil_parallel_partition_sums_to_full_operator(dev, img, three_d, coils):
    """coil_parallel_mri on two emulated ranks of ONE device (no process group: reduce_op=None and the partial images
    added by hand) == the full MultiCoilMRI: the coil slabs tile the coil axis and A^T / A^T A are sums over coils."""
    import deepinv_amd as dinv
    from deepinv_amd.distributed import BatchParallelContext, coil_parallel_mri

    g = _g(9)
    x = torch.randn(2, 2, *img, generator=g).to(dev)
    maps = (torch.randn(1, coils, *img, dtype=torch.complex64, generator=g) / coils ** 0.5).to(dev)
    mask = (torch.rand(1, 1, *img, generator=g) > 0.5).float().to(dev)
    full = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, *img), three_d=three_d, device=dev)
    y_full = full.A(x)
    ranks = []
    for r in range(2):
        ctx = BatchParallelContext(device=dev)
        ctx.world_size, ctx.rank = 2, r
        ranks.append(coil_parallel_mri(ctx, mask, maps, (2, *img), three_d=three_d))
    ys = [p.A(x, gather=False)[0] for p in ranks]
    assert rel_err(torch.cat(ys, dim=2), y_full) < 1e-6                      # coil slabs, in order
    aty = sum(p.A_adjoint(y, reduce_op=None) for p, y in zip(ranks, [[y] for y in ys]))
    assert rel_err(aty, full.A_adjoint(y_full)) < 1e-5
    ata = sum(p.A_adjoint_A(x, reduce_op=None) for p in ranks)
    assert rel_err(ata, full.A_adjoint_A(x)) < 1e-5


@pytest.mark.parametrize("img,three_d,coils", [((320, 320), False, 8), ((64, 128), False, 3), ((128, 64), False, 1),
                                               ((256, 512), False, 2), ((512, 64), False, 2), ((32, 64), False, 2),
                                               ((16, 64, 128), True, 3), ((16, 32, 64), True, 2), ((32, 128, 64), True, 1),
                                               ((17, 64), False, 2)])     # last: no static plan -> composite fallback
@pytest.mark.parametrize("batched", [False, True])
def test_multicoil_normal_operator(dev, img, three_d, coils, batched):
    """fused A^T A (dinv_mri_normal: k-space never written) == oracle A^T(A x) and == the two-kernel-chain composite"""
    import deepinv_amd as dinv

    B = 2
    g = _g(11)
    x = torch.randn(B, 2, *img, generator=g)
    mb = B if batched else 1
    maps = torch.randn(mb, coils, *img, dtype=torch.complex64, generator=g) / coils ** 0.5
    mask = torch.rand(mb, 1, *img, generator=g)
    mask = torch.where(mask > 0.5, mask, torch.zeros_like(mask))      # non-binary weights: M^2 != M
    phys = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, *img), three_d=three_d, device=dev)
    out = phys.A_adjoint_A(x.to(dev))
    ref = O.multicoil_AT(O.multicoil_A(x, maps, mask, three_d), maps, mask, three_d)
    assert rel_err(out, ref) < TOL
    assert rel_err(out, phys.A_adjoint(phys.A(x.to(dev)))) < 1e-5
    # symmetric: <A^T A u, v> == <u, A^T A v>
    v = torch.randn(B, 2, *img, generator=g).to(dev)
    a, b = (out * v).sum().item(), (x.to(dev) * phys.A_adjoint_A(v)).sum().item()
    assert abs(a - b) <= 1e-4 * max(abs(a), abs(b), 1.0)


def test_single_coil_normal_operator_and_autograd(dev):
    import deepinv_amd as dinv

    g = _g(12)
    img = (64, 128)
    mask = (torch.rand(*img, generator=g) > 0.6).float()
    phys = dinv.physics.MRI(mask=mask, img_size=(2, *img), device=dev)
    x = torch.randn(3, 2, *img, generator=g).to(dev).requires_grad_(True)
    out = phys.A_adjoint_A(x)
    assert rel_err(out.detach(), O.mri_AT(O.mri_A(x.detach().cpu(), mask), mask)) < TOL
    w = torch.randn_like(out)
    (out * w).sum().backward()
    assert rel_err(x.grad, phys.A_adjoint_A(w).detach()) < 1e-6        # self-adjoint: backward is the same chain
