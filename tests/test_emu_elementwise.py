"""Loop-algebra kernels (deepinv_amd/csrc/elementwise.hip) on the host emulation: the CG driver of
deepinv_amd/optim/linear.py with its device-side convergence flag against the oracle's CG
(reference deepinv/optim/linear/conjugate_gradient.py:48-75)."""
import ctypes

import pytest
import torch

import emu_lib as E
from oracle import optim_cpu as OO


class EmuEw:
    """deepinv_amd.hip.elementwise with the kernels running in the fiber emulation on CPU tensors"""

    def __init__(self):
        self.l = E.lib()
        self.l.dinv_batched_dot_blocks.restype = ctypes.c_int32

    def lincomb(self, a, x, b=0.0, y=None, c=0.0, z=None):
        out = torch.empty_like(x)
        E.check(self.l.dinv_lincomb(ctypes.c_int64(x.numel()), ctypes.c_float(a), E.p(x), ctypes.c_float(b), E.p(y),
                                    ctypes.c_float(c), E.p(z), E.p(out), None))
        return out

    def batched_dot(self, x, y):
        B, n = x.shape[0], x.numel() // x.shape[0]
        out = torch.empty(B)
        part = torch.empty(B * self.l.dinv_batched_dot_blocks(ctypes.c_int64(n)))
        E.check(self.l.dinv_batched_dot(B, ctypes.c_int64(n), E.p(x), E.p(y), E.p(out), E.p(part), None))
        return out

    def cg_update_xr(self, num, den, eps, x, r, p, Ap, done=None):
        B = x.shape[0]
        E.check(self.l.dinv_cg_update_masked(0, B, ctypes.c_int64(x.numel() // B), E.p(num), E.p(den), ctypes.c_float(eps),
                                             E.p(x), E.p(r), E.p(p), E.p(Ap), E.p(done), None))

    def cg_update_p(self, num, den, eps, p, r, done=None):
        B = p.shape[0]
        E.check(self.l.dinv_cg_update_masked(1, B, ctypes.c_int64(p.numel() // B), E.p(num), E.p(den), ctypes.c_float(eps),
                                             E.p(p), None, E.p(r), None, E.p(done), None))

    def cg_check(self, res, tol2, done):
        E.check(self.l.dinv_cg_check(res.shape[0], E.p(res), E.p(tol2), E.p(done), None))


@pytest.mark.parametrize("n", [1, 7, 1024, 4099])
def test_affine_clamp(n):
    """dinv_affine (DiffPIR's updates, diffusion.py:463-507, one launch each): clamp(a x + b y + c z + d, lo, hi) with optional
    operands and an n that is not a multiple of 4; dinv_lincomb is the same kernel without constant and clamp"""
    l = E.lib()
    g = torch.Generator().manual_seed(n)
    x, y, z = (torch.randn(n, generator=g) for _ in range(3))
    f, i64 = ctypes.c_float, ctypes.c_int64
    inf = float("inf")
    out = torch.empty(n)
    E.check(l.dinv_affine(i64(n), f(0.5), E.p(x), f(-1.25), E.p(y), f(2.0), E.p(z), f(0.75), f(-inf), f(inf), E.p(out), None))
    assert torch.allclose(out, 0.5 * x - 1.25 * y + 2.0 * z + 0.75, atol=1e-6)
    E.check(l.dinv_affine(i64(n), f(1.0), E.p(x), f(0.0), None, f(0.0), None, f(0.0), f(0.0), f(1.0), E.p(out), None))
    assert torch.equal(out, x.clamp(0, 1))
    E.check(l.dinv_affine(i64(n), f(2.0), E.p(x), f(3.0), E.p(y), f(0.0), None, f(-1.0), f(-1.0), f(inf), E.p(out), None))
    assert torch.allclose(out, (2 * x + 3 * y - 1).clamp(min=-1), atol=1e-6)
    assert l.dinv_affine(i64(n), f(1.0), E.p(x), f(0.0), None, f(0.0), None, f(0.0), f(1.0), f(0.0), E.p(out), None) != 0   # lo > hi
    assert torch.allclose(EmuEw().lincomb(0.5, x, -1.25, y, 2.0, z), 0.5 * x - 1.25 * y + 2.0 * z, atol=1e-6)


def test_lincomb_and_affine_propagate_nan():
    """torch.clamp / plain arithmetic in the reference propagate NaN (a diverged iterate stays visibly diverged): so must the
    fused kernel, although fminf / fmaxf alone would return the non-NaN bound"""
    l = E.lib()
    f, i64 = ctypes.c_float, ctypes.c_int64
    x = torch.tensor([0.5, float("nan"), -2.0, float("nan"), 3.0, float("nan"), 0.25])
    y = torch.ones(7)
    out = torch.empty(7)
    o = EmuEw().lincomb(1.0, x, -1.0, y)
    assert torch.equal(torch.isnan(o), torch.isnan(x)) and torch.equal(o[~torch.isnan(o)], (x - y)[~torch.isnan(x)])
    E.check(l.dinv_affine(i64(7), f(1.0), E.p(x), f(0.0), None, f(0.0), None, f(0.0), f(0.0), f(1.0), E.p(out), None))
    ref = x.clamp(0, 1)
    assert torch.equal(torch.isnan(out), torch.isnan(ref)) and torch.equal(out[~torch.isnan(out)], ref[~torch.isnan(ref)])


@pytest.mark.parametrize("shape,dshape", [((2, 3, 8, 5), (1, 1, 8, 5)), ((4, 3, 6, 4), (1, 3, 6, 4)), ((1, 1, 64, 33), (1, 1, 64, 33))])
def test_cdiv_real_is_the_broadcast_complex_division(shape, dshape):
    """dinv_cdiv_real: a complex spectrum over (real symbol + 1/gamma), the symbol shared by the leading dimensions - the division of
    the closed-form proxes (blur.py:331-363), against torch's broadcast arithmetic; in place as well"""
    l = E.lib()
    g = torch.Generator().manual_seed(3)
    s = torch.randn(*shape, 2, generator=g)
    d = torch.rand(*dshape, generator=g) + 0.1
    out = torch.empty_like(s)
    E.check(l.dinv_cdiv_real(ctypes.c_int64(s.numel() // 2), ctypes.c_int64(d.numel()), E.p(s), E.p(d), ctypes.c_float(0.25), E.p(out), None))
    ref = torch.view_as_real(torch.view_as_complex(s) / (d + 0.25))
    assert torch.allclose(out, ref, rtol=2e-7, atol=0)
    E.check(l.dinv_cdiv_real(ctypes.c_int64(s.numel() // 2), ctypes.c_int64(d.numel()), E.p(s), E.p(d), ctypes.c_float(0.25), E.p(s), None))
    assert torch.equal(s, out)
    assert l.dinv_cdiv_real(ctypes.c_int64(4), ctypes.c_int64(0), E.p(s), E.p(d), ctypes.c_float(0.0), E.p(out), None) != 0


def test_mask_solve_is_the_decomposable_prox_and_pseudo_inverse():
    """dinv_mask_solve against the reference's tensor expressions (forward.py:1223-1234: V^T b / (conj(m) m + 1/gamma);
    forward.py:1247-1252: U^T y * where(m > 1e-5, 1/m, 0)), bit for bit"""
    l = E.lib()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 2, 8, 6, generator=g)
    m = (torch.rand(1, 2, 8, 6, generator=g) > 0.4).float() * (0.5 + torch.rand(1, 2, 8, 6, generator=g))
    m.view(-1)[5] = 5e-6                                   # below the pseudo-inverse's threshold
    gamma = 0.37
    out = torch.empty_like(x)
    E.check(l.dinv_mask_solve(0, ctypes.c_int64(x.numel()), ctypes.c_int64(m.numel()), E.p(x), E.p(m), ctypes.c_float(1 / gamma), E.p(out), None))
    assert torch.equal(out, x / (torch.conj(m) * m + 1 / gamma))
    E.check(l.dinv_mask_solve(1, ctypes.c_int64(x.numel()), ctypes.c_int64(m.numel()), E.p(x), E.p(m), ctypes.c_float(0.0), E.p(out), None))
    assert torch.equal(out, x * torch.where(m > 1e-5, m.reciprocal(), 0.0))
    assert l.dinv_mask_solve(2, ctypes.c_int64(4), ctypes.c_int64(4), E.p(x), E.p(m), ctypes.c_float(0.0), E.p(out), None) != 0


@pytest.mark.parametrize("check_every", [1, 4, 1000])
def test_cg_with_device_side_convergence_matches_reference_cg(check_every, monkeypatch):
    """however rarely the host looks at the flag (every iteration, every 4th, never before max_iter), the iterate is
    the one the reference's `break` leaves: updates issued after convergence are no-ops"""
    from deepinv_amd.optim import linear

    monkeypatch.setattr(linear, "CG_CHECK_EVERY", check_every)
    g = torch.Generator().manual_seed(0)
    M = torch.randn(3, 24, 24, generator=g)
    H = lambda v: torch.einsum("bij,bj->bi", M.transpose(1, 2) @ M + 0.5 * torch.eye(24), v.reshape(3, 24)).reshape(v.shape)
    b = torch.randn(3, 2, 12, generator=g)
    calls = {"n": 0}

    def Hc(v):
        calls["n"] += 1
        return H(v)

    x = linear._conjugate_gradient_hip(Hc, b, 60, 1e-5, 1e-8, None, False, ew=EmuEw())
    ref = OO.conjugate_gradient(H, b, max_iter=60, tol=1e-5)
    assert float((x - ref).norm() / ref.norm()) < 1e-5
    if check_every == 1:
        n1 = calls["n"]
        assert n1 < 40      # really stopped early
