"""Linear solvers behind ``LinearPhysics.prox_l2`` / ``A_dagger``
(reference deepinv/optim/linear/{conjugate_gradient,least_squares,utils}.py)."""
from __future__ import annotations

import warnings
from typing import Callable

import torch
from torch.autograd.function import once_differentiable


def dot(a, b, dim):
    """batched dot product keeping dims (linear/utils.py:6-26)"""
    return (a.conj() * b).sum(dim=dim, keepdim=True)


def conjugate_gradient(A: Callable, b, max_iter=1e2, tol=1e-5, eps=1e-8, parallel_dim=0, init=None, verbose=False):
    """Standard CG for A x = b with per-sample stopping (conjugate_gradient.py:7-77).

    The convergence test ``torch.all(res < tol)`` is one host sync per iteration, as in the
    reference (moving it on-device is row (f).1 of the scope table).
    """
    if isinstance(parallel_dim, int):
        parallel_dim = [parallel_dim]
    if parallel_dim is None:
        parallel_dim = []
    if list(parallel_dim) == [0] and b.ndim > 1 and _hip_cg_ok(b, init):
        return _conjugate_gradient_hip(A, b, max_iter, tol, eps, init, verbose)
    dim = [i for i in range(b.ndim) if i not in parallel_dim]
    x = torch.zeros_like(b) if init is None else init
    r = b - A(x)
    p = r
    res_old = dot(r, r, dim=dim).real
    b_norm_sq = dot(b, b, dim=dim).real
    b_norm_sq = torch.where(b_norm_sq > 0, b_norm_sq, torch.ones_like(b_norm_sq))
    tol = b_norm_sq * (tol ** 2)
    for i in range(int(max_iter)):
        Ap = A(p)
        alpha = res_old / (dot(p, Ap, dim=dim) + eps)
        x = x + p * alpha
        r = r - Ap * alpha
        res_new = dot(r, r, dim=dim).real
        if torch.all(res_new < tol):
            if verbose:
                print("CG Converged at iteration", i + 1)
            break
        p = r + p * (res_new / (res_old + eps))
        res_old = res_new
        if i > 0 and i % 100 == 0:
            r = b - A(x)
            res_old = dot(r, r, dim=dim).real
    else:
        if verbose:
            print("CG did not converge")
    return x


def _hip_cg_ok(b, init):
    from ..hip import elementwise as ew

    n = b.numel() // max(b.shape[0], 1)
    return b.is_cuda and n % 4 == 0 and ew.eligible(b, init) and not torch.is_grad_enabled()


# The host looks at the device-side convergence flag every this many iterations.  None = by problem size: reading the flag drains
# the stream (~30 us until the next kernel starts), running past convergence costs up to CG_CHECK_EVERY - 1 operator pairs - for a
# large system (cfg3: 2 ms per A^T A pair on 8 x 512 x 512) the pair is the expensive side and the flag is read after every
# iteration, so A^T A is applied exactly as often as the reference applies it; for small systems every 4th iteration.
CG_CHECK_EVERY = None
CG_LARGE_SYSTEM = 1 << 20     # elements of b from which an operator pair outweighs a stream drain


def cg_check_cadence(b) -> int:
    if CG_CHECK_EVERY is not None:
        return int(CG_CHECK_EVERY)
    return 1 if b.numel() >= CG_LARGE_SYSTEM else 4


def _conjugate_gradient_hip(A, b, max_iter, tol, eps, init, verbose, ew=None):
    """Same recurrence as above (conjugate_gradient.py:48-75) with the vector algebra in csrc/elementwise.hip:
    per-sample scalars stay on the device, dot products are deterministic shuffle reductions, and the stopping test
    `torch.all(res_new < tol)` is evaluated ON THE DEVICE: it raises a flag that turns the following updates into
    no-ops, so the iterate is exactly what the reference's `break` leaves.  The host reads the flag only every
    CG_CHECK_EVERY iterations (one sync per 4 instead of one per iteration; at most 3 operator applications are issued
    past convergence), and never while a HIP graph is being captured (then all max_iter iterations are recorded and
    the flag alone freezes the iterate)."""
    if ew is None:      # (the CPU test-suite injects the same kernels compiled for the host emulation)
        from ..hip import elementwise as ew

    b = b.contiguous()
    x = torch.zeros_like(b) if init is None else init.contiguous().clone()
    r = ew.lincomb(1.0, b, -1.0, A(x).contiguous())
    p = r.clone()
    res_old = ew.batched_dot(r, r)
    b_norm_sq = ew.batched_dot(b, b)
    b_norm_sq = torch.where(b_norm_sq > 0, b_norm_sq, torch.ones_like(b_norm_sq))
    tol2 = (b_norm_sq * (tol ** 2)).contiguous()
    done = torch.zeros(1, dtype=torch.int32, device=b.device)
    capturing = b.is_cuda and torch.cuda.is_current_stream_capturing()
    every = cg_check_cadence(b)
    for i in range(int(max_iter)):
        Ap = A(p).contiguous()
        pAp = ew.batched_dot(p, Ap)
        ew.cg_update_xr(res_old, pAp, eps, x, r, p, Ap, done)    # x += alpha p ; r -= alpha Ap
        res_new = ew.batched_dot(r, r)
        ew.cg_check(res_new, tol2, done)
        if not capturing and (i % every == every - 1 or i == int(max_iter) - 1) and bool(done.item()):
            if verbose:
                print("CG Converged at iteration <=", i + 1)
            break
        ew.cg_update_p(res_new, res_old, eps, p, r, done)         # p = r + beta p
        res_old = res_new
        if i > 0 and i % 100 == 0 and not capturing and not bool(done.item()):
            r = ew.lincomb(1.0, b, -1.0, A(x).contiguous())
            res_old = ew.batched_dot(r, r)
    else:
        if verbose:
            print("CG did not converge")
    return x


def least_squares(A, AT, y, z=0.0, init=None, gamma=None, parallel_dim=0, AAT=None, ATA=None, solver="CG",
                  max_iter=100, tol=1e-6, **kwargs):
    r""":math:`\min_x \frac{\gamma}{2}\|Ax-y\|^2 + \frac12\|x-z\|^2` (least_squares.py:15-197); solvers CG, BiCGStab,
    lsqr, minres."""
    if isinstance(parallel_dim, int):
        parallel_dim = [parallel_dim]
    if gamma is None:
        gamma = torch.tensor(0.0, device=y.device)
        gamma_provided = False
    else:
        gamma_provided = True
        if not isinstance(gamma, torch.Tensor):
            gamma = torch.tensor(gamma, device=y.device)
        if torch.any(gamma <= 0):
            warnings.warn("Regularization parameter of least squares problem (gamma) should be positive.")
    Aty = AT(y)
    if gamma.ndim > 0:
        if gamma.size(0) != Aty.size(0):
            raise ValueError("If gamma is batched, its batch size must match the one of y.")
        if gamma.ndim == 1:
            gamma = gamma.view([gamma.size(0)] + [1] * (Aty.ndim - 1))
        elif gamma.ndim != Aty.ndim:
            raise ValueError(f"gamma should either be 0D, 1D, or match same number of dimensions as ATy, but got "
                             f"ndims {gamma.ndim} and {Aty.ndim}")
    from .linear_solvers import bicgstab, lsqr, minres

    if solver == "lsqr":     # rectangular solver on A itself (least_squares.py:122-134)
        x, _ = lsqr(A, AT, y, x0=z, eta=1 / gamma if gamma_provided else None, max_iter=max_iter, tol=tol,
                    parallel_dim=parallel_dim, **kwargs)
        return x
    if solver not in ("CG", "BiCGStab", "minres"):
        raise ValueError(f"Solver {solver} not recognized. Choose between 'CG', 'lsqr', 'BiCGStab' and 'minres'.")
    complete = Aty.shape == y.shape
    overcomplete = Aty.numel() < y.numel()
    if complete and solver in ("BiCGStab", "minres"):     # square system solved directly (least_squares.py:139-141)
        H, b = (lambda x: A(x)), y
    else:
        if AAT is None:
            AAT = lambda x: A(AT(x))
        if ATA is None:
            ATA = lambda x: AT(A(x))
        if gamma_provided:
            b = Aty + 1 / gamma * z
            H = lambda x: ATA(x) + 1 / gamma * x
            overcomplete = False
        elif not overcomplete:
            H, b = (lambda x: AAT(x)), y
        else:
            H, b = (lambda x: ATA(x)), Aty
    run = {"CG": conjugate_gradient, "BiCGStab": bicgstab, "minres": minres}[solver]
    x = run(A=H, b=b, init=init, max_iter=max_iter, tol=tol, parallel_dim=parallel_dim, **kwargs)
    if not gamma_provided and not overcomplete and not complete:
        x = AT(x)
    return x


class LeastSquaresSolver(torch.autograd.Function):
    """implicit differentiation of the least-squares solve (least_squares.py:200-341)"""

    @staticmethod
    def forward(ctx, physics, y, z, init, gamma, trigger=None, extra_kwargs=None):
        kwargs = extra_kwargs if extra_kwargs is not None else {}
        with torch.no_grad():
            sol = least_squares(A=physics.A, AT=physics.A_adjoint, y=y, z=z, init=init, gamma=gamma,
                                AAT=physics.A_A_adjoint, ATA=physics.A_adjoint_A, **kwargs)
        gshape = gamma.shape
        if gamma.ndim == 1:
            gamma = gamma.view([gamma.size(0)] + [1] * (sol.ndim - 1))
        ctx.save_for_backward(sol, y, z, gamma)
        ctx.physics, ctx.kwargs, ctx.gshape = physics, kwargs, gshape
        return sol

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        h, y, z, gamma = ctx.saved_tensors
        physics = ctx.physics
        with torch.no_grad():
            mv = least_squares(A=physics.A, AT=physics.A_adjoint, y=torch.zeros_like(y), z=grad_output * gamma,
                               gamma=gamma, AAT=physics.A_A_adjoint, ATA=physics.A_adjoint_A, **ctx.kwargs)
        needs = ctx.needs_input_grad
        grads = [None] * len(needs)
        if needs[1]:
            grads[1] = physics.A(mv)
        if needs[2]:
            grads[2] = mv / gamma
        if needs[4]:
            gg = torch.sum(mv.conj() * (h - z), dim=list(range(1, mv.ndim)), keepdim=True).real / gamma ** 2
            grads[4] = gg.view(ctx.gshape) if len(ctx.gshape) > 0 else torch.sum(gg).view(())
        return tuple(grads)


def least_squares_implicit_backward(physics, y, z=None, init=None, gamma=None, **kwargs):
    """least_squares.py:345-469"""
    if z is None:
        z = torch.zeros_like(physics.A_adjoint(y))
    elif isinstance(z, (int, float)):
        z = torch.full_like(physics.A_adjoint(y), fill_value=float(z))
    if init is None:
        init = torch.zeros_like(z)
    trig = y.requires_grad or z.requires_grad or (isinstance(gamma, torch.Tensor) and gamma.requires_grad)
    trigger = torch.ones(1, device=y.device, dtype=y.dtype).requires_grad_(bool(trig))
    dtype = y.dtype if not torch.is_complex(y) else y.real.dtype
    if gamma is None:
        gamma = torch.zeros((), device=y.device, dtype=dtype)
    if isinstance(gamma, torch.Tensor) and gamma.ndim > 0 and gamma.size(0) != y.size(0):
        raise ValueError("If gamma is batched, its batch size must match the one of y.")
    if not isinstance(gamma, torch.Tensor):
        gamma = torch.as_tensor(gamma, device=y.device, dtype=dtype)
    return LeastSquaresSolver.apply(physics, y, z, init, gamma, trigger, kwargs if kwargs else None)
