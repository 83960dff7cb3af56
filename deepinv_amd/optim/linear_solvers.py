"""The other Krylov solvers behind ``LinearPhysics.prox_l2 / A_dagger(solver=...)``: LSQR, BiCGStab, MINRES
(reference deepinv/optim/linear/{lsqr,bicgstab,minres}.py; same signatures, stopping rules and per-sample treatment of
the batch dimension).  They are written on torch tensors - the work is in the operator applications, which are the HIP
kernels of the physics; the recurrences only touch image-sized vectors a handful of times per iteration."""
from __future__ import annotations

from typing import Callable

import torch


def _reduce_dims(t, parallel_dim):
    return [i for i in range(t.ndim) if i not in parallel_dim]


def _pdims(parallel_dim):
    if isinstance(parallel_dim, int):
        return [parallel_dim]
    return [] if parallel_dim is None else list(parallel_dim)


def _bshape(t, parallel_dim):
    """shape that broadcasts a per-sample scalar against t"""
    return [t.shape[i] if i in parallel_dim else 1 for i in range(t.ndim)]


def lsqr(A: Callable, AT: Callable, b, eta=0.0, x0=None, tol=1e-6, conlim=1e8, max_iter=100, parallel_dim=0,
         verbose=False, **kwargs):
    r"""Paige & Saunders' LSQR on :math:`\min_x \|Ax-b\|^2+\eta\|x-x_0\|^2` (Golub-Kahan bidiagonalisation, the
    damping folded in by an extra plane rotation per step); returns ``(x, cond)`` like lsqr.py:8-226.  Stops when every
    sample's residual estimate is below ``tol * ||b||`` or the condition estimate passes ``conlim``."""
    pd = _pdims(parallel_dim)
    dev = b.device
    nrm = lambda u: torch.linalg.vector_norm(u, dim=_reduce_dims(u, pd), keepdim=False)
    Atb = AT(b)
    sb, sx = _bshape(b, pd), _bshape(Atb, pd)
    mul_b = lambda v, a: v * a.view(sb)
    mul_x = lambda v, a: v * a.view(sx)
    eta = torch.as_tensor(0.0 if eta is None else eta, device=dev)
    if eta.ndim > 0:
        if eta.size(0) != b.size(0):
            raise ValueError("If eta is batched, its batch size must match the one of b.")
        eta = eta.squeeze()
    if torch.any(eta < 0):
        raise ValueError("Damping parameter eta must be non-negative. LSQR cannot be applied to problems with negative eta.")
    sqrt_eta, damped = torch.sqrt(eta), bool(torch.any(eta > 0))

    anorm, ddnorm = 0.0, 0.0
    acond = torch.zeros(1, device=dev)
    u = b.clone()
    bnorm = nrm(b)
    if x0 is None:
        x, beta = torch.zeros_like(Atb), bnorm
    else:
        x = x0 * torch.zeros_like(Atb) if isinstance(x0, float) else x0.clone()
        u = u - A(x)
        beta = nrm(u)
    if torch.all(beta > 0):
        u = mul_b(u, 1 / beta)
        v = AT(u)
        alpha = nrm(v)
    else:
        v, alpha = torch.zeros_like(x), torch.zeros(1, device=dev)
    if torch.all(alpha > 0):
        v = mul_x(v, 1 / alpha)
    w = v.clone()
    rhobar, phibar = alpha, beta
    if torch.any(alpha * beta == 0):
        return x, acond
    z, cs2, sn2, xxnorm = 0.0, -1.0, 0.0, 0.0
    converged = False
    for itn in range(int(max_iter)):
        # next pair of Lanczos vectors
        u = A(v) - mul_b(u, alpha)
        beta = nrm(u)
        if torch.all(beta > 0):
            u = mul_b(u, 1 / beta)
            anorm = torch.sqrt(anorm ** 2 + alpha ** 2 + beta ** 2 + eta)
            v = AT(u) - mul_x(v, beta)
            alpha = nrm(v)
            if torch.all(alpha > 0):
                v = mul_x(v, 1 / alpha)
        # rotation that removes the damping term, then the one that removes the sub-diagonal
        if damped:
            rb1 = torch.sqrt(rhobar ** 2 + eta)
            psi = (sqrt_eta / rb1) * phibar
            phibar = (rhobar / rb1) * phibar
        else:
            rb1, psi = rhobar, 0.0
        cs, sn, rho = _givens(rb1, beta)
        theta, rhobar = sn * alpha, -cs * alpha
        phi, phibar = cs * phibar, sn * phibar
        dk = mul_x(w, 1 / rho)
        x = x + mul_x(w, phi / rho)
        w = v + mul_x(w, -theta / rho)
        ddnorm = ddnorm + nrm(dk) ** 2
        # estimate of ||x|| (kept for parity of the recurrences) and the stopping quantities
        delta, gambar = sn2 * rho, -cs2 * rho
        rhs = phi - delta * z
        gamma = torch.sqrt(gambar ** 2 + theta ** 2)
        cs2, sn2, z = gambar / gamma, theta / gamma, rhs / gamma
        xxnorm = xxnorm + z ** 2
        acond = anorm * torch.sqrt(ddnorm).mean()
        rnorm = torch.sqrt(phibar ** 2 + psi ** 2)
        if torch.all(rnorm <= tol * bnorm):
            converged = True
            if verbose:
                print("LSQR converged at iteration", itn)
            break
        if torch.any(acond > conlim):
            converged = True
            if verbose:
                print(f"LSQR reached condition number limit {conlim} at iteration", itn)
            break
    if not converged and verbose:
        print("LSQR did not converge")
    return x, acond.sqrt()


def _givens(a, b):
    """numerically careful plane rotation (c, s, r) with c a + s b = r (Choi's sym-ortho, as scipy / lsqr.py:229-262)"""
    a, b = torch.broadcast_tensors(torch.as_tensor(a), torch.as_tensor(b))
    if torch.any(b == 0):
        return torch.sign(a), 0, a.abs()
    if torch.any(a == 0):
        return 0, torch.sign(b), b.abs()
    if torch.any(b.abs() > a.abs()):
        tau = a / b
        s = torch.sign(b) / torch.sqrt(1 + tau * tau)
        return s * tau, s, b / s
    tau = b / a
    c = torch.sign(a) / torch.sqrt(1 + tau * tau)
    return c, c * tau, a / c


def _dot(a, b, dim):
    return (a.conj() * b).sum(dim=dim, keepdim=True)


def bicgstab(A: Callable, b, init=None, max_iter=1e2, tol=1e-5, parallel_dim=0, verbose=False,
             left_precon=lambda x: x, right_precon=lambda x: x):
    """van der Vorst's BiCGStab for square ``A`` (bicgstab.py:8-107): shadow residual fixed at r0, divisions guarded at
    machine epsilon (a vanishing denominator zeroes the step instead of producing inf)."""
    pd = _pdims(parallel_dim)
    dim = _reduce_dims(b, pd)
    x = init if init is not None else torch.zeros_like(b)
    r = b - A(x)
    shadow = r.clone()
    rho = _dot(r, shadow, dim)
    p = r
    tol2 = _dot(b, b, dim).real * tol ** 2
    tiny = torch.finfo(b.dtype).eps
    safe_div = lambda num, den: torch.where(den.abs() > tiny, num / den, torch.zeros_like(num))
    done = False
    for i in range(int(max_iter)):
        y = right_precon(left_precon(p))
        v = A(y)
        alpha = safe_div(rho, _dot(shadow, v, dim))
        h = x + alpha * y
        s = r - alpha * v
        zz = right_precon(left_precon(s))
        t = A(zz)
        ls, lt = left_precon(s), left_precon(t)
        omega = safe_div(_dot(lt, ls, dim), _dot(lt, lt, dim))
        x = h + omega * zz
        r = s - omega * t
        if torch.all(_dot(r, r, dim).real < tol2):
            done = True
            if verbose:
                print("BiCGSTAB Converged at iteration", i)
            break
        rho_next = _dot(r, shadow, dim)
        ok = (rho.abs() > tiny) & (omega.abs() > tiny)
        beta = torch.where(ok, (rho_next / rho) * (alpha / omega), torch.zeros_like(rho_next))
        p = r + beta * (p - omega * v)
        rho = rho_next
    if not done and verbose:
        print("BiCGSTAB did not converge")
    return x


def minres(A: Callable, b, init=None, max_iter=1e2, tol=1e-5, eps=1e-6, parallel_dim=0, verbose=False,
           precon=lambda x: x.clone()):
    """Paige & Saunders' MINRES for symmetric ``A`` (minres.py:8-173): preconditioned Lanczos three-term recurrence,
    QR of the tridiagonal by two trailing Givens rotations, right-hand side normalised per sample; stops when the last
    update is below ``tol`` relative to the iterate for every sample."""
    pd = _pdims(parallel_dim)
    dim = _reduce_dims(b, pd)
    vnorm = lambda t: torch.linalg.vector_norm(t, dim=dim, keepdim=True, ord=2)
    scale = vnorm(b)
    null_rhs = scale < 1e-10
    scale = scale.masked_fill(null_rhs, 1)
    b = b / scale
    sol = init / scale if init is not None else torch.zeros(b.shape, dtype=b.dtype, device=b.device)
    z_old = torch.zeros(sol.shape, device=b.device)      # residual-space Lanczos vectors
    z_cur = b - A(sol)
    q_cur = precon(z_cur)                                # preconditioned counterpart
    beta = torch.abs(_dot(z_cur, q_cur, dim).sqrt()).clamp_min(eps)
    z_cur, q_cur = z_cur / beta, q_cur / beta
    one = torch.ones_like(beta)
    c_old, s_old, c_cur, s_cur = one, torch.zeros_like(one), one, torch.zeros_like(one)
    d_old, d_cur = torch.zeros_like(sol), torch.zeros_like(sol)   # search directions (columns of Q R^-1)
    eta_k = beta
    unconverged = True
    i = 0
    for i in range(int(max_iter)):
        Aq = A(q_cur)
        alpha = _dot(Aq, q_cur, dim)
        Aq = Aq - alpha * z_cur - beta * z_old
        q_next = precon(Aq)
        beta_next = torch.abs(_dot(Aq, q_next, dim).sqrt()).clamp_min(eps)
        Aq, q_next = Aq / beta_next, q_next / beta_next
        # apply the two previous rotations to the new column (beta, alpha, beta_next) of the tridiagonal
        eps_k = s_old * beta
        delta = c_old * beta
        diag = alpha * c_cur - s_cur * delta
        delta = delta * c_cur + s_cur * alpha
        rad = torch.sqrt(diag * diag + beta_next * beta_next)
        c_new, s_new = diag / rad, beta_next / rad
        diag = diag * c_new + s_new * beta_next
        d_new = (q_cur - delta * d_cur - eps_k * d_old) / diag
        step = d_new * eta_k * c_new
        sol = sol + step
        if (torch.linalg.vector_norm(step, dim=dim, ord=2).unsqueeze(-1)
                / torch.linalg.vector_norm(sol, dim=dim, ord=2).unsqueeze(-1)).max().item() < tol:
            unconverged = False
            if verbose:
                print("MINRES converged at iteration", i + 1)
            break
        eta_k = -eta_k * s_new
        z_old, z_cur, q_cur, beta = z_cur, Aq, q_next, beta_next
        c_old, s_old, c_cur, s_cur = c_cur, s_cur, c_new, s_new
        d_old, d_cur = d_cur, d_new
    sol = sol.masked_fill(null_rhs, 0)
    if unconverged and verbose:
        print(f"MINRES did not converge in {i} iterations!")
    return sol * scale
