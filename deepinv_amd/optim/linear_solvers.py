"""The other Krylov solvers behind ``LinearPhysics.prox_l2 / A_dagger(solver=...)``: LSQR, BiCGStab, MINRES
(reference deepinv/optim/linear/{lsqr,bicgstab,minres}.py; same signatures, stopping rules and per-sample treatment of
the batch dimension).  They are written on torch tensors - the work is in the operator applications, which are the HIP
kernels of the physics; the recurrences only touch image-sized vectors a handful of times per iteration.

No host synchronisation per iteration: the reference's data-dependent branches (`if torch.all(...)`, `if torch.any(...)`,
`.item()`: 7 per LSQR iteration, 1 per BiCGStab / MINRES iteration) are evaluated on the device.  A stopping test raises a
device flag that freezes the iterate (`torch.where`), so the result is exactly what the reference's `break` leaves; the host
reads the flag every `CHECK_EVERY` iterations only (at most `CHECK_EVERY - 1` operator applications are issued past
convergence), as `conjugate_gradient` does (optim/linear.py)."""
from __future__ import annotations

from typing import Callable

import torch


CHECK_EVERY = 4   # the host looks at the device-side convergence flag every this many iterations


def _poll(done, i, max_iter):
    """True when the host should stop: polled every CHECK_EVERY iterations and at the last one"""
    return (i % CHECK_EVERY == CHECK_EVERY - 1 or i == int(max_iter) - 1) and bool(done)


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
    anorm = torch.zeros((), device=dev, dtype=bnorm.dtype)
    done = torch.zeros((), dtype=torch.bool, device=dev)
    safe = lambda t: torch.where(t > 0, t, torch.ones_like(t))
    for itn in range(int(max_iter)):
        # next pair of Lanczos vectors.  The reference normalises only while every sample's beta (alpha) is positive
        # (`if torch.all(beta > 0)`): the same decision, taken on the device
        u_new = A(v) - mul_b(u, alpha)
        beta_new = nrm(u_new)
        ok_b = (beta_new > 0).all()
        u_new = torch.where(ok_b, mul_b(u_new, 1 / safe(beta_new)), u_new)
        anorm_new = torch.where(ok_b, torch.sqrt(anorm ** 2 + alpha ** 2 + beta_new ** 2 + eta), anorm)
        v_new = AT(u_new) - mul_x(v, beta_new)
        alpha_new = nrm(v_new)
        ok_a = (alpha_new > 0).all()
        v_new = torch.where(ok_a, mul_x(v_new, 1 / safe(alpha_new)), v_new)
        u, beta, anorm = u_new, beta_new, anorm_new
        v, alpha = torch.where(ok_b, v_new, v), torch.where(ok_b, alpha_new, alpha)
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
        x = torch.where(done, x, x + mul_x(w, phi / rho))      # frozen once the stopping test has fired
        w = v + mul_x(w, -theta / rho)
        ddnorm = ddnorm + nrm(dk) ** 2
        # estimate of ||x|| (kept for parity of the recurrences) and the stopping quantities
        delta, gambar = sn2 * rho, -cs2 * rho
        rhs = phi - delta * z
        gamma = torch.sqrt(gambar ** 2 + theta ** 2)
        cs2, sn2, z = gambar / gamma, theta / gamma, rhs / gamma
        xxnorm = xxnorm + z ** 2
        acond = torch.where(done, acond, anorm * torch.sqrt(ddnorm).mean())
        rnorm = torch.sqrt(phibar ** 2 + psi ** 2)
        done = done | (rnorm <= tol * bnorm).all() | (acond > conlim).any()
        if _poll(done, itn, max_iter):
            if verbose:
                print("LSQR stopped (tolerance or condition number limit) at iteration <=", itn)
            break
    else:
        if verbose:
            print("LSQR did not converge")
    return x, acond.sqrt()


def _givens(a, b):
    """numerically careful plane rotation (c, s, r) with c a + s b = r (Choi's sym-ortho, as scipy / lsqr.py:229-262).  The
    reference picks one of its four formulas for the whole batch with `torch.any` (a host sync each); here every element
    takes the formula its own (a, b) calls for, chosen without leaving the device.  Intentional difference: in a batch
    that mixes zero and non-zero entries (or |b| > |a| with |b| <= |a|) the reference applies ONE branch to all elements,
    so single elements of such a batch get the rotation from another (equally valid, differently rounded) formula
    there; the rotations agree to rounding and the golden test holds at 1e-7."""
    a, b = torch.broadcast_tensors(torch.as_tensor(a), torch.as_tensor(b))
    one = torch.ones_like(a)
    big_b = b.abs() > a.abs()
    tau = torch.where(big_b, a / torch.where(b == 0, one, b), b / torch.where(a == 0, one, a))
    root = torch.sqrt(1 + tau * tau)
    s3 = torch.sign(b) / root          # |b| > |a|: tau = a / b
    c4 = torch.sign(a) / root          # else:      tau = b / a
    c = torch.where(big_b, s3 * tau, c4)
    sn = torch.where(big_b, s3, c4 * tau)
    r = torch.where(big_b, b / torch.where(s3 == 0, one, s3), a / torch.where(c4 == 0, one, c4))
    zero = torch.zeros_like(a)
    b0, a0 = b == 0, (a == 0) & (b != 0)
    c = torch.where(b0, torch.sign(a), torch.where(a0, zero, c))
    sn = torch.where(b0, zero, torch.where(a0, torch.sign(b), sn))
    r = torch.where(b0, a.abs(), torch.where(a0, b.abs(), r))
    return c, sn, r


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
    done = torch.zeros((), dtype=torch.bool, device=b.device)
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
        x = torch.where(done, x, h + omega * zz)      # frozen once the stopping test has fired
        r = s - omega * t
        done = done | (_dot(r, r, dim).real < tol2).all()
        if _poll(done, i, max_iter):
            if verbose:
                print("BiCGSTAB Converged at iteration <=", i)
            break
        rho_next = _dot(r, shadow, dim)
        ok = (rho.abs() > tiny) & (omega.abs() > tiny)
        beta = torch.where(ok, (rho_next / rho) * (alpha / omega), torch.zeros_like(rho_next))
        p = r + beta * (p - omega * v)
        rho = rho_next
    else:
        if verbose:
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
    done = torch.zeros((), dtype=torch.bool, device=b.device)
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
        sol_new = sol + step
        rel = (torch.linalg.vector_norm(step, dim=dim, ord=2).unsqueeze(-1)
               / torch.linalg.vector_norm(sol_new, dim=dim, ord=2).unsqueeze(-1)).max()
        sol = torch.where(done, sol, sol_new)      # frozen once the stopping test has fired
        done = done | (rel < tol)
        if _poll(done, i, max_iter):
            if verbose:
                print("MINRES converged at iteration <=", i + 1)
            break
        eta_k = -eta_k * s_new
        z_old, z_cur, q_cur, beta = z_cur, Aq, q_next, beta_next
        c_old, s_old, c_cur, s_cur = c_cur, s_cur, c_new, s_new
        d_old, d_cur = d_cur, d_new
    else:
        if verbose:
            print(f"MINRES did not converge in {i} iterations!")
    sol = sol.masked_fill(null_rhs, 0)
    return sol * scale
