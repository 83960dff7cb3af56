"""Torch-CPU restatement of the iteration loops (PnP-PGD, PnP-HQS, CG) — TEST INFRASTRUCTURE.

Follows deepinv/optim/fixed_point.py:262-406 (loop), optim_iterators/pgd.py:137-168,
hqs.py:65-95, optim_iterator.py:76-125 (relaxation, beta = 1), data_fidelity.py:309-338 (L2),
optimizers.py:572 (x0 = A^T y), linear/conjugate_gradient.py:48-75, least_squares.py:100-169.
"""
from __future__ import annotations

import torch


def pnp_pgd(y, A, AT, denoiser, stepsize=1.0, sigma_denoiser=0.05, lam=1.0, max_iter=50, sigma_noise=1.0, x0=None):
    """x <- D_sigma( x - gamma * (A^T A x - A^T y) / sigma_f^2 ),  x0 = A^T y."""
    norm = 1.0 / sigma_noise ** 2
    x = AT(y) if x0 is None else x0
    for _ in range(max_iter):
        grad = norm * (AT(A(x)) - AT(y))          # L2.grad recomputes A^T y every iteration, like the reference
        u = x - stepsize * grad
        x = denoiser(u, sigma_denoiser)
    return x


def conjugate_gradient(H, b, max_iter=100, tol=1e-5, eps=1e-8, init=None):
    """conjugate_gradient.py:48-75 with parallel_dim=[0]"""
    dim = list(range(1, b.ndim))
    dot = lambda a, c: (a.conj() * c).sum(dim=dim, keepdim=True)
    x = torch.zeros_like(b) if init is None else init
    r = b - H(x)
    p = r
    res_old = dot(r, r).real
    bn = dot(b, b).real
    bn = torch.where(bn > 0, bn, torch.ones_like(bn))
    tol2 = bn * tol ** 2
    for i in range(int(max_iter)):
        Hp = H(p)
        alpha = res_old / (dot(p, Hp) + eps)
        x = x + p * alpha
        r = r - Hp * alpha
        res_new = dot(r, r).real
        if torch.all(res_new < tol2):
            break
        p = r + p * (res_new / (res_old + eps))
        res_old = res_new
        if i > 0 and i % 100 == 0:
            r = b - H(x)
            res_old = dot(r, r).real
    return x


def prox_l2_cg(z, y, gamma, A, AT, max_iter=50, tol=1e-4):
    """LinearPhysics.prox_l2 -> least_squares(gamma) -> CG on (A^T A + I/gamma) x = A^T y + z/gamma, init = z
    (forward.py:751-814, least_squares.py:148-169)"""
    b = AT(y) + z / gamma
    H = lambda v: AT(A(v)) + v / gamma
    return conjugate_gradient(H, b, max_iter=max_iter, tol=tol, init=z)


def pnp_hqs(y, prox_f, denoiser, stepsize, sigma_denoiser, lam=1.0, max_iter=8, x0=None):
    """x <- D_sigma_k( prox_{gamma_k f}(x) ) with per-iteration parameter lists (hqs.py:65-95)."""
    x = x0
    for k in range(max_iter):
        g = stepsize[k] if hasattr(stepsize, "__len__") else stepsize
        s = sigma_denoiser[k] if hasattr(sigma_denoiser, "__len__") else sigma_denoiser
        u = prox_f(x, y, g)
        x = denoiser(u, s)
    return x
