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


def diffpir_schedule(sigma, max_iter, lambda_, T=1000, beta_start=0.1 / 1000, beta_end=20 / 1000):
    """DiffPIR.get_alpha_beta / get_noise_schedule (diffusion.py:323-375), element by element like the reference."""
    betas = torch.linspace(beta_start, beta_end, T, dtype=torch.float32)
    ac = torch.cumprod(1.0 - betas, dim=0)
    sqrt_ac, sqrt_1m = torch.sqrt(ac), torch.sqrt(1.0 - ac)
    reduced = torch.div(sqrt_1m, sqrt_ac)
    sigmas, rhos = [], []
    for i in range(T):
        sigmas.append(reduced[T - 1 - i])
        sigma_k = sqrt_1m[i] / sqrt_ac[i]
        rhos.append(lambda_ * (sigma ** 2) / (sigma_k ** 2))
    seq = torch.sqrt(torch.linspace(0.0, T ** 2, max_iter)).type(torch.int32)
    seq[-1] = seq[-1] - 1
    return dict(rhos=torch.tensor(rhos), sigmas=torch.tensor(sigmas), seq=seq, reduced=reduced, sqrt_ac=sqrt_ac,
                sqrt_1m=sqrt_1m, sqrt_recip=torch.sqrt(1.0 / ac))


def diffpir(y, AT, prox, denoiser, draws, sigma=0.05, max_iter=100, zeta=0.1, lambda_=7.0, noise_sigma=None):
    """DiffPIR.forward (diffusion.py:423-513).  `draws` is an iterator over the torch.randn_like samples (so a recorded
    sample path of the reference can be replayed); `prox(z, y, gamma)` is the data-fidelity prox.  `noise_sigma`: the
    physics' noise_model.sigma (a float32 tensor in the reference), which overwrites the schedule (:441-443)."""
    S = diffpir_schedule(sigma if noise_sigma is None else torch.as_tensor(noise_sigma, dtype=torch.float32), max_iter,
                         lambda_)
    near = lambda v: torch.abs(S["reduced"] - v).argmin()
    draws = iter(draws)
    x = 2 * AT(y) - 1
    seq = S["seq"]
    for i in range(len(seq)):
        cs = S["sigmas"][seq[i]]
        t_i = near(cs)
        at = 1 / S["sqrt_recip"][t_i] ** 2
        if i == 0:
            x = (x + (cs ** 2 - 4.0 * sigma ** 2).sqrt() * next(draws)) / S["sqrt_recip"][-1]
        x0 = (2 * denoiser(x / (2 * at.sqrt()) + 0.5, cs / 2) - 1).clamp(-1, 1)
        if not seq[i] == seq[-1]:
            x0 = prox(x0 / 2 + 0.5, y, 1.0 / (2 * S["rhos"][t_i])) * 2 - 1
            t_im1 = near(S["sigmas"][seq[i + 1]])
            eps = (x - S["sqrt_ac"][t_i] * x0) / S["sqrt_1m"][t_i]
            x = (S["sqrt_ac"][t_im1] * x0 + S["sqrt_1m"][t_im1] * (1 - zeta) ** 0.5 * eps
                 + S["sqrt_1m"][t_im1] * zeta ** 0.5 * next(draws))
    return x / 2 + 0.5
