"""Operator base classes: ``Physics`` -> ``LinearPhysics`` -> ``DecomposablePhysics``.

Host-side mirror of the reference's drop-in boundary (deepinv/physics/forward.py:107-351,
354-862, 1000-1252): same method names, argument meaning and error behaviour, so everything
above it (``L2.grad/prox``, ``BaseOptim``, ``DiffPIR``, user code) talks to our operators
exactly as it talks to the reference's.  The arithmetic of concrete operators is in the HIP
kernels; this file is only the algebra that composes ``A`` / ``A_adjoint`` calls.
"""
from __future__ import annotations

import copy
import warnings
from typing import Callable

import torch
import torch.nn as nn
from torch import Tensor

from .noise import NoiseModel, ZeroNoise


def power_method(operator: Callable, x0: Tensor, max_iter: int = 100, tol: float = 1e-6, verbose: bool = False,
                 **kwargs) -> Tensor:
    """Largest eigenvalue of a PSD operator (reference physics/functional/matrix.py:5-44).

    Starts from ``torch.randn_like(x0)`` drawn from the global RNG like the reference; the
    convergence test is evaluated on the host once per iteration as in the reference.
    """
    x = torch.randn_like(x0)
    x = x / torch.linalg.vector_norm(x)
    z_old = torch.zeros((), device=x.device, dtype=x.dtype)
    z = z_old
    for it in range(max_iter):
        y = operator(x, **kwargs)
        z = torch.vdot(x.flatten(), y.flatten()) / torch.linalg.vector_norm(x) ** 2
        if torch.linalg.vector_norm(z - z_old) < tol:
            if verbose:
                print(f"Power iteration converged at iteration {it}, ||A^T A||_2={z.real.item():.2f}")
            break
        z_old = z
        x = y / torch.linalg.vector_norm(y)
    else:
        warnings.warn("Power iteration: convergence not reached")
    return z.real


class Physics(nn.Module):
    r"""Forward model :math:`y = N(A(x))` (reference forward.py:17-351)."""

    def __init__(self, A: Callable = lambda x, **kwargs: x, noise_model: NoiseModel | None = None,
                 sensor_model: Callable = lambda x: x, solver: str = "gradient_descent", max_iter: int = 50,
                 tol: float = 1e-4, **kwargs):
        super().__init__()
        self.noise_model = ZeroNoise() if noise_model is None else noise_model
        self.sensor_model = sensor_model
        self.forw = A
        self.SVD = False
        self.max_iter = max_iter
        self.tol = tol
        self.solver = solver
        if kwargs:
            warnings.warn(f"Arguments {kwargs} are passed to {self.__class__.__name__} but are ignored.")

    def forward(self, x, **kwargs):
        return self.sensor(self.noise(self.A(x, **kwargs), **kwargs))

    def A(self, x, **kwargs):
        return self.forw(x, **kwargs)

    def sensor(self, x):
        return self.sensor_model(x)

    def set_noise_model(self, noise_model, **kwargs):
        self.noise_model = noise_model

    def noise(self, x, **kwargs) -> Tensor:
        return self.noise_model(x, **kwargs)

    def A_vjp(self, x, v):
        _, vjp = torch.func.vjp(self.A, x)
        return vjp(v)[0]

    def A_dagger(self, y, x_init=None):
        if self.solver != "gradient_descent":
            raise NotImplementedError(f"Solver {self.solver} not implemented for A_dagger")
        if x_init is None:
            if not hasattr(self, "A_adjoint"):
                raise ValueError("x_init must be provided for gradient descent solver if the physics does not have "
                                 "A_adjoint defined.")
            x_init = self.A_adjoint(y)
        x, lr = x_init, 0.1
        for _ in range(self.max_iter):
            x = x - lr * self.A_vjp(x, self.A(x) - y)
            if torch.nn.functional.mse_loss(self.A(x), y) < self.tol:
                break
        return x.clone()

    def set_ls_solver(self, solver, max_iter=None, tol=None):
        if max_iter is not None:
            self.max_iter = max_iter
        if tol is not None:
            self.tol = tol
        self.solver = solver

    def update(self, **kwargs):
        self.update_parameters(**kwargs)
        if hasattr(self.noise_model, "update_parameters"):
            self.noise_model.update_parameters(**kwargs)

    def update_parameters(self, **kwargs):
        """Tensor kwargs overwrite same-named attributes, cast to the attribute's device/dtype
        (reference forward.py:249-276)."""
        for key, value in kwargs.items():
            if value is None or not hasattr(self, key) or not isinstance(value, Tensor):
                continue
            cur = getattr(self, key)
            if isinstance(cur, Tensor):
                if value.device.type != cur.device.type:
                    warnings.warn(f"The provided tensor for parameter '{key}' is on a different device "
                                  f"({value.device}) than the current parameter device ({cur.device}). "
                                  "The current device will be used.", stacklevel=2)
                value = value.to(cur)
            setattr(self, key, value)

    def clone(self):
        return copy.deepcopy(self)


class LinearPhysics(Physics):
    r"""Linear operator with adjoint, least-squares prox and pseudo-inverse (forward.py:354-862)."""

    def __init__(self, A=lambda x, **kwargs: x, A_adjoint=None, img_size=None, noise_model=None,
                 sensor_model=lambda x: x, max_iter=50, tol=1e-4, solver="lsqr", implicit_backward_solver: bool = True,
                 device="cpu", **kwargs):
        super().__init__(A=A, noise_model=noise_model, sensor_model=sensor_model, max_iter=max_iter, solver=solver,
                         tol=tol, **kwargs)
        self.A_adj = A_adjoint
        self.img_size = img_size
        self.implicit_backward_solver = implicit_backward_solver
        self.register_buffer("_device_holder", torch.tensor(0.0, device=device), persistent=False)
        self.to(device)

    @property
    def device(self):
        return self._device_holder.device

    def A_adjoint(self, y, **kwargs):
        if self.A_adj is not None:
            return self.A_adj(y, **kwargs)
        if self.img_size is None:
            raise ValueError("img_size must be set for using the automatic A_adjoint implementation.")
        return adjoint_function(self.A, (y.shape[0],) + tuple(self.img_size), device=y.device)(y, **kwargs)

    def A_vjp(self, x, v):
        return self.A_adjoint(v)

    def A_A_adjoint(self, y, **kwargs):
        return self.A(self.A_adjoint(y, **kwargs), **kwargs)

    def A_adjoint_A(self, x, **kwargs):
        return self.A_adjoint(self.A(x, **kwargs), **kwargs)

    def compute_sqnorm(self, x0: Tensor, *, max_iter: int = 100, tol: float = 1e-3, verbose: bool = True,
                       rng=None, **kwargs) -> Tensor:
        return power_method(self.A_adjoint_A, x0, max_iter=max_iter, tol=tol, verbose=verbose, **kwargs)

    def compute_norm(self, x0: Tensor, max_iter: int = 100, tol: float = 1e-3, verbose: bool = True,
                     squared: bool = True, **kwargs) -> Tensor:
        if squared not in (True, False):
            raise ValueError(f"squared must be True or False, got {squared}")
        if squared:
            warnings.warn("Using `compute_norm(squared=True)` is deprecated. Use `compute_sqnorm()` instead.",
                          DeprecationWarning, stacklevel=1)
        sq = self.compute_sqnorm(x0, max_iter=max_iter, tol=tol, verbose=verbose, **kwargs)
        return sq if squared else sq.sqrt()

    def adjointness_test(self, u, **kwargs):
        r""":math:`\langle Au, v\rangle - \langle u, A^\top v\rangle` for a random ``v`` (forward.py:696-730)."""
        Au = self.A(u, **kwargs)
        v = torch.randn_like(Au)
        Atv = self.A_adjoint(v, **kwargs)
        s1 = (v.conj() * Au).flatten().sum()
        s2 = (Atv * u.conj()).flatten().sum()
        return s1.conj() - s2

    def prox_l2(self, z, y, gamma, solver="CG", max_iter=None, tol=None, verbose=False, **kwargs):
        r""":math:`\arg\min_x \frac{\gamma}{2}\|Ax-y\|^2+\frac12\|x-z\|^2` by an iterative solver
        (forward.py:751-814)."""
        from ..optim.linear import least_squares, least_squares_implicit_backward

        if max_iter is not None:
            self.max_iter = max_iter
        if tol is not None:
            self.tol = tol
        if solver is not None:
            self.solver = solver
        if z is None or isinstance(z, (int, float)):
            z = torch.full_like(self.A_adjoint(y), fill_value=0.0 if z is None else float(z))
        if not self.implicit_backward_solver:
            return least_squares(self.A, self.A_adjoint, y, solver=solver, gamma=gamma, verbose=verbose, init=z, z=z,
                                 parallel_dim=[0], ATA=self.A_adjoint_A, AAT=self.A_A_adjoint,
                                 max_iter=self.max_iter, tol=self.tol, **kwargs)
        return least_squares_implicit_backward(self, y, z=z, init=z, solver=solver, gamma=gamma, verbose=verbose,
                                               max_iter=self.max_iter, tol=self.tol, parallel_dim=[0], **kwargs)

    def A_dagger(self, y, solver="CG", max_iter=None, tol=None, verbose=False, **kwargs):
        from ..optim.linear import least_squares, least_squares_implicit_backward

        if max_iter is not None:
            self.max_iter = max_iter
        if tol is not None:
            self.tol = tol
        if solver is not None:
            self.solver = solver
        if not self.implicit_backward_solver:
            return least_squares(self.A, self.A_adjoint, y, parallel_dim=[0], AAT=self.A_A_adjoint, verbose=verbose,
                                 ATA=self.A_adjoint_A, max_iter=self.max_iter, tol=self.tol, solver=self.solver,
                                 **kwargs)
        return least_squares_implicit_backward(self, y, z=None, init=None, parallel_dim=[0], gamma=1e8,
                                               verbose=verbose, max_iter=self.max_iter, tol=self.tol,
                                               solver=self.solver, **kwargs)


class DecomposablePhysics(LinearPhysics):
    r""":math:`A = U\,\mathrm{diag}(s)\,V^\top` with closed-form prox / pseudo-inverse (forward.py:1000-1252)."""

    def __init__(self, U=None, V_adjoint=None, img_size=None, U_adjoint=None, V=None, mask=1.0, device="cpu",
                 **kwargs):
        super().__init__(device=device, **kwargs)
        if U is None and U_adjoint is not None:
            raise ValueError("U must be provided if U_adjoint is provided.")
        if V_adjoint is None and V is not None:
            raise ValueError("V_adjoint must be provided if V is provided.")
        ident = lambda x: x
        self._V_adjoint = ident if V_adjoint is None else V_adjoint
        self._U = ident if U is None else U
        self._U_adjoint = ident if U is None else U_adjoint
        self._V = ident if V_adjoint is None else V
        self.img_size = img_size
        self.register_buffer("mask", mask if isinstance(mask, Tensor) else torch.tensor(mask))
        self.to(device)

    def A(self, x, mask=None, **kwargs) -> Tensor:
        self.update_parameters(mask=mask, **kwargs)
        return self.U(self.mask * self.V_adjoint(x))

    def A_adjoint(self, y, mask=None, **kwargs) -> Tensor:
        self.update_parameters(mask=mask, **kwargs)
        return self.V(torch.conj(self.mask) * self.U_adjoint(y))

    def A_A_adjoint(self, y, mask=None, **kwargs):
        self.update_parameters(mask=mask, **kwargs)
        return self.U(self.mask.conj() * self.mask * self.U_adjoint(y))

    def A_adjoint_A(self, x, mask=None, **kwargs):
        self.update_parameters(mask=mask, **kwargs)
        return self.V(self.mask.conj() * self.mask * self.V_adjoint(x))

    def U(self, x):
        return self._U(x)

    def V(self, x, **kwargs):
        if self._V is None:
            if self.img_size is None:
                raise ValueError("img_size must be set for using the automatic V implementation.")
            return adjoint_function(self.V_adjoint, (x.shape[0],) + tuple(self.img_size), device=x.device)(x, **kwargs)
        return self._V(x)

    def U_adjoint(self, x, **kwargs):
        if self._U_adjoint is None:
            if self.img_size is None:
                raise ValueError("img_size must be set for using the automatic U_adjoint implementation.")
            return adjoint_function(self.U, (x.shape[0],) + tuple(self.img_size), device=x.device)(x, **kwargs)
        return self._U_adjoint(x)

    def V_adjoint(self, x):
        return self._V_adjoint(x)

    def prox_l2(self, z, y, gamma, **kwargs):
        r""":math:`V\big(V^\top(A^\top y + z/\gamma) / (|s|^2 + 1/\gamma)\big)` (forward.py:1212-1234)."""
        from ..hip import elementwise as EW

        aty = self.A_adjoint(y)
        if (not isinstance(gamma, Tensor) and isinstance(self.mask, Tensor) and not self.mask.is_complex()
                and EW.eligible(aty, z, self.mask) and self.mask.numel() > 1):
            vb = self.V_adjoint(EW.lincomb(1.0, aty, 1.0 / float(gamma), z))
            if EW.eligible(vb) and self._mask_is_trailing(vb):      # b and the division on the HIP kernels (real masks: MRI)
                return self.V(EW.mask_solve(vb, self.mask, 1.0 / float(gamma)))
            return self.V(vb / (self.mask * self.mask + 1 / gamma))
        b = aty + 1 / gamma * z
        if isinstance(gamma, Tensor) and gamma.dim() < self.mask.dim():
            gamma = gamma[(...,) + (None,) * (self.mask.dim() - gamma.dim())]
            gamma = gamma.to(device=self.mask.device, dtype=self.mask.real.dtype)
        scaling = torch.conj(self.mask) * self.mask + 1 / gamma
        return self.V(self.V_adjoint(b) / scaling)

    def A_dagger(self, y, mask=None, **kwargs):
        r""":math:`V(U^\top y \cdot s^{-1}[s>10^{-5}])` (forward.py:1236-1252)."""
        self.update_parameters(mask=mask, **kwargs)
        from ..hip import elementwise as EW

        uy = self.U_adjoint(y)
        if (isinstance(self.mask, Tensor) and not self.mask.is_complex() and self.mask.numel() > 1 and EW.eligible(uy, self.mask)
                and self._mask_is_trailing(uy)):
            return self.V(EW.mask_solve(uy, self.mask, dagger=True))
        inv = torch.where(self.mask > 1e-5, self.mask.reciprocal(), 0.0)
        return self.V(uy * inv)

    def _mask_is_trailing(self, t):
        """the mask, up to leading singleton dimensions, is the trailing part of t's shape (shared by t's leading dimensions)"""
        shp = list(self.mask.shape)
        while len(shp) > 1 and shp[0] == 1:
            shp = shp[1:]
        return len(shp) <= t.dim() and tuple(t.shape[t.dim() - len(shp):]) == tuple(shp)


class Denoising(DecomposablePhysics):
    def __init__(self, noise_model: NoiseModel | None = None, device="cpu", **kwargs):
        from .noise import GaussianNoise

        super().__init__(noise_model=GaussianNoise(sigma=0.1) if noise_model is None else noise_model,
                         device=device, **kwargs)


def adjoint_function(A, input_size, device="cpu", dtype=torch.float):
    r"""Adjoint of a linear ``A`` through its vector-Jacobian product (forward.py:1302-1362).

    Returns a callable whose own autograd backward is ``A`` again, so the pair can be nested
    inside unfolded training graphs.
    """
    x = torch.ones(tuple(input_size), device=device, dtype=dtype)
    _, vjp = torch.func.vjp(A, x)
    batches = x.shape[0]

    class Adjoint(torch.autograd.Function):
        @staticmethod
        def forward(y):
            nb = y.shape[0]
            if nb > batches:
                raise ValueError("Batch size of A_adjoint input is larger than expected")
            if nb < batches:
                pad = torch.zeros((batches,) + tuple(y.shape[1:]), device=y.device, dtype=y.dtype)
                pad[:nb] = y
                return vjp(pad)[0][:nb]
            return vjp(y)[0]

        @staticmethod
        def setup_context(ctx, inputs, outputs):
            pass

        @staticmethod
        def backward(ctx, grad_output):
            return A(grad_output)

    return Adjoint.apply
