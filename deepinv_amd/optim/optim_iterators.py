"""One splitting step (reference deepinv/optim/optim_iterators/{optim_iterator,pgd,hqs}.py)."""
from __future__ import annotations

import torch
import torch.nn as nn


class fStep(nn.Module):
    def __init__(self, g_first=False, **kwargs):
        super().__init__()
        self.g_first = g_first


class gStep(nn.Module):
    def __init__(self, g_first=False, **kwargs):
        super().__init__()
        self.g_first = g_first


class OptimIterator(nn.Module):
    """f-step, g-step (order by ``g_first``), relaxation (optim_iterator.py:13-125)."""

    def __init__(self, g_first=False, cost_fn=None, has_cost=True, **kwargs):
        super().__init__()
        self.g_first = g_first
        self.has_cost = has_cost
        if cost_fn is None and has_cost:
            cost_fn = objective_function
        self.cost_fn = cost_fn
        self.f_step = fStep(g_first=g_first)
        self.g_step = gStep(g_first=g_first)

    def relaxation_step(self, u, v, beta, *args, **kwargs):
        """beta*u + (1-beta)*v (optim_iterator.py:76-87); a no-op for the default beta = 1, one fused pass otherwise"""
        if isinstance(beta, (int, float)) and isinstance(u, torch.Tensor) and u.is_cuda and not torch.is_grad_enabled():
            if float(beta) == 1.0:
                return u
            from ..hip import elementwise as ew

            if ew.eligible(u, v) and u.shape == v.shape:
                return ew.lincomb(float(beta), u, 1.0 - float(beta), v)
        return beta * u + (1 - beta) * v

    def forward(self, X, cur_data_fidelity, cur_prior, cur_params, y, physics, *args, **kwargs):
        x_prev = X["est"][0]
        if not self.g_first:
            z = self.f_step(x_prev, cur_data_fidelity, cur_params, y, physics, *args, **kwargs)
            x = self.g_step(z, cur_prior, cur_params, *args, **kwargs)
        else:
            z = self.g_step(x_prev, cur_prior, cur_params, *args, **kwargs)
            x = self.f_step(z, cur_data_fidelity, cur_params, y, physics, *args, **kwargs)
        x = self.relaxation_step(x, x_prev, cur_params["beta"], *args, **kwargs)
        F = (self.cost_fn(x, cur_data_fidelity, cur_prior, cur_params, y, physics)
             if self.cost_fn is not None and self.has_cost and cur_data_fidelity is not None and cur_prior is not None
             else None)
        return {"est": (x, z), "cost": F}


def objective_function(x, data_fidelity, prior, cur_params, y, physics):
    """f(x) + lambda g(x) (deepinv/optim/utils.py objective_function)"""
    return data_fidelity(x, y, physics) + cur_params["lambda"] * prior(x, cur_params["g_param"])


_ATY_CACHE = {}  # id(physics) -> (key, A^T y): one entry per operator, replaced when anything in the key changes


def _cached_adjoint(physics, y):
    """A^T y is the same tensor in every iteration of a reconstruction (the reference recomputes it each time,
    data_fidelity.py:335-338).  Reuse it while y and every parameter / buffer of the operator are untouched
    (storage address + in-place version counter): identical value, one adjoint less per iteration."""
    if not isinstance(y, torch.Tensor) or not isinstance(physics, torch.nn.Module):
        return physics.A_adjoint(y)
    key = (y.data_ptr(), y._version, tuple(y.shape), tuple(y.stride()), y.device,
           tuple((t.data_ptr(), t._version) for t in list(physics.buffers()) + list(physics.parameters())))
    hit = _ATY_CACHE.get(id(physics))
    if hit is not None and hit[0] == key and hit[2]() is physics:
        return hit[1]
    import weakref

    val = physics.A_adjoint(y)
    if len(_ATY_CACHE) > 16:
        _ATY_CACHE.clear()
    _ATY_CACHE[id(physics)] = (key, val, weakref.ref(physics))
    return val


def _fused_l2_gradient_step(x, data_fidelity, stepsize, y, physics):
    """x - gamma/sigma^2 * (A^T A x - A^T y) as ONE pass over the image (hand-written kernel) when the fidelity is
    L2, the physics linear, the operands live on the HIP device and no autograd graph is recorded; same arithmetic
    as fStepPGD + L2.grad (pgd.py:137-139, data_fidelity.py:335-338); A^T y is reused across iterations."""
    from ..physics.forward import LinearPhysics
    from .data_fidelity import L2

    if type(data_fidelity) is not L2 or not isinstance(physics, LinearPhysics) or not isinstance(x, torch.Tensor):
        return None
    if not x.is_cuda or isinstance(stepsize, torch.Tensor):
        return None
    from ..hip import elementwise as ew

    if not ew.eligible(x) or torch.is_grad_enabled():
        return None
    AtAx = physics.A_adjoint_A(x)
    Aty = _cached_adjoint(physics, y)
    if not (ew.eligible(AtAx, Aty) and AtAx.shape == x.shape == Aty.shape):
        return x - stepsize * data_fidelity.norm * (AtAx - Aty)
    g = float(stepsize) * float(data_fidelity.norm)
    return ew.lincomb(1.0, x, -g, AtAx, g, Aty)


# ------------------------------------------------------------------ PGD (pgd.py:12-176)
class fStepPGD(fStep):
    def forward(self, x, cur_data_fidelity, cur_params, y, physics):
        if not self.g_first:
            fused = _fused_l2_gradient_step(x, cur_data_fidelity, cur_params["stepsize"], y, physics)
            if fused is not None:
                return fused
            return x - cur_params["stepsize"] * cur_data_fidelity.grad(x, y, physics)
        return cur_data_fidelity.prox(x, y, physics, gamma=cur_params["stepsize"])


class gStepPGD(gStep):
    def forward(self, x, cur_prior, cur_params):
        if not self.g_first:
            return cur_prior.prox(x, cur_params["g_param"], gamma=cur_params["lambda"] * cur_params["stepsize"])
        return x - cur_params["lambda"] * cur_params["stepsize"] * cur_prior.grad(x, cur_params["g_param"])


class PGDIteration(OptimIterator):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.g_step = gStepPGD(**kwargs)
        self.f_step = fStepPGD(**kwargs)


class FISTAIteration(OptimIterator):
    """pgd.py:36-108"""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.g_step = gStepPGD(**kwargs)
        self.f_step = fStepPGD(**kwargs)

    def forward(self, X, cur_data_fidelity, cur_prior, cur_params, y, physics, *args, **kwargs):
        x_prev, z_prev = X["est"][0], X["est"][1]
        k = 0 if "it" not in X else X["it"]
        a = cur_params["a"]
        alpha = (k + a - 1) / (k + a)
        if not self.g_first:
            z = self.f_step(z_prev, cur_data_fidelity, cur_params, y, physics)
            x = self.g_step(z, cur_prior, cur_params)
        else:
            z = self.g_step(z_prev, cur_prior, cur_params)
            x = self.f_step(z, cur_data_fidelity, cur_params, y, physics)
        z = x + alpha * (x - x_prev)
        F = (self.cost_fn(x, cur_data_fidelity, cur_prior, cur_params, y, physics)
             if self.has_cost and self.cost_fn is not None and cur_data_fidelity is not None and cur_prior is not None
             else None)
        return {"est": (x, z), "cost": F, "it": k + 1}


# ------------------------------------------------------------------ HQS (hqs.py:11-95)
class fStepHQS(fStep):
    def forward(self, x, cur_data_fidelity, cur_params, y, physics, *args, **kwargs):
        return cur_data_fidelity.prox(x, y, physics, *args, gamma=cur_params["stepsize"], **kwargs)


class gStepHQS(gStep):
    def forward(self, x, cur_prior, cur_params, *args, **kwargs):
        return cur_prior.prox(x, cur_params["g_param"], *args, gamma=cur_params["lambda"] * cur_params["stepsize"],
                              **kwargs)


class HQSIteration(OptimIterator):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.g_step = gStepHQS(**kwargs)
        self.f_step = fStepHQS(**kwargs)


# ------------------------------------------------------------------ GD (gradient_descent.py)
class fStepGD(fStep):
    def forward(self, x, cur_data_fidelity, cur_params, y, physics):
        return cur_data_fidelity.grad(x, y, physics)


class gStepGD(gStep):
    def forward(self, x, cur_prior, cur_params):
        return cur_params["lambda"] * cur_prior.grad(x, cur_params["g_param"])


class GDIteration(OptimIterator):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.g_step = gStepGD(**kwargs)
        self.f_step = fStepGD(**kwargs)

    def forward(self, X, cur_data_fidelity, cur_prior, cur_params, y, physics, *args, **kwargs):
        x_prev = X["est"][0]
        grad = cur_params["stepsize"] * (self.g_step(x_prev, cur_prior, cur_params)
                                         + self.f_step(x_prev, cur_data_fidelity, cur_params, y, physics))
        x = x_prev - grad
        F = (self.cost_fn(x, cur_data_fidelity, cur_prior, cur_params, y, physics)
             if self.has_cost and self.cost_fn is not None else None)
        return {"est": (x,), "cost": F}
