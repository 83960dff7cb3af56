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


class CallContext:
    """Scratch shared by the steps of ONE `FixedPoint.forward` call (created at its start, dropped at its end).
    It holds strong references to the measurement and the operator it was filled for, so a hit means "the same
    objects", never "the same address"."""

    def __init__(self):
        self.y = self.physics = self.aty = None

    def adjoint(self, physics, y):
        """A^T y is the same tensor in every iteration of a reconstruction (the reference recomputes it each
        time, data_fidelity.py:335-338): computed once per call, identical value, one adjoint less per iteration."""
        if self.aty is None or self.y is not y or self.physics is not physics:
            self.aty, self.y, self.physics = physics.A_adjoint(y), y, physics
        return self.aty

    def put(self, physics, y, aty):
        self.aty, self.y, self.physics = aty, y, physics


def _fused_l2_gradient_step(x, data_fidelity, stepsize, y, physics, ctx=None):
    """x - gamma/sigma^2 * (A^T A x - A^T y) as ONE pass over the image (hand-written kernel) when the fidelity is
    L2, the physics linear, the operands live on the HIP device and no autograd graph is recorded; same arithmetic
    as fStepPGD + L2.grad (pgd.py:137-139, data_fidelity.py:335-338); A^T y comes from the per-call context."""
    from ..physics.forward import LinearPhysics
    from .data_fidelity import L2

    if type(data_fidelity) is not L2 or not isinstance(physics, LinearPhysics) or not isinstance(x, torch.Tensor):
        return None
    if torch.is_grad_enabled() or isinstance(stepsize, torch.Tensor) or ctx is None:
        return None
    AtAx = physics.A_adjoint_A(x)
    Aty = ctx.adjoint(physics, y)
    from ..hip import elementwise as ew

    if not (x.is_cuda and ew.eligible(x, AtAx, Aty) and AtAx.shape == x.shape == Aty.shape):
        return x - stepsize * (data_fidelity.norm * (AtAx - Aty))   # the reference's expression, A^T y reused
    g = float(stepsize) * float(data_fidelity.norm)
    return ew.lincomb(1.0, x, -g, AtAx, g, Aty)


# ------------------------------------------------------------------ PGD (pgd.py:12-176)
class fStepPGD(fStep):
    def forward(self, x, cur_data_fidelity, cur_params, y, physics):
        if not self.g_first:
            fused = _fused_l2_gradient_step(x, cur_data_fidelity, cur_params["stepsize"], y, physics,
                                            getattr(self, "call_ctx", None))
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
