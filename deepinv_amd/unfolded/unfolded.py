"""Unfolded (trainable) optimisation (reference deepinv/unfolded/unfolded.py:9-226)."""
from __future__ import annotations

from types import MappingProxyType

import torch
import torch.nn as nn

from ..optim.optimizers import BaseOptim, create_iterator


class BaseUnfold(BaseOptim):
    """Same fixed-point loop with autograd enabled and trainable ``params_algo`` (unfolded.py:9-113)."""

    def __init__(self, iterator, params_algo=MappingProxyType({"lambda": 1.0, "stepsize": 1.0}), data_fidelity=None,
                 prior=None, max_iter=5, trainable_params=("lambda", "stepsize"), device=torch.device("cpu"), *args,
                 **kwargs):
        super().__init__(iterator, max_iter=max_iter, data_fidelity=data_fidelity, prior=prior,
                         params_algo=dict(params_algo), **kwargs)
        for k in trainable_params:
            if k in self.init_params_algo:
                self.init_params_algo[k] = nn.ParameterList(
                    [nn.Parameter(torch.tensor(el).float().to(device)) if not isinstance(el, torch.Tensor)
                     else nn.Parameter(el.float().to(device)) for el in self.init_params_algo[k]])
        self.init_params_algo = nn.ParameterDict(self.init_params_algo)
        self.params_algo = self.init_params_algo.copy()
        self.prior = nn.ModuleList(self.prior) if self.prior else None
        self.data_fidelity = nn.ModuleList(self.data_fidelity) if self.data_fidelity else None
        self.unfold = True   # BaseOptim.forward: graph through the loop, or through the equilibrium only with DEQ


def unfolded_builder(iteration, params_algo=MappingProxyType({"lambda": 1.0, "stepsize": 1.0}), data_fidelity=None,
                     prior=None, max_iter=5, trainable_params=("lambda", "stepsize"), device=torch.device("cpu"),
                     cost_fn=None, g_first=False, bregman_potential=None, **kwargs):
    """unfolded.py:116-226"""
    iterator = create_iterator(iteration, prior=prior, cost_fn=cost_fn, g_first=g_first)
    return BaseUnfold(iterator, max_iter=max_iter, trainable_params=trainable_params, has_cost=iterator.has_cost,
                      data_fidelity=data_fidelity, prior=prior, params_algo=dict(params_algo), device=device, **kwargs)


def DEQ_builder(iteration, params_algo=None, data_fidelity=None, prior=None, cost_fn=None, g_first=False,
                bregman_potential=None, max_iter_backward=50, anderson_acceleration_backward=False,
                history_size_backward=5, beta_anderson_acc_backward=1.0, eps_anderson_acc_backward=1e-4,
                jacobian_free=False, **kwargs):
    """Deep-equilibrium variant of :func:`unfolded_builder` (deep_equilibrium.py:150-250, deprecated upstream in favour
    of ``DEQ=...`` on the optimisers): the loop runs without a graph and gradients come from implicit differentiation
    at the fixed point (:class:`deepinv_amd.optim.DEQConfig`)."""
    from ..optim.data_fidelity import L2
    from ..optim.optimizers import DEQConfig

    if params_algo is None:
        params_algo = {"lambda": 1.0, "stepsize": 1.0, "g_param": 0.03}
    if data_fidelity is None:
        data_fidelity = L2()
    cfg = DEQConfig(jacobian_free=jacobian_free, anderson_acceleration_backward=anderson_acceleration_backward,
                    history_size_backward=history_size_backward, beta_backward=beta_anderson_acc_backward,
                    eps_backward=eps_anderson_acc_backward, max_iter_backward=max_iter_backward)
    iterator = create_iterator(iteration, prior=prior, cost_fn=cost_fn, g_first=g_first)
    return BaseUnfold(iterator, has_cost=iterator.has_cost, data_fidelity=data_fidelity, prior=prior,
                      params_algo=dict(params_algo), DEQ=cfg, **kwargs)
