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
        self.unfold = True   # BaseOptim.forward: keep the autograd graph through the loop


def unfolded_builder(iteration, params_algo=MappingProxyType({"lambda": 1.0, "stepsize": 1.0}), data_fidelity=None,
                     prior=None, max_iter=5, trainable_params=("lambda", "stepsize"), device=torch.device("cpu"),
                     cost_fn=None, g_first=False, bregman_potential=None, **kwargs):
    """unfolded.py:116-226"""
    iterator = create_iterator(iteration, prior=prior, cost_fn=cost_fn, g_first=g_first)
    return BaseUnfold(iterator, max_iter=max_iter, trainable_params=trainable_params, has_cost=iterator.has_cost,
                      data_fidelity=data_fidelity, prior=prior, params_algo=dict(params_algo), device=device, **kwargs)
