"""``BaseOptim`` and the PGD / HQS front-ends
(reference deepinv/optim/optimizers.py:94-881, 884-1058, 1459-1734)."""
from __future__ import annotations

import warnings
from collections.abc import Iterable
from contextlib import nullcontext
from dataclasses import dataclass
from types import MappingProxyType

import torch
import torch.nn as nn

from ..models.base import Reconstructor
from . import optim_iterators as _its
from .data_fidelity import ZeroFidelity
from .fixed_point import FixedPoint
from .optim_iterators import OptimIterator
from .prior import ZeroPrior


@dataclass
class BacktrackingConfig:
    gamma: float = 0.1
    eta: float = 0.9
    max_iter: int = 20


def _psnr(x, y, max_pixel=1.0):
    mse = (x - y).pow(2).mean(dim=tuple(range(1, x.ndim)))
    return 10 * torch.log10(max_pixel ** 2 / mse)


class BaseOptim(Reconstructor):
    """Fixed-point driver for splitting algorithms (optimizers.py:94-881)."""

    def __init__(self, iterator, params_algo=MappingProxyType({"lambda": 1.0, "stepsize": 1.0}), data_fidelity=None,
                 prior=None, max_iter=100, crit_conv="residual", thres_conv=1e-5, early_stop=False, has_cost=False,
                 backtracking=None, custom_metrics=None, custom_init=None, get_output=lambda X: X["est"][0],
                 unfold=False, trainable_params=None, verbose=False,
                 show_progress_bar=False, **kwargs):
        super().__init__()
        # the reference's BaseOptim also takes DEQ= and anderson_acceleration= (optimizers.py:295-296); neither is on the
        # accelerated path - refuse them loudly instead of running a plain loop under their name
        for k in ("DEQ", "anderson_acceleration"):
            if kwargs.pop(k, None) not in (None, False):
                raise NotImplementedError(f"deepinv_amd.optim: {k} is not implemented (only the plain fixed-point loop, "
                                          "optionally unfolded, is on the accelerated path)")
        if kwargs:
            raise TypeError(f"BaseOptim got unexpected keyword arguments {sorted(kwargs)}")
        self.early_stop, self.crit_conv, self.verbose = early_stop, crit_conv, verbose
        self.show_progress_bar, self.max_iter = show_progress_bar, max_iter
        if isinstance(backtracking, bool):
            self.backtracking = backtracking
            self.backtracking_config = BacktrackingConfig() if backtracking else None
        else:
            self.backtracking = backtracking is not None
            self.backtracking_config = backtracking or BacktrackingConfig()
        self.has_converged = False
        self.thres_conv, self.custom_metrics, self.custom_init = thres_conv, custom_metrics, custom_init
        self.get_output, self.unfold = get_output, unfold
        self.prior = [ZeroPrior()] if prior is None else ([prior] if not isinstance(prior, Iterable) else prior)
        self.data_fidelity = ([ZeroFidelity()] if data_fidelity is None else
                              ([data_fidelity] if not isinstance(data_fidelity, Iterable) else data_fidelity))
        self.has_cost = self.prior[0].explicit_prior
        iterator.has_cost = self.has_cost
        params_algo = dict(params_algo)
        if "g_param" not in params_algo:
            params_algo["g_param"] = params_algo.pop("sigma_denoiser", None)
        if "lambda" not in params_algo:
            params_algo["lambda"] = params_algo.pop("lambda_reg", 1.0)
        params_algo.setdefault("beta", 1.0)
        for key, value in params_algo.items():
            if not isinstance(value, Iterable):
                params_algo[key] = [value]
            elif 1 < len(value) < self.max_iter:
                raise ValueError(f"The number of elements in the parameter {key} is inferior to max_iter.")
        if "stepsize" in params_algo and len(params_algo["stepsize"]) > 1 and self.backtracking:
            self.backtracking = None
            warnings.warn("Backtracking impossible when stepsize is predefined as a list. Setting backtracking to False.")
        if not self.has_cost and self.backtracking:
            self.backtracking = None
            warnings.warn("Backtracking impossible when no cost function is given. Setting backtracking to False.")
        self.init_params_algo = params_algo
        if self.unfold:
            if trainable_params is not None:
                trainable_params = [{"lambda_reg": "lambda", "sigma_denoiser": "g_param"}.get(p, p) for p in trainable_params]
            else:
                trainable_params = list(params_algo.keys())
            for k in trainable_params:
                if k in self.init_params_algo:
                    self.init_params_algo[k] = nn.ParameterList(
                        [nn.Parameter(torch.tensor(el).float()) if not isinstance(el, torch.Tensor)
                         else nn.Parameter(el.float()) for el in self.init_params_algo[k]])
            self.params_algo = nn.ParameterDict(self.init_params_algo)
            self.init_params_algo = self.params_algo.copy()
            self.prior = nn.ModuleList(self.prior)
            self.data_fidelity = nn.ModuleList(self.data_fidelity)
        self.fixed_point = FixedPoint(
            iterator=iterator, update_params_fn=self.update_params_fn,
            update_data_fidelity_fn=self.update_data_fidelity_fn, update_prior_fn=self.update_prior_fn,
            backtracking_check_fn=self.backtracking_check_fn, check_conv_fn=self.check_conv_fn,
            init_metrics_fn=self.init_metrics_fn, init_iterate_fn=self.init_iterate_fn,
            update_metrics_fn=self.update_metrics_fn, max_iter=max_iter, early_stop=early_stop,
            backtracking_config=self.backtracking_config, verbose=verbose,
            show_progress_bar=show_progress_bar,
            # device-side early stop only for the stock criterion: a subclass that overrides check_conv_fn keeps the host
            # path (its override is what decides); the threshold is read live
            conv_crit_fn=self.conv_crit if type(self).check_conv_fn is BaseOptim.check_conv_fn else None,
            thres_conv=self._get_thres_conv, on_converged=self._set_converged)

    # ---- per-iteration lookups (optimizers.py:464-500)
    def update_params_fn(self, it):
        return {k: (v[it] if len(v) > 1 else v[0]) for k, v in self.params_algo.items()}

    def update_prior_fn(self, it):
        return self.prior[it] if len(self.prior) > 1 else self.prior[0]

    def update_data_fidelity_fn(self, it):
        return self.data_fidelity[it] if len(self.data_fidelity) > 1 else self.data_fidelity[0]

    def init_iterate_fn(self, y, physics, init=None, cost_fn=None):
        """default x0 = z0 = A^T y (optimizers.py:502-587)"""
        self.params_algo = self.init_params_algo.copy()
        init = init if init is not None else self.custom_init
        if init is not None:
            if callable(init):
                init = init(y, physics)
            if isinstance(init, torch.Tensor):
                X = {"est": (init,)}
            elif isinstance(init, tuple):
                X = {"est": init}
            elif isinstance(init, dict):
                X = init
            else:
                raise ValueError(f"Custom initial iterate must be a torch.Tensor, a tuple, or a dict. Got {type(init)}.")
        else:
            # the reference evaluates A^T y twice here and once more per PGD iteration; one evaluation serves the
            # whole call (handed to the data-fidelity step through the FixedPoint call context)
            aty = physics.A_adjoint(y)
            ctx = getattr(self.fixed_point, "call_ctx", None)
            if ctx is not None and not torch.is_grad_enabled():
                ctx.put(physics, y, aty)
            X = {"est": (aty, aty.clone())}
        X["cost"] = (cost_fn(X["est"][0], self.update_data_fidelity_fn(0), self.update_prior_fn(0),
                             self.update_params_fn(0), y, physics) if self.has_cost and cost_fn is not None else None)
        return X

    def init_metrics_fn(self, X_init, x_gt=None):
        x0 = self.get_output(X_init)
        self.batch_size = x0.shape[0]
        m = {"psnr": [[_psnr(x0[i:i + 1], x_gt[i:i + 1]).cpu().item()] if x_gt is not None else []
                      for i in range(self.batch_size)]}
        if self.has_cost:
            m["cost"] = [[] for _ in range(self.batch_size)]
        m["residual"] = [[] for _ in range(self.batch_size)]
        if self.custom_metrics is not None:
            for name in self.custom_metrics:
                m[name] = [[] for _ in range(self.batch_size)]
        return m

    def update_metrics_fn(self, metrics, X_prev, X, x_gt=None):
        if metrics is None:
            return metrics
        x_prev, x = self.get_output(X_prev), self.get_output(X)
        for i in range(self.batch_size):
            metrics["residual"][i].append(((x_prev[i] - x[i]).norm() / (x[i].norm() + 1e-6)).detach().cpu().item())
            if x_gt is not None:
                metrics["psnr"][i].append(_psnr(x[i:i + 1], x_gt[i:i + 1]).cpu().item())
            if self.has_cost:
                metrics["cost"][i].append(X["cost"][i].detach().cpu().item())
            if self.custom_metrics is not None:
                for name, fn in self.custom_metrics.items():
                    metrics[name][i].append(fn(metrics[name], x_prev[i], x[i]))
        return metrics

    def backtracking_check_fn(self, X_prev, X):
        if not (self.backtracking and self.has_cost and X_prev is not None):
            return True
        x_prev = X_prev["est"][0].reshape(X_prev["est"][0].shape[0], -1)
        x = X["est"][0].reshape(X["est"][0].shape[0], -1)
        diff_F = (X_prev["cost"] - X["cost"]).mean()
        diff_x = torch.linalg.vector_norm(x - x_prev, dim=-1, ord=2).pow(2).mean()
        stepsize = self.params_algo["stepsize"][0]
        if diff_F < (self.backtracking_config.gamma / stepsize) * diff_x:
            self.params_algo["stepsize"] = [self.backtracking_config.eta * stepsize]
            return False
        return True

    def _get_thres_conv(self):
        return self.thres_conv       # (a bound method, not a lambda: copy.deepcopy re-binds it to the copy)

    def conv_crit(self, X_prev, X):
        """the convergence criterion as a tensor (a device scalar on the HIP path; optimizers.py:703-739)"""
        if self.crit_conv == "residual":
            x_prev = self.get_output(X_prev).reshape(self.get_output(X_prev).shape[0], -1)
            x = self.get_output(X).reshape(x_prev.shape[0], -1)
            return ((x_prev - x).norm(p=2, dim=-1) / (x.norm(p=2, dim=-1) + 1e-6)).mean()
        if self.crit_conv == "cost":
            return ((X_prev["cost"] - X["cost"]).norm(dim=-1) / (X["cost"].norm(dim=-1) + 1e-6)).mean()
        raise ValueError("convergence criteria not implemented")

    def check_conv_fn(self, it, X_prev, X):
        """host-side decision (one sync): used with metrics / backtracking / CPU tensors; on the HIP path without them
        FixedPoint decides on the device from `conv_crit` and never reads the criterion back per iteration"""
        if self.conv_crit(X_prev, X) < self.thres_conv:
            self.has_converged = True
            return True
        return False

    def _set_converged(self, flag):
        self._converged_flag = flag          # device tensor: read (one sync) only when `has_converged` is looked at

    @property
    def has_converged(self):
        f = self.__dict__.get("_converged_flag")
        if f is not None:
            self.__dict__["_has_converged"] = bool(f)
            self.__dict__["_converged_flag"] = None
        return self.__dict__.get("_has_converged", False)

    @has_converged.setter
    def has_converged(self, v):
        self.__dict__["_converged_flag"] = None
        self.__dict__["_has_converged"] = bool(v)

    def forward(self, y, physics, init=None, x_gt=None, compute_metrics=False, **kwargs):
        """no_grad unless unfolding (optimizers.py:826-881)"""
        with (torch.no_grad() if not self.unfold else nullcontext()):
            X, metrics = self.fixed_point(y, physics, init=init, x_gt=x_gt, compute_metrics=compute_metrics, **kwargs)
        x = self.get_output(X)
        return (x, metrics) if compute_metrics else x


def create_iterator(iteration, prior=None, cost_fn=None, g_first=False, bregman_potential=None, **kwargs):
    """optimizers.py:884-971"""
    if prior is None:
        prior = ZeroPrior()
    explicit = prior[0].explicit_prior if isinstance(prior, list) else prior.explicit_prior
    if cost_fn is None and explicit:
        def cost_fn(x, data_fidelity, prior, cur_params, y, physics):
            pv = prior(x, cur_params["g_param"])
            lam = cur_params["lambda"]
            reg = lam * pv if (pv.dim() == 0 or isinstance(lam, float)) else (lam.flatten(1, -1).to(pv.device) * pv.flatten(1, -1))
            return data_fidelity(x, y, physics) + (reg if reg.dim() == 0 else reg.sum())
        has_cost = True
    else:
        has_cost = False
    if isinstance(iteration, str):
        cls = getattr(_its, iteration + "Iteration", None)
        if cls is None:
            raise NotImplementedError(f"iteration '{iteration}' is not implemented by deepinv_amd.optim (available: PGD, HQS); "
                                      "pass an OptimIterator instance for anything else")
        return cls(g_first=g_first, cost_fn=cost_fn, has_cost=has_cost)
    return iteration


def optim_builder(iteration, max_iter=100, params_algo=MappingProxyType({"lambda": 1.0, "stepsize": 1.0, "g_param": 0.05}),
                  data_fidelity=None, prior=None, cost_fn=None, g_first=False, bregman_potential=None, **kwargs):
    """optimizers.py:974-1058"""
    iterator = create_iterator(iteration, prior=prior, cost_fn=cost_fn, g_first=g_first)
    return BaseOptim(iterator, has_cost=iterator.has_cost, data_fidelity=data_fidelity, prior=prior,
                     params_algo=dict(params_algo), max_iter=max_iter, **kwargs).eval()


def _front_end(iteration_cls):
    class _Algo(BaseOptim):
        def __init__(self, data_fidelity=None, prior=None, lambda_reg=1.0, stepsize=1.0, g_param=None,
                     sigma_denoiser=None, max_iter=100, crit_conv="residual", thres_conv=1e-5, early_stop=False,
                     backtracking=None, custom_metrics=None, custom_init=None, g_first=False, unfold=False,
                     trainable_params=None, cost_fn=None, params_algo=None, **kwargs):
            if g_param is None and sigma_denoiser is not None:
                g_param = sigma_denoiser
            if params_algo is None:
                params_algo = {"lambda": lambda_reg, "stepsize": stepsize, "g_param": g_param}
            super().__init__(iteration_cls(g_first=g_first, cost_fn=cost_fn), data_fidelity=data_fidelity, prior=prior,
                             params_algo=params_algo, max_iter=max_iter, crit_conv=crit_conv, thres_conv=thres_conv,
                             early_stop=early_stop, backtracking=backtracking, custom_metrics=custom_metrics,
                             custom_init=custom_init, unfold=unfold, trainable_params=trainable_params, **kwargs)
    return _Algo


PGD = _front_end(_its.PGDIteration)    # optimizers.py:1596-1734
PGD.__name__ = PGD.__qualname__ = "PGD"
HQS = _front_end(_its.HQSIteration)    # optimizers.py:1459-1593
HQS.__name__ = HQS.__qualname__ = "HQS"
