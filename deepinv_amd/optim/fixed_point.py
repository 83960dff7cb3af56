"""The iteration loop (reference deepinv/optim/fixed_point.py:13-406), including Anderson acceleration
(fixed_point.py:116-260; off by default at optimizers.py:296)."""
from __future__ import annotations

import torch
import torch.nn as nn


class _AndersonMixer:
    """Type-II Anderson mixing of the first iterate, per batch sample (fixed_point.py:116-260).

    Keeps rings of the last `history_size` points x_k and images T(x_k).  With G = T - X (rows = history slots) it
    solves the bordered system  [[0, 1^T], [1, G G^T + eps I]] [nu; p] = [1; 0]  (weights p sum to one and
    minimise |p^T G|^2 + eps |p|^2) and returns  beta p^T T + (1 - beta) p^T X.  Only slots [0, m) with
    m = min(it + 1, history_size) enter, in ring order, exactly as the reference indexes them.  Unless
    `full_backprop`, the history is detached and only the newest slot carries a gradient."""

    def __init__(self, cfg, x):
        self.cfg = cfg
        b, hs, d = x.shape[0], cfg.history_size, x[0].numel()
        opts = dict(dtype=x.dtype, device=x.device)
        self.xs = torch.zeros(b, hs, d, **opts)
        self.ts = torch.zeros(b, hs, d, **opts)
        self.H = torch.zeros(b, hs + 1, hs + 1, **opts)
        self.H[:, 0, 1:] = 1.0
        self.H[:, 1:, 0] = 1.0
        self.q = torch.zeros(b, hs + 1, 1, **opts)
        self.q[:, 0] = 1.0

    def mix(self, it, x_prev, tx_prev):
        cfg = self.cfg
        b = x_prev.shape[0]
        slot, m = it % cfg.history_size, min(it + 1, cfg.history_size)
        xf, tf = x_prev.reshape(b, -1), tx_prev.reshape(b, -1)
        if cfg.full_backprop:   # keep the graph through the whole history: rebuild the rings out of place
            self.xs = self.xs.clone()
            self.ts = self.ts.clone()
            H = self.H.clone()
            self.xs[:, slot] = xf
            self.ts[:, slot] = tf
            X, T = self.xs[:, :m], self.ts[:, :m]
        else:
            H = self.H.clone().detach()
            self.xs[:, slot] = xf.detach()
            self.ts[:, slot] = tf.detach()
            X_old, T_old = self.xs[:, :m].detach(), self.ts[:, :m].detach()
            sel = torch.zeros((1, m, 1), device=x_prev.device, dtype=x_prev.dtype)
            sel[:, slot, :] = 1
            X = X_old + sel * (xf[:, None, :] - X_old[:, slot:slot + 1, :])
            T = T_old + sel * (tf[:, None, :] - T_old[:, slot:slot + 1, :])
        G = T - X
        H[:, 1:m + 1, 1:m + 1] = torch.bmm(G, G.transpose(1, 2)) + cfg.eps * torch.eye(
            m, dtype=tx_prev.dtype, device=tx_prev.device)[None]
        p = torch.linalg.solve(H[:, :m + 1, :m + 1], self.q[:, :m + 1])[:, 1:m + 1, 0]
        x = cfg.beta * (p[:, None] @ T)[:, 0] + (1 - cfg.beta) * (p[:, None] @ X)[:, 0]
        self.H = H
        return x.view_as(x_prev)


class FixedPoint(nn.Module):
    def __init__(self, iterator=None, update_params_fn=None, update_data_fidelity_fn=None, update_prior_fn=None,
                 init_iterate_fn=None, init_metrics_fn=None, update_metrics_fn=None, check_conv_fn=None,
                 backtracking_check_fn=None, max_iter=50, early_stop=True, anderson_acceleration_config=None,
                 backtracking_config=None, verbose=False, show_progress_bar=False):
        super().__init__()
        self.anderson_acceleration_config = anderson_acceleration_config
        self._anderson = None
        self.iterator = iterator
        self.max_iter = max_iter
        self.early_stop = early_stop
        self.update_params_fn = update_params_fn
        self.update_data_fidelity_fn = update_data_fidelity_fn
        self.update_prior_fn = update_prior_fn
        self.init_iterate_fn = init_iterate_fn
        self.init_metrics_fn = init_metrics_fn
        self.update_metrics_fn = update_metrics_fn
        self.check_conv_fn = check_conv_fn
        self.backtracking_check_fn = backtracking_check_fn
        self.backtracking_config = backtracking_config
        self.verbose = verbose
        self.show_progress_bar = show_progress_bar
        self.backtracking_check = True

    def single_iteration(self, X, it, *args, **kwargs):
        """fixed_point.py:363-406"""
        cur_params = self.update_params_fn(it) if self.update_params_fn else None
        cur_df = self.update_data_fidelity_fn(it) if self.update_data_fidelity_fn else None
        cur_prior = self.update_prior_fn(it) if self.update_prior_fn else None
        X_prev = X
        X = self.iterator(X_prev, cur_df, cur_prior, cur_params, *args, **kwargs)
        if self._anderson is not None:   # fixed_point.py:389-398
            x = self._anderson.mix(it, X_prev["est"][0], X["est"][0])
            cost_fn = self.iterator.cost_fn
            F = (cost_fn(x, cur_df, cur_prior, cur_params, *args)
                 if cost_fn is not None and cur_df is not None and cur_prior is not None else None)
            est = list(X["est"])
            est[0] = x
            X = {"est": est, "cost": F}
        self.backtracking_check = self.backtracking_check_fn(X_prev, X) if self.backtracking_check_fn else True
        return X if self.backtracking_check else X_prev

    def forward(self, *args, init=None, compute_metrics=False, x_gt=None, **kwargs):
        """fixed_point.py:262-361"""
        X = self.init_iterate_fn(*args, init, cost_fn=self.iterator.cost_fn) if self.init_iterate_fn else None
        metrics = self.init_metrics_fn(X, x_gt=x_gt) if self.init_metrics_fn and compute_metrics else None
        self.backtracking_check = True
        failed = 0
        self._anderson = (_AndersonMixer(self.anderson_acceleration_config, X["est"][0])
                          if self.anderson_acceleration_config is not None else None)
        for it in range(self.max_iter):
            X_prev = X
            X = self.single_iteration(X, it, *args, **kwargs)
            if self.backtracking_check or self.backtracking_config is None:
                failed = 0
                metrics = (self.update_metrics_fn(metrics, X_prev, X, x_gt=x_gt)
                           if self.update_metrics_fn and compute_metrics else None)
                if self.early_stop and self.check_conv_fn is not None and it > 1 and self.check_conv_fn(it, X_prev, X):
                    break
            else:
                failed += 1
                if failed >= self.backtracking_config.max_iter:
                    break
        return X, metrics
