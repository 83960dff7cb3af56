"""The iteration loop (reference deepinv/optim/fixed_point.py:262-406)."""
from __future__ import annotations

import torch.nn as nn

from .optim_iterators import CallContext


class FixedPoint(nn.Module):
    def __init__(self, iterator=None, update_params_fn=None, update_data_fidelity_fn=None, update_prior_fn=None,
                 init_iterate_fn=None, init_metrics_fn=None, update_metrics_fn=None, check_conv_fn=None,
                 backtracking_check_fn=None, max_iter=50, early_stop=True, backtracking_config=None, verbose=False,
                 show_progress_bar=False):
        super().__init__()
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
        self.call_ctx = None

    def single_iteration(self, X, it, *args, **kwargs):
        """fixed_point.py:363-406"""
        cur_params = self.update_params_fn(it) if self.update_params_fn else None
        cur_df = self.update_data_fidelity_fn(it) if self.update_data_fidelity_fn else None
        cur_prior = self.update_prior_fn(it) if self.update_prior_fn else None
        X_prev = X
        X = self.iterator(X_prev, cur_df, cur_prior, cur_params, *args, **kwargs)
        self.backtracking_check = self.backtracking_check_fn(X_prev, X) if self.backtracking_check_fn else True
        return X if self.backtracking_check else X_prev

    def _set_ctx(self, ctx):
        """hand the per-call scratch (A^T y of THIS call) to the data-fidelity step; nothing outlives the call"""
        self.call_ctx = ctx
        f_step = getattr(self.iterator, "f_step", None)
        if f_step is not None:
            f_step.call_ctx = ctx

    # ------------------------------------------------------------------ HIP-graph replay of the loop
    def _graph_ok(self, X, compute_metrics):
        """one iteration can be captured once and replayed when nothing in it depends on the iteration index or on
        host-side decisions: no autograd, no metrics, no early stop, no backtracking, constant parameters / prior /
        data fidelity, device tensors.  Opt-in: DINV_LOOP_GRAPH=1 (or `self.use_graph = True`)."""
        import os

        import torch

        if not (getattr(self, "use_graph", False) or os.environ.get("DINV_LOOP_GRAPH", "0") == "1"):
            return False
        if torch.is_grad_enabled() or compute_metrics or self.backtracking_config is not None or self.max_iter < 3:
            return False
        if self.early_stop and self.check_conv_fn is not None:
            return False
        if X is None or not all(isinstance(t, torch.Tensor) and t.is_cuda for t in X["est"]) or X.get("cost") is not None:
            return False
        first = [f(0) if f else None for f in (self.update_params_fn, self.update_data_fidelity_fn, self.update_prior_fn)]
        for it in range(1, self.max_iter):
            cur = [f(it) if f else None for f in (self.update_params_fn, self.update_data_fidelity_fn, self.update_prior_fn)]
            if cur[0] != first[0] or cur[1] is not first[1] or cur[2] is not first[2]:
                return False
        return True

    def _run_graph(self, X, *args, **kwargs):
        """iteration 0 runs eagerly (it builds every plan / workspace), iteration 1 is captured into a HIP graph whose
        inputs are static copies of the iterate, and the graph is replayed for the remaining iterations: one host call
        per iteration instead of ~80 kernel launches (fixed_point.py:324-361 with the same arithmetic)."""
        import torch

        X = self.single_iteration(X, 0, *args, **kwargs)
        static = [t.clone() for t in X["est"]]
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                Xo = self.single_iteration({"est": tuple(static), "cost": None}, 1, *args, **kwargs)
                for s, o in zip(static, Xo["est"]):
                    s.copy_(o)
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(1, self.max_iter):
            graph.replay()
        out = {"est": tuple(s.clone() for s in static), "cost": None}
        del graph
        return out

    def forward(self, *args, init=None, compute_metrics=False, x_gt=None, **kwargs):
        """fixed_point.py:262-361"""
        self._set_ctx(CallContext())
        try:
            X = self.init_iterate_fn(*args, init, cost_fn=self.iterator.cost_fn) if self.init_iterate_fn else None
            metrics = self.init_metrics_fn(X, x_gt=x_gt) if self.init_metrics_fn and compute_metrics else None
            self.backtracking_check = True
            failed = 0
            if self._graph_ok(X, compute_metrics):
                return self._run_graph(X, *args, **kwargs), metrics
            for it in range(self.max_iter):
                X_prev = X
                X = self.single_iteration(X, it, *args, **kwargs)
                if self.backtracking_check or self.backtracking_config is None:
                    failed = 0
                    metrics = (self.update_metrics_fn(metrics, X_prev, X, x_gt=x_gt)
                               if self.update_metrics_fn and compute_metrics else None)
                    if (self.early_stop and self.check_conv_fn is not None and it > 1
                            and self.check_conv_fn(it, X_prev, X)):
                        break
                else:
                    failed += 1
                    if failed >= self.backtracking_config.max_iter:
                        break
        finally:
            self._set_ctx(None)
        return X, metrics
