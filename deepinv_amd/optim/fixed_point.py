"""The iteration loop (reference deepinv/optim/fixed_point.py:262-406)."""
from __future__ import annotations

import torch.nn as nn

from .optim_iterators import CallContext


_CAPTURE_STREAMS: dict = {}


def _capture_stream(device):
    """ONE capture stream per device, reused by every call: per-stream scratch of the kernels (the F(4x4) tail-split workspace,
    hip/drunet.py: winograd4_workspace) is then created once - by an eager launch on this stream before the first capture - instead
    of being allocated inside a capture for every fresh stream of the pool"""
    import torch

    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _CAPTURE_STREAMS:
        _CAPTURE_STREAMS[idx] = torch.cuda.Stream(device)
    return _CAPTURE_STREAMS[idx]


class FixedPoint(nn.Module):
    def __init__(self, iterator=None, update_params_fn=None, update_data_fidelity_fn=None, update_prior_fn=None,
                 init_iterate_fn=None, init_metrics_fn=None, update_metrics_fn=None, check_conv_fn=None,
                 backtracking_check_fn=None, max_iter=50, early_stop=True, backtracking_config=None, verbose=False,
                 show_progress_bar=False, conv_crit_fn=None, thres_conv=None, on_converged=None):
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
        # device-side early stop (optimizers.py:703-739 without the per-iteration host sync): `conv_crit_fn(X_prev, X)`
        # returns the criterion as a device scalar, `on_converged(flag)` receives the device flag at the end of the call
        # `thres_conv` may be a callable: the owner's threshold is then read live at every call (a user who changes
        # `model.thres_conv` after construction gets the new value on the device path too)
        self.conv_crit_fn, self.thres_conv, self.on_converged = conv_crit_fn, thres_conv, on_converged
        self.poll_every = 4

    def _thres(self):
        return self.thres_conv() if callable(self.thres_conv) else self.thres_conv

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
        host-side decisions: no autograd, no metrics, no host-side early stop, no backtracking, constant parameters /
        prior / data fidelity, device tensors.  Opt-in: `model.fixed_point.use_graph = True`."""
        import torch

        if not getattr(self, "use_graph", False):
            return False
        if torch.is_grad_enabled() or compute_metrics or self.backtracking_config is not None or self.max_iter < 3:
            return False
        if self.early_stop and self.check_conv_fn is not None and not self._device_stop_ok(X, compute_metrics):
            return False
        if X is None or not all(isinstance(t, torch.Tensor) and t.is_cuda for t in X["est"]) or X.get("cost") is not None:
            return False
        first = [f(0) if f else None for f in (self.update_params_fn, self.update_data_fidelity_fn, self.update_prior_fn)]
        for it in range(1, self.max_iter):
            cur = [f(it) if f else None for f in (self.update_params_fn, self.update_data_fidelity_fn, self.update_prior_fn)]
            if cur[0] != first[0] or cur[1] is not first[1] or cur[2] is not first[2]:
                return False
        return True

    def _device_stop_ok(self, X, compute_metrics):
        """early_stop can be decided on the device when nothing else needs the host each iteration"""
        import torch

        return (self.early_stop and self.conv_crit_fn is not None and self._thres() is not None and not compute_metrics
                and self.backtracking_config is None and not torch.is_grad_enabled() and X is not None
                and all(isinstance(t, torch.Tensor) and t.is_cuda for t in X["est"]))

    def _stop_step(self, it, X_prev, X_new, done):
        """One iteration's share of the device-side early stop.  The reference breaks AFTER the iteration whose criterion
        is below the threshold and returns that iteration's iterate (fixed_point.py:340-352): here the flag `done` is
        raised on the device at that iteration and every later iterate is the frozen one (`where(done, X_prev, X_new)`),
        so running on - until the host notices the flag, or to max_iter under graph replay - changes nothing."""
        import torch

        if done is not None:
            est = tuple(torch.where(done, a, b) for a, b in zip(X_prev["est"], X_new["est"]))
            cost = X_new.get("cost")
            if cost is not None and X_prev.get("cost") is not None:
                cost = torch.where(done, X_prev["cost"], cost)
            X_new = {"est": est, "cost": cost}
        if it > 1:
            c = self.conv_crit_fn(X_prev, X_new) < self._thres()
            done = c if done is None else (done | c)
        return X_new, done

    def _run_graph(self, X, *args, **kwargs):
        """the first iteration(s) run eagerly (they build every plan / workspace), the next one is captured into a HIP
        graph whose inputs are static copies of the iterate, and the graph is replayed for the remaining iterations: one
        host call per iteration instead of ~80 kernel launches (fixed_point.py:324-361 with the same arithmetic).  With
        early_stop the captured iteration carries the device-side convergence test and the freeze."""
        import torch

        stop = self._device_stop_ok(X, False) and self.check_conv_fn is not None
        n_eager = 2 if stop else 1
        graph = torch.cuda.CUDAGraph()
        # the eager iteration(s) run on the capture stream as well: whatever scratch the kernels keep per stream exists before
        # the capture begins (nothing is allocated inside it)
        side = _capture_stream(X["est"][0].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for it in range(n_eager):
                X = self.single_iteration(X, it, *args, **kwargs)
            static = [t.clone() for t in X["est"]]
            done = torch.zeros((), dtype=torch.bool, device=static[0].device) if stop else None
            with torch.cuda.graph(graph, stream=side):
                Xp = {"est": tuple(static), "cost": None}
                Xo = self.single_iteration(Xp, n_eager, *args, **kwargs)
                if stop:
                    Xo, d2 = self._stop_step(n_eager, Xp, Xo, done)
                    done.copy_(d2)
                for s, o in zip(static, Xo["est"]):
                    s.copy_(o)
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(n_eager, self.max_iter):
            graph.replay()
        out = {"est": tuple(s.clone() for s in static), "cost": None}
        del graph
        if stop and self.on_converged is not None:
            self.on_converged(done)
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
            if self._device_stop_ok(X, compute_metrics) and self.check_conv_fn is not None:
                return self._run_device_stop(X, *args, **kwargs), metrics
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


    def _run_device_stop(self, X, *args, **kwargs):
        """eager loop with the convergence decision on the device: the host looks at the flag every `poll_every`
        iterations through a pinned copy it never waits for (an event query), so the stream is never drained; iterates
        computed after the flag went up are frozen copies (see `_stop_step`), hence the result is the reference's."""
        import torch

        done, polls = None, []
        for it in range(self.max_iter):
            X, done = self._stop_step(it, X, self.single_iteration(X, it, *args, **kwargs), done)
            if done is not None and it % self.poll_every == self.poll_every - 1:
                host = torch.empty((), dtype=torch.bool, pin_memory=True)
                host.copy_(done, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                polls.append((ev, host))
            if polls and polls[0][0].query():
                if bool(polls.pop(0)[1]):
                    break
        if self.on_converged is not None and done is not None:
            self.on_converged(done)
        return X
