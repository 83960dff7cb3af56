"""Training loop for reconstruction networks (unfolded models with the HIP-backward DRUNet prior: BASELINE config 4).

Mirrors the part of the reference's ``deepinv.Trainer`` (deepinv/training/trainer.py:27-1492) that drives the hot path:
the same constructor fields for the training loop, ``setup_train`` / ``get_samples`` (online and offline measurements, physics
generators) / ``model_inference`` / ``compute_loss`` / ``step`` / ``train`` / ``test`` with the same call sequence per batch
(``optimizer.zero_grad(set_to_none=True)`` -> forward -> ``loss.mean()`` summed over the losses -> ``backward()`` -> optional
gradient clipping -> ``optimizer.step()``; scheduler stepped once per epoch).  Out of scope here: Weights & Biases /
MLflow logging, image plotting, the no-learning comparison, loss schedulers.  Checkpoints are written only when ``save_path`` is set.
"""
from __future__ import annotations

import os
import warnings
from dataclasses import dataclass, field

import numpy as np
import torch


class SupLoss(torch.nn.Module):
    r"""Supervised loss :math:`\frac{1}{n}\|x - \hat x\|^2` (deepinv/loss/sup.py:16-52; ``metric`` defaults to the MSE)."""

    def __init__(self, metric=None):
        super().__init__()
        self.name = "supervised"
        self.metric = torch.nn.MSELoss() if metric is None else metric

    def forward(self, x_net, x, **kwargs):
        return self.metric(x_net, x)


def _psnr(x_net, x, max_pixel=1.0):
    mse = (x_net - x).pow(2).mean(dim=tuple(range(1, x.ndim)))
    return 10 * torch.log10(max_pixel ** 2 / mse)


class _Avg:
    def __init__(self):
        self.sum, self.n = 0.0, 0

    def update(self, v, n=1):
        self.sum += float(v) * n
        self.n += n

    @property
    def avg(self):
        return self.sum / max(self.n, 1)


@dataclass
class Trainer:
    """Trainer(model, physics, optimizer, train_dataloader, ...) - see the module docstring for what is mirrored."""

    model: torch.nn.Module
    physics: object
    optimizer: torch.optim.Optimizer | None
    train_dataloader: object
    epochs: int = 100
    max_batch_steps: int = 10 ** 10
    losses: object = field(default_factory=SupLoss)
    eval_dataloader: object = None
    early_stop: int | None = None
    scheduler: object = None
    online_measurements: bool = False
    physics_generator: object = None
    optimizer_step_multi_dataset: bool = True
    metrics: object = None                 # callable(x_net, x) -> per-sample values; default PSNR
    compute_train_metrics: bool = True
    device: str | torch.device = "cuda"
    ckpt_pretrained: str | None = None
    save_path: str | None = None
    grad_clip: float | None = None
    check_grad: bool = False
    ckp_interval: int = 1
    eval_interval: int = 1
    compute_eval_losses: bool = False
    verbose: bool = True
    show_progress_bar: bool = False
    non_blocking_transfers: bool = True

    def __post_init__(self):
        self.device = torch.device(self.device)
        if self.non_blocking_transfers and self.device.type != "cuda":
            self.non_blocking_transfers = False
        self.train_loss_history, self.eval_metric_history = [], []

    # ------------------------------------------------------------------ setup (trainer.py:336-566)
    def setup_train(self, train: bool = True, **kwargs):
        as_list = lambda v: list(v) if isinstance(v, (list, tuple)) else [v]  # noqa: E731
        self.train_dataloader = as_list(self.train_dataloader) if self.train_dataloader is not None else []
        self.eval_dataloader = as_list(self.eval_dataloader) if self.eval_dataloader is not None else []
        self.physics = as_list(self.physics)
        self.losses = as_list(self.losses)
        self.metrics = as_list(self.metrics) if self.metrics is not None else [_psnr]
        if self.physics_generator is not None:
            self.physics_generator = as_list(self.physics_generator)
        self.G = len(self.train_dataloader) if train else len(self.eval_dataloader)
        if len(self.physics) == 1 and self.G > 1:
            self.physics = self.physics * self.G
        self.model = self.model.to(self.device)
        self.epoch_start = 0
        if self.ckpt_pretrained is not None:
            ck = torch.load(self.ckpt_pretrained, map_location=self.device)
            self.model.load_state_dict(ck["state_dict"])
            if ck.get("optimizer") is not None and self.optimizer is not None:
                self.optimizer.load_state_dict(ck["optimizer"])
            if ck.get("scheduler") is not None and self.scheduler is not None:     # trainer.py:592-593
                self.scheduler.load_state_dict(ck["scheduler"])
            # the reference stores its histories as {class name: [value per epoch]} (trainer.py:454-480, 1192-1198); this trainer
            # keeps the total loss and the first metric: read either form
            self.train_loss_history = self._history(ck.get("loss"), [l.__class__.__name__ for l in self.losses], self.train_loss_history, total=True)
            self.eval_metric_history = self._history(ck.get("eval_metrics"), [self._metric_name(m) for m in self.metrics],
                                                     self.eval_metric_history)
            self.epoch_start = ck.get("epoch", -1) + 1
        if train and self.optimizer is None:
            raise ValueError("an optimizer is needed for training")
        if self.check_grad:
            self.check_grad_val = _Avg()

    # ------------------------------------------------------------------ samples (trainer.py:662-792)
    def get_samples_online(self, iterators, g):
        data = next(iterators[g])
        params = {}
        if isinstance(data, (tuple, list)):
            x = data[0]
            if len(data) == 2 and isinstance(data[1], dict):
                params = data[1]
            elif len(data) > 1:
                warnings.warn("Generating online measurements requires dataloader to return tensor `x` or (tensor `x`, dict "
                              "`params`). Discarding all data after `x`.")
        else:
            x = data
        if torch.isnan(x).all():
            raise ValueError("Online measurements can't be used if x is all NaN.")
        x = x.to(self.device, non_blocking=self.non_blocking_transfers)
        physics = self.physics[g]
        if self.physics_generator is not None:
            if params:
                warnings.warn("Physics generator is provided but dataloader also returns params. Ignoring params from dataloader.")
            params = self.physics_generator[g].step(batch_size=x.size(0))
        physics.update(**params)
        y = physics(x, **params)
        return x, y, physics

    def get_samples_offline(self, iterators, g):
        data = next(iterators[g])
        if not isinstance(data, (tuple, list)) or len(data) < 2:
            raise ValueError("If online_measurements=False, the dataloader should output a tuple (x, y) or (x, y, params)")
        if len(data) == 2:
            x, y, params = *data, None
            if isinstance(y, dict):
                raise ValueError("If online_measurements=False, measurements y must be provided as a tensor.")
        elif len(data) == 3:
            x, y, params = data
        else:
            raise ValueError("Dataloader returns too many items. For offline learning, dataloader should either return (x, y) "
                             "or (x, y, params).")
        if x.size(0) != y.size(0):
            raise ValueError(f"Data x, y must have same batch size, but got {x.size(0)}, {y.size(0)}")
        x = None if (torch.isnan(x).all() and x.ndim <= 1) else x.to(self.device, non_blocking=self.non_blocking_transfers)
        y = y.to(self.device, non_blocking=self.non_blocking_transfers)
        physics = self.physics[g]
        if params is not None:
            physics.update(**{k: (p.to(self.device) if isinstance(p, torch.Tensor) else p) for k, p in params.items()})
        return x, y, physics

    def get_samples(self, iterators, g):
        return self.get_samples_online(iterators, g) if self.online_measurements else self.get_samples_offline(iterators, g)

    # ------------------------------------------------------------------ one batch (trainer.py:794-1094)
    def model_inference(self, y, physics, x=None, train=True, **kwargs):
        if train:
            self.model.train()
            return self.model(y, physics, **kwargs)
        self.model.eval()
        with torch.no_grad():
            return self.model(y, physics, **kwargs)

    def check_clip_grad(self):
        out = None
        if self.grad_clip is not None:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_clip)
        if self.check_grad:
            grads = [p.grad.detach().flatten() for p in self.model.parameters() if p.grad is not None]
            out = float(torch.cat(grads).abs().pow(2).sum().sqrt())
            self.check_grad_val.update(out)
        return out

    def compute_loss(self, physics, x, y, train=True, epoch=None, step=False):
        logs = {}
        if train and step:
            self.optimizer.zero_grad(set_to_none=True)
        if train or self.compute_eval_losses:
            x_net = self.model_inference(y=y, physics=physics, x=x, train=True)
            loss_total = 0
            for l in self.losses:
                loss = l(x=x, x_net=x_net, y=y, physics=physics, model=self.model, epoch=epoch)
                loss_total = loss_total + loss.mean()
            logs["TotalLoss"] = float(loss_total.detach())
        else:
            loss_total, x_net = 0, None
        if train:
            loss_total.backward()
            norm = self.check_clip_grad()
            if norm is not None:
                logs["gradient_norm"] = self.check_grad_val.avg
            if step:
                self.optimizer.step()
        return loss_total, x_net, logs

    def compute_metrics(self, x, x_net, y, physics, logs, train=True, epoch=None):
        if x_net is None:
            x_net = self.model_inference(y=y, physics=physics, x=x, train=False)
        if x is not None:
            with torch.no_grad():
                for m in self.metrics:
                    v = m(x_net, x)
                    logs[self._metric_name(m)] = float(torch.as_tensor(v).float().mean())
        return x_net, logs

    def step(self, epoch, train_ite=None, train=True, last_batch=False):
        if train and self.optimizer_step_multi_dataset:
            self.optimizer.zero_grad(set_to_none=True)
        loss, logs = 0, {}
        for g in np.random.permutation(self.G):
            x, y, physics_cur = self.get_samples(self.current_train_iterators if train else self.current_eval_iterators, int(g))
            loss_cur, x_net, logs = self.compute_loss(physics_cur, x, y, train=train, epoch=epoch,
                                                      step=(not self.optimizer_step_multi_dataset))
            loss = loss + (loss_cur.detach() if isinstance(loss_cur, torch.Tensor) else loss_cur)
            if self.compute_train_metrics or not train:
                x_net, logs = self.compute_metrics(x, x_net.detach() if x_net is not None else None, y, physics_cur, logs,
                                                   train=train, epoch=epoch)
        if train and self.optimizer_step_multi_dataset:
            self.optimizer.step()
        return loss, logs

    # ------------------------------------------------------------------ loops (trainer.py:1332-1600)
    @staticmethod
    def _history(field, names, default, total=False):
        """a history field of a checkpoint as a flat list: this trainer's own {name: [..]} form with one entry, the reference's
        {class name: [..]} (total = the per-epoch sum over the losses, else the first of `names` found), or a plain list"""
        if field is None:
            return default
        if isinstance(field, dict):
            lists = [list(v) for v in field.values()] if total else ([list(field[n]) for n in names if n in field] or [list(v) for v in field.values()])
            if not lists:
                return []
            if total and len(lists) > 1 and len({len(v) for v in lists}) == 1:
                return [float(sum(vals)) for vals in zip(*lists)]
            return [float(v) for v in lists[0]]
        return [float(v) for v in field]

    def save_model(self, filename, epoch):
        if self.save_path is None:
            return
        os.makedirs(self.save_path, exist_ok=True)
        # the reference's fields and their shapes (trainer.py:1192-1198): epoch, state_dict, optimizer, scheduler, and the histories
        # as {name: [value per epoch]} - here the total training loss and the first evaluation metric
        metric = self._metric_name(self.metrics[0]) if self.metrics else "metric"      # (a plain function has no class name of its own)
        torch.save({"epoch": epoch, "state_dict": self.model.state_dict(), "loss": {"TotalLoss": list(self.train_loss_history)},
                    "optimizer": self.optimizer.state_dict() if self.optimizer else None,
                    "scheduler": self.scheduler.state_dict() if self.scheduler is not None else None,
                    "eval_metrics": {metric: list(self.eval_metric_history)}}, os.path.join(self.save_path, filename))

    def _metric_name(self, m):
        return getattr(m, "__name__", m.__class__.__name__).strip("_")

    def _eval_value(self, logs):
        """the value model selection runs on: the FIRST metric's logged value (not a hard-wired "psnr")"""
        return logs.get(self._metric_name(self.metrics[0]), 0.0)

    def _better(self, val, best):
        """metrics with `lower_better = True` (the reference's Metric attribute) improve downwards"""
        return val < best if getattr(self.metrics[0], "lower_better", False) else val > best

    def train(self):
        self.setup_train()
        best, since_best = None, 0
        for epoch in range(self.epoch_start, self.epochs):
            self.current_train_iterators = [iter(loader) for loader in self.train_dataloader]
            batches = min(min(len(loader) for loader in self.train_dataloader), self.max_batch_steps)
            self.model.train()
            meter = _Avg()
            for i in range(batches):
                loss, logs = self.step(epoch, train_ite=epoch * batches + i, train=True, last_batch=i == batches - 1)
                meter.update(float(loss))
            self.train_loss_history.append(meter.avg)
            if self.verbose:
                print(f"Train epoch {epoch}: TotalLoss={meter.avg:.6g}" + "".join(f", {k}={v:.4g}" for k, v in logs.items()
                                                                                   if k != "TotalLoss"))
            if self.scheduler is not None:
                self.scheduler.step()
            if self.eval_dataloader and (epoch + 1) % self.eval_interval == 0:
                val = self._evaluate(epoch)
                self.eval_metric_history.append(val)
                if best is None or self._better(val, best):
                    best, since_best = val, 0
                    self.save_model("ckp_best.pth.tar", epoch)
                else:
                    since_best += 1
                if self.early_stop is not None and since_best > self.early_stop:
                    break
            if (epoch + 1) % self.ckp_interval == 0 or epoch + 1 == self.epochs:
                self.save_model(f"ckp_{epoch}.pth.tar", epoch)
        return self.model

    def _evaluate(self, epoch):
        self.current_eval_iterators = [iter(loader) for loader in self.eval_dataloader]
        vals = []
        saved_G, self.G = self.G, len(self.eval_dataloader)
        try:
            for _ in range(min(len(loader) for loader in self.eval_dataloader)):
                _, logs = self.step(epoch, train=False)
                vals.append(self._eval_value(logs))
        finally:
            self.G = saved_G
        return float(np.mean(vals)) if vals else 0.0

    def test(self, test_dataloader, **kwargs):
        """average of the first metric over a test set (trainer.py:1494-1600): {"<NAME>": ..., "<NAME>_std": ...}, the name
        upper-cased as the reference prints it ("PSNR" for the default metric)"""
        self.eval_dataloader = test_dataloader
        self.setup_train(train=False)
        self.current_eval_iterators = [iter(loader) for loader in self.eval_dataloader]
        vals = []
        for _ in range(min(len(loader) for loader in self.eval_dataloader)):
            _, logs = self.step(0, train=False)
            vals.append(self._eval_value(logs))
        name = self._metric_name(self.metrics[0]).upper()
        return {name: float(np.mean(vals)), name + "_std": float(np.std(vals))}


def train(model, physics, optimizer, train_dataloader, epochs=100, losses=None, eval_dataloader=None, *args, **kwargs):
    """deepinv.train (trainer.py:1603-1660): build a Trainer and run it"""
    return Trainer(model=model, physics=physics, optimizer=optimizer, train_dataloader=train_dataloader, epochs=epochs,
                   losses=losses if losses is not None else SupLoss(), eval_dataloader=eval_dataloader, *args, **kwargs).train()
