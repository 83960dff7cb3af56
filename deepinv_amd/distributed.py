"""Batch data-parallel execution over the GPUs of one node.

Every operator and every reconstruction loop of the hot path is per-sample (the only cross-sample
reductions, ``check_conv_fn`` / backtracking means, are off by default), so the batch is cut into
contiguous slabs, one process per GPU, and the only data-path collective is one RCCL ``all_gather`` of
the reconstructions.  Conventions (env-driven rank / device selection, 127.0.0.1 rendez-vous, gloo when
no GPU) follow the reference's ``DistributedContext`` (deepinv/distributed/distrib_framework.py:73-173).
``torch.distributed`` with backend ``"nccl"`` *is* RCCL on ROCm.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class BatchParallelContext:
    def __init__(self, backend: str | None = None, device: torch.device | None = None):
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        use_gpu = torch.cuda.is_available() and (device is None or torch.device(device).type == "cuda")
        self.device = torch.device("cuda", self.local_rank) if use_gpu else torch.device("cpu")
        self.backend = backend or ("nccl" if use_gpu else "gloo")
        self._own_pg = False

    def __enter__(self):
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        if self.world_size > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            kw = {"device_id": self.device} if self.backend == "nccl" else {}
            dist.init_process_group(backend=self.backend, rank=self.rank, world_size=self.world_size, **kw)
            self._own_pg = True
        return self

    def __exit__(self, *exc):
        if self._own_pg and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return False

    # ---- slab partition of a batch of `n` units
    def slab(self, n: int) -> slice:
        """contiguous slab of rank r: sizes differ by at most one when n % world_size != 0"""
        q, r = divmod(n, self.world_size)
        start = self.rank * q + min(self.rank, r)
        return slice(start, start + q + (1 if self.rank < r else 0))

    def scatter_batch(self, x: torch.Tensor) -> torch.Tensor:
        return x[self.slab(x.shape[0])].to(self.device)

    def all_gather_batch(self, x_local: torch.Tensor, n_total: int) -> torch.Tensor:
        """gather the per-rank slabs back into the full batch on every rank (one collective)"""
        if self.world_size == 1:
            return x_local
        q, r = divmod(n_total, self.world_size)
        x_local = x_local.contiguous()
        if r == 0:
            out = torch.empty((n_total, *x_local.shape[1:]), device=x_local.device, dtype=x_local.dtype)
            dist.all_gather_into_tensor(out, x_local)
            return out
        # ragged slabs: pad every slab to q+1 rows, gather, drop the padding
        pad = torch.zeros((q + 1, *x_local.shape[1:]), device=x_local.device, dtype=x_local.dtype)
        pad[: x_local.shape[0]] = x_local
        buf = torch.empty((self.world_size * (q + 1), *x_local.shape[1:]), device=x_local.device, dtype=x_local.dtype)
        dist.all_gather_into_tensor(buf, pad)
        parts = [buf[k * (q + 1): k * (q + 1) + q + (1 if k < r else 0)] for k in range(self.world_size)]
        return torch.cat(parts, dim=0)

    def barrier(self):
        if self.world_size > 1:
            dist.barrier()


def reconstruct_batch_parallel(ctx: BatchParallelContext, model, y_full: torch.Tensor, physics, **kwargs):
    """``model(y, physics)`` on this rank's slab of measurements, then all-gather the reconstructions.
    Shared operator parameters (mask, coil maps, filter, angles, denoiser weights) are replicated; per-sample
    parameters must be sliced by the caller with ``ctx.slab``."""
    n = y_full.shape[0]
    x_local = model(ctx.scatter_batch(y_full), physics, **kwargs)
    return ctx.all_gather_batch(x_local, n)
