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


# ----------------------------------------------------------------------------------------------------------------------
# Operator-parallel distribution: a stack of linear operators A = [A_1; ...; A_n] shared out over the ranks
# (reference deepinv/distributed/distrib_framework.py:234-732, DistributedStackedLinearPhysics).  The natural instance
# on the hot path is COIL-parallel MultiCoilMRI: rank r owns a slab of coils, `A` needs no communication at all (each
# rank produces the k-space of its coils) and `A_adjoint` / `A_adjoint_A` end in ONE image-sized RCCL all-reduce
# (sum_n conj(S_n) F^H M y_n is a sum over coils).  One huge volume can thus use all 8 GPUs of a node.
# ----------------------------------------------------------------------------------------------------------------------
from .physics.forward import LinearPhysics  # noqa: E402


class DistributedStackedLinearPhysics(LinearPhysics):
    """`num_operators` linear operators built by ``factory(index, device, factory_kwargs)``; rank r owns the indices
    i with i % world_size == r (round robin, distrib_framework.py:194-203).

    * ``A(x, gather=False)`` -> list of this rank's measurements (no communication); ``gather=True`` -> the full list,
      in operator order, on every rank (one all-gather per operator group; needs equal measurement shapes);
    * ``A_adjoint(y, reduce_op="sum")``: y = full list (length num_operators) or this rank's local list; local partial
      sum, then one all-reduce; ``reduce_op=None`` returns the local contribution;
    * ``A_adjoint_A(x)``: sum_i A_i^T A_i x with one all-reduce (what the PGD / CG loops call)."""

    def __init__(self, ctx: BatchParallelContext, num_operators: int, factory, *, factory_kwargs=None, **kwargs):
        super().__init__(**kwargs)
        self.ctx, self.num_operators = ctx, int(num_operators)
        self.local_indexes = [i for i in range(self.num_operators) if i % ctx.world_size == ctx.rank]
        self.local_physics = torch.nn.ModuleList([factory(i, ctx.device, factory_kwargs) for i in self.local_indexes])
        for p in self.local_physics:
            if not isinstance(p, LinearPhysics):
                raise ValueError("factory must return LinearPhysics instances.")

    def _local(self, y):
        """this rank's share of a measurement list (accepts the full list or the local one)"""
        if len(y) == self.num_operators and self.num_operators != len(self.local_indexes):
            return [y[i] for i in self.local_indexes]
        if len(y) != len(self.local_indexes):
            raise ValueError(f"expected {self.num_operators} (all) or {len(self.local_indexes)} (local) measurements, got {len(y)}")
        return list(y)

    def _reduce(self, t, reduce_op):
        if reduce_op is None or self.ctx.world_size == 1:
            return t
        if reduce_op != "sum":
            raise ValueError("reduce_op must be 'sum' or None")
        t = t.contiguous()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def A(self, x, gather: bool = True, **kwargs):
        local = [p.A(x, **kwargs) for p in self.local_physics]
        if not gather or self.ctx.world_size == 1:
            return local
        out = [None] * self.num_operators
        W = self.ctx.world_size
        for k in range((self.num_operators + W - 1) // W):      # k-th operator of every rank
            mine = local[k] if k < len(local) else None
            if mine is None:     # ranks without a k-th operator contribute an empty slot of the common shape
                ref = local[0] if local else None
                if ref is None:
                    raise RuntimeError("a rank without any operator cannot take part in a gathered A()")
                mine = torch.zeros_like(ref)
            bufs = [torch.empty_like(mine) for _ in range(W)]
            dist.all_gather(bufs, mine.contiguous())
            for r in range(W):
                i = k * W + r
                if i < self.num_operators:
                    out[i] = bufs[r]
        return out

    def A_adjoint(self, y, reduce_op: str | None = "sum", **kwargs):
        ys = self._local(y)
        acc = None
        for p, yi in zip(self.local_physics, ys):
            t = p.A_adjoint(yi, **kwargs)
            acc = t if acc is None else acc + t
        if acc is None:
            raise RuntimeError("this rank owns no operator: cannot shape its (zero) contribution")
        return self._reduce(acc, reduce_op)

    def A_adjoint_A(self, x, reduce_op: str | None = "sum", **kwargs):
        acc = None
        for p in self.local_physics:
            t = p.A_adjoint_A(x, **kwargs)
            acc = t if acc is None else acc + t
        return self._reduce(acc, reduce_op)

    def A_vjp(self, x, v, reduce_op: str | None = "sum", **kwargs):
        return self.A_adjoint(v, reduce_op=reduce_op, **kwargs)


def coil_parallel_mri(ctx: BatchParallelContext, mask, coil_maps, img_size, three_d: bool = False, **kwargs):
    """MultiCoilMRI with its coils dealt out over the ranks (contiguous slabs, so every rank applies its fused
    expand / FFT / combine kernels to a [1, N/P, ...] map tensor); see DistributedStackedLinearPhysics."""
    from .physics.mri import MultiCoilMRI

    n = coil_maps.shape[1]
    q, r = divmod(n, ctx.world_size)
    bounds = [0]
    for k in range(ctx.world_size):
        bounds.append(bounds[-1] + q + (1 if k < r else 0))

    def factory(i, device, _):
        return MultiCoilMRI(mask=mask, coil_maps=coil_maps[:, bounds[i]:bounds[i + 1]].contiguous(), img_size=img_size,
                            three_d=three_d, device=device, **kwargs)

    return DistributedStackedLinearPhysics(ctx, ctx.world_size, factory)


# ----------------------------------------------------------------------------------------------------------------------
# Tile-parallel processing of one large signal (reference deepinv/distributed/distrib_framework.py:734-934 with the
# "overlap_tiling" strategy of distributed/strategies.py:292-457): the signal is reflect-padded by the halo radius,
# cut into equally sized windows (patch + halo on both sides; the last window of an axis is shifted inwards instead of
# being smaller, so that all windows can ride one batch through the denoiser), the windows are dealt out round robin,
# every rank runs `processor` on its windows, keeps the inner patch of each result, and ONE all-reduce of the
# zero-initialised output assembles the signal on every rank.  Every output sample is written by exactly one window.
# ----------------------------------------------------------------------------------------------------------------------
class OverlapTiling:
    def __init__(self, shape, patch_size=256, overlap=32, tiling_dims=None, pad_mode="reflect"):
        shape = tuple(int(s) for s in shape)
        if tiling_dims is None:
            n = len(patch_size) if isinstance(patch_size, (tuple, list)) else 2
            tiling_dims = tuple(range(len(shape) - n, len(shape)))
        elif isinstance(tiling_dims, int):
            tiling_dims = (tiling_dims,)
        self.dims = tuple(d % len(shape) for d in tiling_dims)
        as_tuple = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (int(v),) * len(self.dims)  # noqa: E731
        self.patch, self.halo = list(as_tuple(patch_size)), list(as_tuple(overlap))
        if len(self.patch) != len(self.dims) or len(self.halo) != len(self.dims):
            raise ValueError("patch_size / overlap must have one entry per tiled dimension")
        self.shape, self.pad_mode = shape, pad_mode
        per_dim = []
        for i, d in enumerate(self.dims):
            D = shape[d]
            if self.patch[i] >= D:            # one window spans the axis: shrink the patch so that a halo remains
                self.patch[i] = D
            if self.halo[i] >= D and pad_mode == "reflect":
                self.halo[i] = max(0, D - 1)  # reflect padding needs halo < size
            p = self.patch[i]
            starts = list(range(0, max(D - p, 0) + 1, p))
            if starts[-1] + p < D:
                starts.append(D - p)          # shifted last window
            wins, done = [], 0
            for s in starts:                  # (window start in the padded axis, first owned sample, end)
                wins.append((s, max(s, done), s + p))
                done = s + p
            per_dim.append(wins)
        self.windows = []                     # cartesian product, row-major
        def rec(i, cur):
            if i == len(per_dim):
                self.windows.append(tuple(cur))
                return
            for w in per_dim[i]:
                rec(i + 1, cur + [w])
        rec(0, [])

    def __len__(self):
        return len(self.windows)

    def pad(self, x):
        pads = [0] * (2 * x.ndim)
        for i, d in enumerate(self.dims):
            pads[2 * (x.ndim - 1 - d)] = pads[2 * (x.ndim - 1 - d) + 1] = self.halo[i]
        while len(pads) >= 2 and pads[-1] == 0 and pads[-2] == 0:
            pads = pads[:-2]
        if not any(pads):
            return x
        return torch.nn.functional.pad(x, pads, mode=self.pad_mode)

    def window(self, x_pad, k):
        idx = [slice(None)] * x_pad.ndim
        for i, d in enumerate(self.dims):
            s = self.windows[k][i][0]
            idx[d] = slice(s, s + self.patch[i] + 2 * self.halo[i])
        return x_pad[tuple(idx)]

    def place(self, out, k, processed):
        src, dst = [slice(None)] * out.ndim, [slice(None)] * out.ndim
        for i, d in enumerate(self.dims):
            s, own, end = self.windows[k][i]
            src[d] = slice(self.halo[i] + own - s, self.halo[i] + end - s)
            dst[d] = slice(own, end)
        out[tuple(dst)] = processed[tuple(src)]


class DistributedProcessing:
    """``processor`` (a denoiser, a prior's prox, any shape-preserving map with a bounded receptive field) applied to
    one large signal tile by tile across the ranks; see the block comment above.  ``strategy_kwargs``: ``patch_size``,
    ``overlap`` (halo radius >= receptive-field radius for exact agreement with untiled processing away from the
    signal border), ``tiling_dims``, ``pad_mode``.  ``max_batch_size`` bounds the windows per processor call."""

    def __init__(self, ctx: BatchParallelContext, processor, *, strategy: str | None = None, strategy_kwargs=None,
                 max_batch_size: int | None = None):
        if strategy not in (None, "overlap_tiling"):
            raise ValueError("only the 'overlap_tiling' strategy is implemented")
        self.ctx, self.processor = ctx, processor
        self.kw = dict(strategy_kwargs or {})
        self.max_batch_size = max_batch_size
        self._tiling = None
        if hasattr(processor, "to"):
            processor.to(ctx.device)

    def __call__(self, x, *args, gather: bool = True, **kwargs):
        if self._tiling is None or self._tiling.shape != tuple(x.shape):
            self._tiling = OverlapTiling(x.shape, **self.kw)
        T = self._tiling
        mine = [k for k in range(len(T)) if k % self.ctx.world_size == self.ctx.rank]
        out = torch.zeros_like(x)
        if mine:
            xp = T.pad(x)
            B = x.shape[0]
            step = self.max_batch_size or len(mine)
            for i0 in range(0, len(mine), step):
                group = mine[i0:i0 + step]
                batch = torch.cat([T.window(xp, k) for k in group], dim=0)       # windows ride the batch axis
                res = self.processor(batch, *args, **kwargs)
                for j, k in enumerate(group):
                    T.place(out, k, res[j * B:(j + 1) * B])
        if gather and self.ctx.world_size > 1:
            dist.all_reduce(out, op=dist.ReduceOp.SUM)
        return out
