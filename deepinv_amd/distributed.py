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
    def __init__(self, backend: str | None = None, device: torch.device | None = None, init_always: bool = False):
        """init_always: create the process group and run the collectives even at world size 1 (exercises RCCL initialisation
        and the collective code path on a single GPU; off by default: a single process needs no communicator)"""
        self.init_always = bool(init_always)
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
        if (self.world_size > 1 or self.init_always) and not dist.is_initialized():
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
        if self.world_size == 1 and not (self.init_always and dist.is_initialized()):
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

    def __init__(self, ctx: BatchParallelContext, num_operators: int, factory, *, factory_kwargs=None, img_shape=None,
                 **kwargs):
        super().__init__(**kwargs)
        self.ctx, self.num_operators = ctx, int(num_operators)
        self.img_shape = tuple(img_shape) if img_shape is not None else None
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

    def _empty_contribution(self, like=None):
        """a rank that owns no operator still has to join the all-reduce (the other ranks are already in it): its share is
        a zero image.  The image shape is `like`'s when the caller has an image at hand (A_adjoint_A), else the shape given
        at construction (`img_shape=`)."""
        if like is not None:
            return torch.zeros_like(like)
        if self.img_shape is None:
            raise RuntimeError("this rank owns no operator and the image shape is unknown: construct the distributed "
                               "physics with img_shape=(B, C, ...) when num_operators < world_size")
        return torch.zeros(self.img_shape, device=self.ctx.device, dtype=torch.float32)

    def A_adjoint(self, y, reduce_op: str | None = "sum", **kwargs):
        ys = self._local(y)
        acc = None
        for p, yi in zip(self.local_physics, ys):
            t = p.A_adjoint(yi, **kwargs)
            acc = t if acc is None else acc + t
        if acc is None:
            acc = self._empty_contribution()
        return self._reduce(acc, reduce_op)

    def A_adjoint_A(self, x, reduce_op: str | None = "sum", **kwargs):
        acc = None
        for p in self.local_physics:
            t = p.A_adjoint_A(x, **kwargs)
            acc = t if acc is None else acc + t
        if acc is None:
            acc = self._empty_contribution(like=x)
        return self._reduce(acc, reduce_op)

    def A_vjp(self, x, v, reduce_op: str | None = "sum", **kwargs):
        return self.A_adjoint(v, reduce_op=reduce_op, **kwargs)


def coil_parallel_mri(ctx: BatchParallelContext, mask, coil_maps, img_size, three_d: bool = False, **kwargs):
    """MultiCoilMRI with its coils dealt out over the ranks (contiguous slabs, so every rank applies its fused
    expand / FFT / combine kernels to a [1, N/P, ...] map tensor); see DistributedStackedLinearPhysics."""
    from .physics.mri import MultiCoilMRI

    n = coil_maps.shape[1]
    if n < ctx.world_size:
        raise ValueError(f"coil-parallel MultiCoilMRI needs at least one coil per rank: {n} coils for {ctx.world_size} ranks")
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
                def rep(a):
                    """arguments ride along: spatial maps ([B or 1, C, *image dims], e.g. a noise-level map) are cut into
                    the same windows as the signal, other per-sample tensors ([B, ...]) are repeated per window, everything
                    else ([1, ...] tensors, scalars) broadcasts over the windows as it did over the samples"""
                    if not isinstance(a, torch.Tensor) or a.ndim < 1:
                        return a
                    if a.ndim == x.ndim and a.shape[2:] == x.shape[2:] and a.shape[0] in (1, B):
                        ap = T.pad(a.expand(B, *a.shape[1:]).contiguous())
                        return torch.cat([T.window(ap, k) for k in group], dim=0)
                    if a.shape[0] == B and B > 1:
                        return a.repeat(len(group), *([1] * (a.ndim - 1)))
                    return a
                res = self.processor(batch, *[rep(a) for a in args], **{k: rep(v) for k, v in kwargs.items()})
                for j, k in enumerate(group):
                    T.place(out, k, res[j * B:(j + 1) * B])
        if gather and self.ctx.world_size > 1:
            dist.all_reduce(out, op=dist.ReduceOp.SUM)
        return out


# ----------------------------------------------------------------------------------------------------------------------
# Data fidelity over distributed stacked physics and the `distribute()` factory
# (reference deepinv/distributed/distrib_framework.py:940-1180, distributed/distribute.py:30-420)
# ----------------------------------------------------------------------------------------------------------------------
class DistributedDataFidelity:
    """f(x) = sum_i d(A_i x, y_i) and its gradient sum_i A_i^T grad d(A_i x, y_i) over a `DistributedStackedLinearPhysics`:
    every rank evaluates its own operators (no communication), then ONE all-reduce (`reduction` 'sum' or 'mean' over the
    operators).  `data_fidelity`: one DataFidelity shared by all operators, or a factory
    ``factory(index, device, factory_kwargs) -> DataFidelity`` (then `num_operators` is required)."""

    def __init__(self, ctx: BatchParallelContext, data_fidelity, num_operators: int | None = None, *, factory_kwargs=None,
                 reduction: str = "sum"):
        import copy

        from .optim.data_fidelity import DataFidelity

        if reduction not in ("sum", "mean"):
            raise ValueError("reduction must be 'sum' or 'mean'")
        self.ctx, self.reduction_mode = ctx, reduction
        self.local_data_fidelities, self.single_fidelity = [], None
        if isinstance(data_fidelity, DataFidelity):
            self.single_fidelity = copy.deepcopy(data_fidelity)
            self.single_fidelity.to(ctx.device)
        elif callable(data_fidelity):
            if num_operators is None:
                raise ValueError("num_operators must be provided when using a factory.")
            for i in range(int(num_operators)):
                if i % ctx.world_size == ctx.rank:
                    self.local_data_fidelities.append(data_fidelity(i, ctx.device, factory_kwargs))
        else:
            raise ValueError("data_fidelity must be a DataFidelity instance or a factory callable.")

    def _get_fidelity(self, i: int):
        return self.single_fidelity if self.single_fidelity is not None else self.local_data_fidelities[i]

    def _apply_op(self, local_op, x, y, physics, gather=True, **kwargs):
        if not isinstance(physics, DistributedStackedLinearPhysics):
            raise ValueError("physics must be a DistributedStackedLinearPhysics instance to be used with DistributedDataFidelity.")
        y_local = physics._local(y)
        Ax_local = physics.A(x, gather=False, **kwargs)
        acc = None
        for idx, (Ax_i, y_i) in enumerate(zip(Ax_local, y_local)):
            t = local_op(idx, Ax_i, y_i)
            acc = t if acc is None else acc + t
        if acc is None:        # a rank without operators: a zero of the right shape joins the reduction
            acc = self._zero
        if self.reduction_mode == "mean":
            acc = acc / physics.num_operators
        if gather and self.ctx.world_size > 1:
            acc = acc.contiguous()
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        return acc

    def fn(self, x, y, physics, gather: bool = True, *args, **kwargs):
        """sum_i d(A_i x, y_i), one value per batch element"""
        self._zero = torch.zeros(x.shape[0], device=x.device, dtype=x.dtype)
        return self._apply_op(lambda idx, Ax_i, y_i: self._get_fidelity(idx).d.fn(Ax_i, y_i, *args), x, y, physics, gather, **kwargs)

    def grad(self, x, y, physics, gather: bool = True, *args, **kwargs):
        """sum_i A_i^T grad d(A_i x, y_i)"""
        self._zero = torch.zeros_like(x)
        return self._apply_op(lambda idx, Ax_i, y_i: physics.local_physics[idx].A_vjp(x, self._get_fidelity(idx).d.grad(Ax_i, y_i, *args)),
                              x, y, physics, gather, **kwargs)

    def __call__(self, x, y, physics, *args, **kwargs):
        return self.fn(x, y, physics, *args, **kwargs)


def distribute(obj, ctx: BatchParallelContext, *, num_operators: int | None = None, type_object: str | None = "auto",
               tiling_strategy: str | None = "overlap_tiling", tiling_dims=None, patch_size=256, overlap=64,
               max_batch_size: int | None = None, **kwargs):
    """`deepinv.distributed.distribute` (distribute.py:214-420) for the objects on this library's path:
      * a list of `LinearPhysics` / a factory ``f(index, device, kwargs)`` (+ `num_operators`)  -> `DistributedStackedLinearPhysics`
      * a `DataFidelity` / a list of them / a factory (+ `num_operators`, `type_object="data_fidelity"`)  -> `DistributedDataFidelity`
      * a `Denoiser` (any module called as ``model(x, sigma)``)  -> `DistributedProcessing` (overlap tiling)
    A callable needs an explicit `type_object` ('linear_physics', 'data_fidelity' or 'denoiser'), exactly as in the reference."""
    from .models.base import Denoiser
    from .optim.data_fidelity import DataFidelity

    is_list = isinstance(obj, (list, tuple)) and len(obj) > 0
    if type_object == "auto":
        if is_list and isinstance(obj[0], LinearPhysics):
            type_object = "linear_physics"
        elif isinstance(obj, DataFidelity) or (is_list and isinstance(obj[0], DataFidelity)):
            type_object = "data_fidelity"
        elif isinstance(obj, Denoiser):
            type_object = "denoiser"
        elif callable(obj):
            raise ValueError("For callable objects, you must specify type_object parameter")
        else:
            raise ValueError(f"Cannot auto-detect type for object: {type(obj)}")
    if type_object in ("physics", "linear_physics"):
        if is_list:
            ops = list(obj)
            return DistributedStackedLinearPhysics(ctx, len(ops), lambda i, device, _: ops[i].to(device), **kwargs)
        if not callable(obj) or num_operators is None:
            raise ValueError("a physics factory needs num_operators")
        return DistributedStackedLinearPhysics(ctx, num_operators, obj, **kwargs)
    if type_object == "data_fidelity":
        if is_list:
            dfs = list(obj)
            return DistributedDataFidelity(ctx, lambda i, device, _: dfs[i].to(device), num_operators=len(dfs), **kwargs)
        return DistributedDataFidelity(ctx, obj, num_operators=num_operators, **kwargs)
    if type_object == "denoiser":
        if tiling_strategy not in (None, "overlap_tiling"):
            raise ValueError("only the 'overlap_tiling' strategy is implemented")
        return DistributedProcessing(ctx, obj, strategy=tiling_strategy,
                                     strategy_kwargs={"patch_size": patch_size, "overlap": overlap, "tiling_dims": tiling_dims},
                                     max_batch_size=max_batch_size)
    raise ValueError(f"Unsupported type_object: {type_object}")
