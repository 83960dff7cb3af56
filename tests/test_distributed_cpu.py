"""Batch data-parallel path on 2 CPU processes over gloo (the GPU run uses RCCL through the same code).
Pattern follows the reference's test_distributed.py:194-296 (spawned local workers, 127.0.0.1, free port)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import deepinv_amd as dinv

    torch.manual_seed(0)
    M = torch.randn(6, 4)

    class P(dinv.physics.LinearPhysics):
        def A(self, x, **k):
            return x @ M.T

        def A_adjoint(self, y, **k):
            return y @ M

    phys = P()
    y_full = torch.randn(n_total, 6, generator=torch.Generator().manual_seed(1))
    model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), stepsize=0.02, max_iter=20)
    with dinv.distributed.BatchParallelContext(backend="gloo", device="cpu") as ctx:
        sl = ctx.slab(n_total)
        rec = dinv.distributed.reconstruct_batch_parallel(ctx, model, y_full, phys)
        q.put((rank, sl.start, sl.stop, rec.clone().numpy()))   # by value: a tensor would travel as a shared-memory
        # handle that can vanish when this process exits before the parent maps it


@pytest.mark.parametrize("n_total", [8, 7])
def test_batch_parallel_equals_single_process(n_total):
    import deepinv_amd as dinv

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference
    torch.manual_seed(0)
    M = torch.randn(6, 4)

    class P(dinv.physics.LinearPhysics):
        def A(self, x, **k):
            return x @ M.T

        def A_adjoint(self, y, **k):
            return y @ M

    y_full = torch.randn(n_total, 6, generator=torch.Generator().manual_seed(1))
    ref = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), stepsize=0.02, max_iter=20)(y_full, P())
    slabs = sorted((s, e) for _, s, e, _ in results)
    assert slabs[0][0] == 0 and slabs[-1][1] == n_total and slabs[0][1] == slabs[1][0]   # contiguous cover
    for _, _, _, rec in results:
        rec = torch.from_numpy(rec)
        assert rec.shape == ref.shape
        assert torch.allclose(rec, ref, atol=1e-6)   # every rank holds the full gathered batch


def _stack_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import deepinv_amd as dinv
    from deepinv_amd.distributed import BatchParallelContext, DistributedStackedLinearPhysics

    g = torch.Generator().manual_seed(0)
    Ms = [torch.randn(5, 4, generator=g) for _ in range(5)]      # 5 operators over 2 ranks: ragged (3 + 2)

    class Mat(dinv.physics.LinearPhysics):
        def __init__(self, M):
            super().__init__()
            self.M = M

        def A(self, x, **k):
            return x @ self.M.T

        def A_adjoint(self, y, **k):
            return y @ self.M

    x = torch.randn(3, 4, generator=g)
    with BatchParallelContext(backend="gloo", device="cpu") as ctx:
        phys = DistributedStackedLinearPhysics(ctx, 5, lambda i, dev, kw: Mat(Ms[i]))
        y_local = phys.A(x, gather=False)
        y_all = phys.A(x, gather=True)
        aty = phys.A_adjoint(y_all)                 # full list in, all-reduced image out
        aty2 = phys.A_adjoint(y_local)              # local list in: same result
        ata = phys.A_adjoint_A(x)
        part = phys.A_adjoint(y_local, reduce_op=None)
        # a PGD loop on the distributed operator: L2.grad = A^T A x - A^T y, both all-reduced
        ident = dinv.optim.PnP(denoiser=lambda x, sigma=None, **k: x)      # implicit prior: no cost on the list y
        model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=ident, stepsize=0.01, max_iter=10)
        rec = model(y_local, phys)
        q.put((rank, [t.numpy() for t in y_all], aty.numpy(), aty2.numpy(), ata.numpy(), part.numpy(), rec.numpy(),
               phys.local_indexes))


def test_operator_parallel_stack_matches_single_process():
    """DistributedStackedLinearPhysics on 2 gloo ranks (ragged 3 + 2 operators) == the stacked operator in one process:
    gathered A, all-reduced A^T / A^T A, and a PGD reconstruction through them (distrib_framework.py:387-560)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stack_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    Ms = [torch.randn(5, 4, generator=g) for _ in range(5)]
    x = torch.randn(3, 4, generator=g)
    ys = [x @ M.T for M in Ms]
    aty = sum(y @ M for y, M in zip(ys, Ms))
    rec = aty.clone()
    for _ in range(10):
        rec = rec - 0.01 * (sum((rec @ M.T) @ M for M in Ms) - aty)
    assert results[0][7] == [0, 2, 4] and results[1][7] == [1, 3]
    parts = []
    for rank, y_all, a1, a2, ata, part, r, _ in results:
        for yi, ref in zip(y_all, ys):
            assert torch.allclose(torch.from_numpy(yi), ref, atol=1e-6)
        assert torch.allclose(torch.from_numpy(a1), aty, atol=1e-5) and torch.allclose(torch.from_numpy(a2), aty, atol=1e-5)
        assert torch.allclose(torch.from_numpy(ata), aty, atol=1e-5)
        assert torch.allclose(torch.from_numpy(r), rec, atol=1e-5)
        parts.append(torch.from_numpy(part))
    assert torch.allclose(parts[0] + parts[1], aty, atol=1e-5) and not torch.allclose(parts[0], aty, atol=1e-3)


def _blur5(x, sigma=None):
    """shape-preserving map with receptive-field radius 2 and reflect boundary handling"""
    k = torch.tensor([1., 4., 6., 4., 1.]) / 16
    k2 = (k[:, None] * k[None, :])[None, None].repeat(x.shape[1], 1, 1, 1)
    return torch.nn.functional.conv2d(torch.nn.functional.pad(x, (2, 2, 2, 2), mode="reflect"), k2, groups=x.shape[1]) + x ** 2


def test_overlap_tiling_partitions_the_signal():
    from deepinv_amd.distributed import OverlapTiling

    for shape, p, h in (((2, 3, 37, 50), 16, 4), ((1, 1, 16, 16), 16, 2), ((1, 2, 8, 40, 33), (4, 16, 16), (1, 3, 2))):
        T = OverlapTiling(shape, patch_size=p, overlap=h)
        x = torch.arange(float(torch.tensor(shape).prod())).reshape(shape)
        xp = T.pad(x)
        out = torch.full_like(x, float("nan"))
        count = torch.zeros_like(x)
        sizes = set()
        for k in range(len(T)):
            w = T.window(xp, k)
            sizes.add(tuple(w.shape))
            T.place(out, k, w)          # identity processor: placing the inner part of each window rebuilds x
            one = torch.zeros_like(x)
            T.place(one, k, torch.ones_like(w))
            count += one
        assert len(sizes) == 1          # equal windows: they can share one batch
        assert torch.equal(out, x) and torch.all(count == 1)


def _tile_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from deepinv_amd.distributed import BatchParallelContext, DistributedProcessing

    x = torch.rand(2, 3, 45, 70, generator=torch.Generator().manual_seed(3))
    with BatchParallelContext(backend="gloo", device="cpu") as ctx:
        proc = DistributedProcessing(ctx, _blur5, strategy="overlap_tiling",
                                     strategy_kwargs={"patch_size": 16, "overlap": 2}, max_batch_size=4)
        y = proc(x, 0.1)
        y_local = proc(x, 0.1, gather=False)
        q.put((rank, y.numpy(), y_local.numpy()))


def test_tile_parallel_processing_matches_untiled():
    """DistributedProcessing on 2 gloo ranks: halo = receptive-field radius => identical to processing the whole signal
    (distrib_framework.py:734-934); the local contributions are disjoint and sum to it"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = torch.rand(2, 3, 45, 70, generator=torch.Generator().manual_seed(3))
    ref = _blur5(x)
    for _, y, _ in results:
        assert torch.allclose(torch.from_numpy(y), ref, atol=1e-6)
    parts = [torch.from_numpy(r[2]) for r in results]
    assert torch.allclose(parts[0] + parts[1], ref, atol=1e-6)
    assert torch.all((parts[0] == 0) | (parts[1] == 0))


def _distribute_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import deepinv_amd as dinv
    from deepinv_amd.distributed import BatchParallelContext, DistributedProcessing, distribute

    g = torch.Generator().manual_seed(0)
    Ms = [torch.randn(5, 4, generator=g) for _ in range(3)]

    class Mat(dinv.physics.LinearPhysics):
        def __init__(self, M):
            super().__init__()
            self.M = M

        def A(self, x, **k):
            return x @ self.M.T

        def A_adjoint(self, y, **k):
            return y @ self.M

    x = torch.randn(2, 4, generator=g)
    with BatchParallelContext(backend="gloo", device="cpu") as ctx:
        phys = distribute([Mat(M) for M in Ms], ctx)                       # list of LinearPhysics -> stacked distributed
        df = distribute(dinv.optim.L2(sigma=0.5), ctx)                      # one fidelity for every operator
        df_list = distribute([dinv.optim.L2(sigma=0.5) for _ in Ms], ctx)   # ... or one per operator
        y = phys.A(x + 1.0, gather=True)
        f, gr = df.fn(x, y, phys), df.grad(x, y, phys)
        f2, gr2 = df_list.fn(x, y, phys), df_list.grad(x, y, phys)
        fm = distribute(dinv.optim.L2(sigma=0.5), ctx, reduction="mean").fn(x, y, phys)
        # ONE operator on two ranks: rank 1 owns nothing and must still join the all-reduce with a zero image
        single = distribute([Mat(Ms[0])], ctx, img_shape=(2, 4))
        y1 = single.A(x, gather=False)
        aty = single.A_adjoint(y1 if y1 else [])
        ata = single.A_adjoint_A(x)
        # per-sample arguments follow the windows on the batch axis
        sig = torch.tensor([0.5, 2.0]).view(2, 1, 1, 1)
        proc = DistributedProcessing(ctx, lambda z, s: z * s.view(-1, 1, 1, 1), strategy_kwargs={"patch_size": 8, "overlap": 2})
        img = torch.rand(2, 1, 16, 24, generator=g)
        scaled = proc(img, sig.view(2))
        try:
            distribute(lambda i, d, k: Mat(Ms[i]), ctx)
            err = ""
        except ValueError as e:
            err = str(e)
        q.put((rank, f.numpy(), gr.numpy(), f2.numpy(), gr2.numpy(), fm.numpy(), aty.numpy(), ata.numpy(), scaled.numpy(),
               (img * sig).numpy(), err, type(phys).__name__, type(df).__name__))


def test_distribute_factory_and_distributed_data_fidelity():
    """distribute() (distribute.py:214-420) and DistributedDataFidelity (distrib_framework.py:940-1180) on 2 gloo ranks: value
    and gradient equal the single-process sums; a rank without operators contributes zeros instead of dead-locking the
    all-reduce; per-sample tensor arguments of a tiled processor are repeated with the windows"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_distribute_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    Ms = [torch.randn(5, 4, generator=g) for _ in range(3)]
    x = torch.randn(2, 4, generator=g)
    ys = [(x + 1.0) @ M.T for M in Ms]
    f_ref = sum(0.5 * ((x @ M.T - y) ** 2).sum(1) / 0.25 for M, y in zip(Ms, ys))
    g_ref = sum(((x @ M.T - y) / 0.25) @ M for M, y in zip(Ms, ys))
    for r in results:
        _, f, gr, f2, gr2, fm, aty, ata, scaled, scaled_ref, err, tp, td = r
        assert tp == "DistributedStackedLinearPhysics" and td == "DistributedDataFidelity"
        for a, b in ((f, f_ref), (f2, f_ref), (gr, g_ref), (gr2, g_ref), (fm, f_ref / 3)):
            assert torch.allclose(torch.from_numpy(a), b, atol=1e-5)
        assert torch.allclose(torch.from_numpy(aty), (x @ Ms[0].T) @ Ms[0], atol=1e-5)
        assert torch.allclose(torch.from_numpy(ata), (x @ Ms[0].T) @ Ms[0], atol=1e-5)
        assert torch.allclose(torch.from_numpy(scaled), torch.from_numpy(scaled_ref), atol=1e-6)
        assert "type_object" in err
