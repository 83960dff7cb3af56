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
