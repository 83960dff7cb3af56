#!/usr/bin/env python
"""Golden vectors for loop options of `BaseOptim` / `FixedPoint` from the REAL reference (deepinv v0.4.1 at
/root/reference through oracle/ref_shim.py): plain PGD and backtracking (optimizers.py:655-701).  Plain matrix physics in float64, so the vectors pin the loop logic itself.

    python tests/golden/make_golden_optim.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ref_shim import import_reference  # noqa: E402

dinv = import_reference()
OUT = os.path.dirname(os.path.abspath(__file__))


class MatPhysics(dinv.physics.LinearPhysics):
    def __init__(self, M):
        super().__init__()
        self.M = M

    def A(self, x, **kw):
        return x @ self.M.T

    def A_adjoint(self, y, **kw):
        return y @ self.M


class L1(dinv.optim.Prior):
    def __init__(self):
        super().__init__()
        self.explicit_prior = True

    def fn(self, x, *a, **k):
        return x.abs().sum(dim=-1)

    def prox(self, x, *a, gamma=1.0, **k):
        return torch.sign(x) * torch.clamp(x.abs() - gamma, min=0)


g = torch.Generator().manual_seed(11)
M = torch.randn(9, 6, generator=g, dtype=torch.float64)
y = torch.randn(3, 9, generator=g, dtype=torch.float64)
step = 0.9 / float(torch.linalg.matrix_norm(M, 2) ** 2)
phys = MatPhysics(M)
out = {"M": M.numpy(), "y": y.numpy(), "step": np.float64(step)}

from deepinv.optim.optimizers import BacktrackingConfig  # noqa: E402

cases = {
    "pgd_plain": dict(),
    "pgd_backtracking": dict(backtracking=BacktrackingConfig(gamma=0.1, eta=0.5, max_iter=20), stepsize_scale=8.0),
}
for name, kw in cases.items():
    kw = dict(kw)
    scale = kw.pop("stepsize_scale", 1.0)
    model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=L1(), lambda_reg=0.2, stepsize=step * scale, max_iter=12,
                           early_stop=False, **kw)
    with torch.no_grad():
        x = model(y, phys)
    out[name] = x.numpy()
    print(name, x[0, :3].numpy())
np.savez_compressed(os.path.join(OUT, "optim_loop_options.npz"), **out)
