#!/usr/bin/env python
"""Golden vectors for the least-squares solvers from the REAL reference (deepinv v0.4.1, oracle/ref_shim.py):
`least_squares(solver=CG | BiCGStab | lsqr | minres)` with and without the proximal term, batched gamma, rectangular
and square operators (reference tests: test_optim.py:1110-1180).  Plain matrix operators, float64.

    python tests/golden/make_golden_solvers.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ref_shim import import_reference  # noqa: E402

dinv = import_reference()
from deepinv.optim.linear import least_squares  # noqa: E402

g = torch.Generator().manual_seed(5)
out = {}
for tag, (m, n) in (("tall", (14, 9)), ("wide", (7, 12)), ("square", (10, 10))):
    M = torch.randn(3, m, n, generator=g, dtype=torch.float64)          # one operator per batch sample
    if tag == "square":
        M = M @ M.transpose(1, 2) + 0.5 * torch.eye(n, dtype=torch.float64)   # symmetric positive definite
    y = torch.randn(3, m, generator=g, dtype=torch.float64)
    z = torch.randn(3, n, generator=g, dtype=torch.float64)
    A = lambda x, M=M: torch.einsum("bij,bj->bi", M, x)
    AT = lambda v, M=M: torch.einsum("bij,bi->bj", M, v)
    out[f"{tag}_M"], out[f"{tag}_y"], out[f"{tag}_z"] = M.numpy(), y.numpy(), z.numpy()
    for solver in ("CG", "BiCGStab", "lsqr", "minres"):
        for gname, gamma in (("none", None), ("scalar", 0.7), ("batched", torch.tensor([0.3, 1.0, 2.5], dtype=torch.float64))):
            x = least_squares(A, AT, y, z=z if gamma is not None else 0.0, init=z if gamma is not None else None, gamma=gamma,
                              solver=solver, max_iter=200, tol=1e-10)
            out[f"{tag}_{solver}_{gname}"] = x.numpy()
            print(tag, solver, gname, x[0, :3].numpy())
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ls_solvers.npz"), **out)
