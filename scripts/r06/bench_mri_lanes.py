"""MultiCoilMRI A / A^T / A^T A at cfg2 (8 coils, 320 x 320) with the batch cut into 1 / 2 / 3 / 4 lanes (hip/mri.py: MRI_LANES): ms per
call over 50 back-to-back calls, and bit-identity with the single launch sequence.   python scripts/r06/bench_mri_lanes.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import deepinv_amd as dinv  # noqa: E402
from deepinv_amd.hip import mri as M  # noqa: E402

dev = torch.device("cuda:0")


def t_op(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B in (32, 16, 8, 4):
    physics, x, y, maps, mask = bench.make_problem(dinv, B, 0, 320, 320, 8, dev)
    M.MRI_LANES = 1
    ref = (physics.A(x), physics.A_adjoint(y), physics.A_adjoint_A(x))
    for lanes in (1, 2, 3, 4):
        if lanes > B:
            continue
        M.MRI_LANES = lanes
        out = (physics.A(x), physics.A_adjoint(y), physics.A_adjoint_A(x))
        same = all(torch.equal(a, b) for a, b in zip(out, ref))
        row = {"batch": B, "lanes": lanes, "A_ms": round(t_op(lambda: physics.A(x)), 4), "AT_ms": round(t_op(lambda: physics.A_adjoint(y)), 4),
               "ATA_ms": round(t_op(lambda: physics.A_adjoint_A(x)), 4), "bit_identical_to_one_lane": same}
        print(json.dumps(row), flush=True)
    del physics, x, y
