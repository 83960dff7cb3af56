"""Blur / BlurFFT / single-coil MRI operator rows alone (the rows bench.py prints, without the headline): quick A/B of csrc/blur.hip.
    python scripts/r06/bench_blur.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import deepinv_amd as dinv  # noqa: E402

dev = torch.device("cuda:0")
rows = []


def time_op(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def op_row(name, cfg, batch, fn, alg, n=50, **extra):
    t = time_op(fn, n)
    row = {"op": name, "config": cfg, "batch": batch, "ms": round(t * 1e3, 4), "alg_MB": round(alg / 1e6, 2), "GBps": round(alg / t / 1e9, 1),
           "frac_hbm_peak": round(alg / t / bench.HBM_PEAK, 4)}
    for k, v in extra.items():
        row[k] = round(v / t, 2) if k.endswith("_per_s") else v
    rows.append(row)
    print(json.dumps(row), flush=True)


bench.blur_and_single_coil_rows(dinv, dev, op_row)
# Downsampling rows of cfg5 (strided: the general gather kernels)
g = torch.Generator().manual_seed(0)
B, img = 16, (3, 256, 256)
phys = dinv.physics.Downsampling(img_size=img, filter="bicubic", factor=4, padding="circular", device=dev)
x = torch.rand(B, *img, generator=g).to(dev)
y = phys.A(x)
alg = B * 3 * (256 * 256 + 64 * 64) * 4
op_row("Downsampling.A", "cfg5", B, lambda: phys.A(x), alg)
op_row("Downsampling.A_adjoint", "cfg5", B, lambda: phys.A_adjoint(y), alg)
