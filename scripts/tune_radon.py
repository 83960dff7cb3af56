"""Tomography operator timings at BASELINE config 3 (512x512, 720 angles, 8 images) for the tuning knobs of the tiled
kernels: run as   DINV_RADON_KW=4 DINV_RADON_NB=4 python scripts/tune_radon.py   (knobs are read once per process)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepinv_amd as dinv  # noqa: E402

dev = torch.device("cuda:0")
B, W, nang = int(os.environ.get("TUNE_B", "8")), int(os.environ.get("TUNE_W", "512")), int(os.environ.get("TUNE_A", "720"))
phys = dinv.physics.Tomography(angles=nang, img_width=W, normalize=False, device=dev)
x = torch.rand(B, 1, W, W, device=dev)
y = phys.A(x)


def timeit(fn, iters=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


geo = phys._geometry(dev)
res = {"B": B, "W": W, "angles": nang, "kw": geo.plan.kw if geo.plan else None, "win_w": geo.plan.win_w if geo.plan else None,
       "env": {k: v for k, v in os.environ.items() if k.startswith("DINV_R")},
       "A_ms": timeit(lambda: phys.A(x)), "AT_ms": timeit(lambda: phys.A_adjoint(y)), "ramp_ms": timeit(lambda: phys.filter(y)),
       "fbp_ms": timeit(lambda: phys.A_dagger(y, fbp=True))}
print(json.dumps(res))
