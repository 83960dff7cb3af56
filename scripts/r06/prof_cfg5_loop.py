"""cfg5's 100-step DiffPIR loop (16 images 3 x 256 x 256) once more under rocprofv3 --kernel-trace: wall time of the loop against
the sum (and the union) of its kernel intervals - is the loop bound by the GPU or by the host's launch rate?"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepinv_amd as dinv  # noqa: E402

dev = torch.device("cuda:0")
img, B = (3, 256, 256), 16
g = torch.Generator().manual_seed(0)
x = torch.rand(B, *img, generator=g).to(dev)
p = dinv.physics.Downsampling(img_size=img, filter="bicubic", factor=4, padding="circular", device=dev,
                              noise_model=dinv.physics.GaussianNoise(0.05))
y = p.A(x)
den = dinv.models.DRUNet(3, 3, pretrained=None).to(dev).eval()
sampler = dinv.sampling.DiffPIR(den, dinv.optim.L2(), sigma=0.05, max_iter=100, zeta=0.1, lambda_=7.0, device=dev)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sampler(y, p)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("rep", rep, "wall_ms", round((t1 - t0) * 1e3, 1), "lanes", getattr(den, "_lane_calibration", None), flush=True)
