"""BASELINE config 5's loop at its per-GPU shard (DiffPIR 100 steps, DRUNet(3->3), x4 super-resolution of 16 images 3x256x256) for
rocprofv3: one warm-up call, then `reps` timed calls; prints ms per call (scripts/prof.sh r04_cfg5 scripts/r04/prof_cfg5.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import deepinv_amd as dinv

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, img = 16, (3, 256, 256)
torch.manual_seed(0)
den = dinv.models.DRUNet(3, 3, pretrained=None).to(dev).eval()
phys = dinv.physics.Downsampling(img_size=img, filter="bicubic", factor=4, padding="circular", device=dev,
                                 noise_model=dinv.physics.GaussianNoise(0.05))
sampler = dinv.sampling.DiffPIR(den, dinv.optim.L2(), sigma=0.05, max_iter=steps, zeta=0.1, lambda_=7.0, device=dev)
x = torch.rand(B, *img, generator=g).to(dev)
y = phys(x)
out = sampler(y, phys, seed=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = sampler(y, phys, seed=0)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / reps
with torch.no_grad():
    den_ms = []
    u = torch.rand(B, *img, device=dev)
    for _ in range(3):
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(10):
            den(u, 0.05)
        torch.cuda.synchronize(); den_ms.append((time.perf_counter() - t1) * 100)
print({"loop_ms": round(ms, 2), "ms_per_step": round(ms / steps, 3), "denoiser_ms": round(min(den_ms), 3), "finite": bool(torch.isfinite(out).all()),
       "checksum": float(out.double().sum())})
