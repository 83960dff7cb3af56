"""BASELINE config 3's loop at its per-GPU shard (FBP-initialised PnP-HQS, 30 iterations with the prox by CG, DRUNet(1->1), 8 images
512x512, 720 angles) for rocprofv3: one warm-up call, then `reps` timed calls (scripts/prof.sh r04_cfg3 scripts/r04/prof_cfg3.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import deepinv_amd as dinv

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, W, A = 8, 512, 720
phys = dinv.physics.Tomography(angles=A, img_width=W, circle=False, normalize=False, device=dev)
x = torch.rand(B, 1, W, W, generator=g).to(dev)
y = phys.A(x)
torch.manual_seed(0)
den = dinv.models.DRUNet(1, 1, pretrained=None).to(dev).eval()
s = np.logspace(np.log10(49 / 255.0), np.log10(0.02), 30).astype("float32")[:iters]
st = ((s / 0.02) ** 2 / 0.23).astype("float32")
hqs = dinv.optim.HQS(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den), stepsize=list(map(float, st)),
                     g_param=list(map(float, s)), max_iter=iters, early_stop=False,
                     custom_init=lambda yy, p: p.A_dagger(yy, fbp=True))
with torch.no_grad():
    out = hqs(y, phys)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = hqs(y, phys)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    u = torch.rand(B, 1, W, W, device=dev)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(10):
        den(u, 0.05)
    torch.cuda.synchronize(); dms = (time.perf_counter() - t1) * 100
print({"loop_ms": round(ms, 2), "ms_per_iteration": round(ms / iters, 3), "denoiser_ms": round(dms, 3), "finite": bool(torch.isfinite(out).all())})
