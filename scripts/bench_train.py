"""One training step (forward + backward) of unfolded PnP-PGD with a DRUNet prior on 2-D multi-coil MRI:
the hand-written DRUNet backward (models/drunet_train.py) vs the PyTorch-ROCm graph (MIOpen).
Usage: python scripts/bench_train.py [B] [iters]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepinv_amd as dinv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
img, coils = (320, 320), 8
x = torch.rand(B, 2, *img, generator=g).to(dev)
maps = (torch.randn(1, coils, *img, dtype=torch.complex64, generator=g) / coils ** 0.5).to(dev)
mask = (torch.rand(*img, generator=g) > 0.75).float().to(dev)
phys = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, *img), device=dev)
y = phys.A(x)
torch.manual_seed(0)
den = dinv.models.DRUNet(2, 2, pretrained=None).to(dev)
model = dinv.unfolded.unfolded_builder("PGD", data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den),
                                       params_algo={"stepsize": 0.8, "g_param": 0.05, "lambda": 1.0}, max_iter=iters,
                                       trainable_params=["stepsize", "g_param"], device=dev).to(dev)


def step():
    model.zero_grad()
    loss = (model(y, phys) - x).pow(2).mean()
    loss.backward()
    return loss


for mode, prec in (("torch", ""), ("hip", "fp32"), ("hip", "bf16split")):
    den.__dict__.pop("forward", None)
    if mode == "torch":     # the PyTorch-ROCm graph of the same module lives with the tests (the product has no such path)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        from torch_drunet import torch_forward
        den.forward = lambda xx, ss, _d=den: torch_forward(_d, xx, ss)
    if prec:
        den.train_forward_precision = prec
    step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"mode": mode, "forward_precision": prec or "fp32 (MIOpen)", "B": B, "unrolled_iterations": iters,
                      "s_per_step": round(dt, 4), "loss": float(loss), "peak_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))
