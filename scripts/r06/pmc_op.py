"""ONE operator, `calls` times, in a process of its own - the unit of the round-6 counter passes (scripts/r06/pmc_ops.sh): every launch of
the product's kernels in this process belongs to the named operator, so bytes per call = sum over those launches / calls.
    python scripts/r06/pmc_op.py <op name as in bench.py> [calls]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepinv_amd as dinv  # noqa: E402

name, calls = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
B = 32
if name.startswith("Blur"):
    img = (3, 256, 256)
    x = torch.rand(B, *img, generator=g).to(dev)
    k = dinv.physics.functional.gaussian_blur(psf_size=(9, 9), sigma=(2.0, 2.0)).to(dev)
    if name.startswith("BlurFFT"):
        p = dinv.physics.BlurFFT(img_size=img, filter=k, device=dev)
    else:
        p = dinv.physics.Blur(filter=k, padding=name[name.index("(") + 1:name.index(")")], device=dev)
else:
    H = W = 320
    x = torch.rand(B, 2, H, W, generator=g).to(dev)
    p = dinv.physics.MRI(mask=dinv.utils.radial_mask(H, W, 80).to(dev), img_size=(2, H, W), device=dev)
torch.cuda.synchronize()
# the operator under the counters: exactly `calls` applications
op = name.split(".")[-1]
if op == "A":
    for _ in range(calls):
        p.A(x)
elif op == "A_adjoint":
    # the measurement is a filled buffer of the operator's output shape (no forward launch under the counters)
    hh = x.shape[-2] - (8 if "valid" in name else 0)
    y = (torch.empty(B, 3, hh, hh, device=dev) if name.startswith("Blur(") else torch.empty_like(x)).uniform_()
    for _ in range(calls):
        p.A_adjoint(y)
elif op == "prox_l2":
    y = torch.empty_like(x).uniform_()
    for _ in range(calls):
        p.prox_l2(x, y, 1.3)
torch.cuda.synchronize()
print("done", name, calls)
