#!/usr/bin/env python
"""Golden vectors for DRUNet(dim=3) from the REAL reference (deepinv v0.4.1 at /root/reference through
oracle/ref_shim.py; deepinv/models/drunet.py:39-263 with Conv3d / ConvTranspose3d): BASELINE config 4's denoiser in
miniature (nc = 16, 32, 64, 128; nb = 1), weights from `torch.manual_seed(7)` + the reference's own initialisation (the
product module reproduces them bit for bit from the same seed: same parameter order, same state_dict keys, including
`m_body.res.*` for nb = 1).

* `drunet3d.npz`  x [1,2,16,32,32], noise map, y = model(x, sigma); for L = sum(y * v): dL/dx, dL/dsigma, and the
                  gradients of the head, tail, one 3x3x3 ResBlock conv, one 2x2x2 down conv, one 2x2x2 up conv, plus the
                  norms of ALL weight gradients

    python tests/golden/make_golden_drunet3d.py
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
KEEP = ("m_head.weight", "m_tail.weight", "m_down1.0.res.0.weight", "m_down1.1.weight", "m_up1.0.weight", "m_up2.1.res.2.weight")

torch.manual_seed(7)
model = dinv.models.DRUNet(in_channels=2, out_channels=2, nc=(16, 32, 64, 128), nb=1, pretrained=None, dim=3)
g = torch.Generator().manual_seed(11)
x = torch.rand(1, 2, 16, 32, 32, generator=g).requires_grad_(True)
sigma = (0.05 + 0.1 * torch.rand(1, 1, 16, 32, 32, generator=g)).requires_grad_(True)
v = torch.randn(1, 2, 16, 32, 32, generator=g)
model.train()            # gradients; the forward is the same function (no dropout / batch norm in DRUNet)
y = model(x, sigma)
(y * v).sum().backward()
arrs = {"x": x.detach().numpy(), "sigma": sigma.detach().numpy(), "v": v.numpy(), "y": y.detach().numpy(),
        "gx": x.grad.numpy(), "gsigma": sigma.grad.numpy()}
names, norms = [], []
for n, p in model.named_parameters():
    names.append(n)
    norms.append(float(p.grad.norm()))
    if n in KEEP:
        arrs["gw_" + n] = p.grad.numpy()
arrs["names"] = np.array(names)
arrs["gw_norms"] = np.array(norms)
np.savez_compressed(os.path.join(OUT, "drunet3d.npz"), **arrs)
print({k: getattr(a, "shape", None) for k, a in arrs.items()})
