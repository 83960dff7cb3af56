"""cfg5's Downsampling operators (16 images 3 x 256 x 256, bicubic x4, circular) under rocprofv3 --kernel-trace: which launches make up
A (0.060 ms), A_adjoint (0.093 ms) and prox_l2 (0.186 ms)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepinv_amd as dinv  # noqa: E402

dev = torch.device("cuda:0")
img = (3, 256, 256)
p = dinv.physics.Downsampling(img_size=img, filter="bicubic", factor=4, padding="circular", device=dev)
x = torch.rand(16, *img, device=dev)
y = p.A(x)
which = sys.argv[1]
torch.cuda.synchronize()
for _ in range(20):
    if which == "A":
        p.A(x)
    elif which == "AT":
        p.A_adjoint(y)
    else:
        p.prox_l2(x, y, 1.3)
torch.cuda.synchronize()
