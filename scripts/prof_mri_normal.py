"""Run the fused MRI normal operator (and the A / A^T chain) a few times for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepinv_amd as dinv  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "2d"
B, coils, img, three_d = (32, 8, (320, 320), False) if cfg == "2d" else (2, 12, (16, 256, 256), True)
x = torch.rand(B, 2, *img, generator=g).to(dev)
maps = (torch.randn(1, coils, *img, dtype=torch.complex64, generator=g) / coils ** 0.5).to(dev)
mask = (torch.rand(*img, generator=g) > 0.75).float().to(dev)
phys = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, *img), three_d=three_d, device=dev)
for _ in range(20):
    phys.A_adjoint_A(x)
if len(sys.argv) > 2:
    for _ in range(20):
        phys.A_adjoint(phys.A(x))
torch.cuda.synchronize()
