#!/usr/bin/env python
"""Golden for the training loop from the REAL reference (deepinv.Trainer, deepinv/training/trainer.py:1002-1492):
unfolded PGD (3 iterations, trainable stepsize / g_param / denoiser conv) on 2-D 4-coil MRI, two epochs over two batches
(offline measurements), Adam: per-epoch training loss and every parameter after training.

    python tests/golden/make_golden_trainer.py
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
g = lambda s: torch.Generator().manual_seed(s)  # noqa: E731

H = W = 32
coils, B, nb = 4, 2, 2
x = torch.rand(nb * B, 2, H, W, generator=g(80))
maps = torch.randn(1, coils, H, W, dtype=torch.complex64, generator=g(81)) / coils ** 0.5
mask = (torch.rand(H, W, generator=g(82)) > 0.5).float()
phys = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, H, W))
y = phys.A(x)
wden = torch.randn(2, 2, 3, 3, generator=g(83)) * 0.1
bden = torch.randn(2, generator=g(84)) * 0.01


class Den(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.c = torch.nn.Conv2d(2, 2, 3, padding=1)
        with torch.no_grad():
            self.c.weight.copy_(wden)
            self.c.bias.copy_(bden)

    def forward(self, u, s):
        return u - s * self.c(u)


model = dinv.unfolded.unfolded_builder("PGD", data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(Den()),
                                       params_algo={"stepsize": 0.8, "g_param": 0.05, "lambda": 1.0}, max_iter=3,
                                       trainable_params=["stepsize", "g_param"])
data = [(x[i * B:(i + 1) * B], y[i * B:(i + 1) * B]) for i in range(nb)]


class DS(torch.utils.data.Dataset):
    def __len__(self):
        return nb * B

    def __getitem__(self, i):
        return x[i], y[i]


loader = torch.utils.data.DataLoader(DS(), batch_size=B, shuffle=False)
opt = torch.optim.Adam(model.parameters(), lr=1e-2)
np.random.seed(0)
trainer = dinv.Trainer(model=model, physics=phys, optimizer=opt, train_dataloader=loader, epochs=2, losses=dinv.loss.SupLoss(),
                       device="cpu", save_path=None, verbose=False, show_progress_bar=False, plot_images=False,
                       compute_train_metrics=False, ckp_interval=10 ** 6, metrics=None)
trainer.train()
out = {"x": x, "maps": torch.view_as_real(maps), "mask": mask, "y": y, "wden": wden, "bden": bden,
       "loss_history": np.asarray(trainer.loss_history["SupLoss"], dtype=np.float64)}
for n, p in model.named_parameters():
    out["param_" + n.replace(".", "_")] = p.detach()
np.savez_compressed(os.path.join(OUT, "trainer.npz"), **{k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in out.items()})
print({k: getattr(v, "shape", None) for k, v in out.items()})
