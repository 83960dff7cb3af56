"""deepinv_amd.Trainer driving unfolded training on the HIP path against the REAL reference's Trainer (golden trainer.npz,
tests/golden/make_golden_trainer.py): per-epoch training loss and every trained parameter after two epochs of Adam."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_trainer_unfolded_pgd_matches_reference(dev):
    import deepinv_amd as dinv

    raw = np.load(os.path.join(G, "trainer.npz"))
    t = lambda k: torch.from_numpy(raw[k])
    x, y = t("x"), t("y")
    H = W = 32
    phys = dinv.physics.MultiCoilMRI(mask=t("mask").to(dev), coil_maps=torch.view_as_complex(t("maps").contiguous()).to(dev),
                                     img_size=(2, H, W), device=dev)
    assert rel_err(phys.A(x.to(dev)), y) < 1e-4

    class Den(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv2d(2, 2, 3, padding=1)
            with torch.no_grad():
                self.c.weight.copy_(t("wden"))
                self.c.bias.copy_(t("bden"))

        def forward(self, u, s):
            return u - s * self.c(u)

    model = dinv.unfolded.unfolded_builder("PGD", data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(Den()),
                                           params_algo={"stepsize": 0.8, "g_param": 0.05, "lambda": 1.0}, max_iter=3,
                                           trainable_params=["stepsize", "g_param"], device=dev)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 4

        def __getitem__(self, i):
            return x[i], y[i]

    loader = torch.utils.data.DataLoader(DS(), batch_size=2, shuffle=False)
    model.to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    trainer = dinv.Trainer(model=model, physics=phys, optimizer=opt, train_dataloader=loader, epochs=2, losses=dinv.training.SupLoss(),
                           device=dev, verbose=False, compute_train_metrics=False)
    trainer.train()
    assert np.allclose(trainer.train_loss_history, raw["loss_history"], rtol=1e-4)
    for n, p in model.named_parameters():
        ref = t("param_" + n.replace(".", "_"))
        assert rel_err(p.detach(), ref) < 1e-4, n
    # evaluation entry point
    res = trainer.test(loader)
    assert 5 < res["PSNR"] < 60


def test_trainer_online_measurements_and_generator(dev):
    """online measurements with a physics generator (trainer.py:662-707): every batch draws a new Cartesian mask on the
    device, measurements are synthesised by the operator incl. its Gaussian noise model, the loss decreases"""
    import deepinv_amd as dinv

    H = W = 64
    torch.manual_seed(0)
    gen = dinv.physics.generator.GaussianMaskGenerator((2, H, W), acceleration=4, device=dev)
    phys = dinv.physics.MRI(img_size=(2, H, W), device=dev, noise_model=dinv.physics.GaussianNoise(0.01))
    x = torch.rand(8, 2, H, W)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x), batch_size=4)

    class Wrap(torch.utils.data.Dataset):
        def __len__(self):
            return 8

        def __getitem__(self, i):
            return x[i]

    loader = torch.utils.data.DataLoader(Wrap(), batch_size=4)

    class Den(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv2d(2, 2, 3, padding=1)

        def forward(self, u, s):
            return u - s * self.c(u)

    model = dinv.unfolded.unfolded_builder("PGD", data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(Den()),
                                           params_algo={"stepsize": 1.0, "g_param": 0.05, "lambda": 1.0}, max_iter=2,
                                           trainable_params=["stepsize", "g_param"], device=dev).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    tr = dinv.Trainer(model=model, physics=phys, optimizer=opt, train_dataloader=loader, epochs=6, online_measurements=True,
                      physics_generator=gen, device=dev, verbose=False, grad_clip=1.0, check_grad=True)
    tr.train()
    assert len(tr.train_loss_history) == 6 and tr.train_loss_history[-1] < tr.train_loss_history[0]
