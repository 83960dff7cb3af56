#!/usr/bin/env python
"""Round-2 golden vectors from the REAL reference (deepinv v0.4.1 at /root/reference through oracle/ref_shim.py):

* `tomo_applyradon.npz`  Tomography(adjoint_via_backprop=False): A, A_adjoint (ApplyRadon / IRadon, radon.py:396-531),
                         fbp; circle False/True
* `tomo_normalized.npz`  Tomography(normalize=True) at 16x16 / 16 angles: operator_norm, A, A_adjoint, fbp
* `diffpir.npz`          DiffPIR (diffusion.py:289-513) on x4 super-resolution, 6 steps, every torch.randn_like draw
                         recorded so that the sample path can be replayed; schedule (rhos, sigmas, seq) stored
* `unfolded_pgd.npz`     unfolded_builder("PGD") (unfolded.py:116-226) on 3-D multi-coil MRI with a small conv
                         denoiser: loss and gradients of the trainable parameters and the denoiser weights
* `drunet_gain1.npz`     DRUNet(2->2) with O(1)-gain ResBlock weights (orthogonal gain 1.0, regenerated from a seed)

    python tests/golden/make_golden_r2.py
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


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach()
            v = torch.view_as_real(v).numpy() if v.is_complex() else v.numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items()})


def g(seed):
    return torch.Generator().manual_seed(seed)


# ---------------------------------------------------------------- Tomography, inexact-adjoint branch
x = torch.rand(2, 1, 16, 16, generator=g(30))
arrs = {"x": x}
for circle in (False, True):
    p = dinv.physics.Tomography(angles=12, img_width=16, circle=circle, normalize=False, adjoint_via_backprop=False,
                                device="cpu")
    y = p.A(x)
    v = torch.randn(y.shape, generator=g(31))
    c = int(circle)
    arrs.update({f"angles": p.angles, f"y_c{c}": y, f"v_c{c}": v, f"vadj_c{c}": p.A_adjoint(v),
                 f"fbp_c{c}": p.A_dagger(y, fbp=True)})
save("tomo_applyradon", **arrs)

# ---------------------------------------------------------------- Tomography, normalised
x = torch.rand(2, 1, 16, 16, generator=g(32))
p = dinv.physics.Tomography(angles=16, img_width=16, circle=False, normalize=True, device="cpu")
y = p.A(x)
v = torch.randn(y.shape, generator=g(33))
save("tomo_normalized", x=x, angles=p.angles, operator_norm=p.operator_norm, y=y, v=v, vadj=p.A_adjoint(v),
     fbp=p.A_dagger(y, fbp=True))

# ---------------------------------------------------------------- DiffPIR, x4 super-resolution, 6 steps
from oracle import drunet_cpu as OD  # noqa: E402

img, f, B, n_it = (3, 32, 32), 4, 2, 6
sd = OD.init_state_dict(3, 3, seed=5)
den = dinv.models.DRUNet(in_channels=3, out_channels=3, pretrained=None)
den.load_state_dict(sd)
den.eval()
x = torch.rand(B, *img, generator=g(34))
phys = dinv.physics.Downsampling(img_size=img, filter="bicubic", factor=f, padding="circular",
                                 noise_model=dinv.physics.GaussianNoise(0.05))
y = phys.A(x) + 0.05 * torch.randn(B, 3, 8, 8, generator=g(35))
draws = []
gen = g(36)
_orig = torch.randn_like


def recording_randn_like(t, **kw):
    d = torch.randn(t.shape, generator=gen)
    draws.append(d.clone())
    return d


torch.randn_like = recording_randn_like
sampler = dinv.sampling.DiffPIR(den, dinv.optim.L2(), sigma=0.05, max_iter=n_it, zeta=0.1, lambda_=7.0, device="cpu")
out = sampler(y, phys)
torch.randn_like = _orig
save("diffpir", x=x, y=y, k=phys.filter, out=out, draws=torch.stack(draws), rhos=sampler.rhos, sigmas=sampler.sigmas,
     seq=sampler.seq, drunet_seed=5)
# the schedule for two more settings (pins get_noise_schedule / get_alpha_beta, diffusion.py:323-375)
for tag, kw in (("a", dict(sigma=0.1, max_iter=100, lambda_=7.0)), ("b", dict(sigma=0.02, max_iter=20, lambda_=3.0))):
    s = dinv.sampling.DiffPIR(den, dinv.optim.L2(), zeta=0.1, device="cpu", **kw)
    save("diffpir_schedule_" + tag, rhos=s.rhos, sigmas=s.sigmas, seq=s.seq, reduced=s.reduced_alpha_cumprod,
         sigma=np.float32(kw["sigma"]), max_iter=kw["max_iter"], lambda_=np.float32(kw["lambda_"]))

# ---------------------------------------------------------------- unfolded PGD, 3-D multi-coil MRI
vol, coils, B = (4, 16, 16), 3, 2
x = torch.rand(B, 2, *vol, generator=g(37))
maps = torch.randn(1, coils, *vol, dtype=torch.complex64, generator=g(38)) / coils ** 0.5
mask = (torch.rand(*vol, generator=g(39)) > 0.5).float()
phys = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, *vol), three_d=True)
y = phys.A(x)
wden = torch.randn(2, 2, 3, 3, 3, generator=g(40)) * 0.1
bden = torch.randn(2, generator=g(41)) * 0.01


class Den(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.c = torch.nn.Conv3d(2, 2, 3, padding=1)
        with torch.no_grad():
            self.c.weight.copy_(wden)
            self.c.bias.copy_(bden)

    def forward(self, u, s):
        return u - s * self.c(u)


model = dinv.unfolded.unfolded_builder("PGD", data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(Den()),
                                       params_algo={"stepsize": 0.8, "g_param": 0.05, "lambda": 1.0}, max_iter=3,
                                       trainable_params=["stepsize", "g_param"])
rec = model(y, phys)
loss = (rec - x).pow(2).mean()
loss.backward()
grads = {"grad_" + n.replace(".", "_"): p.grad for n, p in model.named_parameters()}
print(sorted(grads))
save("unfolded_pgd", x=x, maps=maps, mask=mask, y=y, wden=wden, bden=bden, rec=rec, loss=loss, **grads)

# ---------------------------------------------------------------- DRUNet with O(1)-gain ResBlock weights
sd = OD.init_state_dict(2, 2, seed=321, res_gain=1.0)
m = dinv.models.DRUNet(in_channels=2, out_channels=2, pretrained=None)
m.load_state_dict(sd)
m.eval()
x = torch.rand(1, 2, 32, 40, generator=g(42))
with torch.no_grad():
    save("drunet_gain1", x=x, sigma=np.float32(0.05), y=m(x, 0.05), w_body=sd["m_body.0.res.0.weight"][:4, :4])
print("done")
