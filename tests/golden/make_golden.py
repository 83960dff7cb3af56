#!/usr/bin/env python
"""Generate golden input/output vectors from the REAL reference (deepinv v0.4.1 at /root/reference,
imported through oracle/ref_shim.py).  Run in the build container only; the .npz files are committed
because the reference cannot travel to the GPU box.

    python tests/golden/make_golden.py
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


# ---------------------------------------------------------------- MRI (test_physics.py:180-232 sizes)
x = torch.randn(2, 2, 17, 11, generator=g(0))
mask = (torch.rand(17, 11, generator=g(1)) > 0.5).float()
p = dinv.physics.MRI(mask=mask, img_size=(2, 17, 11))
y = p.A(x)
z = torch.randn(2, 2, 17, 11, generator=g(2))
save("mri_2d", x=x, mask=mask, y=y, xadj=p.A_adjoint(y), prox=p.prox_l2(z, y, 0.7), z=z, dagger=p.A_dagger(y))

x = torch.randn(1, 2, 5, 17, 11, generator=g(3))
mask = (torch.rand(5, 17, 11, generator=g(4)) > 0.5).float()
p = dinv.physics.MRI(mask=mask, img_size=(2, 5, 17, 11), three_d=True)
y = p.A(x)
save("mri_3d", x=x, mask=mask, y=y, xadj=p.A_adjoint(y))

x = torch.randn(2, 2, 17, 11, generator=g(5))
maps = torch.randn(1, 7, 17, 11, dtype=torch.complex64, generator=g(6)) / 7 ** 0.5
mask = (torch.rand(17, 11, generator=g(7)) > 0.4).float()
p = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, 17, 11))
y = p.A(x)
save("multicoil_2d", x=x, mask=mask, maps=maps, y=y, xadj=p.A_adjoint(y), rss=p.A_adjoint(y, rss=True))

x = torch.randn(1, 2, 4, 16, 12, generator=g(8))
maps = torch.randn(1, 5, 4, 16, 12, dtype=torch.complex64, generator=g(9)) / 5 ** 0.5
mask = (torch.rand(4, 16, 12, generator=g(10)) > 0.4).float()
p = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, 4, 16, 12), three_d=True)
y = p.A(x)
save("multicoil_3d", x=x, mask=mask, maps=maps, y=y, xadj=p.A_adjoint(y))

# literal docstring vector (deepinv/physics/mri.py:52-76)
torch.manual_seed(0)
x = torch.randn(1, 2, 2, 2)
p = dinv.physics.MRI(mask=1 - torch.eye(2))
save("mri_doctest", x=x, mask=1 - torch.eye(2), y=p(x))

# fastMRI-equality FFT test input (test_physics.py:1575-1648): [4,2,16,8]
x = torch.randn(4, 2, 16, 8, generator=g(11))
from deepinv.utils.mixins import MRIMixin
save("mri_fft", x=x, k=MRIMixin().im_to_kspace(x), back=MRIMixin().kspace_to_im(x))

# ---------------------------------------------------------------- Tomography (test_physics.py:266-272)
x = torch.rand(2, 1, 16, 16, generator=g(12))
for circle in (False, True):
    p = dinv.physics.Tomography(angles=16, img_width=16, circle=circle, normalize=False, device="cpu")
    y = p.A(x)
    v = torch.randn(y.shape, generator=g(13))
    save(f"tomo_16_circle{int(circle)}", x=x, angles=p.angles, y=y, v=v, vadj=p.A_adjoint(v),
         ramp=p.iradon.filter(y), fbp=p.A_dagger(y, fbp=True))
# docstring (tomography.py:91-114)
torch.manual_seed(0)
x = torch.randn(1, 1, 4, 4)
angles = torch.linspace(0, 45, steps=3)
p = dinv.physics.Tomography(angles=angles, img_width=4, circle=True, normalize=True)
save("tomo_doctest", x=x, angles=angles, y=p(x), operator_norm=p.operator_norm)

# ---------------------------------------------------------------- Blur / BlurFFT / Downsampling
x = torch.randn(2, 3, 17, 19, generator=g(14))
k = dinv.physics.functional.gaussian_blur(sigma=(2.0, 1.0), angle=30.0)[..., :5, :4].contiguous()
k = k / k.sum()
arrs = {"x": x, "k": k}
for pad in ("valid", "circular", "reflect", "replicate", "constant"):
    p = dinv.physics.Blur(filter=k, padding=pad)
    y = p.A(x)
    v = torch.randn(y.shape, generator=g(15))
    arrs.update({f"y_{pad}": y, f"v_{pad}": v, f"vadj_{pad}": p.A_adjoint(v)})
save("blur_paddings", **arrs)

x = torch.rand(1, 3, 17, 19, generator=g(16))
k = dinv.physics.functional.bicubic_filter(2)
p = dinv.physics.BlurFFT(img_size=(3, 17, 19), filter=k)
y = p.A(x)
z = torch.rand(1, 3, 17, 19, generator=g(17))
save("blurfft", x=x, k=k, mask=p.mask, angle=p.angle, y=y, xadj=p.A_adjoint(y), prox=p.prox_l2(z, y, 1.3), z=z)

x = torch.rand(2, 3, 32, 24, generator=g(18))
p = dinv.physics.Downsampling(img_size=(3, 32, 24), filter="bicubic", factor=4, padding="circular")
y = p.A(x)
z = torch.rand(2, 3, 32, 24, generator=g(19))
save("downsampling", x=x, k=p.filter, y=y, yadj=p.A_adjoint(y), prox=p.prox_l2(z, y, 0.8), z=z)

# ---------------------------------------------------------------- DRUNet (weights regenerated from the seed)
from oracle import drunet_cpu as OD  # noqa: E402

sd = OD.init_state_dict(2, 2, seed=123)
m = dinv.models.DRUNet(in_channels=2, out_channels=2, pretrained=None)
m.load_state_dict(sd)
m.eval()
x = torch.rand(1, 2, 32, 40, generator=g(20))
with torch.no_grad():
    save("drunet_2ch", x=x, sigma=np.float32(0.05), y=m(x, 0.05), w_head=sd["m_head.weight"])

# ---------------------------------------------------------------- optim: PGD doctest + cfg1 + PnP-PGD/HQS on MRI
x = torch.rand(1, 3, 256, 256, generator=g(21))
h = dinv.physics.functional.gaussian_blur(psf_size=(9, 9), sigma=(2.0, 2.0))
p = dinv.physics.BlurFFT(img_size=(3, 256, 256), filter=h)
y = p.A(x)
ident = lambda u, s: u


class _Id(torch.nn.Module):
    def forward(self, u, s):
        return u


model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(_Id()), stepsize=1.0, g_param=0.05,
                       max_iter=20)
xr = model(y, p)
save("cfg1_blurfft_pgd", x_crop=x[..., :64, :64], y_crop=y[..., :64, :64], rec_crop=xr[..., :64, :64],
     rec_sum=xr.double().sum(), rec_norm=xr.double().norm(), seed=21)

x = torch.rand(1, 2, 32, 32, generator=g(22))
maps = torch.randn(1, 4, 32, 32, dtype=torch.complex64, generator=g(23)) / 2
mask = (torch.rand(32, 32, generator=g(24)) > 0.6).float()
p = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, 32, 32))
y = p.A(x)
model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(m), stepsize=1.0, g_param=0.05, max_iter=3)
with torch.no_grad():
    r_pgd = model(y, p)
model = dinv.optim.HQS(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(m), stepsize=[2.0, 1.0, 0.5],
                       g_param=[0.1, 0.05, 0.02], max_iter=3)
with torch.no_grad():
    r_hqs = model(y, p)
save("pnp_mri", x=x, maps=maps, mask=mask, y=y, rec_pgd=r_pgd, rec_hqs=r_hqs)
print("done")
