#!/usr/bin/env python
"""Golden vectors for the fan-beam Tomography from the REAL reference (deepinv v0.4.1 at /root/reference through
oracle/ref_shim.py): fan_beam_grid / Radon(fan_beam=True) (functional/radon.py:16-52, 205-342), Tomography.A,
A_adjoint (autograd adjoint), fbp (tomography.py:229-350).

* `tomo_fan.npz`  16x16 image, 10 angles over 360 degrees, 20 detector pixels whose fan covers the image, circle False/True;
                  plus the reference's DEFAULT fan parameters (258 detector pixels, most rays miss the image) once.

    python tests/golden/make_golden_fan.py
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
FAN = {"pixel_spacing": 0.1, "source_radius": 6.0, "detector_radius": 6.0, "n_detector_pixels": 20, "detector_spacing": 0.34}


def g(seed):
    return torch.Generator().manual_seed(seed)


x = torch.rand(2, 1, 16, 16, generator=g(40))
angles = torch.linspace(0, 360, 11)[:-1]
arrs = {"x": x.numpy(), "angles": angles.numpy(), "fan": np.array([FAN[k] for k in sorted(FAN)]), "fan_keys": np.array(sorted(FAN))}
for circle in (False, True):
    p = dinv.physics.Tomography(angles=angles, img_width=16, circle=circle, normalize=False, fan_beam=True,
                                fan_parameters=dict(FAN), device="cpu")
    y = p.A(x)
    v = torch.randn(y.shape, generator=g(41))
    c = int(circle)
    arrs.update({f"y_c{c}": y.numpy(), f"v_c{c}": v.numpy(), f"vadj_c{c}": p.A_adjoint(v).detach().numpy(),
                 f"fbp_c{c}": p.A_dagger(y, fbp=True).detach().numpy()})
p = dinv.physics.Tomography(angles=6, img_width=16, normalize=False, fan_beam=True, device="cpu")   # default parameters
y = p.A(x)
v = torch.randn(y.shape, generator=g(42))
arrs.update({"y_default": y.numpy(), "v_default": v.numpy(), "vadj_default": p.A_adjoint(v).detach().numpy()})
np.savez_compressed(os.path.join(OUT, "tomo_fan.npz"), **arrs)
print({k: getattr(a, "shape", None) for k, a in arrs.items()})
