"""Torch-CPU restatement of the reference operators (the oracle "port").

TEST INFRASTRUCTURE.  Every function follows the reference's ATen call sequence and cites it;
inputs/outputs are CPU tensors with the reference's layouts.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# MRI  (deepinv/utils/mixins.py:149-206, deepinv/physics/mri.py:99-163, 254-324)
# --------------------------------------------------------------------------------------


def to_complex(x):
    """[B,2,...] -> complex [B,...]  (mixins.py:149-151)"""
    return torch.view_as_complex(x.moveaxis(1, -1).contiguous())


def from_complex(x):
    """complex [B,...] -> [B,2,...]  (mixins.py:154-156)"""
    return torch.view_as_real(x).moveaxis(-1, 1)


def cfft(x, dim):
    """centred orthonormal fft (mixins.py:171-180)"""
    x = torch.fft.ifftshift(x, dim=dim)
    x = torch.fft.fftn(x, dim=dim, norm="ortho")
    return torch.fft.fftshift(x, dim=dim)


def cifft(x, dim):
    """centred orthonormal ifft (mixins.py:159-168)"""
    x = torch.fft.ifftshift(x, dim=dim)
    x = torch.fft.ifftn(x, dim=dim, norm="ortho")
    return torch.fft.fftshift(x, dim=dim)


def _dims(three_d):
    return (-3, -2, -1) if three_d else (-2, -1)


def im_to_kspace(x, three_d=False):
    """mixins.py:182-193"""
    return from_complex(cfft(to_complex(x), _dims(three_d)))


def kspace_to_im(y, three_d=False):
    """mixins.py:195-206"""
    return from_complex(cifft(to_complex(y), _dims(three_d)))


def check_mask(mask, three_d=False):
    """mixins.py:127-146"""
    while mask.ndim < (5 if three_d else 4):
        mask = mask.unsqueeze(0)
    if mask.shape[1] == 1:
        mask = torch.cat([mask, mask], dim=1)
    return mask


def mri_A(x, mask, three_d=False):
    """MRI.A = U(mask * V_adjoint(x)), U = id (forward.py:1080-1096, mri.py:99-100)"""
    return check_mask(mask, three_d) * im_to_kspace(x, three_d)


def mri_AT(y, mask, three_d=False):
    """MRI.A_adjoint = V(conj(mask) * U_adjoint(y)) (forward.py:1098-1117, mri.py:102-104)"""
    return kspace_to_im(torch.conj(check_mask(mask, three_d)) * y, three_d)


def mri_prox_l2(z, y, gamma, mask, three_d=False):
    """DecomposablePhysics.prox_l2 (forward.py:1212-1234)"""
    mask = check_mask(mask, three_d)
    b = mri_AT(y, mask, three_d) + 1 / gamma * z
    scaling = torch.conj(mask) * mask + 1 / gamma
    return kspace_to_im(im_to_kspace(b, three_d) / scaling, three_d)


def mri_dagger(y, mask, three_d=False):
    """DecomposablePhysics.A_dagger (forward.py:1236-1252)"""
    mask = check_mask(mask, three_d)
    inv = torch.where(mask > 1e-5, mask.reciprocal(), 0.0)
    return kspace_to_im(y * inv, three_d)


def multicoil_A(x, coil_maps, mask, three_d=False):
    """MultiCoilMRI.A (mri.py:254-272): coil_maps [1|B,N,...] complex, mask [1|B,2,...]"""
    mask = check_mask(mask, three_d)
    Sx = coil_maps * to_complex(x)[:, None]
    FSx = cfft(Sx, _dims(three_d))
    return mask[:, :, None] * from_complex(FSx)


def multicoil_AT(y, coil_maps, mask, three_d=False):
    """MultiCoilMRI.A_adjoint, rss=False (mri.py:284-324)"""
    mask = check_mask(mask, three_d)
    My = to_complex(mask[:, :, None] * y)
    FiMy = cifft(My, _dims(three_d))
    return from_complex(torch.sum(torch.conj(coil_maps) * FiMy, dim=1))


def multicoil_AT_rss(y, mask, three_d=False):
    """MultiCoilMRI.A_adjoint, rss=True (mri.py:316-318 + mixins.py:248-286)"""
    mask = check_mask(mask, three_d)
    x = from_complex(cifft(to_complex(mask[:, :, None] * y), _dims(three_d)))
    return x.pow(2).sum(dim=1, keepdim=True).sum(dim=2).sqrt()


def radial_mask(H, W, n_spokes):
    """Synthetic config-2 mask: `n_spokes` lines through the k-space centre at
    angles k*pi/n_spokes, rasterised with integer index arithmetic -> bit-reproducible {0,1}.
    (The reference has no radial generator: generator/mri.py only offers Cartesian masks.)"""
    mask = torch.zeros(H, W)
    cy, cx = H // 2, W // 2
    L = int(math.ceil(math.hypot(H, W)))
    t = torch.arange(-L, L + 1, dtype=torch.float64)
    for k in range(n_spokes):
        a = math.pi * k / n_spokes
        yy = torch.round(cy + t * math.sin(a)).long()
        xx = torch.round(cx + t * math.cos(a)).long()
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        mask[yy[ok], xx[ok]] = 1.0
    return mask
