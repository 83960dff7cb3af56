"""``MRI`` and ``MultiCoilMRI`` on HIP kernels.

API mirror of deepinv/physics/mri.py:11-163 (``MRI``), :166-397 (``MultiCoilMRI``) and of
``MRIMixin`` (deepinv/utils/mixins.py:118-300).  Buffer names (``mask``, ``coil_maps``) are
kept so reference ``state_dict``s load unchanged.
"""
from __future__ import annotations

from warnings import warn

import numpy as np
import torch
from torch import Tensor

from ..hip import fft as hfft
from ..hip import mri as hmri
from .forward import DecomposablePhysics, LinearPhysics


class MRIMixin:
    """FFT + mask helpers shared by the MRI operators (mixins.py:118-300)."""

    @staticmethod
    def check_mask(mask: Tensor = None, three_d: bool = False, **kwargs) -> Tensor | None:
        """Bring the mask to ``(B,2,(D,)H,W)`` (mixins.py:127-146)."""
        if mask is None:
            return None
        if isinstance(mask, np.ndarray):
            mask = torch.from_numpy(mask)
        while mask.ndim < (5 if three_d else 4):
            mask = mask.unsqueeze(0)
        if mask.shape[1] == 1:
            mask = torch.cat([mask, mask], dim=1)
        return mask

    @staticmethod
    def to_torch_complex(x: Tensor) -> Tensor:
        """[B,2,...] real -> [B,...] complex (mixins.py:149-151)."""
        return torch.view_as_complex(x.moveaxis(1, -1).contiguous())

    @staticmethod
    def from_torch_complex(x: Tensor) -> Tensor:
        """[B,...] complex -> [B,2,...] real (mixins.py:154-156)."""
        return torch.view_as_real(x).moveaxis(-1, 1)

    @staticmethod
    def fft(x: Tensor, dim=(-2, -1), norm="ortho") -> Tensor:
        """Centred orthonormal FFT of a complex tensor (mixins.py:171-180)."""
        return hfft.fftn(x, dim=dim, norm=norm, centered=True)

    @staticmethod
    def ifft(x: Tensor, dim=(-2, -1), norm="ortho") -> Tensor:
        """Centred orthonormal inverse FFT of a complex tensor (mixins.py:159-168)."""
        return hfft.ifftn(x, dim=dim, norm=norm, centered=True)

    def im_to_kspace(self, x: Tensor, three_d: bool = False) -> Tensor:
        """(B,2,...) image -> (B,2,...) k-space, one fused launch per axis (mixins.py:182-193)."""
        self._check_ndim(x, three_d)
        return hmri.mri_forward(x, None, None, coil_dim=False)

    def kspace_to_im(self, y: Tensor, three_d: bool = False) -> Tensor:
        """(B,2,...) k-space -> (B,2,...) image (mixins.py:195-206)."""
        self._check_ndim(y, three_d)
        return hmri.mri_adjoint(y, None, None, coil_dim=False)

    @staticmethod
    def _check_ndim(x, three_d):
        want = 5 if three_d else 4
        if x.ndim != want:
            raise ValueError(f"expected a {want}-D (B,2,{'D,' if three_d else ''}H,W) tensor, got shape {tuple(x.shape)}")

    def crop(self, x: Tensor, crop: bool = True, shape: tuple = None, rescale: bool = False) -> Tensor:
        """Centre crop of the last two dims to ``img_size`` (mixins.py:208-246; same rounding as
        torchvision ``CenterCrop``; odd heights adjusted by one pixel to match FastMRI)."""
        if rescale and crop:
            raise ValueError("Only one of rescale or crop can be used.")
        if not crop and not rescale:
            return x
        ch, cw = (shape[-2:] if shape is not None else self.img_size[-2:])
        odd_h = ch % 2 == 1
        if odd_h:
            ch += 1
        if rescale:
            # the reference calls torchvision.transforms.Resize on the (..., H, W) tensor (mixins.py:236-240): for tensors that
            # is bilinear interpolation with half-pixel centres and, since torchvision 0.17, antialiasing when shrinking -
            # the same ATen kernel, called directly (display-side helper, not on the hot path; torchvision is not installed
            # in the build container, so this branch has no reference fixture: tests pin it to the defining sums instead)
            flat = x.reshape(-1, 1, *x.shape[-2:])
            out = torch.nn.functional.interpolate(flat, size=(ch, cw), mode="bilinear", align_corners=False, antialias=True)
            out = out.reshape(*x.shape[:-2], ch, cw)
            return out[..., :-1, :] if odd_h else out
        H, W = x.shape[-2:]
        if ch > H or cw > W:  # CenterCrop zero-pads when the crop is larger
            ph, pw = max(ch - H, 0), max(cw - W, 0)
            x = torch.nn.functional.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
            H, W = x.shape[-2:]
        top = int(round((H - ch) / 2.0))
        left = int(round((W - cw) / 2.0))
        out = x[..., top:top + ch, left:left + cw]
        if odd_h:
            out = out[..., :-1, :]
        return out

    @staticmethod
    def rss(x: Tensor, multicoil: bool = True, mag: bool = True, three_d: bool = False) -> Tensor:
        """Root-sum-of-squares over (complex-pair[, coil]) dims (mixins.py:248-300)."""
        if x.shape[1] != 2 or x.is_complex():
            raise ValueError("x should be of shape (B,2,...) and not of complex dtype.")
        if x.ndim != 4 + int(multicoil) + int(three_d):
            raise ValueError("x should be of shape (B,2,...) for singlecoil data or (B,2,N,...) for multicoil data.")
        ss = x.pow(2)
        if mag:
            ss = ss.sum(dim=1, keepdim=True)
        if multicoil:
            ss = ss.sum(dim=2)
        return ss.sqrt()


class MRI(MRIMixin, DecomposablePhysics):
    r"""Single-coil accelerated MRI :math:`y = M F x` (mri.py:11-163)."""

    def __init__(self, mask: Tensor | None = None, img_size: tuple | None = (320, 320), three_d: bool = False,
                 device="cpu", **kwargs):
        super().__init__(device=device, **kwargs)
        self.three_d = three_d
        self.img_size = img_size
        if mask is None:
            mask = torch.ones(*img_size, device=device)
        self.register_buffer("mask", self.check_mask(mask, three_d=three_d))
        self.img_size = self.mask.shape[1:]
        self.to(device)

    # singular vectors: U = I, V^T = centred orthonormal FFT
    def V_adjoint(self, x: Tensor) -> Tensor:
        return self.im_to_kspace(x, three_d=self.three_d)

    def V(self, x: Tensor) -> Tensor:
        return self.kspace_to_im(x, three_d=self.three_d)

    def A(self, x: Tensor, mask: Tensor = None, **kwargs) -> Tensor:
        """``mask * F x`` with the mask multiply fused into the last FFT pass."""
        self.update_parameters(mask=mask, **kwargs)
        self._check_ndim(x, self.three_d)
        return hmri.mri_forward(x, None, self.mask, coil_dim=False)

    def A_adjoint(self, y: Tensor, mask: Tensor = None, mag: bool = False, crop: bool = False, **kwargs) -> Tensor:
        self.update_parameters(mask=mask, **kwargs)
        self._check_ndim(y, self.three_d)
        x = hmri.mri_adjoint(y, None, self.mask, coil_dim=False)
        if mag:
            x = self.rss(x, multicoil=False, three_d=self.three_d)
        if crop:
            x = self.crop(x, crop=crop)
        return x

    def A_adjoint_A(self, x: Tensor, mask: Tensor = None, **kwargs) -> Tensor:
        """``F^H M^2 F x`` without materialising k-space (fused normal operator, csrc/mri.hip: dinv_mri_normal)."""
        self.update_parameters(mask=mask, **kwargs)
        self._check_ndim(x, self.three_d)
        return hmri.mri_normal(x, None, self.mask, coil_dim=False)

    def noise(self, x, **kwargs):
        return self.U(self.noise_model(x, **kwargs) * self.mask)

    def update_parameters(self, mask: Tensor = None, check_mask: bool = True, **kwargs):
        if mask is not None and check_mask:
            mask = self.check_mask(mask=mask, three_d=getattr(self, "three_d", False))
        super().update_parameters(mask=mask, **kwargs)


class MultiCoilMRI(MRIMixin, LinearPhysics):
    r"""Multi-coil MRI :math:`y_n = M F (S_n \odot x)` (mri.py:166-397)."""

    def __init__(self, mask: Tensor | None = None, coil_maps: Tensor | int | None = None,
                 img_size: tuple | None = (320, 320), three_d: bool = False, device=torch.device("cpu"), **kwargs):
        super().__init__(device=device, **kwargs)
        self.img_size = img_size
        self.three_d = three_d
        if mask is None:
            mask = torch.ones(*img_size, device=device)
        if coil_maps is None:
            coil_maps = torch.ones(tuple(img_size[-3:] if three_d else img_size[-2:]), dtype=torch.complex64,
                                   device=device)
        elif isinstance(coil_maps, int):
            coil_maps = self.simulate_birdcage_csm(n_coils=coil_maps).to(device)
        self.register_buffer("mask", self.check_mask(mask, three_d=three_d))
        self.register_buffer("coil_maps", self.check_coil_maps(coil_maps, three_d=three_d))
        self.to(device)

    def A(self, x: Tensor, mask: Tensor = None, coil_maps: Tensor = None, **kwargs) -> Tensor:
        self.update_parameters(mask=mask, coil_maps=coil_maps, **kwargs)
        self._check_ndim(x, self.three_d)
        return hmri.mri_forward(x, self.coil_maps, self.mask, coil_dim=True)

    def A_adjoint_A(self, x: Tensor, mask: Tensor = None, coil_maps: Tensor = None, **kwargs) -> Tensor:
        """``sum_n conj(S_n) F^H M^2 F (S_n x)``: what ``L2.grad`` and the CG prox evaluate every iteration, as one
        kernel chain that never writes the k-space tensor (csrc/mri.hip: dinv_mri_normal)."""
        self.update_parameters(mask=mask, coil_maps=coil_maps, **kwargs)
        self._check_ndim(x, self.three_d)
        return hmri.mri_normal(x, self.coil_maps, self.mask, coil_dim=True)

    def noise(self, x, **kwargs) -> Tensor:
        return self.mask[:, :, None] * self.noise_model(x, **kwargs)

    def A_adjoint(self, y: Tensor, mask: Tensor = None, coil_maps: Tensor = None, rss: bool = False,
                  crop: bool = False, **kwargs) -> Tensor:
        if y.shape[1] != 2:
            raise ValueError("y must be of shape (B,2,N,...,H,W)")
        self.update_parameters(mask=mask, coil_maps=coil_maps, **kwargs)
        if rss:
            # per-coil images F^H(M y_n) (no coil combination), then root-sum-of-squares
            B, _, N = y.shape[:3]
            vol = y.shape[3:]
            ones = torch.ones((1, 1, *vol), dtype=torch.complex64, device=y.device)
            m = self.mask
            yy = y.transpose(1, 2).reshape(B * N, 2, 1, *vol)
            mm = m if m.shape[0] == 1 else m.repeat_interleave(N, dim=0)
            imgs = hmri.mri_adjoint(yy, ones, mm, coil_dim=True).reshape(B, N, 2, *vol).transpose(1, 2)
            x = self.rss(imgs, multicoil=True, three_d=self.three_d)
        else:
            x = hmri.mri_adjoint(y, self.coil_maps, self.mask, coil_dim=True)
        return self.crop(x, crop=crop)

    def A_dagger(self, y: Tensor, mask: Tensor = None, coil_maps: Tensor = None, **kwargs) -> Tensor:
        self.update_parameters(mask=mask, coil_maps=coil_maps)
        return super().A_dagger(y, **kwargs)

    def update_parameters(self, mask: Tensor = None, coil_maps: Tensor = None, check_mask: bool = True,
                          check_coil_maps: bool = True, **kwargs):
        if mask is not None and check_mask:
            mask = self.check_mask(mask=mask, three_d=self.three_d)
        if coil_maps is not None and check_coil_maps:
            coil_maps = self.check_coil_maps(coil_maps, three_d=self.three_d)
        super().update_parameters(mask=mask, coil_maps=coil_maps, **kwargs)
        self.img_size = self.mask.shape[1:]
        if self.coil_maps is not None and self.coil_maps.shape[2:] != self.img_size[1:]:
            warn(f"After updating parameters, img_size {self.img_size} in MultiCoilMRI is incompatible with "
                 f"coil_maps shape {self.coil_maps.shape} in the spatial dims.")

    @staticmethod
    def check_coil_maps(coil_maps: Tensor, three_d: bool) -> Tensor:
        while coil_maps.ndim < (5 if three_d else 4):
            coil_maps = coil_maps.unsqueeze(0)
        if not coil_maps.is_complex():
            raise ValueError("coil_maps should be of torch complex dtype.")
        return coil_maps

    def simulate_birdcage_csm(self, n_coils: int) -> Tensor:
        try:
            from sigpy.mri import birdcage_maps
        except ImportError:  # pragma: no cover
            raise ImportError("sigpy is required to simulate coil maps. Install it using pip install sigpy")
        maps = birdcage_maps((n_coils,) + tuple(self.img_size[-3:] if self.three_d else self.img_size[-2:]))
        return torch.tensor(maps).type(torch.complex64)
