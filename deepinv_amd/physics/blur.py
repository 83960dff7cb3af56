"""``Blur``, ``BlurFFT`` and ``Downsampling`` on HIP kernels — API mirror of
deepinv/physics/blur.py:15-389 (Downsampling), :443-561 (Blur), :564-737 (BlurFFT).
Buffer names (``filter``, ``mask``, ``angle``, ``Fh``, ``Fhc``, ``Fh2``) follow the reference.
"""
from __future__ import annotations

from warnings import warn

import torch
from torch import Tensor

from ..hip import conv as hc
from ..hip import fft as hfft
from . import functional as dF
from .forward import DecomposablePhysics, LinearPhysics


class Blur(LinearPhysics):
    r"""``y = w * x`` (true convolution) with padding valid|circular|reflect|replicate|constant (blur.py:443-561)."""

    def __init__(self, filter: Tensor = None, padding: str = "valid", use_fft: bool = False,
                 device=torch.device("cpu"), **kwargs):
        super().__init__(device=device, **kwargs)
        assert isinstance(filter, Tensor) or filter is None, \
            f"The filter must be a torch.Tensor or None, got filter of type {type(filter)}."
        self.padding = padding
        self.register_buffer("filter", filter)
        self.use_fft = use_fft
        self.to(device)

    def A(self, x: Tensor, filter: Tensor = None, **kwargs) -> Tensor:
        self.update_parameters(filter=filter, **kwargs)
        if x.dim() not in (4, 5):       # images or volumes (blur.py:535-546)
            raise ValueError(f"Expected Tensor dimension to be 4 or 5, is {x.dim()}")
        fn = {4: dF.conv2d_fft if self.use_fft else dF.conv2d, 5: dF.conv3d_fft if self.use_fft else dF.conv3d}[x.dim()]
        return fn(x, filter=self.filter, padding=self.padding)

    def A_adjoint(self, y: Tensor, filter: Tensor = None, **kwargs) -> Tensor:
        self.update_parameters(filter=filter, **kwargs)
        if y.dim() not in (4, 5):
            raise ValueError(f"Expected Tensor dimension to be 4 or 5, is {y.dim()}")
        fn = {4: dF.conv_transpose2d_fft if self.use_fft else dF.conv_transpose2d,
              5: dF.conv_transpose3d_fft if self.use_fft else dF.conv_transpose3d}[y.dim()]
        return fn(y, filter=self.filter, padding=self.padding)


class BlurFFT(DecomposablePhysics):
    r"""Circular blur diagonalised by the real FFT: ``A = U diag(mask) V^T`` with ``V^T = rfft2`` and
    ``U = irfft2(angle * .)`` (blur.py:564-737)."""

    def __init__(self, img_size, filter: Tensor = None, device="cpu", **kwargs):
        super().__init__(device=device, **kwargs)
        self.img_size = img_size
        assert isinstance(filter, Tensor) or filter is None, \
            f"The filter must be a torch.Tensor or None, got filter of type {type(filter)}."
        p = self.get_filter_parameters(img_size=img_size, filter=filter, device=device)
        self.register_buffer("filter", p["filter"])
        self.register_buffer("angle", p["angle"])
        self.register_buffer("mask", p["mask"])
        self.to(device)

    # ---- the operators of DecomposablePhysics (forward.py:1080-1117, 1212-1252) as ONE library call each: rfft2 rows, the column
    # transform pair with the symbol applied in LDS between them, irfft2 rows (hip/conv.py: blurfft_apply).  Tensors that record a
    # gradient, non-fp32 inputs and symbol buffers of another layout take the composed U / V / mask expressions of the base class.
    def _fused(self, x, flags, add=0.0):
        if isinstance(self.mask, Tensor) and isinstance(self.angle, Tensor) and tuple(x.shape[-2:]) == tuple(self.img_size[-2:]) \
                and hc.blurfft_supported(x, self.mask, self.angle):
            return hc.blurfft_apply(x, self.mask, self.angle, flags, add, norm="ortho")
        return None

    def A(self, x, filter=None, **kwargs):
        self.update_parameters(filter=filter)
        out = self._fused(x, hc.SYM_MASK | hc.SYM_POST_ANGLE)
        return out if out is not None else DecomposablePhysics.A(self, x)

    def A_adjoint(self, x, filter=None, **kwargs):
        self.update_parameters(filter=filter)
        out = self._fused(x, hc.SYM_PRE_CONJ_ANGLE | hc.SYM_MASK)
        return out if out is not None else DecomposablePhysics.A_adjoint(self, x)

    def A_adjoint_A(self, x, filter=None, **kwargs):
        self.update_parameters(filter=filter)
        out = self._fused(x, hc.SYM_MASK2)
        return out if out is not None else DecomposablePhysics.A_adjoint_A(self, x)

    def A_A_adjoint(self, y, filter=None, **kwargs):
        self.update_parameters(filter=filter)
        out = self._fused(y, hc.SYM_PRE_CONJ_ANGLE | hc.SYM_MASK2 | hc.SYM_POST_ANGLE)
        return out if out is not None else DecomposablePhysics.A_A_adjoint(self, y)

    def prox_l2(self, z, y, gamma, **kwargs):
        from ..hip import elementwise as EW

        if not isinstance(gamma, Tensor) and EW.eligible(z, y):
            aty = self._fused(y, hc.SYM_PRE_CONJ_ANGLE | hc.SYM_MASK)
            if aty is not None:     # b = A^T y + z / gamma, then V(V^T b / (m m + 1 / gamma))
                return self._fused(EW.lincomb(1.0, aty, 1.0 / float(gamma), z), hc.SYM_PROX, 1.0 / float(gamma))
        return DecomposablePhysics.prox_l2(self, z, y, gamma, **kwargs)

    def A_dagger(self, y, filter=None, **kwargs):
        self.update_parameters(filter=filter)
        out = self._fused(y, hc.SYM_PRE_CONJ_ANGLE | hc.SYM_DAGGER)
        return out if out is not None else DecomposablePhysics.A_dagger(self, y)

    def _symbol_ok(self, spec):
        from ..hip import elementwise as EW

        return (isinstance(self.angle, Tensor) and spec.dtype == torch.complex64 and EW.eligible(torch.view_as_real(spec))
                and not (torch.is_grad_enabled() and self.angle.requires_grad)
                and hc._symbol_planes(spec.shape[:-2], None, self.angle, *spec.shape[-2:]) is not None)

    def V_adjoint(self, x):
        return torch.view_as_real(hc.rfft2(x, norm="ortho"))

    def U(self, x):
        spec = torch.view_as_complex(x.contiguous())
        if self._symbol_ok(spec):
            spec = hc.spectrum_symbol(spec, None, self.angle, hc.SYM_POST_ANGLE)
        else:
            spec = spec * self.angle
        return hc.irfft2(spec, self.img_size[-2:], norm="ortho")

    def U_adjoint(self, x):
        spec = hc.rfft2(x, norm="ortho")
        if self._symbol_ok(spec):
            return torch.view_as_real(hc.spectrum_symbol(spec, None, self.angle, hc.SYM_PRE_CONJ_ANGLE))
        return torch.view_as_real(spec * torch.conj(self.angle))

    def V(self, x):
        return hc.irfft2(torch.view_as_complex(x.contiguous()), self.img_size[-2:], norm="ortho")

    @staticmethod
    def get_filter_parameters(img_size, filter, device="cpu"):
        """singular values / phases of the circular blur (blur.py:659-690)"""
        if filter is None or not isinstance(filter, Tensor):
            return {"filter": None, "angle": None, "mask": None}
        filter = filter.to(device)
        if img_size[0] > filter.shape[1]:
            filter = filter.repeat(1, img_size[0], 1, 1)
        if filter.is_cuda:
            spec = dF.filter_fft(filter, img_size, dims=(-2, -1), real_fft=True)
        else:
            # operator still being constructed on the host (module not moved yet): the filter spectrum is a
            # one-off parameter computation, done where the filter lives
            spec = _host_filter_fft(filter, img_size)
        angle = torch.angle(spec)
        mask = torch.abs(spec).unsqueeze(-1)
        return {"filter": filter, "angle": torch.exp(1j * angle), "mask": torch.cat([mask, mask], dim=-1)}

    def update_parameters(self, filter: Tensor = None, **kwargs):
        device = self.filter.device if isinstance(self.filter, Tensor) else (
            filter.device if isinstance(filter, Tensor) else self._device_holder.device)
        if self.filter is None and isinstance(filter, Tensor):
            self.to(device)
        p = self.get_filter_parameters(img_size=self.img_size, filter=filter, device=device)
        if kwargs.get("mask") is None and "mask" in kwargs:
            kwargs.pop("mask")
        super().update_parameters(**p)


def _host_filter_fft(filter, img_size):
    import torch.nn.functional as F

    h, w = filter.shape[-2:]
    f = F.pad(filter, (0, img_size[-1] - w, 0, img_size[-2] - h))
    f = torch.roll(f, shifts=(-int(h / 2), -int(w / 2)), dims=(-2, -1))
    return torch.fft.rfft2(f)


class Downsampling(LinearPhysics):
    r"""``y = (h * x)[::f, ::f]`` with closed-form FFT prox for circular padding (blur.py:15-389)."""

    def __init__(self, img_size=None, filter="warn", factor=2, device="cpu", padding="circular", **kwargs):
        if isinstance(filter, str) and filter == "warn":
            warn("Leaving the filter as default is deprecated and will be removed in future versions. Please specify "
                 "filter=None for bare decimation, or one of the available filters (gaussian, bilinear, bicubic, "
                 "sinc) for filtered downsampling.", stacklevel=2)
            filter = None
        super().__init__(device=device, **kwargs)
        self.imsize = tuple(img_size) if isinstance(img_size, list) else img_size
        self.imsize_dynamic = (3, 128, 128)
        self.padding = padding
        imsize = self.imsize if self.imsize is not None else self.imsize_dynamic
        p = self.get_filter_parameters(img_size=imsize, filter=filter, factor=factor, device=device)
        self.factor = p["factor"]
        for k in ("filter", "Fh", "Fhc", "Fh2"):
            self.register_buffer(k, p[k])
        self.to(device)

    @staticmethod
    def check_factor(factor) -> int:
        if isinstance(factor, (int, float)):
            return int(factor)
        if isinstance(factor, Tensor):
            if factor.ndim > 1:
                raise ValueError("Factor tensor must be 1D.")
            u = torch.unique(factor)
            if len(u) > 1:
                raise ValueError(f"Downsampling only supports one unique factor per batch, but got factors {u.tolist()}.")
            return int(u.item())
        raise ValueError(f"Factor must be an integer, got {factor} of type {type(factor)}.")

    @staticmethod
    def get_filter_parameters(img_size=None, filter=None, factor=None, device="cpu"):
        """blur.py:130-192"""
        out = {"factor": Downsampling.check_factor(factor) if factor is not None else None}
        if filter is None:
            out.update(filter=None, Fh=None, Fhc=None, Fh2=None)
            return out
        assert factor is not None, "factor must be provided when filter is not None."
        assert img_size is not None, "img_size must be provided when filter is not None."
        f = out["factor"]
        if isinstance(filter, list):
            if len(set(filter)) == 1 and isinstance(filter[0], str):
                filter = filter[0]
            else:
                raise ValueError(f"Downsampling supports filter string lists if they are identical, but got unique "
                                 f"filters {set(filter)}.")
        if isinstance(filter, Tensor):
            filter = filter.to(device)
        elif filter == "gaussian":
            filter = dF.gaussian_blur(sigma=(f, f), device=device)
        elif filter == "bilinear":
            filter = dF.bilinear_filter(f, device=device)
        elif filter == "bicubic":
            filter = dF.bicubic_filter(f, device=device)
        elif filter == "sinc":
            filter = dF.sinc_filter(f, length=4 * f, device=device)
        else:
            raise ValueError(f"unknown filter {filter}")
        if filter.is_cuda:
            Fh = dF.filter_fft(filter, img_size, real_fft=False)
        else:
            import torch.nn.functional as F
            h, w = filter.shape[-2:]
            ff = F.pad(filter, (0, img_size[-1] - w, 0, img_size[-2] - h))
            Fh = torch.fft.fft2(torch.roll(ff, shifts=(-int(h / 2), -int(w / 2)), dims=(-2, -1)))
        out.update(filter=filter, Fh=Fh, Fhc=torch.conj(Fh), Fh2=torch.conj(Fh) * Fh)
        return out

    def update_parameters(self, filter=None, factor=None, device=None, **kwargs):
        """blur.py:194-253"""
        if factor is not None and filter is None and self.filter is not None:
            warn("Updating factor but not filter. Filter will not be valid for new factor. Pass filter string or new "
                 "filter to resolve this.")
        if filter is None and factor is None:
            LinearPhysics.update_parameters(self, **kwargs)
            return
        imsize = self.imsize if self.imsize is not None else self.imsize_dynamic
        if isinstance(self.filter, Tensor):
            device = self.filter.device
        elif isinstance(filter, Tensor):
            device = filter.device
            self.to(device)
        elif device is None:
            device = self._device_holder.device
        p = self.get_filter_parameters(img_size=imsize, filter=filter,
                                       factor=factor if factor is not None else self.factor, device=device)
        if p["factor"] is not None:
            self.factor = p.pop("factor")
        else:
            p.pop("factor")
        if p["filter"] is None:
            p = {}
        p.update(**kwargs)
        for k, v in p.items():
            if v is not None and k in ("filter", "Fh", "Fhc", "Fh2") and getattr(self, k) is None:
                setattr(self, k, v)  # buffer was registered as None: install the tensor
        LinearPhysics.update_parameters(self, **p)

    def A(self, x: Tensor, filter=None, factor=None, **kwargs) -> Tensor:
        self.imsize_dynamic = x.shape[-3:]
        self.update_parameters(filter=filter, factor=factor, device=x.device, **kwargs)
        if self.filter is not None:
            return hc.conv2d_strided(x, self.filter, self.padding, self.factor)   # conv + decimation fused
        from ..hip import require_hip
        require_hip(x)
        return x[:, :, ::self.factor, ::self.factor]

    def A_adjoint(self, y: Tensor, filter=None, factor=None, **kwargs) -> Tensor:
        self.imsize_dynamic = (y.shape[-3], y.shape[-2] * self.factor, y.shape[-1] * self.factor)
        self.update_parameters(filter=filter, factor=factor, device=y.device, **kwargs)
        imsize = self.imsize if self.imsize is not None else self.imsize_dynamic
        if self.filter is not None:
            H, W = imsize[-2:]
            return hc.conv2d_strided_transpose(y, self.filter, self.padding, self.factor, H, W)
        from ..hip import require_hip
        require_hip(y)
        x = torch.zeros((y.shape[0],) + tuple(imsize[:3]), device=y.device, dtype=y.dtype)
        x[:, :, ::self.factor, ::self.factor] = y
        return x

    def prox_l2(self, z, y, gamma, use_fft=True, **kwargs):
        r"""Closed-form prox for circular padding (blur.py:331-363, Zhao et al. 2016), evaluated in its RESIDUAL form

            x = z + A^T (A A^T + I / gamma)^{-1} (y - A z),

        which is the same minimiser as the reference's ``(z_hat - F^{-1}[conj(K) tile(mean(K F z_hat) / (mean |K|^2 +
        1/gamma))]) * gamma`` (Woodbury), but has no ``gamma``-fold cancellation: ``A A^T`` is diagonalised by the
        low-resolution DFT with symbol ``mean_blocks(|K|^2)``, so only ONE small (H/f x W/f) transform pair is needed
        instead of two full-size ones.  The reference's form subtracts two nearly equal images and multiplies by gamma:
        in fp32 it is 5e-3 off the exact minimiser at DiffPIR's gamma = 7e5 (this form: 3e-7; both measured against an
        fp64 evaluation in tests/test_oracle_golden.py::test_downsampling_prox_forms)."""
        if not (use_fft and self.padding == "circular" and self.filter is not None):
            return LinearPhysics.prox_l2(self, z, y, gamma, **kwargs)
        sf = self.factor
        key = (self.Fh2.data_ptr(), self.Fh2._version, sf)
        if getattr(self, "_alias_key", None) != key:
            a = self.Fh2.real if self.Fh2.is_complex() else self.Fh2
            b = torch.stack(torch.chunk(a, sf, dim=2), dim=4)
            self._alias_mean = torch.mean(torch.cat(torch.chunk(b, sf, dim=3), dim=4), dim=-1).contiguous()
            self._alias_key = key
        from ..hip import conv as hc
        from ..hip import elementwise as EW

        Az = self.A(z)
        if EW.eligible(z, y, Az) and not isinstance(gamma, Tensor) and y.shape[-1] % 2 == 0:
            # all on the HIP kernels: the residual, a REAL-input transform pair of the low-resolution grid (half the spectrum:
            # the aliased symbol of a real filter is even), the division by the symbol, the update
            hkey = key + (tuple(y.shape[:2]),)
            if getattr(self, "_alias_half_key", None) != hkey:
                half = self._alias_mean[..., : y.shape[-1] // 2 + 1]
                # the kernel indexes the symbol as the TRAILING part of the spectrum [B, C, h, w/2+1]: a per-sample symbol
                # [B, 1, h, .] (filters from a physics generator) is expanded over the channels first (low-resolution grid: small)
                if half.dim() == 4 and half.shape[0] > 1 and half.shape[1] != y.shape[1]:
                    half = half.expand(half.shape[0], y.shape[1], *half.shape[2:])
                self._alias_half = half.contiguous()
                self._alias_half_key = hkey
            r = EW.lincomb(1.0, y, -1.0, Az)
            S = EW.cdiv_real(hc.rfft2(r, norm="backward"), self._alias_half, 1.0 / float(gamma))
            s = hc.irfft2(S, y.shape[-2:], norm="backward")
            return EW.lincomb(1.0, z, 1.0, self.A_adjoint(s))
        r = y - Az
        S = hfft.fftn(r.to(torch.complex64), dim=(-2, -1), norm="backward") / (self._alias_mean + 1 / gamma)
        s = torch.real(hfft.ifftn(S, dim=(-2, -1), norm="backward"))
        return z + self.A_adjoint(s.contiguous())
