"""``Tomography`` (parallel beam) on HIP kernels — API mirror of deepinv/physics/tomography.py:26-350.

Differences from the reference, all internal: no precomputed sampling grids (3 GB at 512/720 angles),
the exact adjoint is a deterministic gather kernel instead of an autograd replay
(``adjoint_via_backprop=True`` keeps its meaning: *exact* adjoint), the ramp filter is the reference's zero-padded
rFFT product done in one LDS-resident kernel, and the operator norm of ``normalize=True`` stays on the device.
``fan_beam=True`` (fan_beam_grid, functional/radon.py:16-52) runs on gather kernels with the same exact-adjoint
construction; as in the reference it always uses the exact adjoint and ``RampFilter`` + adjoint for FBP.
"""
from __future__ import annotations

from warnings import warn

import torch
from numpy import ndarray

from ..hip import radon as hr
from .forward import LinearPhysics


class RampFilter(torch.nn.Module):
    """radon.py:74-173 (dim=-2 only)"""

    def __init__(self, dtype=torch.float32):
        super().__init__()
        self.dtype = dtype

    def forward(self, x, dim=-2):
        if dim not in (-2, 2):
            raise NotImplementedError("the HIP ramp filter acts on the detector axis (dim=-2)")
        return hr.ramp_filter(x)


class Tomography(LinearPhysics):
    def __init__(self, angles, img_width, circle=False, parallel_computation=True, adjoint_via_backprop=True,
                 fbp_interpolate_boundary=False, normalize=None, fan_beam=False, fan_parameters=None,
                 device=torch.device("cpu"), dtype=torch.float, **kwargs):
        super().__init__(device=device, **kwargs)
        if isinstance(angles, int):
            angles = torch.linspace(0, 180, steps=angles + 1, device=device)[:-1].to(device)
        elif isinstance(angles, (list, tuple, ndarray)):
            angles = torch.tensor(angles).to(device)
        elif not isinstance(angles, torch.Tensor):
            raise ValueError(f"angles must be int, float, iterable or Tensor, but got {type(angles)}")
        self.register_buffer("angles", angles)
        self.fan_beam = bool(fan_beam)
        self.fan_parameters = dict(fan_parameters) if fan_parameters is not None else None
        self.adjoint_via_backprop = adjoint_via_backprop
        if circle and fbp_interpolate_boundary:
            warn("The argument fbp_interpolate_boundary=True is not applicable if circle=True. The value "
                 "fbp_interpolate_boundary will be changed to False...")
            fbp_interpolate_boundary = False
        self.fbp_interpolate_boundary = fbp_interpolate_boundary
        self.img_width = img_width
        self.circle = circle
        self.dtype = dtype
        self.parallel_computation = parallel_computation  # kept for API compatibility; no effect here
        self.filter = RampFilter(dtype=dtype)
        self._geo = self._geo_key = None
        if normalize is None:
            warn("The default value of `normalize` is not specified and will be automatically set to `True`. "
                 "Set `normalize` explicitly to `True` or `False` to avoid this warning.")
            normalize = True
        self.normalize = False
        if normalize:
            # (the power method runs on the HIP kernels: on a CPU device their wrappers raise, there is no CPU path)
            x0 = torch.randn((1, img_width, img_width), generator=torch.Generator(device).manual_seed(0),
                             device=device)[None]
            operator_norm = self.compute_norm(x0, squared=False, verbose=False)
            self.register_buffer("operator_norm", operator_norm)
            self.normalize = True
        self.to(device)

    # ---- geometry tables live on the device of the data; rebuilt if the module moved or the angles changed
    def _geometry(self, device):
        a = self.angles
        ver = None if a.is_inference() else a._version
        key = (torch.device(device), a.data_ptr(), ver, a.numel())
        if self._geo is None or self._geo_key != key:
            if self.fan_beam:
                self._geo = hr.FanGeometry(a, self.img_width, self.circle, device, self.fan_parameters)
            else:
                self._geo = hr.RadonGeometry(a, self.img_width, self.circle, device)
            self._geo_key = key
        return self._geo

    def update_parameters(self, **kwargs):
        """new angles (same count or not) rebuild the geometry tables (forward.py:249-276 semantics for buffers)"""
        super().update_parameters(**kwargs)
        self._geo = None

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._geo = None

    def _norm(self):
        """the operator norm as a DEVICE scalar handed to the kernels (no host read-back per call)"""
        return self.operator_norm.reshape(1) if self.normalize else None

    def _scale_host(self):
        """1/||A|| as a host float for the ApplyRadon branch only; read back once per value of the buffer"""
        if not self.normalize:
            return 1.0
        n = self.operator_norm
        key = (n.data_ptr(), None if n.is_inference() else n._version)
        if getattr(self, "_scale_key", None) != key:
            self._scale_key, self._scale_val = key, 1.0 / float(n)
        return self._scale_val

    def A(self, x, **kwargs):
        if not x.shape[-2:] == (self.img_width, self.img_width):
            raise ValueError(f"Input image size {x.shape[-2:]} does not match the operator image size "
                             f"{(self.img_width, self.img_width)}.")
        if self.fan_beam:
            return hr.fan_forward(x, self._geometry(x.device), self._norm())
        if self.adjoint_via_backprop:
            return hr.radon_forward(x, self._geometry(x.device), self._norm())
        return hr.apply_radon(x, self._geometry(x.device), self._scale_host(), False)   # ApplyRadon (radon.py:493-531)

    def A_adjoint(self, y, **kwargs):
        if self.fan_beam:
            return hr.fan_adjoint(y, self._geometry(y.device), self._norm())
        if self.adjoint_via_backprop:
            return hr.radon_adjoint(y, self._geometry(y.device), self._norm())
        # ApplyRadon(adjoint=True) = iradon(y, filtering=False) / pi * 2A = plain interpolated sum; / operator_norm
        return hr.apply_radon(y, self._geometry(y.device), self._scale_host(), True)

    def fbp(self, y, **kwargs):
        """filtered back-projection (tomography.py:258-293)"""
        if not self.adjoint_via_backprop and not self.fan_beam:
            out = hr.apply_radon(self.filter(y), self._geometry(y.device), 1.0, True) * torch.pi / (2 * self.angles.numel())
            if self.normalize:
                out = out * self.operator_norm
            if self.fbp_interpolate_boundary:
                out = torch.nn.functional.pad(out[:, :, 2:-2, 2:-2], (2, 2, 2, 2), mode="replicate")
            return out
        y = self.filter(y)
        out = self.A_adjoint(y, **kwargs) * torch.pi / (2 * self.angles.numel())
        if self.normalize:
            out = out * self.operator_norm ** 2
        if self.fbp_interpolate_boundary:
            out = torch.nn.functional.pad(out[:, :, 2:-2, 2:-2], (2, 2, 2, 2), mode="replicate")
        return out

    def A_dagger(self, y, fbp=False, **kwargs):
        if fbp:
            return self.fbp(y, **kwargs)
        return super().A_dagger(y, **kwargs)
