"""ctypes + autograd wrappers for the Radon kernels (csrc/radon.hip)."""
from __future__ import annotations

import ctypes

import torch

from . import check, f32c, lib, ptr, require_hip, stream_ptr


class RadonDesc(ctypes.Structure):
    _fields_ = [("n_img", ctypes.c_int32), ("width", ctypes.c_int32), ("grid", ctypes.c_int32),
                ("pad_before", ctypes.c_int32), ("n_angles", ctypes.c_int32), ("circle", ctypes.c_int32),
                ("scale", ctypes.c_float), ("reserved", ctypes.c_int32)]


_declared = False


def _l():
    global _declared
    l = lib()
    if not _declared:
        vp, i32, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_size_t
        D = ctypes.POINTER(RadonDesc)
        l.dinv_radon_workspace_bytes.restype = sz
        l.dinv_radon_workspace_bytes.argtypes = [D, i32]
        l.dinv_radon_forward.argtypes = [D, vp, vp, vp, vp, vp, sz, vp]
        l.dinv_radon_adjoint.argtypes = [D, vp, vp, vp, vp, vp, sz, vp]
        l.dinv_radon_ramp.argtypes = [i32, i32, i32, vp, vp, vp]
        l.dinv_radon_backproject.argtypes = [D, vp, vp, vp, vp, vp, vp]
        _declared = True
    return l


class RadonGeometry:
    """Host-side tables built exactly like the reference builds its grids (radon.py:70-71, 242-250, 334-341)."""

    def __init__(self, angles_deg: torch.Tensor, width: int, circle: bool, device):
        sqrt2 = (2 * torch.ones(1)).sqrt()
        self.W = int(width)
        if circle:
            self.G, self.pad_before = self.W, 0
        else:
            self.G = int((sqrt2 * self.W).ceil())
            pad = int((sqrt2 * self.W - self.W).ceil())
            self.pad_before = (self.W + pad) // 2 - self.W // 2
        self.circle = bool(circle)
        a = angles_deg.detach().to("cpu", torch.float32)
        theta = a * 4 * torch.ones(1).atan() / 180           # deg2rad (radon.py:70-71)
        self.A = int(a.numel())
        self.cs = torch.stack([theta.cos(), theta.sin()], dim=1).contiguous().to(device)
        self.xn = torch.linspace(-1, 1, self.G).to(device)   # affine_grid base grid, align_corners=True
        # IRadon grid x-coordinate of angle column a (radon.py:474-489) -> unnormalised like grid_sample does
        X = torch.arange(self.A, dtype=torch.float32) * 2.0 / (self.A - 1) - 1.0 if self.A > 1 else torch.zeros(1)
        self.ixtab = (((X + 1.0) / 2) * (self.A - 1)).contiguous().to(device)
        self.device = torch.device(device)

    def desc(self, n_img: int, scale: float) -> RadonDesc:
        return RadonDesc(n_img, self.W, self.G, self.pad_before, self.A, int(self.circle), float(scale), 0)


def _fwd(x, geo: RadonGeometry, scale):
    dev = require_hip(x, geo.xn)
    x = f32c(x)
    B, C, H, W = x.shape
    d = geo.desc(B * C, scale)
    sino = torch.empty((B, C, geo.G, geo.A), device=dev, dtype=torch.float32)
    ws = torch.empty(_l().dinv_radon_workspace_bytes(ctypes.byref(d), 0), device=dev, dtype=torch.uint8)
    check(_l().dinv_radon_forward(ctypes.byref(d), ptr(x), ptr(geo.xn), ptr(geo.cs), ptr(sino), ptr(ws), ws.numel(),
                                  stream_ptr(dev)))
    return sino


def _adj(y, geo: RadonGeometry, scale):
    dev = require_hip(y, geo.xn)
    y = f32c(y)
    B, C, G, A = y.shape
    if G != geo.G or A != geo.A:
        raise ValueError(f"sinogram of shape {tuple(y.shape)} does not match the operator ({geo.G} detectors, {geo.A} angles)")
    d = geo.desc(B * C, scale)
    x = torch.empty((B, C, geo.W, geo.W), device=dev, dtype=torch.float32)
    ws = torch.empty(_l().dinv_radon_workspace_bytes(ctypes.byref(d), 1), device=dev, dtype=torch.uint8)
    check(_l().dinv_radon_adjoint(ctypes.byref(d), ptr(y), ptr(geo.xn), ptr(geo.cs), ptr(x), ptr(ws), ws.numel(),
                                  stream_ptr(dev)))
    return x


class _RadonFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, geo, scale):
        ctx.geo, ctx.scale = geo, scale
        return _fwd(x, geo, scale)

    @staticmethod
    def backward(ctx, g):
        return _RadonAdj.apply(g, ctx.geo, ctx.scale), None, None


class _RadonAdj(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, geo, scale):
        ctx.geo, ctx.scale = geo, scale
        return _adj(y, geo, scale)

    @staticmethod
    def backward(ctx, g):
        return _RadonFwd.apply(g, ctx.geo, ctx.scale), None, None


def iradon_backproject(y, geo: RadonGeometry, scale=1.0):
    """sum_a interp(sino[:, a], x cos - y sin) on the image grid (IRadon.forward without filter / pi/(2A) factor)"""
    dev = require_hip(y, geo.xn)
    y = f32c(y)
    B, C, G, A = y.shape
    if G != geo.G or A != geo.A:
        raise ValueError(f"sinogram of shape {tuple(y.shape)} does not match the operator ({geo.G} detectors, {geo.A} angles)")
    out = torch.empty((B, C, geo.W, geo.W), device=dev, dtype=torch.float32)
    n = B * C
    for s0 in range(0, n, 65535):
        e = min(n, s0 + 65535)
        d = geo.desc(e - s0, scale)
        check(_l().dinv_radon_backproject(ctypes.byref(d), ptr(y.view(n, G, A)[s0:e]), ptr(geo.xn), ptr(geo.cs),
                                          ptr(geo.ixtab), ptr(out.view(n, geo.W, geo.W)[s0:e]), stream_ptr(dev)))
    return out


class _ApplyRadon(torch.autograd.Function):
    """Radon / interpolating back-projection pair of the reference's ``ApplyRadon`` (radon.py:493-531): each one's
    autograd backward is the other (an *inexact* adjoint pair by design)."""

    @staticmethod
    def forward(ctx, x, geo, scale, adjoint):
        ctx.geo, ctx.scale, ctx.adjoint = geo, scale, adjoint
        return iradon_backproject(x, geo, scale) if adjoint else _fwd(x, geo, scale)

    @staticmethod
    def backward(ctx, g):
        return _ApplyRadon.apply(g, ctx.geo, ctx.scale, not ctx.adjoint), None, None, None


def apply_radon(x, geo, scale, adjoint):
    return _ApplyRadon.apply(x, geo, float(scale), bool(adjoint))


def radon_forward(x, geo, scale=1.0):
    return _RadonFwd.apply(x, geo, float(scale))


def radon_adjoint(y, geo, scale=1.0):
    return _RadonAdj.apply(y, geo, float(scale))


class _Ramp(torch.autograd.Function):
    """the ramp kernel h is symmetric, so the filter is self-adjoint"""

    @staticmethod
    def forward(ctx, y):
        dev = require_hip(y)
        y = f32c(y)
        B, C, N, A = y.shape
        out = torch.empty_like(y)
        n = B * C
        step = 65535
        for s in range(0, n, step):
            e = min(n, s + step)
            yy, oo = y.view(n, N, A)[s:e], out.view(n, N, A)[s:e]
            check(_l().dinv_radon_ramp(e - s, N, A, ptr(yy), ptr(oo), stream_ptr(dev)))
        return out

    @staticmethod
    def backward(ctx, g):
        return _Ramp.apply(g)


def ramp_filter(y):
    """RampFilter along dim -2 of a sinogram [B,C,N_det,A] (radon.py:74-173)"""
    return _Ramp.apply(y)
