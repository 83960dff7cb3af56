"""ctypes + autograd wrappers for the Radon kernels (csrc/radon_tiled.hip, csrc/radon.hip).

Default path: the LDS-tiled forward / adjoint kernels and the FFT ramp filter; the operator norm of a normalised
``Tomography`` is handed to the kernels as a DEVICE scalar (no host read-back per call).  ``DINV_RADON_TILED=0``
selects the round-1 gather kernels (kept for detector counts above the tiled kernels' limit and as a cross-check)."""
from __future__ import annotations

import ctypes
import os

import torch

from . import FftPlan, check, f32c, fft_plan, fft_plan_host_table, lib, ptr, require_hip, stream_ptr


class RadonDesc(ctypes.Structure):
    _fields_ = [("n_img", ctypes.c_int32), ("width", ctypes.c_int32), ("grid", ctypes.c_int32),
                ("pad_before", ctypes.c_int32), ("n_angles", ctypes.c_int32), ("circle", ctypes.c_int32),
                ("scale", ctypes.c_float), ("reserved", ctypes.c_int32)]


class RadonPlan(ctypes.Structure):
    _fields_ = [("grid", ctypes.c_int32), ("n_angles", ctypes.c_int32), ("kw", ctypes.c_int32), ("band_h", ctypes.c_int32),
                ("win_w", ctypes.c_int32), ("n_jblocks", ctypes.c_int32), ("n_bands", ctypes.c_int32),
                ("n_chunks_plain", ctypes.c_int32), ("n_chunks_swap", ctypes.c_int32), ("fits", ctypes.c_int32),
                ("blob_words", ctypes.c_int32), ("widest_window", ctypes.c_int32), ("reserved", ctypes.c_int32 * 4)]


TILED_MAX_GRID = 4096   # MAXG in csrc/radon_tiled.hip
_declared = False


def _l():
    global _declared
    l = lib()
    if not _declared:
        vp, i32, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_size_t
        D, P, F = ctypes.POINTER(RadonDesc), ctypes.POINTER(RadonPlan), ctypes.POINTER(FftPlan)
        l.dinv_radon_workspace_bytes.restype = sz
        l.dinv_radon_workspace_bytes.argtypes = [D, i32]
        l.dinv_radon_forward.argtypes = [D, vp, vp, vp, vp, vp, sz, vp]
        l.dinv_radon_adjoint.argtypes = [D, vp, vp, vp, vp, vp, sz, vp]
        l.dinv_radon_ramp.argtypes = [i32, i32, i32, vp, vp, vp]
        l.dinv_radon_backproject.argtypes = [D, vp, vp, vp, vp, vp, vp]
        l.dinv_radon_plan_bytes.restype = sz
        l.dinv_radon_plan_bytes.argtypes = [D]
        l.dinv_radon_plan_init.argtypes = [D, vp, P, vp]
        l.dinv_radon_tiled_workspace_bytes.restype = sz
        l.dinv_radon_tiled_workspace_bytes.argtypes = [D, i32]
        l.dinv_radon_forward_tiled.argtypes = [D, P, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        l.dinv_radon_adjoint_tiled.argtypes = [D, vp, vp, vp, vp, vp, vp, sz, vp]
        l.dinv_radon_ramp_padded_size.restype = i32
        l.dinv_radon_ramp_padded_size.argtypes = [i32]
        l.dinv_radon_ramp_filter_init.argtypes = [i32, vp, vp]
        l.dinv_radon_ramp_fft.argtypes = [i32, i32, i32, i32, F, vp, vp, vp, vp, vp]
        l.dinv_radon_fan_workspace_bytes.restype = sz
        l.dinv_radon_fan_workspace_bytes.argtypes = [D, i32, i32]
        l.dinv_radon_fan_forward.argtypes = [D, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        l.dinv_radon_fan_adjoint.argtypes = [D, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        _declared = True
    return l


# test hooks (module attributes, patched by tests/test_tomography_gpu.py to compare kernel generations; not configuration)
ENABLE_TILED = True        # False: the first-generation gather kernels everywhere
FORCE_TILED = False        # True: the LDS-tiled forward even for angle lists with fewer than 4 angles per workgroup
ENABLE_RAMP_FFT = True     # False: the direct-convolution ramp filter


def _use_tiled(grid: int) -> bool:
    return grid <= TILED_MAX_GRID and ENABLE_TILED


class RadonGeometry:
    """Host-side tables built exactly like the reference builds its grids (radon.py:70-71, 242-250, 334-341), plus the
    window plan of the tiled forward kernel (angle chunks, per-band LDS windows; ``dinv_radon_plan_init``)."""

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
        cs_host = torch.stack([theta.cos(), theta.sin()], dim=1).contiguous()
        self.cs = cs_host.to(device)
        self.xn = torch.linspace(-1, 1, self.G).to(device)   # affine_grid base grid, align_corners=True
        # IRadon grid x-coordinate of angle column a (radon.py:474-489) -> unnormalised like grid_sample does
        X = torch.arange(self.A, dtype=torch.float32) * 2.0 / (self.A - 1) - 1.0 if self.A > 1 else torch.zeros(1)
        self.ixtab = (((X + 1.0) / 2) * (self.A - 1)).contiguous().to(device)
        self.device = torch.device(device)
        self.plan = self.plan_dev = None
        if self.device.type == "cuda" and self.G <= TILED_MAX_GRID:
            d = self.desc(1, 1.0)
            blob = torch.zeros(_l().dinv_radon_plan_bytes(ctypes.byref(d)) // 4, dtype=torch.int32)
            self.plan = RadonPlan()
            check(_l().dinv_radon_plan_init(ctypes.byref(d), ptr(cs_host), ctypes.byref(self.plan), ptr(blob)))
            self.plan_dev = blob[: self.plan.blob_words].to(device)

    def desc(self, n_img: int, scale: float) -> RadonDesc:
        return RadonDesc(n_img, self.W, self.G, self.pad_before, self.A, int(self.circle), float(scale), 0)


def _norm_ptr(norm, dev):
    if norm is None:
        return ptr(None)
    if norm.device != dev or norm.dtype != torch.float32:
        raise RuntimeError("the operator norm must be a float32 scalar on the operator's device")
    return ptr(norm)


def _fwd(x, geo: RadonGeometry, norm, scale=1.0):
    """x [B,C,W,W] -> [B,C,G,A] / norm * scale"""
    dev = require_hip(x, geo.xn)
    x = f32c(x)
    B, C, H, W = x.shape
    sino = torch.empty((B, C, geo.G, geo.A), device=dev, dtype=torch.float32)
    # few angles per workgroup (coarse or irregular angle lists) leave the LDS-tiled forward kernel under-occupied:
    # below 4 the gather kernel is the faster one (measured at 512^2: 60 angles, kw = 2)
    if _use_tiled(geo.G) and geo.plan is not None and (geo.plan.kw >= 4 or FORCE_TILED):
        d = geo.desc(B * C, scale)
        ws = torch.empty(_l().dinv_radon_tiled_workspace_bytes(ctypes.byref(d), 0), device=dev, dtype=torch.uint8)
        check(_l().dinv_radon_forward_tiled(ctypes.byref(d), ctypes.byref(geo.plan), ptr(geo.plan_dev), ptr(x), ptr(geo.xn),
                                            ptr(geo.cs), _norm_ptr(norm, dev), ptr(sino), ptr(ws), ws.numel(),
                                            stream_ptr(dev)))
        return sino
    d = geo.desc(B * C, scale)
    ws = torch.empty(_l().dinv_radon_workspace_bytes(ctypes.byref(d), 0), device=dev, dtype=torch.uint8)
    check(_l().dinv_radon_forward(ctypes.byref(d), ptr(x), ptr(geo.xn), ptr(geo.cs), ptr(sino), ptr(ws), ws.numel(),
                                  stream_ptr(dev)))
    return sino if norm is None else sino.div_(norm)   # the gather kernels take a host scalar only: divide on the device


def _adj(y, geo: RadonGeometry, norm, scale=1.0):
    dev = require_hip(y, geo.xn)
    y = f32c(y)
    B, C, G, A = y.shape
    if G != geo.G or A != geo.A:
        raise ValueError(f"sinogram of shape {tuple(y.shape)} does not match the operator ({geo.G} detectors, {geo.A} angles)")
    x = torch.empty((B, C, geo.W, geo.W), device=dev, dtype=torch.float32)
    if _use_tiled(geo.G):
        d = geo.desc(B * C, scale)
        ws = torch.empty(_l().dinv_radon_tiled_workspace_bytes(ctypes.byref(d), 1), device=dev, dtype=torch.uint8)
        check(_l().dinv_radon_adjoint_tiled(ctypes.byref(d), ptr(y), ptr(geo.xn), ptr(geo.cs), _norm_ptr(norm, dev), ptr(x),
                                            ptr(ws), ws.numel(), stream_ptr(dev)))
        return x
    d = geo.desc(B * C, scale)
    ws = torch.empty(_l().dinv_radon_workspace_bytes(ctypes.byref(d), 1), device=dev, dtype=torch.uint8)
    check(_l().dinv_radon_adjoint(ctypes.byref(d), ptr(y), ptr(geo.xn), ptr(geo.cs), ptr(x), ptr(ws), ws.numel(),
                                  stream_ptr(dev)))
    return x if norm is None else x.div_(norm)


class _RadonFwd(torch.autograd.Function):
    @staticmethod
    def forward(x, geo, norm):
        return _fwd(x, geo, norm)

    @staticmethod
    def setup_context(ctx, inputs, output):
        _, ctx.geo, ctx.norm = inputs

    @staticmethod
    def backward(ctx, g):
        return _RadonAdj.apply(g, ctx.geo, ctx.norm), None, None


class _RadonAdj(torch.autograd.Function):
    @staticmethod
    def forward(y, geo, norm):
        return _adj(y, geo, norm)

    @staticmethod
    def setup_context(ctx, inputs, output):
        _, ctx.geo, ctx.norm = inputs

    @staticmethod
    def backward(ctx, g):
        return _RadonFwd.apply(g, ctx.geo, ctx.norm), None, None


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
    def forward(x, geo, scale, adjoint):
        return iradon_backproject(x, geo, scale) if adjoint else _fwd(x, geo, None, scale)

    @staticmethod
    def setup_context(ctx, inputs, output):
        _, ctx.geo, ctx.scale, ctx.adjoint = inputs

    @staticmethod
    def backward(ctx, g):
        return _ApplyRadon.apply(g, ctx.geo, ctx.scale, not ctx.adjoint), None, None, None


def apply_radon(x, geo, scale, adjoint):
    return _ApplyRadon.apply(x, geo, float(scale), bool(adjoint))


def radon_forward(x, geo, norm=None):
    """Radon transform divided by the device scalar `norm` (None: unnormalised)"""
    return _RadonFwd.apply(x, geo, norm)


def radon_adjoint(y, geo, norm=None):
    return _RadonAdj.apply(y, geo, norm)


# --------------------------------------------------------------------------- fan-beam geometry
FAN_DEFAULTS = {"source_radius": 57.5, "detector_radius": 57.5, "n_detector_pixels": 258, "detector_spacing": 0.077}


def fan_tables(grid: int, width: int, fan_parameters=None):
    """Host tables of the fan-beam sampling lattice (fan_beam_grid, functional/radon.py:16-52; defaults of
    Radon.__init__ :224-240): xm [G] runs along the central ray, yd [n_det] across the detector, sc [G] is the stretch of
    the detector axis at march position xm (it grows linearly from the source to the detector).  Pure torch, CPU."""
    fp = dict(fan_parameters or {})
    fp.setdefault("pixel_spacing", 0.5 / width)
    for k, v in FAN_DEFAULTS.items():
        fp.setdefault(k, v)
    n_det = int(fp["n_detector_pixels"])
    if n_det < 2:
        raise ValueError("fan-beam geometry needs at least 2 detector pixels")
    unit = 2.0 / (grid * fp["pixel_spacing"])               # physical length -> normalised [-1, 1] image coordinates
    r_src, r_det = fp["source_radius"] * unit, fp["detector_radius"] * unit
    det_len = fp["detector_spacing"] * unit * (n_det - 1)
    xm = torch.linspace(-1, 1, grid)
    yd = torch.linspace(-1, 1, n_det)
    sc = 0.5 * det_len * (xm + r_src) / (r_src + r_det)
    return fp, xm.contiguous(), sc.contiguous(), yd.contiguous()


class FanGeometry(RadonGeometry):
    """RadonGeometry (padding, angle table) plus the fan-beam lattice tables; no tiled plan (gather kernels)."""

    def __init__(self, angles_deg, width, circle, device, fan_parameters=None):
        super().__init__(angles_deg, width, circle, "cpu")      # host part only: no window plan for this geometry
        self.fan_parameters, xm, sc, yd = fan_tables(self.G, self.W, fan_parameters)
        self.n_det = int(yd.numel())
        self.device = torch.device(device)
        self.cs, self.xn = self.cs.to(device), self.xn.to(device)
        self.xm, self.sc, self.yd = xm.to(device), sc.to(device), yd.to(device)


def _fan_fwd(x, geo: FanGeometry, norm):
    dev = require_hip(x, geo.xm)
    x = f32c(x)
    B, C = x.shape[:2]
    sino = torch.empty((B, C, geo.n_det, geo.A), device=dev, dtype=torch.float32)
    d = geo.desc(B * C, 1.0)
    ws = torch.empty(_l().dinv_radon_fan_workspace_bytes(ctypes.byref(d), geo.n_det, 0), device=dev, dtype=torch.uint8)
    check(_l().dinv_radon_fan_forward(ctypes.byref(d), geo.n_det, ptr(x), ptr(geo.xm), ptr(geo.sc), ptr(geo.yd), ptr(geo.cs),
                                      ptr(sino), ptr(ws), ws.numel(), stream_ptr(dev)))
    return sino if norm is None else sino.div_(norm)


def _fan_adj(y, geo: FanGeometry, norm):
    dev = require_hip(y, geo.xm)
    y = f32c(y)
    B, C, N, A = y.shape
    if N != geo.n_det or A != geo.A:
        raise ValueError(f"sinogram of shape {tuple(y.shape)} does not match the operator ({geo.n_det} detector pixels, "
                         f"{geo.A} angles)")
    x = torch.empty((B, C, geo.W, geo.W), device=dev, dtype=torch.float32)
    d = geo.desc(B * C, 1.0)
    ws = torch.empty(_l().dinv_radon_fan_workspace_bytes(ctypes.byref(d), geo.n_det, 1), device=dev, dtype=torch.uint8)
    check(_l().dinv_radon_fan_adjoint(ctypes.byref(d), geo.n_det, ptr(y), ptr(geo.xm), ptr(geo.sc), ptr(geo.yd), ptr(geo.cs),
                                      ptr(x), ptr(ws), ws.numel(), stream_ptr(dev)))
    return x if norm is None else x.div_(norm)


class _FanFwd(torch.autograd.Function):
    @staticmethod
    def forward(x, geo, norm):
        return _fan_fwd(x, geo, norm)

    @staticmethod
    def setup_context(ctx, inputs, output):
        _, ctx.geo, ctx.norm = inputs

    @staticmethod
    def backward(ctx, g):
        return _FanAdj.apply(g, ctx.geo, ctx.norm), None, None


class _FanAdj(torch.autograd.Function):
    @staticmethod
    def forward(y, geo, norm):
        return _fan_adj(y, geo, norm)

    @staticmethod
    def setup_context(ctx, inputs, output):
        _, ctx.geo, ctx.norm = inputs

    @staticmethod
    def backward(ctx, g):
        return _FanFwd.apply(g, ctx.geo, ctx.norm), None, None


def fan_forward(x, geo: FanGeometry, norm=None):
    return _FanFwd.apply(x, geo, norm)


def fan_adjoint(y, geo: FanGeometry, norm=None):
    return _FanAdj.apply(y, geo, norm)


# --------------------------------------------------------------------------- ramp filter
_ramp_cache: dict = {}


def _ramp_tables(n_det: int, device):
    """(P, fft plan, device fft table, device filter) of the reference's zero-padded rFFT ramp filter (radon.py:79-162)"""
    device = torch.device(device)
    key = (int(n_det), device.index if device.index is not None else torch.cuda.current_device())
    hit = _ramp_cache.get(key)
    if hit is None:
        P = int(_l().dinv_radon_ramp_padded_size(int(n_det)))
        plan, table = fft_plan(P, device)
        filt = torch.empty(P, dtype=torch.float32)
        check(_l().dinv_radon_ramp_filter_init(P, ptr(fft_plan_host_table(P)), ptr(filt)))
        hit = _ramp_cache[key] = (P, plan, table, filt.to(device))
    return hit


RAMP_FFT_MAX_P = 8192   # one complex column of P points + the twiddles must fit the 160 KiB LDS


class _Ramp(torch.autograd.Function):
    """the ramp kernel h is symmetric, so the filter is self-adjoint"""

    @staticmethod
    def setup_context(ctx, inputs, output):
        pass

    @staticmethod
    def forward(y):
        dev = require_hip(y)
        y = f32c(y)
        B, C, N, A = y.shape
        out = torch.empty_like(y)
        n = B * C
        P = int(_l().dinv_radon_ramp_padded_size(int(N)))
        use_fft = P <= RAMP_FFT_MAX_P and ENABLE_RAMP_FFT
        if use_fft:
            P, plan, table, filt = _ramp_tables(N, dev)
        step = 65535
        for s in range(0, n, step):
            e = min(n, s + step)
            yy, oo = y.view(n, N, A)[s:e], out.view(n, N, A)[s:e]
            if use_fft:
                check(_l().dinv_radon_ramp_fft(e - s, N, A, P, ctypes.byref(plan), ptr(table), ptr(filt), ptr(yy), ptr(oo),
                                               stream_ptr(dev)))
            else:
                check(_l().dinv_radon_ramp(e - s, N, A, ptr(yy), ptr(oo), stream_ptr(dev)))
        return out

    @staticmethod
    def backward(ctx, g):
        return _Ramp.apply(g)


def ramp_filter(y):
    """RampFilter along dim -2 of a sinogram [B,C,N_det,A] (radon.py:74-173)"""
    return _Ramp.apply(y)
