"""TEST INFRASTRUCTURE: ctypes access to tests/emu/libdeepinv_amd_emu.so, i.e. the product's kernel SOURCES
(deepinv_amd/csrc/*.hip) compiled for the host against a fiber-based emulation of the HIP execution model
(tests/emu/include/hip/hip_runtime.h).  It lets the GPU-less CPU suite execute the real kernel code on small problems
and compare it with the oracle.  The product never loads this library."""
import ctypes
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "emu")
LIB = os.path.join(EMU_DIR, "libdeepinv_amd_emu.so")

MAX_STAGES = 16


class FftPlan(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("nstages", ctypes.c_int32), ("generic", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("radix", ctypes.c_int32 * MAX_STAGES)]


class RadonDesc(ctypes.Structure):
    _fields_ = [("n_img", ctypes.c_int32), ("width", ctypes.c_int32), ("grid", ctypes.c_int32),
                ("pad_before", ctypes.c_int32), ("n_angles", ctypes.c_int32), ("circle", ctypes.c_int32),
                ("scale", ctypes.c_float), ("reserved", ctypes.c_int32)]


class RadonPlan(ctypes.Structure):
    _fields_ = [("grid", ctypes.c_int32), ("n_angles", ctypes.c_int32), ("kw", ctypes.c_int32), ("band_h", ctypes.c_int32),
                ("win_w", ctypes.c_int32), ("n_jblocks", ctypes.c_int32), ("n_bands", ctypes.c_int32),
                ("n_chunks_plain", ctypes.c_int32), ("n_chunks_swap", ctypes.c_int32), ("fits", ctypes.c_int32),
                ("blob_words", ctypes.c_int32), ("widest_window", ctypes.c_int32), ("reserved", ctypes.c_int32 * 4)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-C", EMU_DIR, "-j4"], check=True, stdout=subprocess.DEVNULL)
        l = ctypes.CDLL(LIB)
        l.dinv_last_error.restype = ctypes.c_char_p
        for name in ("dinv_radon_plan_bytes", "dinv_fft_table_bytes"):
            getattr(l, name).restype = ctypes.c_size_t
        l.dinv_radon_tiled_workspace_bytes.restype = ctypes.c_size_t
        l.dinv_radon_workspace_bytes.restype = ctypes.c_size_t
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(f"emu lib error {rc}: {lib().dinv_last_error().decode()}")


def p(a):
    """pointer to a numpy array / torch CPU tensor (None -> NULL)"""
    if a is None:
        return ctypes.c_void_p(0)
    if isinstance(a, torch.Tensor):
        assert a.is_contiguous() and a.device.type == "cpu"
        return ctypes.c_void_p(a.data_ptr())
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


class RadonGeom:
    """the host tables exactly as deepinv_amd.hip.radon.RadonGeometry builds them"""

    def __init__(self, angles_deg, width, circle):
        sqrt2 = (2 * torch.ones(1)).sqrt()
        self.W = int(width)
        if circle:
            self.G, self.pad = self.W, 0
        else:
            self.G = int((sqrt2 * self.W).ceil())
            pad = int((sqrt2 * self.W - self.W).ceil())
            self.pad = (self.W + pad) // 2 - self.W // 2
        self.circle = bool(circle)
        a = torch.as_tensor(angles_deg, dtype=torch.float32)
        theta = a * 4 * torch.ones(1).atan() / 180
        self.A = int(a.numel())
        self.cs = torch.stack([theta.cos(), theta.sin()], dim=1).contiguous()
        self.xn = torch.linspace(-1, 1, self.G).contiguous()

    def desc(self, n_img, scale=1.0):
        return RadonDesc(n_img, self.W, self.G, self.pad, self.A, int(self.circle), float(scale), 0)

    def plan(self, n_img, kw=0):
        """kw = 1, 2, 4, 8 forces the angles per workgroup (dinv_radon_plan_init reads plan.kw on entry), 0 = automatic"""
        l = lib()
        d = self.desc(n_img)
        nbytes = l.dinv_radon_plan_bytes(ctypes.byref(d))
        blob = np.zeros(nbytes // 4, np.int32)
        pl = RadonPlan()
        pl.kw = kw
        check(l.dinv_radon_plan_init(ctypes.byref(d), p(self.cs), ctypes.byref(pl), p(blob)))
        return pl, blob


def radon_forward_tiled(x, geo, norm=None, scale=1.0, kw=0):
    l = lib()
    B, C, W, _ = x.shape
    x = x.contiguous().float()
    d = geo.desc(B * C, scale)
    pl, blob = geo.plan(B * C, kw)
    sino = torch.full((B, C, geo.G, geo.A), float("nan"))
    ws = np.zeros(l.dinv_radon_tiled_workspace_bytes(ctypes.byref(d), 0), np.uint8)
    check(l.dinv_radon_forward_tiled(ctypes.byref(d), ctypes.byref(pl), p(blob), p(x), p(geo.xn), p(geo.cs), p(norm), p(sino),
                                     p(ws), ctypes.c_size_t(ws.size), None))
    return sino, pl


def radon_adjoint_tiled(y, geo, norm=None, scale=1.0):
    l = lib()
    B, C, G, A = y.shape
    y = y.contiguous().float()
    d = geo.desc(B * C, scale)
    x = torch.full((B, C, geo.W, geo.W), float("nan"))
    ws = np.zeros(l.dinv_radon_tiled_workspace_bytes(ctypes.byref(d), 1), np.uint8)
    check(l.dinv_radon_adjoint_tiled(ctypes.byref(d), p(y), p(geo.xn), p(geo.cs), p(norm), p(x), p(ws),
                                     ctypes.c_size_t(ws.size), None))
    return x


def radon_forward_gather(x, geo, scale=1.0):
    l = lib()
    B, C, W, _ = x.shape
    x = x.contiguous().float()
    d = geo.desc(B * C, scale)
    sino = torch.full((B, C, geo.G, geo.A), float("nan"))
    ws = np.zeros(l.dinv_radon_workspace_bytes(ctypes.byref(d), 0), np.uint8)
    check(l.dinv_radon_forward(ctypes.byref(d), p(x), p(geo.xn), p(geo.cs), p(sino), p(ws), ctypes.c_size_t(ws.size), None))
    return sino


def fft_plan(n):
    l = lib()
    plan = FftPlan()
    table = np.zeros(l.dinv_fft_table_bytes(ctypes.c_int32(n)), np.uint8)
    check(l.dinv_fft_plan_init(ctypes.c_int32(n), ctypes.byref(plan), p(table)))
    return plan, table


def ramp_fft(y):
    l = lib()
    B, C, N, A = y.shape
    y = y.contiguous().float()
    P = l.dinv_radon_ramp_padded_size(ctypes.c_int32(N))
    plan, table = fft_plan(P)
    filt = np.zeros(P, np.float32)
    check(l.dinv_radon_ramp_filter_init(ctypes.c_int32(P), p(table), p(filt)))
    out = torch.full_like(y, float("nan"))
    check(l.dinv_radon_ramp_fft(B * C, N, A, P, ctypes.byref(plan), p(table), p(filt), p(y), p(out), None))
    return out


class FanGeom(RadonGeom):
    """fan-beam tables from the product's own host code (deepinv_amd.hip.radon.fan_tables: pure torch)"""

    def __init__(self, angles_deg, width, circle, fan_parameters=None):
        super().__init__(angles_deg, width, circle)
        import sys
        sys.path.insert(0, os.path.dirname(HERE))
        from deepinv_amd.hip.radon import fan_tables
        self.fp, self.xm, self.sc, self.yd = fan_tables(self.G, self.W, fan_parameters)
        self.n_det = int(self.yd.numel())


def radon_fan_forward(x, geo):
    l = lib()
    l.dinv_radon_fan_workspace_bytes.restype = ctypes.c_size_t
    B, C, W, _ = x.shape
    x = x.contiguous().float()
    d = geo.desc(B * C)
    sino = torch.full((B, C, geo.n_det, geo.A), float("nan"))
    ws = np.zeros(l.dinv_radon_fan_workspace_bytes(ctypes.byref(d), geo.n_det, 0), np.uint8)
    check(l.dinv_radon_fan_forward(ctypes.byref(d), geo.n_det, p(x), p(geo.xm), p(geo.sc), p(geo.yd), p(geo.cs), p(sino), p(ws),
                                   ctypes.c_size_t(ws.size), None))
    return sino


def radon_fan_adjoint(y, geo):
    l = lib()
    l.dinv_radon_fan_workspace_bytes.restype = ctypes.c_size_t
    B, C, N, A = y.shape
    y = y.contiguous().float()
    d = geo.desc(B * C)
    x = torch.full((B, C, geo.W, geo.W), float("nan"))
    ws = np.zeros(l.dinv_radon_fan_workspace_bytes(ctypes.byref(d), geo.n_det, 1), np.uint8)
    check(l.dinv_radon_fan_adjoint(ctypes.byref(d), geo.n_det, p(y), p(geo.xm), p(geo.sc), p(geo.yd), p(geo.cs), p(x), p(ws),
                                   ctypes.c_size_t(ws.size), None))
    return x
