"""ctypes + autograd wrappers for csrc/blur.hip (padded convolution, its transpose, rfft2/irfft2)."""
from __future__ import annotations

import ctypes
import math

import torch

from . import FftPlan, check, f32c, fft_plan, lib, ptr, require_hip, stream_ptr

_MODES = {"valid": 0, "circular": 1, "reflect": 2, "replicate": 3, "constant": 4, "zeros": 4}


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("batch", "channels", "height", "width", "fbatch", "fchannels", "fh",
                                              "fw", "mode", "stride")]


class Conv3dDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("batch", "channels", "depth", "height", "width", "fbatch", "fchannels", "fd", "fh",
                                              "fw", "mode", "reserved")]


_declared = False


def _l():
    global _declared
    l = lib()
    if not _declared:
        vp, i32, i64, f32, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
        D, P = ctypes.POINTER(ConvDesc), ctypes.POINTER(FftPlan)
        l.dinv_conv2d_out_size.argtypes = [D, ctypes.POINTER(i32), ctypes.POINTER(i32)]
        l.dinv_conv2d.argtypes = [D, vp, vp, vp, vp]
        l.dinv_conv2d_transpose.argtypes = [D, vp, vp, vp, vp]
        l.dinv_conv2d_filter_grad.argtypes = [D, vp, vp, vp, vp]
        D3 = ctypes.POINTER(Conv3dDesc)
        l.dinv_conv3d_out_size.argtypes = [D3, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]
        for name in ("dinv_conv3d", "dinv_conv3d_transpose", "dinv_conv3d_filter_grad"):
            getattr(l, name).argtypes = [D3, vp, vp, vp, vp]
        l.dinv_rfft2.argtypes = [vp, vp, i64, P, vp, P, vp, f32, vp]
        l.dinv_irfft2.argtypes = [vp, vp, i64, P, vp, P, vp, f32, vp, sz, vp]
        l.dinv_blurfft_workspace_bytes.restype = sz
        l.dinv_blurfft_workspace_bytes.argtypes = [i64, i32, i32]
        l.dinv_blurfft_apply.argtypes = [vp, vp, i64, P, vp, P, vp, vp, vp, i64, i32, f32, f32, vp, sz, vp]
        l.dinv_spectrum_symbol.argtypes = [vp, vp, i64, i32, i32, vp, vp, i64, i32, f32, vp]
        _declared = True
    return l


def pad_mode(padding: str) -> int:
    p = padding.lower()
    if p not in _MODES:
        raise ValueError(f"padding = '{padding}' not implemented. Please use one of 'valid', 'circular', "
                         "'replicate', 'reflect', 'constant' or 'zeros'.")
    return _MODES[p]


def _check_broadcast(B, C, filt):
    b, c = filt.shape[:2]
    if c != C and c != 1:
        raise AssertionError(f"Number of channels of the kernel is not matched for broadcasting, got c={c} and C={C}")
    if b != B and b != 1:
        raise AssertionError(f"Batch size of the kernel is not matched for broadcasting, got b={b} and B={B}")


def _desc(B, C, spatial, filt, padding, stride):
    """descriptor + output spatial size of y = conv(x, filt) for x [B, C, *spatial] (2 or 3 spatial dims)"""
    _check_broadcast(B, C, filt)
    if len(spatial) == 2:
        d = ConvDesc(B, C, *spatial, *filt.shape, pad_mode(padding), int(stride))
        ho, wo = ctypes.c_int32(), ctypes.c_int32()
        check(_l().dinv_conv2d_out_size(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)))
        return d, (ho.value, wo.value)
    if int(stride) != 1:
        raise ValueError("strided convolution is 2-D only (Downsampling)")
    d = Conv3dDesc(B, C, *spatial, *filt.shape, pad_mode(padding), 0)
    do, ho, wo = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    check(_l().dinv_conv3d_out_size(ctypes.byref(d), ctypes.byref(do), ctypes.byref(ho), ctypes.byref(wo)))
    return d, (do.value, ho.value, wo.value)


def _entry(nd, name):
    return getattr(_l(), f"dinv_conv{nd}d{name}")


def _conv_fwd(x, filt, padding, stride):
    dev = require_hip(x, filt)
    x, filt = f32c(x), f32c(filt)
    B, C, *sp = x.shape
    d, out = _desc(B, C, sp, filt, padding, stride)
    y = torch.empty((B, C, *out), device=dev, dtype=torch.float32)
    check(_entry(len(sp), "")(ctypes.byref(d), ptr(x), ptr(filt), ptr(y), stream_ptr(dev)))
    return y


def _conv_adj(y, filt, padding, stride, size):
    dev = require_hip(y, filt)
    y, filt = f32c(y), f32c(filt)
    B, C = y.shape[:2]
    d, out = _desc(B, C, tuple(size), filt, padding, stride)
    if out != tuple(y.shape[2:]):
        raise ValueError(f"measurement of spatial size {tuple(y.shape[2:])} does not match the operator output {out}")
    x = torch.empty((B, C, *size), device=dev, dtype=torch.float32)
    check(_entry(len(size), "_transpose")(ctypes.byref(d), ptr(y), ptr(filt), ptr(x), stream_ptr(dev)))
    return x


def _filter_grad(x, gy, filt, padding, stride):
    """d<gy, conv(x, k)>/dk in the shape of `filt` (the planes a broadcast filter is shared by are summed)"""
    dev = require_hip(x, gy)
    x, gy = f32c(x), f32c(gy)
    B, C, *sp = x.shape
    d, _ = _desc(B, C, sp, filt, padding, stride)
    planes = torch.empty((B, C, *filt.shape[2:]), device=dev, dtype=torch.float32)
    check(_entry(len(sp), "_filter_grad")(ctypes.byref(d), ptr(x), ptr(gy), ptr(planes), stream_ptr(dev)))
    if filt.shape[0] == 1 and B > 1:
        planes = planes.sum(0, keepdim=True)
    if filt.shape[1] == 1 and C > 1:
        planes = planes.sum(1, keepdim=True)
    return planes


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(x, filt, padding, stride):
        return _conv_fwd(x, filt, padding, stride)

    @staticmethod
    def setup_context(ctx, inputs, output):
        x, filt, padding, stride = inputs
        ctx.save_for_backward(filt, x)
        ctx.cfg = (padding, stride, tuple(x.shape[2:]))

    @staticmethod
    def backward(ctx, g):
        filt, x = ctx.saved_tensors
        padding, stride, size = ctx.cfg
        gx = _ConvT.apply(g, filt, padding, stride, size) if ctx.needs_input_grad[0] else None
        gk = _FilterGrad.apply(x, g, filt, padding, stride) if ctx.needs_input_grad[1] else None
        return gx, gk, None, None


class _ConvT(torch.autograd.Function):
    @staticmethod
    def forward(y, filt, padding, stride, size):
        return _conv_adj(y, filt, padding, stride, size)

    @staticmethod
    def setup_context(ctx, inputs, output):
        y, filt, padding, stride, _ = inputs
        ctx.save_for_backward(filt, y)
        ctx.cfg = (padding, stride)

    @staticmethod
    def backward(ctx, g):
        filt, y = ctx.saved_tensors
        padding, stride = ctx.cfg
        gy = _Conv.apply(g, filt, padding, stride) if ctx.needs_input_grad[0] else None
        # <g, A_k^T y> = <A_k g, y>: the filter gradient of the forward convolution with image g and output gradient y
        gk = _FilterGrad.apply(g, y, filt, padding, stride) if ctx.needs_input_grad[1] else None
        return gy, gk, None, None, None


class _FilterGrad(torch.autograd.Function):
    """k -> d<gy, conv(x, k)>/dk is bilinear in (x, gy) and does not depend on k: its own derivatives are convolutions again"""

    @staticmethod
    def forward(x, gy, filt, padding, stride):
        return _filter_grad(x, gy, filt, padding, stride)

    @staticmethod
    def setup_context(ctx, inputs, output):
        x, gy, filt, padding, stride = inputs
        ctx.save_for_backward(x, gy)
        ctx.cfg = (padding, stride, tuple(filt.shape))

    @staticmethod
    def backward(ctx, gk):
        x, gy = ctx.saved_tensors
        padding, stride, fshape = ctx.cfg
        gk = gk.expand(fshape) if tuple(gk.shape) != fshape else gk
        # <gk, dk(x, gy)> = <gy, conv(x, gk)>: linear in x (transpose conv of gy with gk) and in gy (conv of x with gk)
        gx = _ConvT.apply(gy, gk, padding, stride, tuple(x.shape[2:])) if ctx.needs_input_grad[0] else None
        ggy = _Conv.apply(x, gk, padding, stride) if ctx.needs_input_grad[1] else None
        return gx, ggy, None, None, None


def conv2d_strided(x, filt, padding="valid", stride=1):
    return _Conv.apply(x, filt, padding, int(stride))


def conv2d_strided_transpose(y, filt, padding, stride, H, W):
    return _ConvT.apply(y, filt, padding, int(stride), (int(H), int(W)))


def conv3d(x, filt, padding="valid"):
    return _Conv.apply(x, filt, padding, 1)


def conv3d_transpose(y, filt, padding, size):
    return _ConvT.apply(y, filt, padding, 1, tuple(int(v) for v in size))


# --------------------------------------------------------------------------- rfft2 / irfft2
def _scale(norm, n, inverse):
    if norm == "ortho":
        return 1.0 / math.sqrt(n)
    if norm == "backward":
        return 1.0 / n if inverse else 1.0
    if norm == "forward":
        return 1.0 if inverse else 1.0 / n
    raise ValueError(f"unknown norm {norm}")


def _rfft2_raw(x, norm):
    dev = require_hip(x)
    x = f32c(x)
    H, W = x.shape[-2:]
    P = x.numel() // (H * W) if x.numel() else 0
    out = torch.empty((*x.shape[:-1], W // 2 + 1), device=dev, dtype=torch.complex64)
    ph, th = fft_plan(H, dev)
    pw, tw = fft_plan(W, dev)
    check(_l().dinv_rfft2(ptr(x), ptr(out), P, ctypes.byref(ph), ptr(th), ctypes.byref(pw), ptr(tw),
                          _scale(norm, H * W, False), stream_ptr(dev)))
    return out


def _irfft2_raw(xc, s, norm):
    dev = require_hip(xc)
    xc = xc.to(torch.complex64).contiguous()
    H, W = int(s[0]), int(s[1])
    if xc.shape[-2] != H or xc.shape[-1] != W // 2 + 1:
        raise ValueError(f"half spectrum of shape {tuple(xc.shape[-2:])} does not match output size {(H, W)}")
    P = xc.numel() // (H * (W // 2 + 1)) if xc.numel() else 0
    out = torch.empty((*xc.shape[:-2], H, W), device=dev, dtype=torch.float32)
    ws = torch.empty(max(xc.numel(), 1) * 8, device=dev, dtype=torch.uint8)
    ph, th = fft_plan(H, dev)
    pw, tw = fft_plan(W, dev)
    check(_l().dinv_irfft2(ptr(xc), ptr(out), P, ctypes.byref(ph), ptr(th), ctypes.byref(pw), ptr(tw),
                           _scale(norm, H * W, True), ptr(ws), ws.numel(), stream_ptr(dev)))
    return out


class _Rfft2(torch.autograd.Function):
    """ortho rfft2; its backward is the *adjoint* (not the inverse) of the real-to-half-complex map."""

    @staticmethod
    def forward(x, norm):
        return _rfft2_raw(x, norm)

    @staticmethod
    def setup_context(ctx, inputs, output):
        x, norm = inputs
        ctx.cfg = (norm, x.shape[-2], x.shape[-1])

    @staticmethod
    def backward(ctx, g):
        norm, H, W = ctx.cfg
        # adjoint of rfft: zero-extend the half spectrum to the full one and take Re(ifft) with flipped norm;
        # equivalently irfft of the spectrum with interior bins halved.
        flip = {"ortho": "ortho", "backward": "forward", "forward": "backward"}[norm]
        w = torch.full((W // 2 + 1,), 0.5, device=g.device)
        w[0] = 1.0
        if W % 2 == 0:
            w[-1] = 1.0
        return _Irfft2.apply(g * w, (H, W), flip), None


class _Irfft2(torch.autograd.Function):
    @staticmethod
    def forward(xc, s, norm):
        return _irfft2_raw(xc, s, norm)

    @staticmethod
    def setup_context(ctx, inputs, output):
        _, s, norm = inputs
        ctx.cfg = (norm, s)

    @staticmethod
    def backward(ctx, g):
        norm, (H, W) = ctx.cfg
        flip = {"ortho": "ortho", "backward": "forward", "forward": "backward"}[norm]
        w = torch.full((W // 2 + 1,), 2.0, device=g.device)
        w[0] = 1.0
        if W % 2 == 0:
            w[-1] = 1.0
        return _Rfft2.apply(g, flip) * w, None, None


def rfft2(x, norm="backward"):
    return _Rfft2.apply(x, norm)


def irfft2(xc, s, norm="backward"):
    return _Irfft2.apply(xc, tuple(s), norm)


# --------------------------------------------------------------------------- BlurFFT: symbol of the spectrum, fused operator
SYM_PRE_CONJ_ANGLE, SYM_POST_ANGLE = 1, 2
SYM_NONE, SYM_MASK, SYM_MASK2, SYM_PROX, SYM_DAGGER = (m << 4 for m in range(5))


def _symbol_planes(x_planes_shape, mask, angle, H, Wh):
    """number of symbol planes Ps (spectrum plane p uses symbol plane p % Ps), or None when the buffers do not have the shape the
    kernels index: mask [1|B, C, H, Wh, 2] fp32 and angle [1|B, C, H, Wh] complex64 of one common leading shape, contiguous"""
    lead = None
    for t, tail, dt in ((mask, (H, Wh, 2), torch.float32), (angle, (H, Wh), torch.complex64)):
        if t is None:
            continue
        if not (isinstance(t, torch.Tensor) and t.dtype == dt and t.is_contiguous() and tuple(t.shape[t.dim() - len(tail):]) == tail):
            return None
        l = tuple(t.shape[: t.dim() - len(tail)])
        while len(l) > 1 and l[0] == 1:
            l = l[1:]
        if lead is not None and l != lead:
            return None
        lead = l
    lead = tuple(lead or ())
    xs = tuple(x_planes_shape)
    if lead and lead != (1,) and xs[len(xs) - len(lead):] != lead:
        return None
    return max(int(math.prod(lead)), 1)


def blurfft_supported(x, mask, angle):
    """the fused BlurFFT operator takes this image / symbol pair (plain fp32 device tensors outside autograd recording)"""
    from . import elementwise as EW

    if not (EW.eligible(x) and x.dim() >= 2 and x.shape[-1] >= 2):
        return False
    H, W = x.shape[-2:]
    for t in (mask, angle):
        if t is not None and ((torch.is_grad_enabled() and t.requires_grad) or t.device != x.device):
            return False
    return _symbol_planes(x.shape[:-2], mask, angle, H, W // 2 + 1) is not None


def blurfft_apply(x, mask, angle, flags: int, add: float = 0.0, norm: str = "ortho"):
    """irfft2(SYMBOL(rfft2(x))) in one call of dinv_blurfft_apply (csrc/blur.hip): flags = SYM_* scale mode | SYM_PRE_CONJ_ANGLE |
    SYM_POST_ANGLE; `mask` / `angle` are BlurFFT's buffers (blur.py:677-686)"""
    dev = require_hip(x)
    H, W = x.shape[-2:]
    Ps = _symbol_planes(x.shape[:-2], mask, angle, H, W // 2 + 1)
    if Ps is None:
        raise ValueError(f"blurfft_apply: symbol buffers {None if mask is None else tuple(mask.shape)} / "
                         f"{None if angle is None else tuple(angle.shape)} do not fit an image of shape {tuple(x.shape)}")
    P = x.numel() // (H * W) if x.numel() else 0
    out = torch.empty_like(x)
    nb = int(_l().dinv_blurfft_workspace_bytes(P, H, W))
    ws = torch.empty(max(nb, 16), device=dev, dtype=torch.uint8)
    ph, th = fft_plan(H, dev)
    pw, tw = fft_plan(W, dev)
    scale = _scale(norm, H * W, False) * _scale(norm, H * W, True)
    check(_l().dinv_blurfft_apply(ptr(x), ptr(out), P, ctypes.byref(ph), ptr(th), ctypes.byref(pw), ptr(tw), ptr(mask),
                                  ptr(None if angle is None else torch.view_as_real(angle)), Ps, int(flags), float(add), scale,
                                  ptr(ws), ws.numel(), stream_ptr(dev)))
    return out


def spectrum_symbol(spec, mask, angle, flags: int, add: float = 0.0):
    """SYMBOL(spec) for a contiguous complex64 half spectrum [..., H, Wh] (the multiplications of BlurFFT.U / U_adjoint)"""
    dev = require_hip(spec)
    H, Wh = spec.shape[-2:]
    Ps = _symbol_planes(spec.shape[:-2], mask, angle, H, Wh)
    if Ps is None or spec.dtype != torch.complex64 or not spec.is_contiguous():
        raise ValueError(f"spectrum_symbol: symbol buffers do not fit a spectrum of shape {tuple(spec.shape)} / {spec.dtype}")
    out = torch.empty_like(spec)
    P = spec.numel() // (H * Wh) if spec.numel() else 0
    check(_l().dinv_spectrum_symbol(ptr(torch.view_as_real(spec)), ptr(torch.view_as_real(out)), P, H, Wh, ptr(mask),
                                    ptr(None if angle is None else torch.view_as_real(angle)), Ps, int(flags), float(add),
                                    stream_ptr(dev)))
    return out
