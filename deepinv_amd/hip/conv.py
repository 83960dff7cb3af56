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
        l.dinv_rfft2.argtypes = [vp, vp, i64, P, vp, P, vp, f32, vp]
        l.dinv_irfft2.argtypes = [vp, vp, i64, P, vp, P, vp, f32, vp, sz, vp]
        _declared = True
    return l


def pad_mode(padding: str) -> int:
    p = padding.lower()
    if p not in _MODES:
        raise ValueError(f"padding = '{padding}' not implemented. Please use one of 'valid', 'circular', "
                         "'replicate', 'reflect', 'constant' or 'zeros'.")
    return _MODES[p]


def _desc(B, C, H, W, filt, padding, stride):
    b, c, h, w = filt.shape
    if c != C and c != 1:
        raise AssertionError(f"Number of channels of the kernel is not matched for broadcasting, got c={c} and C={C}")
    if b != B and b != 1:
        raise AssertionError(f"Batch size of the kernel is not matched for broadcasting, got b={b} and B={B}")
    d = ConvDesc(B, C, H, W, b, c, h, w, pad_mode(padding), int(stride))
    ho, wo = ctypes.c_int32(), ctypes.c_int32()
    check(_l().dinv_conv2d_out_size(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)))
    return d, ho.value, wo.value


def _chunks(n, step=65535):
    return [(s, min(n, s + step)) for s in range(0, n, step)]


def _conv_fwd(x, filt, padding, stride):
    dev = require_hip(x, filt)
    x, filt = f32c(x), f32c(filt)
    B, C, H, W = x.shape
    d, ho, wo = _desc(B, C, H, W, filt, padding, stride)
    y = torch.empty((B, C, ho, wo), device=dev, dtype=torch.float32)
    check(_l().dinv_conv2d(ctypes.byref(d), ptr(x), ptr(filt), ptr(y), stream_ptr(dev)))
    return y


def _conv_adj(y, filt, padding, stride, H, W):
    dev = require_hip(y, filt)
    y, filt = f32c(y), f32c(filt)
    B, C = y.shape[:2]
    d, ho, wo = _desc(B, C, H, W, filt, padding, stride)
    if (ho, wo) != tuple(y.shape[-2:]):
        raise ValueError(f"measurement of spatial size {tuple(y.shape[-2:])} does not match the operator output {(ho, wo)}")
    x = torch.empty((B, C, H, W), device=dev, dtype=torch.float32)
    check(_l().dinv_conv2d_transpose(ctypes.byref(d), ptr(y), ptr(filt), ptr(x), stream_ptr(dev)))
    return x


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(x, filt, padding, stride):
        return _conv_fwd(x, filt, padding, stride)

    @staticmethod
    def setup_context(ctx, inputs, output):
        x, filt, padding, stride = inputs
        ctx.save_for_backward(filt)
        ctx.cfg = (padding, stride, x.shape[-2], x.shape[-1])

    @staticmethod
    def backward(ctx, g):
        (filt,) = ctx.saved_tensors
        padding, stride, H, W = ctx.cfg
        return _ConvT.apply(g, filt, padding, stride, H, W), None, None, None


class _ConvT(torch.autograd.Function):
    @staticmethod
    def forward(y, filt, padding, stride, H, W):
        return _conv_adj(y, filt, padding, stride, H, W)

    @staticmethod
    def setup_context(ctx, inputs, output):
        _, filt, padding, stride, _, _ = inputs
        ctx.save_for_backward(filt)
        ctx.cfg = (padding, stride)

    @staticmethod
    def backward(ctx, g):
        (filt,) = ctx.saved_tensors
        padding, stride = ctx.cfg
        return _Conv.apply(g, filt, padding, stride), None, None, None, None, None


def _no_filter_grad(filt):
    """the reference gets d/d(filter) from autograd through F.conv2d; these kernels differentiate w.r.t. the image
    only, so a filter that asks for a gradient fails loudly instead of silently receiving None"""
    if torch.is_grad_enabled() and isinstance(filt, torch.Tensor) and filt.requires_grad:
        raise NotImplementedError("gradients w.r.t. the blur filter are not implemented on the HIP path "
                                  "(blind / learned kernels): detach the filter or differentiate w.r.t. the image")


def conv2d_strided(x, filt, padding="valid", stride=1):
    _no_filter_grad(filt)
    return _Conv.apply(x, filt, padding, int(stride))


def conv2d_strided_transpose(y, filt, padding, stride, H, W):
    _no_filter_grad(filt)
    return _ConvT.apply(y, filt, padding, int(stride), int(H), int(W))


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
