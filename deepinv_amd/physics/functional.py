"""Functional layer of the blur operators — API mirror of ``deepinv.physics.functional``
(convolution.py: conv2d / conv_transpose2d / conv2d_fft / conv_transpose2d_fft / filter_fft;
blur.py: gaussian_blur / bilinear_filter / bicubic_filter / sinc_filter / kaiser_window).

The convolutions run on the HIP kernels of csrc/blur.hip; the filter constructors are tiny
host-side tensors (SURVEY.md §2.1 marks them "host-side filter construction only").
"""
from __future__ import annotations

from math import pi, sqrt

import torch
import torch.nn.functional as F

from ..hip import conv as hc
from ..hip import fft as hfft


# --------------------------------------------------------------------------- convolutions
def _check4(x, filter):
    if x.dim() != filter.dim() or filter.dim() != 4:
        raise ValueError("Input and filter must be 4D tensors")


def conv2d(x, filter, padding="valid", correlation=False):
    """True convolution (kernel flipped) with deepinv's padding conventions (convolution.py:42-107)."""
    _check4(x, filter)
    if correlation:
        filter = filter.flip(dims=(-2, -1))
    return hc.conv2d_strided(x, filter, padding, 1)


def conv_transpose2d(y, filter, padding="valid", correlation=False):
    """Exact transpose of :func:`conv2d` (convolution.py:110-164, 689-758)."""
    _check4(y, filter)
    if correlation:
        filter = filter.flip(dims=(-2, -1))
    h, w = filter.shape[-2:]
    if hc.pad_mode(padding) == 0:
        H, W = y.shape[-2] + h - 1, y.shape[-1] + w - 1
    else:
        H, W = y.shape[-2:]
    return hc.conv2d_strided_transpose(y, filter, padding, 1, H, W)


def filter_fft(filter, img_size, real_fft=True, dims=(-1, -2)):
    """FFT of a filter zero-padded to ``img_size`` and rolled so its centre sits at index 0
    (convolution.py:790-812).  Tiny tensor: the pad/roll are torch ops, the transform is ours."""
    dims = sorted(dims)
    f_size = tuple(filter.shape[d] for d in dims)
    i_size = tuple(img_size[d] for d in dims)
    pad = []
    for f, i in zip(reversed(f_size), reversed(i_size)):
        pad += [0, i - f]
    filter = F.pad(filter, tuple(pad), mode="constant", value=0)
    filter = torch.roll(filter, shifts=tuple(-int(f / 2) for f in f_size), dims=dims)
    if dims != [-2, -1] and dims != [filter.ndim - 2, filter.ndim - 1]:
        raise NotImplementedError("only 2-D (last two dims) filter FFTs are on the accelerated path")
    if real_fft:
        return hc.rfft2(filter, norm="backward")
    return hfft.fftn(filter.to(torch.complex64), dim=(-2, -1), norm="backward")


def _circular_conv_fft(x, filter, s, real_fft=True, shift_filter=False, transpose=False):
    """convolution.py:837-865 (2-D)"""
    H, W = x.shape[-2:]
    if (H, W) != tuple(s):  # zero-pad the signal to s (rfftn(x, s=s))
        x = F.pad(x, (0, s[1] - W, 0, s[0] - H))
    fx = hc.rfft2(x, norm="backward")
    if shift_filter:
        ff = filter_fft(filter, img_size=(*filter.shape[:-2], *s), real_fft=True, dims=(-2, -1))
    else:
        fh, fw = filter.shape[-2:]
        ff = hc.rfft2(F.pad(filter, (0, s[1] - fw, 0, s[0] - fh)), norm="backward")
    prod = fx * (torch.conj(ff) if transpose else ff)
    return hc.irfft2(prod, s, norm="backward")


def conv2d_fft(x, filter, real_fft=True, padding="valid"):
    """convolution.py:167-240"""
    _check4(x, filter)
    hc.pad_mode(padding)
    B, C, H, W = x.shape
    h, w = filter.shape[-2:]
    ph, pw = h // 2, w // 2
    if padding == "circular":
        return _circular_conv_fft(x, filter, (H, W), shift_filter=True).contiguous()
    if padding == "valid":
        full = _circular_conv_fft(x, filter, (H + h - 1, W + w - 1), shift_filter=False)
        return full[:, :, h - 1:H, w - 1:W].contiguous()
    mode = "constant" if padding in ("zeros", "constant") else padding
    xp = F.pad(x, (pw, pw, ph, ph), mode=mode, value=0)
    out = _circular_conv_fft(xp, filter, xp.shape[-2:], shift_filter=True)
    return out[:, :, _crop(ph, 0), _crop(pw, 0)].contiguous()


def conv_transpose2d_fft(y, filter, real_fft=True, padding="valid"):
    """convolution.py:243-330"""
    _check4(y, filter)
    hc.pad_mode(padding)
    B, C, H, W = y.shape
    h, w = filter.shape[-2:]
    ph, pw, ih, iw = h // 2, w // 2, (h - 1) % 2, (w - 1) % 2
    if padding == "circular":
        return _circular_conv_fft(y, filter, (H, W), shift_filter=True, transpose=True).contiguous()
    if padding == "valid":
        yf = F.pad(y, (w - 1, w - 1, h - 1, h - 1))
        out = _circular_conv_fft(yf, filter, (H + h - 1, W + w - 1), transpose=True)
        return out[:, :, :H + h - 1, :W + w - 1].contiguous()
    yb = F.pad(y, (pw, pw, ph, ph))
    z = _circular_conv_fft(yb, filter, (H + 2 * ph, W + 2 * pw), shift_filter=True, transpose=True)
    z = z[..., ih:, iw:]
    return _fold_padding(z, "constant" if padding == "zeros" else padding, (ph, pw), (ih, iw)).contiguous()


def _crop(p, i):
    return slice(None) if p == 0 and i == 0 else slice(p - i, -p if p > 0 else None)


def _fold_padding(x, padding, p, i):
    """adjoint of the padding: fold the border of the full transposed convolution back
    (convolution.py:689-758), written axis by axis."""
    for axis, (pk, ik) in zip(range(-len(p), 0), zip(p, i)):
        n_full = x.shape[axis]
        lo, hi = pk - ik, pk           # border widths on each side
        n = n_full - lo - hi
        core = x.narrow(axis, lo, n).clone()
        if padding == "constant" or (lo == 0 and hi == 0):
            x = core
            continue
        left = x.narrow(axis, 0, lo) if lo > 0 else None
        right = x.narrow(axis, n_full - hi, hi) if hi > 0 else None
        if padding == "circular":
            if lo > 0:
                core.narrow(axis, n - lo, lo).add_(left)
            if hi > 0:
                core.narrow(axis, 0, hi).add_(right)
        elif padding == "reflect":
            if lo > 0:
                core.narrow(axis, 1, lo).add_(left.flip(dims=(axis,)))
            if hi > 0:
                core.narrow(axis, n - 1 - hi, hi).add_(right.flip(dims=(axis,)))
        elif padding == "replicate":
            if lo > 0:
                core.narrow(axis, 0, 1).add_(left.sum(dim=axis, keepdim=True))
            if hi > 0:
                core.narrow(axis, n - 1, 1).add_(right.sum(dim=axis, keepdim=True))
        else:
            raise ValueError(f"padding = '{padding}' not implemented.")
        x = core
    return x


# --------------------------------------------------------------------------- volumes (convolution.py:333-640)
def _check5(x, filter):
    if x.dim() != filter.dim() or filter.dim() != 5:
        raise ValueError("Input and filter must be 5D tensors")


def conv3d(x, filter, padding="valid", correlation=False):
    """True 3-D convolution with the 2-D padding conventions on every axis (convolution.py:333-393)."""
    _check5(x, filter)
    if correlation:
        filter = filter.flip(dims=(-3, -2, -1))
    return hc.conv3d(x, filter, padding)


def conv_transpose3d(y, filter, padding="valid", correlation=False):
    """Exact transpose of :func:`conv3d` (convolution.py:396-452, 689-758)."""
    _check5(y, filter)
    if correlation:
        filter = filter.flip(dims=(-3, -2, -1))
    size = tuple(y.shape[2:]) if hc.pad_mode(padding) != 0 else tuple(n + f - 1 for n, f in zip(y.shape[2:], filter.shape[2:]))
    return hc.conv3d_transpose(y, filter, padding, size)


def _rfft3(x):
    """rfftn over the last three dims (backward norm): the 2-D real transform of every slice, then a complex pass along depth"""
    return hfft.fftn(hc.rfft2(x, norm="backward"), dim=(-3,), norm="backward")


def _irfft3(xc, s):
    return hc.irfft2(hfft.ifftn(xc, dim=(-3,), norm="backward").contiguous(), s[-2:], norm="backward")


def _circular_conv_fft3(x, filter, s, shift_filter=False, transpose=False):
    """convolution.py:837-865 with dims = (-3, -2, -1)"""
    D, H, W = x.shape[-3:]
    if (D, H, W) != tuple(s):
        x = F.pad(x, (0, s[2] - W, 0, s[1] - H, 0, s[0] - D))
    fd, fh, fw = filter.shape[-3:]
    k = F.pad(filter, (0, s[2] - fw, 0, s[1] - fh, 0, s[0] - fd))
    if shift_filter:    # centre of the filter to index 0 on every axis (filter_fft, convolution.py:790-812)
        k = torch.roll(k, shifts=(-(fd // 2), -(fh // 2), -(fw // 2)), dims=(-3, -2, -1))
    ff = _rfft3(k)
    prod = _rfft3(x) * (torch.conj(ff) if transpose else ff)
    return _irfft3(prod, s)


def conv3d_fft(x, filter, real_fft=True, padding="valid"):
    """convolution.py:455-541"""
    _check5(x, filter)
    hc.pad_mode(padding)
    D, H, W = x.shape[-3:]
    d, h, w = filter.shape[-3:]
    pd, ph, pw = d // 2, h // 2, w // 2
    if padding == "circular":
        return _circular_conv_fft3(x, filter, (D, H, W), shift_filter=True).contiguous()
    if padding == "valid":
        full = _circular_conv_fft3(x, filter, (D + d - 1, H + h - 1, W + w - 1))
        return full[:, :, d - 1:D, h - 1:H, w - 1:W].contiguous()
    mode = "constant" if padding in ("zeros", "constant") else padding
    xp = F.pad(x, (pw, pw, ph, ph, pd, pd), mode=mode, value=0)
    out = _circular_conv_fft3(xp, filter, xp.shape[-3:], shift_filter=True)
    return out[:, :, _crop(pd, 0), _crop(ph, 0), _crop(pw, 0)].contiguous()


def conv_transpose3d_fft(y, filter, real_fft=True, padding="valid"):
    """convolution.py:544-640"""
    _check5(y, filter)
    hc.pad_mode(padding)
    D, H, W = y.shape[-3:]
    d, h, w = filter.shape[-3:]
    pd, ph, pw = d // 2, h // 2, w // 2
    idp, ih, iw = (d - 1) % 2, (h - 1) % 2, (w - 1) % 2
    if padding == "circular":
        return _circular_conv_fft3(y, filter, (D, H, W), shift_filter=True, transpose=True).contiguous()
    if padding == "valid":
        yf = F.pad(y, (w - 1, w - 1, h - 1, h - 1, d - 1, d - 1))
        out = _circular_conv_fft3(yf, filter, (D + d - 1, H + h - 1, W + w - 1), transpose=True)
        return out[:, :, :D + d - 1, :H + h - 1, :W + w - 1].contiguous()
    yb = F.pad(y, (pw, pw, ph, ph, pd, pd))
    z = _circular_conv_fft3(yb, filter, (D + 2 * pd, H + 2 * ph, W + 2 * pw), shift_filter=True, transpose=True)
    z = z[..., idp:, ih:, iw:]
    return _fold_padding(z, "constant" if padding == "zeros" else padding, (pd, ph, pw), (idp, ih, iw)).contiguous()


# --------------------------------------------------------------------------- filters (host side)
def gaussian_blur(psf_size=None, sigma=(1.0, 1.0), angle=0.0, device="cpu", dtype=torch.float32):
    """2-D anisotropic Gaussian PSF, shape (1,1,h,w) (functional/blur.py:136-269, 2-D case)."""
    if isinstance(sigma, (int, float)):
        sigma = (float(sigma), float(sigma))
    sigma = torch.as_tensor(sigma, dtype=dtype, device=device).flatten()
    if sigma.numel() != 2:
        raise ValueError("only 2-D Gaussian kernels are on the accelerated path")
    if psf_size is None:
        c = int(float(sigma.max()) / 0.3 + 1)
        psf_size = (2 * c + 1, 2 * c + 1)
    axes = [torch.linspace(-((n - 1) / 2), (n - 1) / 2, n, device=device, dtype=dtype) for n in psf_size]
    yy, xx = torch.meshgrid(*axes, indexing="ij")
    coords = torch.stack([xx, yy], dim=-1)  # (x, y) order as the reference (mesh[::-1])
    sig = torch.flip(sigma, dims=(0,))
    th = torch.as_tensor(float(angle) if not isinstance(angle, torch.Tensor) else angle, dtype=dtype, device=device)
    th = torch.deg2rad(th)  # _resolve_angle (functional/blur.py:118-133)
    ct, st = torch.cos(th), torch.sin(th)
    rot = torch.stack([ct, -st, st, ct]).reshape(2, 2)
    coords = torch.einsum("ij,...j->...i", rot, coords)
    k = torch.ones(psf_size, device=device, dtype=dtype)
    for d in range(2):
        k = k * torch.exp(-0.5 * coords[..., d] ** 2 / sig[d] ** 2) / (sqrt(2 * pi) * sig[d])
    k = k / k.sum()
    return k[None, None]


def kaiser_window(beta, length, device="cpu"):
    if beta < 0:
        raise ValueError("beta must be greater than 0")
    if length < 1:
        raise ValueError("length must be greater than 0")
    if length == 1:
        return torch.tensor([1.0])
    half = (length - 1) / 2
    n = torch.arange(length, device=device)
    beta = torch.tensor(beta, device=device)
    return torch.i0(beta * torch.sqrt(1 - ((n - half) / half) ** 2)) / torch.i0(beta)


def sinc_filter(factor=2, length=11, windowed=True, device="cpu"):
    if isinstance(factor, torch.Tensor):
        factor = factor.cpu().item()
    deltaf = 2 * (2 - 1.4142136) / factor
    n = torch.arange(length, device=device) - (length - 1) / 2
    f = torch.sinc(n / factor)
    if windowed:
        A = 2.285 * (length - 1) * 3.14159 * deltaf + 7.95
        beta = 0 if A <= 21 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21) if A <= 50 else 0.1102 * (A - 8.7))
        f = f * kaiser_window(beta, length, device=device)
    f = f.unsqueeze(0)
    f = f * f.T
    f = f[None, None]
    return f / f.sum()


def bilinear_filter(factor=2, device="cpu"):
    if isinstance(factor, torch.Tensor):
        factor = factor.cpu().item()
    x = torch.arange(start=-factor + 0.5, end=factor, step=1, device=device) / factor
    w = 1 - x.abs()
    w = torch.outer(w, w)
    return (w / w.sum())[None, None]


def bicubic_filter(factor=2, device="cpu"):
    if isinstance(factor, torch.Tensor):
        factor = factor.cpu().item()
    x = (torch.arange(start=-2 * factor + 0.5, end=2 * factor, step=1, device=device) / factor).abs()
    a = -0.5
    w = ((a + 2) * x.pow(3) - (a + 3) * x.pow(2) + 1) * (x <= 1)
    w = w + (a * x.pow(3) - 5 * a * x.pow(2) + 8 * a * x - 4 * a) * (x > 1) * (x < 2)
    w = torch.outer(w, w)
    return (w / w.sum())[None, None]
