"""Centred / plain complex FFTs on HIP (replaces ``MRIMixin.fft/ifft``, mixins.py:159-180)."""
from __future__ import annotations

import ctypes
import math

import torch

from . import check, fft_plan, lib, ptr, require_hip, stream_ptr


def _c2c_axis(buf: torch.Tensor, axis: int, inverse: bool, centered: bool, scale: float):
    """In-place transform of a contiguous complex64 tensor along ``axis``."""
    shape = buf.shape
    n = shape[axis]
    outer = int(math.prod(shape[:axis]))
    inner = int(math.prod(shape[axis + 1:]))
    plan, table = fft_plan(n, buf.device)
    check(lib().dinv_fft_c2c_axis(ptr(buf), ptr(buf), outer, inner, ctypes.byref(plan), ptr(table),
                                  int(inverse), int(centered), float(scale), stream_ptr(buf.device)))


class _FftN(torch.autograd.Function):
    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.cfg = tuple(inputs[1:])

    @staticmethod
    def forward(x, dims, inverse, centered, norm):
        require_hip(x)
        out = x.to(torch.complex64).contiguous().clone()
        view = torch.view_as_real(out)  # same storage; kernels see interleaved fp32 pairs
        for d in dims:
            d = d % out.ndim
            n = out.shape[d]
            if norm == "ortho":
                s = 1.0 / math.sqrt(n)
            elif norm == "backward":
                s = 1.0 / n if inverse else 1.0
            elif norm == "forward":
                s = 1.0 if inverse else 1.0 / n
            else:
                raise ValueError(f"unknown norm {norm}")
            _c2c_axis(out, d, inverse, centered, s)
        del view
        return out

    @staticmethod
    def backward(ctx, g):
        dims, inverse, centered, norm = ctx.cfg
        # adjoint of a (scaled) DFT is the conjugate transform with the same scaling
        flip = {"ortho": "ortho", "backward": "forward", "forward": "backward"}[norm]
        return _FftN.apply(g, dims, not inverse, centered, flip), None, None, None, None


def fftn(x, dim=(-2, -1), norm="ortho", centered=False):
    return _FftN.apply(x, tuple(dim), False, centered, norm)


def ifftn(x, dim=(-2, -1), norm="ortho", centered=False):
    return _FftN.apply(x, tuple(dim), True, centered, norm)
