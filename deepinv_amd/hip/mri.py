"""Fused MRI forward / adjoint (``dinv_mri_forward`` / ``dinv_mri_adjoint``).

Replaces the ATen sequences of ``MultiCoilMRI.A`` (mri.py:254-272), ``MultiCoilMRI.A_adjoint``
(mri.py:284-324), ``MRI`` via ``DecomposablePhysics`` (forward.py:1080-1117) and
``MRIMixin.im_to_kspace / kspace_to_im`` (mixins.py:182-206).
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import MriDesc, check, f32c, fft_plan, lib, ptr, require_hip, stream_ptr


# Test hook (dinv_mri_desc.reserved = 1): take the wave-autonomous 2-D pipelines (csrc/mri_wave.hpp) whenever the sizes allow,
# also below the batch size from which the library picks them for A^T / A^T A on its own.
FORCE_WAVE_PIPELINES = False


def _desc(batch, coils, vol, mask, maps, coil_dim, device):
    d = MriDesc()
    d.reserved = 1 if FORCE_WAVE_PIPELINES else 0
    d.batch, d.coils, d.ndim = int(batch), int(coils), len(vol)
    keep = []
    for i, n in enumerate(vol):
        d.dims[i] = int(n)
        plan, table = fft_plan(int(n), device)
        d.plan[i] = plan
        d.table[i] = table.data_ptr()
        keep.append(table)
    d.mask_batch = 0 if mask is None else int(mask.shape[0])
    d.maps_batch = 0 if maps is None else int(maps.shape[0])
    d.coil_dim = int(coil_dim)
    return d, keep


def _prep_mask(mask, vol, batch):
    if mask is None:
        return None
    mask = f32c(mask)
    if tuple(mask.shape[2:]) != tuple(vol) or mask.shape[1] != 2 or mask.shape[0] not in (1, batch):
        raise ValueError(f"mask of shape {tuple(mask.shape)} incompatible with data volume {tuple(vol)} / batch {batch}")
    return mask


def _prep_maps(maps, vol, batch):
    if maps is None:
        return None
    if not maps.is_complex():
        raise ValueError("coil_maps should be of torch complex dtype.")
    maps = maps.to(torch.complex64).contiguous()
    if tuple(maps.shape[2:]) != tuple(vol) or maps.shape[0] not in (1, batch):
        raise ValueError(f"coil_maps of shape {tuple(maps.shape)} incompatible with volume {tuple(vol)} / batch {batch}")
    return maps


def _forward_raw(x, maps, mask, coil_dim):
    dev = require_hip(x, maps, mask)
    x = f32c(x)
    if x.shape[1] != 2:
        raise ValueError("x must be of shape (B,2,...,H,W)")
    B, vol = x.shape[0], tuple(x.shape[2:])
    maps = _prep_maps(maps, vol, B)
    mask = _prep_mask(mask, vol, B)
    N = 1 if maps is None else maps.shape[1]
    d, keep = _desc(B, N, vol, mask, maps, coil_dim, dev)
    yshape = (B, 2, N, *vol) if coil_dim else (B, 2, *vol)
    y = torch.empty(yshape, device=dev, dtype=torch.float32)
    ws = torch.empty(lib().dinv_mri_workspace_bytes(ctypes.byref(d)), device=dev, dtype=torch.uint8)
    check(lib().dinv_mri_forward(ctypes.byref(d), ptr(x), ptr(None if maps is None else torch.view_as_real(maps)),
                                 ptr(mask), ptr(y), ptr(ws), ws.numel(), stream_ptr(dev)))
    return y


def _adjoint_raw(y, maps, mask, coil_dim):
    dev = require_hip(y, maps, mask)
    y = f32c(y)
    if y.shape[1] != 2:
        raise ValueError("y must be of shape (B,2,N,...,H,W)")
    B = y.shape[0]
    vol = tuple(y.shape[3:]) if coil_dim else tuple(y.shape[2:])
    maps = _prep_maps(maps, vol, B)
    mask = _prep_mask(mask, vol, B)
    N = y.shape[2] if coil_dim else 1
    if maps is not None and maps.shape[1] != N:
        raise ValueError(f"y has {N} coils but coil_maps has {maps.shape[1]}")
    d, keep = _desc(B, N, vol, mask, maps, coil_dim, dev)
    x = torch.empty((B, 2, *vol), device=dev, dtype=torch.float32)
    ws = torch.empty(lib().dinv_mri_workspace_bytes(ctypes.byref(d)), device=dev, dtype=torch.uint8)
    check(lib().dinv_mri_adjoint(ctypes.byref(d), ptr(y), ptr(None if maps is None else torch.view_as_real(maps)),
                                 ptr(mask), ptr(x), ptr(ws), ws.numel(), stream_ptr(dev)))
    return x


def _as_complex(t):   # [B,2,...] real pairs -> complex [B,...]
    return torch.complex(t[:, 0], t[:, 1])


def _per_coil_adjoint(y, mask, coil_dim):
    """u_n = F^H(mask * y_n) for every coil, without the coil combination: the coils ride along the batch axis of the
    single-coil kernel.  y [B,2,N,vol] (or [B,2,vol]) -> complex [B,N,vol]."""
    if not coil_dim:
        return _as_complex(_adjoint_raw(y, None, mask, False))[:, None]
    B, _, N = y.shape[:3]
    vol = tuple(y.shape[3:])
    yb = y.permute(0, 2, 1, *range(3, y.ndim)).reshape(B * N, 2, *vol)
    mb = mask
    if mask is not None and mask.shape[0] == B and B > 1:
        mb = mask.repeat_interleave(N, dim=0)
    return _as_complex(_adjoint_raw(yb, None, mb, False)).reshape(B, N, *vol)


def _reduce_like(g, ref):
    """sum a per-sample gradient over the batch when the parameter is shared by the batch (leading size 1)"""
    return g.sum(0, keepdim=True) if ref.shape[0] == 1 and g.shape[0] != 1 else g


def _mask_grad(prod, mask, coil_dim):   # prod = upstream * unmasked k-space, [B,2,N,vol] or [B,2,vol]
    if coil_dim:
        prod = prod.sum(2)
    return _reduce_like(prod, mask).to(mask.dtype)


class _MriForward(torch.autograd.Function):
    """y = M F S x ; backward is the adjoint kernel (same trick as ApplyRadon, radon.py:493-531).  Gradients with
    respect to `coil_maps` and `mask` (the reference gets them from autograd, e.g. for learned sampling patterns)
    are assembled from the same kernels: dL/dM = sum g * F(S x),  dL/dS_n = F^H(M g_n) * conj(x)."""

    # (forward / setup_context form: usable under torch.func transforms - the reference's adjoint_function is torch.func.vjp)
    @staticmethod
    def forward(x, maps, mask, coil_dim):
        return _forward_raw(x, maps, mask, coil_dim)

    @staticmethod
    def setup_context(ctx, inputs, output):
        x, maps, mask, ctx.coil_dim = inputs
        ctx.save_for_backward(x, maps, mask)

    @staticmethod
    def backward(ctx, g):
        x, maps, mask = ctx.saved_tensors
        gx = _MriAdjoint.apply(g, maps, mask, ctx.coil_dim) if ctx.needs_input_grad[0] else None
        gmaps = gmask = None
        if maps is not None and ctx.needs_input_grad[1]:
            u = _per_coil_adjoint(g, mask, ctx.coil_dim)
            gmaps = _reduce_like(u * _as_complex(x).conj()[:, None], maps).to(maps.dtype)
        if mask is not None and ctx.needs_input_grad[2]:
            gmask = _mask_grad(g * _forward_raw(x, maps, None, ctx.coil_dim), mask, ctx.coil_dim)
        return gx, gmaps, gmask, None


class _MriAdjoint(torch.autograd.Function):
    """x = sum_n conj(S_n) F^H(M y_n); dL/dM = sum F(S g) * y,  dL/dS_n = conj(g) * F^H(M y_n)."""

    @staticmethod
    def forward(y, maps, mask, coil_dim):
        return _adjoint_raw(y, maps, mask, coil_dim)

    @staticmethod
    def setup_context(ctx, inputs, output):
        y, maps, mask, ctx.coil_dim = inputs
        ctx.save_for_backward(y, maps, mask)

    @staticmethod
    def backward(ctx, g):
        y, maps, mask = ctx.saved_tensors
        gy = _MriForward.apply(g, maps, mask, ctx.coil_dim) if ctx.needs_input_grad[0] else None
        gmaps = gmask = None
        if maps is not None and ctx.needs_input_grad[1]:
            u = _per_coil_adjoint(y, mask, ctx.coil_dim)
            gmaps = _reduce_like(_as_complex(g).conj()[:, None] * u, maps).to(maps.dtype)
        if mask is not None and ctx.needs_input_grad[2]:
            gmask = _mask_grad(_forward_raw(g, maps, None, ctx.coil_dim) * y, mask, ctx.coil_dim)
        return gy, gmaps, gmask, None


def _normal_raw(x, maps, mask, coil_dim):
    """A^T A x through ``dinv_mri_normal`` (no k-space tensor); None when the sizes have no static plan."""
    dev = require_hip(x, maps, mask)
    x = f32c(x)
    if x.shape[1] != 2:
        raise ValueError("x must be of shape (B,2,...,H,W)")
    B, vol = x.shape[0], tuple(x.shape[2:])
    maps = _prep_maps(maps, vol, B)
    mask = _prep_mask(mask, vol, B)
    N = 1 if maps is None else maps.shape[1]
    d, keep = _desc(B, N, vol, mask, maps, coil_dim, dev)
    if not lib().dinv_mri_normal_supported(ctypes.byref(d)):
        return None
    out = torch.empty_like(x)
    ws = torch.empty(lib().dinv_mri_workspace_bytes(ctypes.byref(d)), device=dev, dtype=torch.uint8)
    check(lib().dinv_mri_normal(ctypes.byref(d), ptr(x), ptr(None if maps is None else torch.view_as_real(maps)),
                                ptr(mask), ptr(out), ptr(ws), ws.numel(), stream_ptr(dev)))
    return out


class _MriNormal(torch.autograd.Function):
    """A^T A is self-adjoint: the backward of the fused normal operator is the same kernel chain."""

    @staticmethod
    def forward(x, maps, mask, coil_dim):
        return _normal_raw(x, maps, mask, coil_dim)

    @staticmethod
    def setup_context(ctx, inputs, output):
        _, maps, mask, ctx.coil_dim = inputs
        ctx.save_for_backward(maps, mask)

    @staticmethod
    def backward(ctx, g):
        maps, mask = ctx.saved_tensors
        return _MriNormal.apply(g, maps, mask, ctx.coil_dim), None, None, None


ENABLE_FUSED_NORMAL = True    # test hook: False evaluates A^T A as adjoint(forward(x))


def mri_normal(x, coil_maps=None, mask=None, coil_dim=True):
    """``A^T A x = sum_n conj(S_n) F^H(mask^2 F(S_n x))`` in one kernel chain that never writes k-space.  Falls back to
    adjoint(forward(x)) for sizes without a static FFT plan, or when the mask / coil maps themselves need gradients."""
    needs_param_grad = torch.is_grad_enabled() and any(p is not None and p.requires_grad for p in (coil_maps, mask))
    if not needs_param_grad and ENABLE_FUSED_NORMAL:
        B, vol = x.shape[0], tuple(x.shape[2:])
        if all(_static_ok(n) for n in vol) and vol[-1] >= 64 and vol[0] >= 16:
            return _MriNormal.apply(x, coil_maps, mask, bool(coil_dim))
    return mri_adjoint(mri_forward(x, coil_maps, mask, coil_dim), coil_maps, mask, coil_dim)


def _static_ok(n: int) -> bool:   # sizes with a compile-time plan (csrc/fft_static.hpp: has_static_plan)
    return n in (16, 32, 64, 128, 256, 320, 512)


def mri_forward(x, coil_maps=None, mask=None, coil_dim=True):
    """``mask * F(coil_maps * x)``; x ``[B,2,vol]`` -> ``[B,2,N,vol]`` (or ``[B,2,vol]`` if not coil_dim)."""
    return _MriForward.apply(x, coil_maps, mask, bool(coil_dim))


def mri_adjoint(y, coil_maps=None, mask=None, coil_dim=True):
    """``sum_n conj(coil_maps_n) * F^H(mask * y_n)``."""
    return _MriAdjoint.apply(y, coil_maps, mask, bool(coil_dim))
