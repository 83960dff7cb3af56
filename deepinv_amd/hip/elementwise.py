"""ctypes wrappers of csrc/elementwise.hip: fused linear combinations, deterministic per-sample dot products and
the CG vector updates.  Used by the iteration drivers when their operands live on the HIP device and no autograd
graph is being recorded (otherwise the drivers use differentiable torch expressions)."""
from __future__ import annotations

import ctypes

import torch

from . import check, lib, ptr, stream_ptr

_declared = False


def _l():
    global _declared
    l = lib()
    if not _declared:
        vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
        l.dinv_lincomb.argtypes = [i64, f32, vp, f32, vp, f32, vp, vp, vp]
        l.dinv_affine.argtypes = [i64, f32, vp, f32, vp, f32, vp, f32, f32, f32, vp, vp]
        l.dinv_batched_dot_blocks.restype = i32
        l.dinv_batched_dot_blocks.argtypes = [i64]
        l.dinv_batched_dot.argtypes = [i32, i64, vp, vp, vp, vp, vp]
        l.dinv_cg_update.argtypes = [i32, i32, i64, vp, vp, f32, vp, vp, vp, vp, vp]
        l.dinv_cg_update_masked.argtypes = [i32, i32, i64, vp, vp, f32, vp, vp, vp, vp, vp, vp]
        l.dinv_cg_check.argtypes = [i32, vp, vp, vp, vp]
        l.dinv_cdiv_real.argtypes = [i64, i64, vp, vp, f32, vp, vp]
        l.dinv_mask_solve.argtypes = [i32, i64, i64, vp, vp, f32, vp, vp]
        _declared = True
    return l


def eligible(*tensors) -> bool:
    """fast path only for plain fp32 contiguous device tensors outside autograd recording"""
    for t in tensors:
        if t is None:
            continue
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            return False
        if torch.is_grad_enabled() and t.requires_grad:
            return False
        if t.data_ptr() % 16:
            return False
    return True


def lincomb(a: float, x, b: float = 0.0, y=None, c: float = 0.0, z=None, out=None):
    """a*x + b*y + c*z in one pass (into `out` when given: same shape, contiguous)"""
    if out is None:
        out = torch.empty_like(x)
    check(_l().dinv_lincomb(x.numel(), float(a), ptr(x), float(b), ptr(y), float(c), ptr(z), ptr(out), stream_ptr(x.device)))
    return out


def affine(a: float, x, b: float = 0.0, y=None, c: float = 0.0, z=None, d: float = 0.0, lo: float = -float("inf"),
           hi: float = float("inf"), out=None):
    """clamp(a*x + b*y + c*z + d, lo, hi) in one pass (into `out` when given: same shape, contiguous)"""
    if out is None:
        out = torch.empty_like(x)
    check(_l().dinv_affine(x.numel(), float(a), ptr(x), float(b), ptr(y), float(c), ptr(z), float(d), float(lo), float(hi),
                           ptr(out), stream_ptr(x.device)))
    return out


def cdiv_real(s, d, add: float = 0.0):
    """s / (d + add) for a complex64 tensor s whose trailing dimensions are those of the real fp32 tensor d up to d's LEADING singleton
    dimensions (d is shared by the leading batch / channel dimensions of s): one launch of dinv_cdiv_real, in place of a broadcast
    add and a complex division.  Only leading singletons are stripped (like mask_solve): a symbol [B, 1, h, w] against a spectrum
    [B, C, h, w] is NOT a trailing part and is rejected - the caller expands it."""
    tail = list(d.shape)
    while len(tail) > 1 and tail[0] == 1:
        tail = tail[1:]
    if not (s.dtype == torch.complex64 and s.is_contiguous() and d.dtype == torch.float32 and d.is_contiguous()
            and d.numel() > 0 and len(tail) <= s.dim() and list(s.shape[s.dim() - len(tail):]) == tail):
        raise ValueError(f"cdiv_real: spectrum {tuple(s.shape)} / {s.dtype} does not end in the symbol's shape {tuple(d.shape)} / {d.dtype}")
    out = torch.empty_like(s)
    check(_l().dinv_cdiv_real(s.numel(), d.numel(), ptr(torch.view_as_real(s)), ptr(d), float(add), ptr(torch.view_as_real(out)),
                              stream_ptr(s.device)))
    return out


def mask_solve(x, m, add: float = 0.0, dagger: bool = False):
    """x / (m*m + add) (dagger = False: DecomposablePhysics.prox_l2) or x * (m > 1e-5 ? 1/m : 0) (dagger = True: A_dagger) for a real
    mask m whose shape is the trailing part of x's (shared by x's leading dimensions): one launch of dinv_mask_solve"""
    tail = list(m.shape)
    while len(tail) > 1 and tail[0] == 1:
        tail = tail[1:]
    if not (m.dtype == torch.float32 and m.is_contiguous() and x.dtype == torch.float32 and x.is_contiguous() and len(tail) <= x.dim()
            and list(x.shape[x.dim() - len(tail):]) == tail):
        raise ValueError(f"mask_solve: the mask {tuple(m.shape)} is not the trailing part of {tuple(x.shape)}")
    out = torch.empty_like(x)
    check(_l().dinv_mask_solve(1 if dagger else 0, x.numel(), m.numel(), ptr(x), ptr(m), float(add), ptr(out), stream_ptr(x.device)))
    return out


def batched_dot(x, y):
    """<x[b], y[b]> per sample, shape [B] (deterministic summation order)"""
    B = x.shape[0]
    n = x.numel() // max(B, 1)
    out = torch.empty(B, device=x.device, dtype=torch.float32)
    part = torch.empty(B * _l().dinv_batched_dot_blocks(n), device=x.device, dtype=torch.float32)
    check(_l().dinv_batched_dot(B, n, ptr(x), ptr(y), ptr(out), ptr(part), stream_ptr(x.device)))
    return out


def cg_update_xr(num, den, eps, x, r, p, Ap, done=None):
    """x += s p ; r -= s Ap  with s_b = num_b / (den_b + eps), in place (skipped once the device flag `done` is set)"""
    B = x.shape[0]
    if done is None:
        check(_l().dinv_cg_update(0, B, x.numel() // B, ptr(num), ptr(den), float(eps), ptr(x), ptr(r), ptr(p), ptr(Ap),
                                  stream_ptr(x.device)))
    else:
        check(_l().dinv_cg_update_masked(0, B, x.numel() // B, ptr(num), ptr(den), float(eps), ptr(x), ptr(r), ptr(p),
                                         ptr(Ap), ptr(done), stream_ptr(x.device)))


def cg_update_p(num, den, eps, p, r, done=None):
    """p = r + s p  with s_b = num_b / (den_b + eps), in place (skipped once `done` is set)"""
    B = p.shape[0]
    if done is None:
        check(_l().dinv_cg_update(1, B, p.numel() // B, ptr(num), ptr(den), float(eps), ptr(p), None, ptr(r), None,
                                  stream_ptr(p.device)))
    else:
        check(_l().dinv_cg_update_masked(1, B, p.numel() // B, ptr(num), ptr(den), float(eps), ptr(p), None, ptr(r), None,
                                         ptr(done), stream_ptr(p.device)))


def cg_check(res, tol2, done):
    """done |= all(res < tol2), on the device (no host round trip)"""
    check(_l().dinv_cg_check(res.shape[0], ptr(res), ptr(tol2), ptr(done), stream_ptr(res.device)))
