"""ctypes wrappers of csrc/random.hip: fused additive Gaussian noise and the Cartesian MRI mask-line generator."""
from __future__ import annotations

import ctypes

import torch

from . import check, lib, ptr, require_hip, stream_ptr

_declared = False


def _l():
    global _declared
    l = lib()
    if not _declared:
        vp, i32, i64, u64, f32, f64 = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float,
                                       ctypes.c_double)
        l.dinv_gaussian_noise.argtypes = [i64, i64, vp, vp, f32, u64, u64, vp, vp]
        l.dinv_mri_mask_lines.argtypes = [i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, f64, i32, u64, u64, vp, vp]
        _declared = True
    return l


def philox_state(gen: torch.Generator | None, device, n_blocks: int):
    """(seed, offset) for a kernel that consumes `n_blocks` Philox counters, taken from - and advanced on - the torch
    generator that the reference would have drawn from (the given one, else the device's default generator), so that
    torch.manual_seed / Generator.manual_seed keep their meaning: same seed -> same numbers, successive calls differ."""
    if gen is None:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        gen = torch.cuda.default_generators[idx]
    seed, off = gen.initial_seed(), gen.get_offset()
    gen.set_offset(off + 4 * ((int(n_blocks) + 3) // 4))     # torch keeps Philox offsets in multiples of 4
    return seed & 0xFFFFFFFFFFFFFFFF, off


def gaussian_noise(x: torch.Tensor, sigma, gen: torch.Generator | None = None) -> torch.Tensor:
    """x + sigma * N(0, I) in one pass (sigma: float, 0-dim tensor, or one value per batch sample)"""
    dev = require_hip(x)
    xc = x.contiguous().float()
    y = torch.empty_like(xc)
    n = xc.numel()
    per, sig_t, sig_f = n, None, 0.0
    if isinstance(sigma, torch.Tensor) and sigma.numel() > 1:
        if sigma.numel() != xc.shape[0]:
            raise ValueError(f"sigma has {sigma.numel()} entries for a batch of {xc.shape[0]}")
        sig_t = sigma.reshape(-1).to(dev, torch.float32).contiguous()
        per = n // xc.shape[0]
    elif isinstance(sigma, torch.Tensor):
        if sigma.is_cuda:            # keep a device scalar on the device: one-entry per-"sample" table over the whole tensor
            sig_t = sigma.reshape(1).to(dev, torch.float32)
        else:
            sig_f = float(sigma)
    else:
        sig_f = float(sigma)
    seed, off = philox_state(gen, dev, (n + 3) // 4)
    check(_l().dinv_gaussian_noise(n, max(per, 1), ptr(xc), ptr(sig_t), sig_f, seed, off, ptr(y), stream_ptr(dev)))
    return y


def mri_mask_lines(batch, channels, times, height, width, n_lines, center, mode, pdf, accel, n_offsets, device,
                   gen: torch.Generator | None = None) -> torch.Tensor:
    """mask [batch, channels, times, height, width] of sampled k-space columns (csrc/random.hip)"""
    device = torch.device(device)
    mask = torch.empty((batch, channels, times, height, width), device=device, dtype=torch.float32)
    require_hip(mask)
    seed, off = philox_state(gen, device, batch * times * 1024 + batch)
    check(_l().dinv_mri_mask_lines(batch, channels, times, height, width, int(n_lines), int(center[0]), int(center[1]), int(mode),
                                   ptr(pdf), float(accel), int(n_offsets), seed, off, ptr(mask), stream_ptr(device)))
    return mask
