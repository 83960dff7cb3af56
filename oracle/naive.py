"""fp64 numpy restatements straight from the defining sums.

TEST INFRASTRUCTURE.  O(N^2): use only at the small sizes of the reference's own unit tests.
"""
from __future__ import annotations

import numpy as np


def centered_dft_matrix(n: int, inverse: bool = False) -> np.ndarray:
    """A.1: X[k] = n^-1/2 sum_m x[m] exp(-+2 pi i (m-c)(k-c)/n), c = n//2 (odd and even n);
    equals ifftshift -> (i)fft(norm='ortho') -> fftshift (deepinv/utils/mixins.py:159-180)."""
    c = n // 2
    idx = np.arange(n) - c
    sign = 1.0 if inverse else -1.0
    return np.exp(sign * 2j * np.pi * np.outer(idx, idx) / n) / np.sqrt(n)


def centered_dftn(x: np.ndarray, ndim: int, inverse: bool = False) -> np.ndarray:
    """Separable centred DFT over the last `ndim` axes of a complex array."""
    x = x.astype(np.complex128)
    for ax in range(-ndim, 0):
        F = centered_dft_matrix(x.shape[ax], inverse)
        x = np.moveaxis(np.tensordot(F, np.moveaxis(x, ax, 0), axes=(1, 0)), 0, ax)
    return x


def _cplx(x):  # [B,2,...] -> complex
    return x[:, 0].astype(np.float64) + 1j * x[:, 1].astype(np.float64)


def _planar(z):  # complex [B,...] -> [B,2,...]
    return np.stack([z.real, z.imag], axis=1)


def multicoil_A(x, maps, mask, ndim=2):
    """A.2: y[b,:,n] = M . F_c(S[n] . x_c[b]) ; x [B,2,vol], maps [1|B,N,vol] c, mask [1|B,2,vol]"""
    xc = _cplx(x)[:, None] * maps.astype(np.complex128)
    k = centered_dftn(xc, ndim)
    y = np.stack([k.real, k.imag], axis=1)  # [B,2,N,vol]
    return mask[:, :, None].astype(np.float64) * y


def multicoil_AT(y, maps, mask, ndim=2):
    """A.2: x[b] = sum_n conj(S[n]) . F_c^-1(M . y_c[b,n])"""
    my = mask[:, :, None].astype(np.float64) * y.astype(np.float64)
    kc = my[:, 0] + 1j * my[:, 1]
    im = centered_dftn(kc, ndim, inverse=True)
    return _planar(np.sum(np.conj(maps.astype(np.complex128)) * im, axis=1))


def radon_forward(x, angles_deg, circle=False):
    """A.5 as explicit fp64 loops; x [W,W] -> [G,A].  Coordinates follow affine_grid/grid_sample with
    align_corners=True (deepinv/physics/functional/radon.py:7-10, 252-342)."""
    W = x.shape[0]
    if circle:
        ax = 2 * np.arange(W) / (W - 1) - 1.0
        x = x * ((ax[:, None] ** 2 + ax[None, :] ** 2) <= 1)
        G, pb = W, 0
    else:
        G = int(np.ceil(np.float32(np.sqrt(np.float32(2))) * np.float32(W)))
        pad = int(np.ceil(np.float32(np.sqrt(np.float32(2))) * np.float32(W) - W))
        pb = (W + pad) // 2 - W // 2
    xp = np.zeros((G, G))
    xp[pb:pb + W, pb:pb + W] = x
    c = (G - 1) / 2.0
    out = np.zeros((G, len(angles_deg)))
    for a, deg in enumerate(angles_deg):
        th = np.deg2rad(deg)
        ct, st = np.cos(th), np.sin(th)
        for j in range(G):
            for i in range(G):
                px = ct * (j - c) + st * (i - c) + c
                py = -st * (j - c) + ct * (i - c) + c
                x0, y0 = int(np.floor(px)), int(np.floor(py))
                tx, ty = px - x0, py - y0
                for (yy, xx, w) in ((y0, x0, (1 - tx) * (1 - ty)), (y0, x0 + 1, tx * (1 - ty)),
                                    (y0 + 1, x0, (1 - tx) * ty), (y0 + 1, x0 + 1, tx * ty)):
                    if 0 <= yy < G and 0 <= xx < G:
                        out[j, a] += w * xp[yy, xx]
    return out


def resize_bilinear_antialias(x: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """What torchvision.transforms.Resize does to a (..., H, W) tensor since 0.17 (the call MRIMixin.crop(rescale=True) makes,
    deepinv/utils/mixins.py:236-240): separable triangle-filter resampling with half-pixel centres; the filter support is
    1 output pixel when enlarging and `scale` input pixels when shrinking (antialiasing), weights normalised per output sample.
    fp64, straight from the defining sums."""
    def matrix(n_in, n_out):
        scale = n_in / n_out
        support = max(scale, 1.0)
        M = np.zeros((n_out, n_in))
        for o in range(n_out):
            centre = (o + 0.5) * scale
            lo, hi = max(int(centre - support + 0.5), 0), min(int(centre + support + 0.5), n_in)
            for i in range(lo, hi):
                M[o, i] = max(0.0, 1.0 - abs((i + 0.5 - centre) / support))
            M[o] /= M[o].sum()
        return M

    x = np.asarray(x, dtype=np.float64)
    return np.einsum("oh,...hw,pw->...op", matrix(x.shape[-2], out_h), x, matrix(x.shape[-1], out_w))
