"""csrc/mri.hip (the fused MRI pipelines: coil expansion + column pass, rows pass with mask / planar conversion, the fused
normal operator, the 3-D depth pass) compiled for the HOST against the HIP emulation (tests/emu) and checked against fp64
centred orthonormal FFTs (deepinv/physics/mri.py:254-324, deepinv/utils/mixins.py:159-180) - CPU-only coverage of the kernel
sources, incl. the static plans (16 / 32 / 64 / 128) and generic lengths (odd, prime)."""
import ctypes

import numpy as np
import pytest
import torch

import emu_lib as E


class MriDesc(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int32), ("coils", ctypes.c_int32), ("ndim", ctypes.c_int32), ("dims", ctypes.c_int32 * 3),
                ("mask_batch", ctypes.c_int32), ("maps_batch", ctypes.c_int32), ("coil_dim", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("plan", E.FftPlan * 3), ("table", ctypes.c_void_p * 3)]


def _desc(B, N, vol, mask, maps, coil_dim=1):
    d = MriDesc()
    d.batch, d.coils, d.ndim = B, N, len(vol)
    keep = []
    for i, n in enumerate(vol):
        d.dims[i] = n
        plan, table = E.fft_plan(n)
        d.plan[i] = plan
        d.table[i] = table.ctypes.data
        keep.append(table)
    d.mask_batch = 0 if mask is None else mask.shape[0]
    d.maps_batch = 0 if maps is None else maps.shape[0]
    d.coil_dim = coil_dim
    d.reserved = 1      # reach the wave-autonomous 2-D pipelines (csrc/mri_wave.hpp) at these small batches too
    return d, keep


def _cfft(z, dims, inverse=False):
    f = torch.fft.ifftn if inverse else torch.fft.fftn
    return torch.fft.fftshift(f(torch.fft.ifftshift(z, dim=dims), dim=dims, norm="ortho"), dim=dims)


def _ref_forward(x, maps, mask):
    xc = torch.complex(x[:, 0], x[:, 1]).to(torch.complex128)            # [B, vol]
    dims = tuple(range(-(x.ndim - 2), 0))
    k = _cfft(maps.to(torch.complex128) * xc[:, None], dims)               # [B, N, vol]
    y = torch.stack([k.real, k.imag], 1)                                   # [B, 2, N, vol]
    return y * mask.double()[:, :, None]


def _ref_adjoint(y, maps, mask):
    dims = tuple(range(-(y.ndim - 3), 0))
    ym = y.double() * mask.double()[:, :, None]
    k = torch.complex(ym[:, 0], ym[:, 1])
    u = _cfft(k, dims, inverse=True)
    xc = (maps.to(torch.complex128).conj() * u).sum(1)
    return torch.stack([xc.real, xc.imag], 1)


def _run(fn, d, *ptrs):
    l = E.lib()
    l.dinv_mri_workspace_bytes.restype = ctypes.c_size_t
    ws = np.zeros(l.dinv_mri_workspace_bytes(ctypes.byref(d)), np.uint8)
    E.check(getattr(l, fn)(ctypes.byref(d), *ptrs, E.p(ws), ctypes.c_size_t(ws.size), None))


CASES = [((32, 64), 2, 3, 1), ((64, 32), 1, 2, 2), ((16, 32, 32), 1, 2, 1), ((17, 11), 2, 3, 1), ((128, 32), 1, 1, 1),
         ((24, 40), 1, 2, 1), ((320, 64), 1, 2, 1), ((64, 256), 1, 1, 1), ((32, 320), 2, 2, 2), ((32, 512), 1, 2, 1), ((256, 256), 1, 2, 1), ((320, 320), 2, 2, 2), ((320, 256), 1, 1, 1),
         ((256, 320), 1, 2, 1), ((512, 256), 1, 1, 1),
         ((16, 16, 256), 1, 2, 1)]


@pytest.mark.parametrize("vol,B,N,mask_b", CASES)
def test_mri_forward_adjoint_normal_emulated(vol, B, N, mask_b):
    gen = torch.Generator().manual_seed(sum(vol) + N)
    x = torch.randn(B, 2, *vol, generator=gen)
    maps = torch.complex(torch.randn(1, N, *vol, generator=gen), torch.randn(1, N, *vol, generator=gen)).to(torch.complex64)
    mb = B if mask_b == 2 else 1
    m = (torch.rand(mb, 1, *vol, generator=gen) > 0.4).float()
    mask = m.expand(mb, 2, *vol).contiguous()
    d, keep = _desc(B, N, vol, mask, maps)
    maps_r = torch.view_as_real(maps).contiguous()
    # forward
    y = torch.full((B, 2, N, *vol), float("nan"))
    _run("dinv_mri_forward", d, E.p(x), E.p(maps_r), E.p(mask), E.p(y))
    ref = _ref_forward(x, maps, mask)
    assert float((y.double() - ref).norm() / ref.norm()) < 2e-6
    assert torch.equal(y == 0, (mask[:, :, None].expand_as(y) == 0) | (ref.float() == 0))      # masked samples are exact zeros
    # adjoint
    v = torch.randn(B, 2, N, *vol, generator=gen)
    xa = torch.full((B, 2, *vol), float("nan"))
    _run("dinv_mri_adjoint", d, E.p(v), E.p(maps_r), E.p(mask), E.p(xa))
    refa = _ref_adjoint(v, maps, mask)
    assert float((xa.double() - refa).norm() / refa.norm()) < 2e-6
    # <A x, v> = <x, A^T v>
    lhs, rhs = float((y.double() * v.double()).sum()), float((x.double() * xa.double()).sum())
    assert abs(lhs - rhs) < 1e-5 * max(abs(lhs), 1.0)
    # fused normal operator where the library offers it, else it must say so
    l = E.lib()
    if l.dinv_mri_normal_supported(ctypes.byref(d)):
        xn = torch.full((B, 2, *vol), float("nan"))
        _run("dinv_mri_normal", d, E.p(x), E.p(maps_r), E.p(mask), E.p(xn))
        refn = _ref_adjoint(ref.float(), maps, mask)
        assert float((xn.double() - refn).norm() / refn.norm()) < 3e-6


def test_mri_single_coil_without_maps_emulated():
    """MRI (no coil maps, no coil dimension): y = M F x, x = F^H M y"""
    vol, B = (32, 32), 2
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, 2, *vol, generator=gen)
    mask = (torch.rand(1, 1, *vol, generator=gen) > 0.5).float().expand(1, 2, *vol).contiguous()
    d, keep = _desc(B, 1, vol, mask, None, coil_dim=0)
    y = torch.full((B, 2, *vol), float("nan"))
    _run("dinv_mri_forward", d, E.p(x), None, E.p(mask), E.p(y))
    xc = torch.complex(x[:, 0], x[:, 1]).to(torch.complex128)
    k = _cfft(xc, (-2, -1))
    ref = torch.stack([k.real, k.imag], 1) * mask.double()
    assert float((y.double() - ref).norm() / ref.norm()) < 2e-6
    xa = torch.full((B, 2, *vol), float("nan"))
    _run("dinv_mri_adjoint", d, E.p(y), None, E.p(mask), E.p(xa))
    ym = ref * mask.double()
    u = _cfft(torch.complex(ym[:, 0], ym[:, 1]), (-2, -1), inverse=True)
    refa = torch.stack([u.real, u.imag], 1)
    assert float((xa.double() - refa).norm() / refa.norm()) < 2e-6
