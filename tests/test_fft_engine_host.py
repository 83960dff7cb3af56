"""The device FFT engine's butterflies / stages / permutation (deepinv_amd/csrc/fft_core.hpp) executed on
the host by tests/csrc/libhost_fft_emul.so (single emulated thread) and checked against numpy.fft in fp64.
This is how the kernel arithmetic is validated in the GPU-less build container."""
import ctypes
import os

import numpy as np
import pytest

LIB = os.path.join(os.path.dirname(__file__), "csrc", "libhost_fft_emul.so")


@pytest.fixture(scope="module")
def emul():
    if not os.path.exists(LIB):
        import __graft_entry__ as g

        g.build()
    lib = ctypes.CDLL(LIB)
    lib.emul_fft_lines.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_float]
    return lib


SIZES = [1, 2, 3, 4, 5, 6, 7, 8, 11, 16, 17, 19, 20, 30, 45, 49, 64, 77, 100, 121, 256, 320, 343, 725, 1024, 2048]


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("inverse", [0, 1])
@pytest.mark.parametrize("centered", [0, 1])
def test_engine_matches_numpy(emul, n, inverse, centered):
    rng = np.random.default_rng(n * 4 + inverse * 2 + centered)
    x = (rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))).astype(np.complex64)
    out = np.empty_like(x)
    rc = emul.emul_fft_lines(x.ctypes.data, out.ctypes.data, 3, n, inverse, centered, 1.0 / np.sqrt(n))
    assert rc == 0
    xx = x.astype(np.complex128)
    if centered:
        xx = np.fft.ifftshift(xx, axes=-1)
    ref = (np.fft.ifft if inverse else np.fft.fft)(xx, axis=-1, norm="ortho")
    if centered:
        ref = np.fft.fftshift(ref, axes=-1)
    assert np.abs(out - ref).max() / np.abs(ref).max() < 1e-6
