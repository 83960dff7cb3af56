"""TEST INFRASTRUCTURE: run the PRODUCT's Python layer (deepinv_amd.physics / optim / ...) on CPU tensors by pointing its
ctypes binding at tests/emu/libdeepinv_amd_emu.so - the product's kernel SOURCES compiled for the host (tests/emu).

    with emu_backend():
        physics = deepinv_amd.physics.MultiCoilMRI(..., device="cpu")
        y = physics.A(x)            # csrc/mri.hip executed by the fiber emulation

The product itself has no CPU path (hip.require_hip raises for CPU tensors and hip.lib() loads only the gfx950 library); this
hook exists so that the GPU-less build container can drive the real deepinv optimizers over the product's operator objects
(tests/test_dropin_reference.py).  Small problems only: a kernel launch costs seconds."""
import contextlib
import ctypes
import importlib

import torch

import emu_lib


@contextlib.contextmanager
def emu_backend():
    import deepinv_amd.hip as H

    mods = [H] + [importlib.import_module("deepinv_amd.hip." + m) for m in ("mri", "fft", "radon", "conv", "elementwise", "drunet", "random")]
    emu_lib.lib()                                    # builds the emulation library if needed
    emu = H._DeviceGuardedLib(ctypes.CDLL(emu_lib.LIB))
    H._declare(emu)

    def require(*tensors):
        dev = None
        for t in tensors:
            if t is None:
                continue
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise H.HipExtensionError(f"operands on different devices: {dev} vs {t.device}")
        return dev

    patches = {"lib": lambda: emu, "require_hip": require, "stream_ptr": lambda device: ctypes.c_void_p(0)}
    saved = []
    for m in mods:
        for name, fn in patches.items():
            if hasattr(m, name):
                saved.append((m, name, getattr(m, name)))
                setattr(m, name, fn)
    saved.append((H, "_lib", H._lib))
    H._lib = emu
    EW = importlib.import_module("deepinv_amd.hip.elementwise")

    def eligible_on_host(*tensors):   # the fast-path predicate of the loop algebra with "on the HIP device" read as "on the host"
        for t in tensors:
            if t is None:
                continue
            if not (isinstance(t, torch.Tensor) and not t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                return False
            if (torch.is_grad_enabled() and t.requires_grad) or t.data_ptr() % 16:
                return False
        return True

    saved.append((EW, "eligible", EW.eligible))
    EW.eligible = eligible_on_host
    declared = [(m, getattr(m, "_declared")) for m in mods if hasattr(m, "_declared")]
    for m, _ in declared:
        m._declared = False                          # optional symbol groups are declared per library
    plan_cache, plan_host = dict(H._plan_cache), dict(H._plan_host)
    H._plan_cache.clear()
    cur = torch.cuda.current_device
    torch.cuda.current_device = lambda: 0
    try:
        yield emu
    finally:
        torch.cuda.current_device = cur
        for m, name, fn in saved:
            setattr(m, name, fn)
        for m, v in declared:
            m._declared = False
        H._plan_cache.clear()
        H._plan_cache.update(plan_cache)
        H._plan_host.clear()
        H._plan_host.update(plan_host)
