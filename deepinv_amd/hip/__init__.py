"""ctypes binding of ``libdeepinv_amd.so`` (the C-ABI declared in ``include/deepinv_amd.h``).

Only plumbing lives here: library loading, error translation, stream / pointer helpers.
There is deliberately **no CPU fallback**: every operator wrapper calls :func:`require_hip`
and raises if the tensor is not on a HIP device or the extension is missing.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libdeepinv_amd.so")

MAX_STAGES = 16


class FftPlan(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int32),
        ("nstages", ctypes.c_int32),
        ("generic", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("radix", ctypes.c_int32 * MAX_STAGES),
    ]


class MriDesc(ctypes.Structure):
    _fields_ = [
        ("batch", ctypes.c_int32),
        ("coils", ctypes.c_int32),
        ("ndim", ctypes.c_int32),
        ("dims", ctypes.c_int32 * 3),
        ("mask_batch", ctypes.c_int32),
        ("maps_batch", ctypes.c_int32),
        ("coil_dim", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("plan", FftPlan * 3),
        ("table", ctypes.c_void_p * 3),
    ]


_lib = None
_lock = threading.Lock()


class HipExtensionError(RuntimeError):
    pass


def _declare(lib):
    vp, i32, i64, f32, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
    lib.dinv_last_error.restype = ctypes.c_char_p
    lib.dinv_last_error.argtypes = []
    lib.dinv_version.restype = ctypes.c_int
    lib.dinv_device_count.argtypes = [ctypes.POINTER(ctypes.c_int)]
    lib.dinv_fft_table_bytes.restype = sz
    lib.dinv_fft_table_bytes.argtypes = [i32]
    lib.dinv_fft_plan_init.argtypes = [i32, ctypes.POINTER(FftPlan), vp]
    lib.dinv_fft_c2c_axis.argtypes = [vp, vp, i64, i64, ctypes.POINTER(FftPlan), vp, i32, i32, f32, vp]
    lib.dinv_mri_workspace_bytes.restype = sz
    lib.dinv_mri_workspace_bytes.argtypes = [ctypes.POINTER(MriDesc)]
    for name in ("dinv_mri_forward", "dinv_mri_adjoint"):
        getattr(lib, name).argtypes = [ctypes.POINTER(MriDesc), vp, vp, vp, vp, vp, sz, vp]
    # optional symbol groups are declared by the modules that own them (radon, conv, drunet)


_tls = threading.local()


class _DeviceGuardedLib:
    """The ctypes library with every call issued on the device of that call's operands (`ptr` / `stream_ptr` note it).
    A kernel launch goes to the CURRENT HIP device whatever stream it is given, so operands on cuda:1 while cuda:0 is
    current would fail (or use cuda:0's configuration caches); PyTorch ops guard against that, and so do these."""

    def __init__(self, cdll):
        object.__setattr__(self, "_cdll", cdll)
        object.__setattr__(self, "_wrapped", {})

    def __getattr__(self, name):
        w = self._wrapped.get(name)
        if w is None:
            fn = getattr(self._cdll, name)

            class _Fn:
                __slots__ = ()

                def __call__(_, *args):
                    # the device of THIS call's operands: recorded by ptr() / stream_ptr() while the arguments were
                    # built (every launching entry point takes at least a stream), dropped once the call is issued
                    dev = getattr(_tls, "dev", None)
                    _tls.dev = None
                    if dev is not None and dev.index is not None and dev.index != torch.cuda.current_device():
                        with torch.cuda.device(dev):
                            return fn(*args)
                    return fn(*args)

                def __setattr__(_, k, v):       # argtypes / restype declarations go to the real function
                    setattr(fn, k, v)

                def __getattr__(_, k):
                    return getattr(fn, k)

            w = self._wrapped[name] = _Fn()
        return w


def lib():
    """Load (once) and return the C-ABI library; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise HipExtensionError(
                        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or `make -C deepinv_amd/csrc`). deepinv_amd has no CPU fallback."
                    )
                l = _DeviceGuardedLib(ctypes.CDLL(LIB_PATH))
                _declare(l)
                _lib = l
    return _lib


def check(rc: int):
    if rc != 0:
        msg = lib().dinv_last_error()
        raise RuntimeError(f"libdeepinv_amd error {rc}: {msg.decode() if msg else '?'}")


def require_hip(*tensors: torch.Tensor):
    """Every operand must live on one HIP device (torch device type 'cuda' on ROCm)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HipExtensionError(
                "deepinv_amd operators run only on a HIP device (got a tensor on "
                f"'{t.device}'); there is no CPU fallback. Move inputs and physics with .to('cuda')."
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise HipExtensionError(f"operands on different devices: {dev} vs {t.device}")
    lib()
    _tls.dev = dev
    return dev


def ptr(t: torch.Tensor | None):
    if t is not None and t.is_cuda:
        _tls.dev = t.device
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def stream_ptr(device) -> ctypes.c_void_p:
    device = torch.device(device)
    if device.type == "cuda":
        _tls.dev = device if device.index is not None else torch.device("cuda", torch.cuda.current_device())
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def f32c(t: torch.Tensor) -> torch.Tensor:
    """contiguous fp32 view/copy (the reference calls .contiguous() itself, e.g. mixins.py:151)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# --------------------------------------------------------------------------- FFT plans
_plan_cache: dict = {}


def fft_plan(n: int, device) -> tuple[FftPlan, torch.Tensor]:
    """(plan struct, device table tensor) for length ``n`` on ``device`` (cached)."""
    device = torch.device(device)
    key = (int(n), device.type, device.index if device.index is not None else torch.cuda.current_device())
    hit = _plan_cache.get(key)
    if hit is not None:
        return hit
    l = lib()
    plan = FftPlan()
    nbytes = l.dinv_fft_table_bytes(int(n))
    host = torch.empty(nbytes, dtype=torch.uint8)
    check(l.dinv_fft_plan_init(int(n), ctypes.byref(plan), ctypes.c_void_p(host.data_ptr())))
    table = host.to(device)
    _plan_cache[key] = (plan, table)
    _plan_host[int(n)] = host
    return plan, table


_plan_host: dict = {}


def fft_plan_host_table(n: int) -> torch.Tensor:
    """the host copy of the table dinv_fft_plan_init wrote for length n (twiddles + digit-reversal positions)"""
    if int(n) not in _plan_host:
        l = lib()
        plan = FftPlan()
        host = torch.empty(l.dinv_fft_table_bytes(int(n)), dtype=torch.uint8)
        check(l.dinv_fft_plan_init(int(n), ctypes.byref(plan), ctypes.c_void_p(host.data_ptr())))
        _plan_host[int(n)] = host
    return _plan_host[int(n)]


# --------------------------------------------------------------------------- batch lanes
# (device index, n) -> the n HIP streams on which independent parts of a batch run concurrently, or None when concurrent lanes were
# measured NOT to pay on this device in this process (models/drunet.py: DRUNet._calibrated_lane_streams decides, once).  One set per
# process: HIP multiplexes the streams of a process onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default; this package asks
# for 8 at import), and two lanes whose streams share a queue run one after the other - 400 ms per step at 4 slices where one
# lane takes 327 and two overlapping lanes 302 (profiles/r06_lanes_hw_queues.txt).  Which streams share a queue depends on what
# else the process created before (RCCL, graph capture, user code), so the choice is made by timing the real launch sequence.
_LANE_STREAMS: dict = {}


def lane_key(device, n: int):
    device = torch.device(device)
    return (device.index if device.index is not None else torch.cuda.current_device(), int(n))


def split_batch(B: int, n: int):
    """n contiguous slabs (start, stop) of a batch of B units, sizes differing by at most one"""
    q, r = divmod(B, n)
    out, b0 = [], 0
    for i in range(n):
        b1 = b0 + q + (1 if i < r else 0)
        out.append((b0, b1))
        b0 = b1
    return out
