"""Import shim that makes the *real* reference (``/root/reference``, deepinv v0.4.1)
importable inside the build container.

TEST INFRASTRUCTURE ONLY.  Nothing under ``deepinv_amd/`` may import this module.
It exists so that
``tests/golden/make_golden*.py`` can generate golden input/output vectors from the
reference itself (the vectors are committed; the reference cannot travel to the GPU
box); ``tests/test_oracle_golden.py`` pins the oracle restatement against those vectors.

The reference hard-imports ``torchvision``, ``torchmetrics``, ``h5py`` and ``natsort``
(deepinv/__init__.py:3, utils/mixins.py:8, ...), none of which is installed here, and
reads its own package metadata (deepinv/__about__.py:3).  We stub exactly those.
"""
from __future__ import annotations

import importlib.machinery as _mach
import importlib.metadata as _md
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("DEEPINV_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "deepinv"))


class _Any:
    """Permissive placeholder for anything imported from the stub modules."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, n):
        return _Any()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _Stub(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return sys.modules.get(self.__name__ + "." + n, _Any())


_STUBS = [
    "torchvision", "torchvision.utils", "torchvision.transforms",
    "torchvision.transforms.functional", "torchvision.transforms.v2",
    "torchvision.datasets", "torchvision.datasets.folder", "torchvision.datasets.utils",
    "torchmetrics", "torchmetrics.functional", "torchmetrics.image", "h5py", "natsort",
]

_installed = False


def install():
    """Install the stubs and put the reference on ``sys.path`` (idempotent)."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise ImportError(f"reference not found at {REFERENCE_ROOT}")
    real = _md.metadata

    class _M(dict):
        def get(self, k, d=None):
            return dict.get(self, k, d)

    def fake(name):
        if name == "deepinv":
            return _M({"Name": "deepinv", "Summary": "s", "Version": "0.4.1", "Author": "a",
                       "License": "BSD-3-Clause", "Project-URL": "u"})
        return real(name)

    _md.metadata = fake
    for name in _STUBS:
        if name in sys.modules:
            continue
        m = _Stub(name)
        m.__path__ = []
        m.__spec__ = _mach.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = m
    sys.modules["torchvision.datasets.folder"].IMG_EXTENSIONS = (".png",)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def import_reference():
    """Return the reference package (``import deepinv``)."""
    install()
    import deepinv  # noqa: WPS433

    return deepinv
