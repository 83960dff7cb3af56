import os
import sys

# (as deepinv_amd/__init__.py: the HIP runtime reads it when it initialises - `torch.cuda.is_available()` below comes first here)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    has_gpu = torch.cuda.is_available()
    skip_gpu = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(skip_gpu)


@pytest.fixture
def dev():
    return torch.device("cuda:0")


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """||a-b||_2 / ||b||_2 in fp64 on the CPU."""
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def dot_test(physics, x, y_like) -> float:
    """|<Ax,v> - <x,A^T v>| / (||Ax|| ||v||)  (BASELINE.md target <= 1e-5)."""
    Ax = physics.A(x)
    v = torch.randn_like(y_like)
    Atv = physics.A_adjoint(v)
    s1 = (Ax.double() * v.double()).sum()
    s2 = (x.double() * Atv.double()).sum()
    return float((s1 - s2).abs() / (Ax.double().norm() * v.double().norm()))
