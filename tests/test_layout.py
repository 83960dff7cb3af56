"""Repository contract checks: the product never imports the oracle or the reference."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(top):
    for d, _, fs in os.walk(os.path.join(ROOT, top)):
        for f in fs:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_product_does_not_import_oracle_or_reference():
    bad = []
    pat = re.compile(r"^\s*(from|import)\s+(oracle|deepinv)(\.|\s|$)", re.M)
    for f in _py_files("deepinv_amd"):
        if pat.search(open(f).read()):
            bad.append(os.path.relpath(f, ROOT))
    assert not bad, f"product files importing the oracle / the reference: {bad}"


def test_no_reference_path_in_gpu_side_code():
    """nothing that runs on the GPU box may read /root/reference"""
    for f in list(_py_files("deepinv_amd")) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]:
        assert "/root/reference" not in open(f).read(), f


def test_required_top_level_files():
    for name in ("bench.py", "__graft_entry__.py", "DESIGN.md", "INTEGRATION.md", "include/deepinv_amd.h",
                 "oracle/__init__.py", "tests/golden/make_golden.py"):
        assert os.path.exists(os.path.join(ROOT, name)), name

