"""bench_ops rows (MRI cfg2, Radon cfg3) with an experimental build of the library: python run_variant.py <name|main>"""
import os, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(os.path.dirname(here)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "scripts"))
from deepinv_amd import hip
name = sys.argv[1]
if name != "main":
    hip.LIB_PATH = os.path.join(here, f"libdeepinv_amd_{name}.so")
import bench_ops
print('{"variant": "%s"}' % name)
bench_ops.bench_mri(32, 8, (320, 320), False)
bench_ops.bench_radon(8, 512, 720)
if len(sys.argv) > 2:
    bench_ops.bench_mri(2, 12, (16, 256, 256), True)
