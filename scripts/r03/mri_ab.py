"""A/B of the MRI pipelines on one MI355X: variant 0 = fused first / last pass (round 3), 1 = round-2 three-pass forms"""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepinv_amd as dinv
from deepinv_amd import hip
from bench_ops import timeit
dev = torch.device("cuda:0")
lib = hip.lib()
def run(B, coils, img, three_d):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 2, *img, generator=g).to(dev)
    maps = (torch.randn(1, coils, *img, dtype=torch.complex64, generator=g) / coils ** 0.5).to(dev)
    mask = (torch.rand(*img, generator=g) > 0.75).float().to(dev)
    phys = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, *img), three_d=three_d, device=dev)
    res = {}
    outs = {}
    for v in (1, 0):
        lib.dinv_mri_debug_variant(v)
        y = phys.A(x); xa = phys.A_adjoint(y); xn = phys.A_adjoint_A(x)
        outs[v] = (y.clone(), xa.clone(), xn.clone())
        res[f"A_ms_v{v}"] = round(timeit(lambda: phys.A(x)) * 1e3, 4)
        res[f"AT_ms_v{v}"] = round(timeit(lambda: phys.A_adjoint(y)) * 1e3, 4)
        res[f"ATA_ms_v{v}"] = round(timeit(lambda: phys.A_adjoint_A(x)) * 1e3, 4)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    res.update(B=B, coils=coils, img=img, diff_A=rel(outs[0][0], outs[1][0]), diff_AT=rel(outs[0][1], outs[1][1]), diff_ATA=rel(outs[0][2], outs[1][2]))
    print(json.dumps(res), flush=True)
run(32, 8, (320, 320), False)
run(4, 8, (320, 320), False)
run(2, 12, (16, 256, 256), True)
run(8, 4, (256, 256), False)
run(3, 5, (64, 128), False)
