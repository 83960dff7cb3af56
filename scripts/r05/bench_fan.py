"""Tomography rows at cfg3's geometry (8 images 512x512, 720 angles): parallel beam and fan beam, A / A_adjoint, ms per call"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deepinv_amd as dinv

dev = torch.device("cuda:0")
B, W, A = 8, 512, 720
x = torch.rand(B, 1, W, W, generator=torch.Generator().manual_seed(0)).to(dev)

def t(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 4)

for fan in (False, True):
    p = dinv.physics.Tomography(angles=A, img_width=W, circle=False, normalize=False, fan_beam=fan, device=dev)
    y = p.A(x)
    v = torch.randn_like(y)
    xa = p.A_adjoint(v)
    dot = abs(float((y.double() * v.double()).sum() - (x.double() * xa.double()).sum())) / float(y.double().norm() * v.double().norm())
    print(json.dumps({"fan_beam": fan, "sino": list(y.shape), "A_ms": t(lambda: p.A(x)), "AT_ms": t(lambda: p.A_adjoint(v)), "dot_test": dot}))
