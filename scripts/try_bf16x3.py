"""experimental bf16x3 conv: parity vs fp64 conv2d and the direct / Winograd kernels, and a timing at one level"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from deepinv_amd.hip import drunet as K

dev = torch.device("cuda:0")
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
for (B, H, W, cin, cout, mode) in ((2, 24, 40, 64, 64, "plain"), (1, 17, 33, 32, 128, "relu"), (2, 16, 16, 128, 64, "res")):
    g = torch.Generator().manual_seed(H)
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)).to(dev)
    r = torch.randn(B, cout, H, W, generator=g).to(dev)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    if mode == "relu": ref = ref.relu()
    if mode == "res": ref = ref + r.double()
    geo = K.geom(B, H, W)
    def to_act(t):
        a = K.alloc(geo, t.shape[1], dev)
        av = a[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
        av[:, :, 1:H + 1, 1:W + 1] = t.view(B, -1, 8, H, W).permute(1, 0, 3, 4, 2)
        return a
    def from_act(a, c):
        av = a[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
        return av[:, :, 1:H + 1, 1:W + 1].permute(1, 0, 4, 2, 3).reshape(B, c, H, W)
    xa, ra, ya, yd = to_act(x), to_act(r), K.alloc(geo, cout, dev), K.alloc(geo, cout, dev)
    y2 = K.alloc(geo, cout, dev)
    K.conv3x3_bf16x3(geo, xa, K.pack_bf16x3_weight(w), cin, cout, ya, res1=ra if mode == "res" else None, relu=mode == "relu")
    K.conv3x3_bf16x3(geo, xa, K.pack_bf16x3_weight(w), cin, cout, y2, res1=ra if mode == "res" else None, relu=mode == "relu", planes=2)
    wd, ci, co = K.pack_conv3x3_weight(w)
    K.conv3x3(geo, xa, wd, ci, co, yd, res1=ra if mode == "res" else None, relu=mode == "relu")
    print(json.dumps({"case": [B, H, W, cin, cout, mode], "bf16x3_vs_fp64": rel(from_act(ya, cout), ref),
                      "bf16x2_vs_fp64": rel(from_act(y2, cout), ref),
                      "direct_fp32_vs_fp64": rel(from_act(yd, cout), ref)}))
# timing, DRUNet level 1 (128 ch, 160x160, B=32)
B, H, c = 32, 160, 128
geo = K.geom(B, H, H)
x, y = K.alloc(geo, c, dev), K.alloc(geo, c, dev)
x.normal_()
w = torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)
ws, ww = K.pack_bf16x3_weight(w), K.pack_winograd_weight(w)
def timeit(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print(json.dumps({"bf16x3_ms": timeit(lambda: K.conv3x3_bf16x3(geo, x, ws, c, c, y, relu=True)),
                  "bf16x2_ms": timeit(lambda: K.conv3x3_bf16x3(geo, x, ws, c, c, y, relu=True, planes=2)),
                  "winograd_fp32_ms": timeit(lambda: K.conv3x3_winograd(geo, x, ww, c, c, y, relu=True))}))
