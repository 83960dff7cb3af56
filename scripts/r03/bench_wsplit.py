"""Winograd F(2,3) bf16-split conv (csrc/drunet_wsplit.hip) against the direct bf16-split conv (csrc/drunet_split2d.hip) at
the four DRUNet levels on one MI355X: the two convolutions of a ResBlock, random and zero data, plus the error of both
against an fp64 convolution of two images.  Usage: python scripts/r03/bench_wsplit.py [B]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinv_amd.hip import drunet as K  # noqa: E402
from bench_ops import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
tot = {"direct": 0.0, "wsplit": 0.0}
for lvl, c in enumerate((64, 128, 256, 512)):
    H = 320 >> lvl
    g = K.geom(B, H, H)
    x, y, r, t = K.alloc(g, c, dev), K.alloc(g, c, dev), K.alloc(g, c, dev), K.alloc(g, c, dev)
    inner = lambda a: a[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:H + 1, 1:H + 1]
    inner(x).normal_()
    inner(r).normal_()
    w = torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)
    w2, ww = K.pack_split2d_weight(w), K.pack_wsplit_weight(w)
    fl = 2.0 * 9 * c * c * B * H * H
    row = {"lvl": lvl, "B": B, "c": c, "H": H}
    try:
        # error of conv2 (x -> y + r) of both kernels against fp64 on the first two images
        nb = min(B, 2)
        xn = inner(x)[:, :nb].permute(1, 0, 4, 2, 3).reshape(nb, c, H, H).double()
        rn = inner(r)[:, :nb].permute(1, 0, 4, 2, 3).reshape(nb, c, H, H).double()
        ref = torch.nn.functional.conv2d(xn, w.double(), padding=1) + rn
        K.conv3x3_split(g, x, w2, c, c, y, res1=r)
        out = inner(y)[:, :nb].permute(1, 0, 4, 2, 3).reshape(nb, c, H, H).double()
        row["err_direct"] = float((out - ref).norm() / ref.norm())
        y.zero_()
        K.conv3x3_wsplit(g, x, ww, c, c, y, res1=r)
        out = inner(y)[:, :nb].permute(1, 0, 4, 2, 3).reshape(nb, c, H, H).double()
        row["err_wsplit"] = float((out - ref).norm() / ref.norm())
        n = 8 if lvl == 3 else 16
        d1 = timeit(lambda: K.conv3x3_split(g, x, w2, c, c, t, relu=True, y_presplit=True), iters=30, warmup=3)
        d2 = timeit(lambda: K.conv3x3_split(g, t, w2, c, c, y, res1=r, x_presplit=True), iters=30, warmup=3)
        w1 = timeit(lambda: K.conv3x3_wsplit(g, x, ww, c, c, t, relu=True), iters=30, warmup=3)
        w2t = timeit(lambda: K.conv3x3_wsplit(g, t, ww, c, c, y, res1=r), iters=30, warmup=3)
        row.update(direct_conv1_ms=round(d1 * 1e3, 4), direct_conv2_ms=round(d2 * 1e3, 4), wsplit_conv1_ms=round(w1 * 1e3, 4),
                   wsplit_conv2_ms=round(w2t * 1e3, 4), wsplit_conv2_direct_equiv_TF=round(fl / w2t / 1e12, 1),
                   direct_conv2_direct_equiv_TF=round(fl / d2 / 1e12, 1))
        tot["direct"] += n / 2 * (d1 + d2)
        tot["wsplit"] += n / 2 * (w1 + w2t)
        z = torch.zeros_like(x)
        row["wsplit_conv2_zero_data_ms"] = round(timeit(lambda: K.conv3x3_wsplit(g, z, ww, c, c, y, res1=z), iters=30, warmup=3) * 1e3, 4)
    except Exception as e:  # noqa: BLE001
        row["error"] = repr(e)[:300]
    print(json.dumps(row), flush=True)
print(json.dumps({"B": B, "resblock_convs_ms_per_drunet": {k: round(v * 1e3, 2) for k, v in tot.items()}}))
