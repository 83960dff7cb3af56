"""Opt-in Winograd bf16-split conv (csrc/drunet_wbf16.hip) against the direct bf16-split conv at the four DRUNet levels:
time and relative difference of the outputs.  Usage: python scripts/bench_wbf16.py [B]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinv_amd.hip import drunet as K  # noqa: E402
from bench_ops import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
for lvl, c in enumerate((64, 128, 256, 512)):
    H = 320 >> lvl
    g = K.geom(B, H, H)
    x, y, y2, r = K.alloc(g, c, dev), K.alloc(g, c, dev), K.alloc(g, c, dev), K.alloc(g, c, dev)
    x[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:H + 1, 1:H + 1].normal_()
    r.normal_()
    w = torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)
    ws, wu = K.pack_bf16s_weight(w), K.pack_wbf16_weight(w)
    K.conv3x3_bf16s(g, x, ws, c, c, y, res1=r)
    K.conv3x3_wbf16(g, x, wu, c, c, y2, res1=r)
    inner = lambda t: t[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:H + 1, 1:H + 1]
    diff = float((inner(y2) - inner(y)).norm() / inner(y).norm())
    t1 = timeit(lambda: K.conv3x3_bf16s(g, x, ws, c, c, y, relu=True), iters=10, warmup=2)
    t2 = timeit(lambda: K.conv3x3_wbf16(g, x, wu, c, c, y2, relu=True), iters=10, warmup=2)
    print(json.dumps({"lvl": lvl, "B": B, "bf16s_ms": round(t1 * 1e3, 4), "wbf16_ms": round(t2 * 1e3, 4), "rel_diff": diff}))
